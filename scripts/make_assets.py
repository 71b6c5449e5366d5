#!/usr/bin/env python3
"""Generate the in-tree model assets from the reference's data files.

Run in the authoring container only (needs /root/reference/assets):

    python scripts/make_assets.py

Outputs (derived DATA, no reference source code):

  trex-gym_amd/assets/trex_collide.urdf   joints + inertials + <visual> elements of assets/trex.urdf (the 252
                                          visual origins and mesh file NAMES are what the rollout export places
                                          meshes with; the 32 MB of .obj files themselves are not shipped),
                                          <collision> elements ADDED (the v1 URDF has none,
                                          SURVEY F3): each of the 28 convex hulls in
                                          assets/collisions/*.dae is attached to the link that shows
                                          the same-named visual mesh, at that visual's <origin>
                                          (SURVEY F4).
  trex-gym_amd/assets/collisions/COL_*.obj  hull vertices + triangles re-written as Wavefront OBJ
                                          (plain text, loadable by pybullet for cross-checks).
  trex-gym_amd/assets/floor.urdf          the 1000 x 1000 x 0.001 static box (floor.urdf:18-23),
                                          visual plane mesh dropped.

All numeric attribute strings of joints / inertials are carried over verbatim so the
model compiled from trex_collide.urdf is bit-identical to the one compiled from
assets/trex.urdf + assets/collisions (tests/test_model.py checks this when the reference is present).
"""
import glob
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = "/root/reference/assets"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd", "assets")
NS = {"c": "http://www.collada.org/2005/11/COLLADASchema"}


def hull_to_mesh_name(hull):
    """COL_tibia_L_convex_hull -> tibia_left (SURVEY F2/F4 rename rule)."""
    n = hull[len("COL_"):-len("_convex_hull")]
    n = re.sub(r"_L(_|$)", r"_left\1", n)
    n = re.sub(r"_R(_|$)", r"_right\1", n)
    return n


def read_dae(path):
    root = ET.parse(path).getroot()
    verts = None
    for src in root.iter("{%s}source" % NS["c"]):
        if src.get("id").endswith("positions"):
            verts = np.array(src.find("c:float_array", NS).text.split(), float).reshape(-1, 3)
    tri = root.find(".//c:triangles", NS)
    inputs = tri.findall("c:input", NS)
    stride = max(int(i.get("offset")) for i in inputs) + 1
    v_off = [int(i.get("offset")) for i in inputs if i.get("semantic") == "VERTEX"][0]
    idx = np.array(tri.find("c:p", NS).text.split(), int).reshape(-1, stride)[:, v_off].reshape(-1, 3)
    # the scene node must carry no transform (F4)
    for node in root.iter("{%s}node" % NS["c"]):
        assert node.find("c:matrix", NS) is None, path
    return verts, idx


def read_obj_verts(path):
    return np.array([[float(x) for x in l.split()[1:4]] for l in open(path) if l.startswith("v ")])


def best_sign_flip(hv, ov):
    """The hull must sit in the visual mesh's local frame. 27 of 28 do; ilium_L is a point
    reflection. Pick the axis sign pattern whose bbox matches the visual best."""
    def err(h):
        return max(np.abs(h.min(0) - ov.min(0)).max(), np.abs(h.max(0) - ov.max(0)).max())
    if err(hv) < 0.005:  # already in frame: never flip a (near-)symmetric hull
        return err(hv), np.ones(3)
    best = None
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (1, -1):
                s = np.array([sx, sy, sz], float)
                h = hv * s
                d = max(np.abs(h.min(0) - ov.min(0)).max(), np.abs(h.max(0) - ov.max(0)).max())
                if best is None or d < best[0] - 1e-9:
                    best = (d, s)
    return best


def main():
    os.makedirs(os.path.join(OUT, "collisions"), exist_ok=True)
    tree = ET.parse(os.path.join(REF, "trex.urdf"))
    robot = tree.getroot()

    # visual mesh name -> (link element, origin attrs)
    vis = {}
    for link in robot.findall("link"):
        for v in link.findall("visual"):
            mesh = v.find("geometry/mesh")
            name = os.path.basename(mesh.get("filename"))[:-4]
            assert name not in vis
            vis[name] = (link, dict(v.find("origin").attrib))

    collisions = {}  # link name -> list of (obj file, origin attrs)
    report = []
    for dae in sorted(glob.glob(os.path.join(REF, "collisions", "*.dae"))):
        hull = os.path.basename(dae)[:-4]
        mesh_name = hull_to_mesh_name(hull)
        link, origin = vis[mesh_name]
        hv, tri = read_dae(dae)
        ov = read_obj_verts(os.path.join(REF, "meshes", mesh_name + ".obj"))
        d, s = best_sign_flip(hv, ov)
        assert d < 0.005, (hull, d)
        hv = hv * s
        if np.prod(s) < 0:
            tri = tri[:, ::-1]  # keep outward winding under a reflection
        obj_name = "COL_" + hull[len("COL_"):-len("_convex_hull")] + ".obj"
        with open(os.path.join(OUT, "collisions", obj_name), "w") as f:
            f.write("# convex hull of %s, frame of visual mesh %s.obj; sign pattern %s\n"
                    % (hull, mesh_name, s.astype(int).tolist()))
            for v in hv:
                f.write("v %.9g %.9g %.9g\n" % tuple(v))
            for t in tri:
                f.write("f %d %d %d\n" % tuple(t + 1))
        collisions.setdefault(link.get("name"), []).append((obj_name, origin))
        report.append((hull, link.get("name"), len(hv), d, s.astype(int).tolist()))

    out = ET.Element("robot", {"name": robot.get("name")})
    out.append(ET.Comment(
        " generated by scripts/make_assets.py from assets/trex.urdf + assets/collisions/*.dae of "
        "bingjeff/trex-gym v1: visual elements kept (mesh files not shipped), collision hulls attached (see DESIGN.md) "))
    for j in robot.findall("joint"):
        out.append(j)
    n_col = 0
    for link in robot.findall("link"):
        nl = ET.SubElement(out, "link", {"name": link.get("name")})
        nl.append(link.find("inertial"))
        for v in link.findall("visual"):      # origin + mesh file name, verbatim (rollout export)
            nl.append(v)
        for obj_name, origin in collisions.get(link.get("name"), []):
            c = ET.SubElement(nl, "collision")
            ET.SubElement(c, "origin", origin)
            g = ET.SubElement(c, "geometry")
            ET.SubElement(g, "mesh", {"filename": "collisions/" + obj_name, "scale": "1.0 1.0 1.0"})
            n_col += 1
    ET.indent(out, space="  ")
    ET.ElementTree(out).write(os.path.join(OUT, "trex_collide.urdf"), encoding="utf-8",
                              xml_declaration=True)

    with open(os.path.join(OUT, "floor.urdf"), "w") as f:
        f.write("""<?xml version="1.0" ?>
<!-- static ground box, same geometry as bingjeff/trex-gym assets/floor.urdf (top face z = +0.0005) -->
<robot name="floor">
  <link name="baseLink">
    <inertial>
      <origin rpy="0 0 0" xyz="0 0 0"/>
      <mass value="0.0"/>
      <inertia ixx="0" ixy="0" ixz="0" iyy="0" iyz="0" izz="0"/>
    </inertial>
    <collision>
      <origin rpy="0 0 0" xyz="0 0 0"/>
      <geometry>
        <box size="1000 1000 0.001"/>
      </geometry>
    </collision>
  </link>
</robot>
""")
    for r in report:
        print("%-40s -> %-32s %3d verts  bbox err %.4f  sign %s" % r)
    print("collision elements:", n_col, "total verts:", sum(r[2] for r in report))


if __name__ == "__main__":
    sys.exit(main())
