#!/usr/bin/env python3
"""Generate tests/golden/urdf_reference.json by IMPORTING the reference's own Python
(tools/urdf_parsing.py + tools/geometry.py) in the authoring container.

    PYTHONDONTWRITEBYTECODE=1 python -B scripts/make_golden.py

The reference cannot travel to the GPU box, so its outputs are committed as a small
fixture. What is recorded, all computed by the reference's code:

  joints   name, type, parent, child, axis, origin xyz + quaternion (xyzw), limits
           (urdf_parsing.py:49-59,272-279,320-326)
  links    inertial origin xyz + quaternion, 3x3 inertia (urdf_parsing.py:77-86,282-296)
           mass is taken from the raw XML <mass value> because the reference parser reads
           a non-existent attribute and returns 0.0 - both are recorded.
  tree     root_link_names, branch_link_names, joint_chains key -> length
           (urdf_parsing.py:133-199)
  fk       pose of every link frame relative to the root link frame at the start pose
           (hips -0.6, knees 0.4, ankles -1.2; trex_env.py:81-87 after the F2 rename),
           composed with geometry.Transform.__mul__ (geometry.py:17-21)
  visuals  every <visual> mesh as the reference parser reads it (UrdfLink.visual_shapes,
           urdf_parsing.py:93-120,299-307): link, mesh file name, origin xyz + quaternion; and its pose at the
           start pose relative to the root link frame, fk[link] * shape.origin (what a renderer places the mesh
           with, trex_env.py:156-181)
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from tools import geometry, urdf_parsing  # noqa: E402
from scipy.spatial import transform  # noqa: E402

URDF = "/root/reference/assets/trex.urdf"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "urdf_reference.json")

START = {
    "joint_femur_left": -0.6, "joint_tibia_left": 0.4, "joint_tarsometatarsus_left": -1.2,
    "joint_femur_right": -0.6, "joint_tibia_right": 0.4, "joint_tarsometatarsus_right": -1.2,
}


def tf(t):
    return {"xyz": t.translation.tolist(), "quat_xyzw": t.rotation.as_quat().tolist()}


def main():
    root = urdf_parsing.read_root_node_from_urdf(URDF)
    urdf = urdf_parsing.Urdf.from_element(root)
    raw_mass = {l.get("name"): float(l.find("inertial/mass").get("value"))
                for l in ET.parse(URDF).getroot().findall("link")}
    out = {"joints": [], "links": [], "tree": {}, "fk_start_pose": {}, "start_pose": START}
    for j in urdf.joints.values():
        out["joints"].append({
            "name": j.name, "type": j.type, "parent": j.parent_name, "child": j.child_name,
            "axis": j.axis.tolist(), "origin": tf(j.origin),
            "lower": float(j.limits.position[0]), "upper": float(j.limits.position[1]),
        })
    for l in urdf.links.values():
        out["links"].append({
            "name": l.name, "inertial_origin": tf(l.inertia.origin),
            "inertia": l.inertia.inertia.tolist(),
            "mass_reference_parser": l.inertia.mass,  # 0.0: SURVEY F8
            "mass_xml": raw_mass[l.name],
        })
    out["tree"]["root_link_names"] = sorted(urdf.root_link_names)
    out["tree"]["branch_link_names"] = sorted(urdf.branch_link_names)
    out["tree"]["joint_chains"] = {k: len(v) for k, v in sorted(urdf.joint_chains.items())}

    # forward kinematics with the reference's Transform algebra
    by_parent = urdf.parent_link_name_to_joint
    root_name = urdf.root_link_names[0]
    poses = {root_name: geometry.Transform()}
    stack = [root_name]
    while stack:
        p = stack.pop()
        for j in by_parent.get(p, []):
            q = START.get(j.name, 0.0) if j.type == "revolute" else 0.0
            rot = geometry.Transform(rotation=transform.Rotation.from_rotvec(q * j.axis))
            poses[j.child_name] = poses[p] * j.origin * rot
            stack.append(j.child_name)
    out["fk_start_pose"] = {k: tf(v) for k, v in sorted(poses.items())}
    # world position of every link COM in the root link frame
    out["com_start_pose"] = {
        l.name: (poses[l.name].apply(l.inertia.origin.translation)).tolist()
        for l in urdf.links.values()}
    out["visuals"] = []
    for l in urdf.links.values():
        for shape in l.visual_shapes:
            out["visuals"].append({"link": l.name, "file": shape.filename, "origin": tf(shape.origin),
                                   "pose_start": tf(poses[l.name] * shape.origin)})
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("joints", len(out["joints"]), "links", len(out["links"]),
          "visuals", len(out["visuals"]), "chains", len(out["tree"]["joint_chains"]), "branch", len(out["tree"]["branch_link_names"]),
          "bytes", os.path.getsize(OUT))


if __name__ == "__main__":
    main()
