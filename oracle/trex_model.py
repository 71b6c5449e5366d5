"""ORACLE (test infrastructure, not product code): numpy restatement of the host-side model
compiler - URDF -> reduced articulated tree ("model blob").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path is the C++ loader in trex-gym_amd/csrc/model_loader.cpp; tests compare the
two field by field.

What it restates
  * URDF conventions of the reference's parser: rpy = extrinsic x-y-z Euler angles
    (tools/urdf_parsing.py:267-269), symmetric inertia fill (:282-296), joint origin / axis /
    limit attributes (:49-59,272-279,320-326) - with <mass value> read correctly.
  * what pybullet's loadURDF + the env do with that file (trex_robot.py:39-65,300-320,
    trex_env.py:81-87,105-106): revolute joints sorted by name give the observation order,
    link_atlas_axis COM is the reward point, reset poses the ROOT LINK'S INERTIAL FRAME at
    (0,0,3)/identity, the six start angles after the F2 rename.
  * fixed joints are merged into their parent bodies (parallel-axis composite inertia); this is
    dynamically equivalent to the un-merged tree pybullet simulates.
  * collision hulls: <collision><mesh .obj> of the generated trex_collide.urdf, or - for the
    reference's own trex.urdf, which has no <collision> - assets/collisions/*.dae
    placed with the <visual><origin> of the same-named mesh.

Frames.  Body 0 (base) frame = inertial frame of the root link (COM of that single link,
principal axes).  Body i>0 frame = URDF link frame of the child link of its revolute joint;
the joint axis passes through that frame's origin.
"""
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

START_CONFIGURATION = {  # trex_env.py:81-87, F2 rename applied (both spellings accepted)
    "joint_femur_left": -0.6, "joint_tibia_left": 0.4, "joint_tarsometatarsus_left": -1.2,
    "joint_femur_right": -0.6, "joint_tibia_right": 0.4, "joint_tarsometatarsus_right": -1.2,
}
HEAD_LINK = "link_atlas_axis"  # trex_robot.py:316 (b'atlas_axis_link' before the rename)
_DAE_NS = {"c": "http://www.collada.org/2005/11/COLLADASchema"}


def rename_v0_name(name):
    """'femur_L_joint' -> 'joint_femur_left', 'atlas_axis_link' -> 'link_atlas_axis' (SURVEY F2)."""
    m = re.match(r"^(.*)_(joint|link)$", name)
    if not m:
        return name
    stem, kind = m.groups()
    stem = re.sub(r"_L$", "_left", stem)
    stem = re.sub(r"_R$", "_right", stem)
    return "%s_%s" % (kind, stem)


def rpy_to_matrix(rpy):
    """Extrinsic x-y-z (scipy from_euler('xyz')): R = Rz(yaw) @ Ry(pitch) @ Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def axis_angle_matrix(axis, q):
    a = np.asarray(axis, float)
    k = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * k + (1 - np.cos(q)) * (k @ k)


def quat_to_matrix(q):
    """q = (x, y, z, w), pybullet order."""
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def matrix_to_quat(m):
    """Rotation matrix -> (x, y, z, w), w >= 0."""
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        q = [0, 0, 0, (m[k, j] - m[j, k]) / s]
        q[i] = 0.25 * s
        q[j] = (m[j, i] + m[i, j]) / s
        q[k] = (m[k, i] + m[i, k]) / s
    q = np.array(q)
    return q if q[3] >= 0 else -q


class Tf:
    """Rigid transform p_parent = R p_child + t; compose = geometry.Transform.__mul__."""

    def __init__(self, R=None, t=None):
        self.R = np.eye(3) if R is None else np.asarray(R, float)
        self.t = np.zeros(3) if t is None else np.asarray(t, float)

    def __mul__(self, o):
        return Tf(self.R @ o.R, self.R @ o.t + self.t)

    def inverse(self):
        return Tf(self.R.T, -self.R.T @ self.t)

    def apply(self, v):
        v = np.asarray(v, float)
        return (self.R @ v.T).T + self.t


def _vec(s):
    return np.array([float(x) for x in s.split()])[:3]


def _origin(node):
    o = None if node is None else node.find("origin")
    if o is None:
        return Tf()
    return Tf(rpy_to_matrix(_vec(o.get("rpy", "0 0 0"))), _vec(o.get("xyz", "0 0 0")))


def _read_obj_vertices(path):
    return np.array([[float(x) for x in l.split()[1:4]] for l in open(path) if l.startswith("v ")])


def _read_dae_vertices(path):
    root = ET.parse(path).getroot()
    for src in root.iter("{%s}source" % _DAE_NS["c"]):
        if src.get("id").endswith("positions"):
            return np.array(src.find("c:float_array", _DAE_NS).text.split(), float).reshape(-1, 3)
    raise ValueError(path)


def parse_urdf(urdf_path):
    """-> (joints, links) as plain dict lists in document order."""
    root = ET.parse(urdf_path).getroot()
    joints, links = [], []
    for j in root.findall("joint"):
        lim = j.find("limit")
        dyn = j.find("dynamics")
        ax = j.find("axis")
        joints.append(dict(
            name=j.get("name"), type=j.get("type"),
            parent=j.find("parent").get("link"), child=j.find("child").get("link"),
            origin=_origin(j), axis=_vec(ax.get("xyz")) if ax is not None else np.array([0., 0., 1.]),
            lower=float(lim.get("lower", "0")) if lim is not None else 0.0,
            upper=float(lim.get("upper", "0")) if lim is not None else 0.0,
            damping=float(dyn.get("damping", "0")) if dyn is not None else 0.0))
    for l in root.findall("link"):
        ine = l.find("inertial")
        mass, inertia, io = 0.0, np.zeros((3, 3)), Tf()
        if ine is not None:
            io = _origin(ine)
            mass = float(ine.find("mass").get("value"))  # NOT an attribute of <inertial> (F8)
            i = ine.find("inertia")
            g = lambda k: float(i.get(k, "0"))
            inertia = np.array([[g("ixx"), g("ixy"), g("ixz")],
                                [g("ixy"), g("iyy"), g("iyz")],
                                [g("ixz"), g("iyz"), g("izz")]])
        visuals = [(v.find("geometry/mesh").get("filename"), _origin(v))
                   for v in l.findall("visual") if v.find("geometry/mesh") is not None]
        collisions = [(c.find("geometry/mesh").get("filename"), _origin(c))
                      for c in l.findall("collision") if c.find("geometry/mesh") is not None]
        links.append(dict(name=l.get("name"), mass=mass, inertia=inertia, inertial_origin=io,
                          visuals=visuals, collisions=collisions))
    return joints, links


def _hulls_for_links(urdf_path, links, collisions_dir):
    """link name -> list of (vertices [n,3] in link frame, source tag)."""
    base = os.path.dirname(os.path.abspath(urdf_path))
    out = {}
    have_collision = any(l["collisions"] for l in links)
    if have_collision:
        for l in links:
            for fn, tf in l["collisions"]:
                v = _read_obj_vertices(os.path.join(base, fn))
                out.setdefault(l["name"], []).append((tf.apply(v), os.path.basename(fn)))
        return out
    if collisions_dir is None:
        return out
    vis = {}
    for l in links:
        for fn, tf in l["visuals"]:
            vis[os.path.basename(fn)[:-4]] = (l["name"], tf)
    for f in sorted(os.listdir(collisions_dir)):
        if not f.endswith("_convex_hull.dae"):
            continue
        n = f[len("COL_"):-len("_convex_hull.dae")]
        n = re.sub(r"_L(_|$)", r"_left\1", n)
        n = re.sub(r"_R(_|$)", r"_right\1", n)
        link_name, tf = vis[n]
        v = _read_dae_vertices(os.path.join(collisions_dir, f))
        if f == "COL_ilium_L_convex_hull.dae":
            v = -v  # point-reflected relative to ilium_left.obj (scripts/make_assets.py)
        out.setdefault(link_name, []).append((tf.apply(v), f))
    return out


def compile_model(urdf_path, collisions_dir=None, start_configuration=None):
    """URDF -> dict of numpy arrays (float64). See module docstring for frames."""
    joints, links = parse_urdf(urdf_path)
    link_by_name = {l["name"]: l for l in links}
    children = {}
    child_names = set()
    for j in joints:
        children.setdefault(j["parent"], []).append(j)
        child_names.add(j["child"])
    roots = [l["name"] for l in links if l["name"] not in child_names]
    assert len(roots) == 1, roots
    root = roots[0]
    hulls = _hulls_for_links(urdf_path, links, collisions_dir)

    bodies = []  # dict(head, parent, joint, members[(link, Tf body<-link)])

    def new_body(head, parent, joint, tf_head):
        b = dict(head=head, parent=parent, joint=joint, members=[])
        bodies.append(b)
        idx = len(bodies) - 1
        stack = [(head, tf_head)]
        pending = []
        while stack:
            ln, tf = stack.pop(0)
            b["members"].append((ln, tf))
            for j in children.get(ln, []):
                if j["type"] == "fixed":
                    stack.append((j["child"], tf * j["origin"]))
                elif j["type"] == "revolute":
                    pending.append((j, tf))
                else:
                    raise ValueError("unsupported joint type %s" % j["type"])
        return idx, pending

    # body 0: frame = inertial frame of the root link
    root_tf = link_by_name[root]["inertial_origin"].inverse()
    work = [(root, -1, None, root_tf, None)]
    # depth-first in document order so that parent index < child index
    def build(head, parent, joint, tf_head, tf_parent_joint):
        idx, pending = new_body(head, parent, joint, tf_head)
        bodies[idx]["tf_parent"] = tf_parent_joint
        # document order of revolute joints among this body's members
        order = {id(j): k for k, j in enumerate(joints)}
        for j, tf in sorted(pending, key=lambda p: order[id(p[0])]):
            build(j["child"], idx, j, Tf(), tf * j["origin"])

    build(root, -1, None, root_tf, None)
    nb = len(bodies)
    m = dict(nb=nb)
    m["body_names"] = [b["head"] for b in bodies]
    m["joint_names"] = [b["joint"]["name"] if b["joint"] else "" for b in bodies]
    m["parent"] = np.array([b["parent"] for b in bodies], np.int32)
    depth = np.zeros(nb, np.int32)
    for i in range(1, nb):
        depth[i] = depth[m["parent"][i]] + 1
    m["depth"] = depth
    m["joint_axis"] = np.zeros((nb, 3))
    m["joint_pos"] = np.zeros((nb, 3))
    m["joint_rot"] = np.tile(np.eye(3).reshape(1, 9), (nb, 1))
    m["q_lower"] = np.zeros(nb)
    m["q_upper"] = np.zeros(nb)
    m["joint_damping"] = np.zeros(nb)
    m["mass"] = np.zeros(nb)
    m["com"] = np.zeros((nb, 3))
    m["inertia"] = np.zeros((nb, 6))  # xx xy xz yy yz zz about COM, body axes
    m["n_merged"] = np.zeros(nb, np.int32)
    hull_xyz, hull_start, hull_groups = [], [0], [0]
    sphere_c, sphere_r = np.zeros((nb, 3)), np.zeros(nb)
    head_body, head_point = -1, np.zeros(3)
    for i, b in enumerate(bodies):
        if b["joint"] is not None:
            j = b["joint"]
            a = j["axis"] / np.linalg.norm(j["axis"])
            m["joint_axis"][i] = a
            m["joint_pos"][i] = b["tf_parent"].t
            m["joint_rot"][i] = b["tf_parent"].R.reshape(9)
            m["q_lower"][i], m["q_upper"][i] = j["lower"], j["upper"]
            m["joint_damping"][i] = j["damping"]
        mass, mc = 0.0, np.zeros(3)
        parts = []
        for ln, tf in b["members"]:
            l = link_by_name[ln]
            c = tf.apply(l["inertial_origin"].t)
            R = tf.R @ l["inertial_origin"].R
            parts.append((l["mass"], c, R @ l["inertia"] @ R.T))
            mass += l["mass"]
            mc += l["mass"] * c
            if ln == HEAD_LINK:
                head_body, head_point = i, c
        com = mc / mass
        I = np.zeros((3, 3))
        for ml, c, Il in parts:
            d = c - com
            I += Il + ml * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        m["mass"][i], m["com"][i] = mass, com
        m["inertia"][i] = [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]
        m["n_merged"][i] = len(b["members"])
        verts = []
        for ln, tf in b["members"]:
            for v, _tag in hulls.get(ln, []):
                verts.append(tf.apply(v))
                hull_groups.append(hull_groups[-1] + len(v))
        if verts:
            v = np.concatenate(verts)
            hull_xyz.append(v)
            lo, hi = v.min(0), v.max(0)
            sphere_c[i] = 0.5 * (lo + hi)
            sphere_r[i] = np.sqrt(((v - sphere_c[i]) ** 2).sum(1).max())
        hull_start.append(hull_start[-1] + (len(verts) and sum(len(x) for x in verts)))
    # every URDF link (document order): body it was merged into, link frame in that body's frame
    m["link_names"] = [l["name"] for l in links]
    lb, lR, lt = {}, {}, {}
    for i, b in enumerate(bodies):
        for ln, tf in b["members"]:
            lb[ln], lR[ln], lt[ln] = i, tf.R.reshape(9), tf.t
    m["link_body"] = np.array([lb[n] for n in m["link_names"]], np.int32)
    m["link_tf"] = np.array([np.concatenate([lR[n], lt[n]]) for n in m["link_names"]])
    m["hull_xyz"] = np.concatenate(hull_xyz) if hull_xyz else np.zeros((0, 3))
    m["hull_start"] = np.array(hull_start, np.int32)
    m["hull_group_start"] = np.array(hull_groups, np.int32)   # one group per original convex hull
    m["hull_radius"] = np.zeros(len(m["hull_xyz"]))           # > 0 only for fitted sphere / capsule ends
    m["sphere_center"], m["sphere_radius"] = sphere_c, sphere_r
    m["head_body"], m["head_point"] = head_body, head_point

    # observation / action order: revolute joint names sorted (trex_robot.py:311-314)
    names = sorted(n for n in m["joint_names"] if n)
    m["obs_joint_names"] = names
    m["obs_order"] = np.array([m["joint_names"].index(n) for n in names], np.int32)
    # pybullet joint indices of the revolute joints (= document index of the joint, since
    # loadURDF numbers links depth-first; reported only through the model shim)
    doc_index = {j["name"]: k for k, j in enumerate(joints)}
    m["revolute_joint_indices"] = np.array([doc_index[n] for n in names], np.int32)
    m["num_urdf_joints"] = len(joints)
    # trex_robot.py:318-320 sums links 0..n-1 and skips the base link
    m["total_mass"] = float(sum(l["mass"] for l in links))
    m["total_mass_excluding_base"] = float(sum(l["mass"] for l in links if l["name"] != root))

    q0 = np.zeros(nb)
    if start_configuration is None:   # the env's default pose, where those joints exist (generic URDFs: zeros)
        for k, v in START_CONFIGURATION.items():
            if k in m["joint_names"]:
                q0[m["joint_names"].index(k)] = v
    else:
        for k, v in start_configuration.items():
            q0[m["joint_names"].index(rename_v0_name(k))] = v  # unknown joint raises, like KeyError at trex_robot.py:307
    m["q_start"] = q0
    m["base_start_pos"] = np.array([0.0, 0.0, 3.0])  # trex_env.py:105
    m["base_start_quat"] = np.array([0.0, 0.0, 0.0, 1.0])  # rpy = 0, trex_env.py:106
    return m


# ---------------------------------------------------------------- collision primitives
_OCTANTS = [(1, 1, 1), (1, -1, 1), (1, -1, -1), (1, 1, -1), (-1, 1, 1), (-1, -1, 1), (-1, -1, -1), (-1, 1, -1)]


def _principal_box(pts):
    """PCA-aligned bounding box as tools/mesh_primitives.py:323-344: z = dominant axis, y = second,
    x = y x z; centre = mid-range in that frame. Axis signs are canonicalised (largest component
    positive) so that every implementation enumerates the octants in the same order."""
    c = pts.mean(0)
    X = pts - c
    u, _, _ = np.linalg.svd(X.T @ X)
    z, y = u[:, 0].copy(), u[:, 1].copy()
    for a in (z, y):
        if a[np.argmax(np.abs(a))] < 0:
            a *= -1
    x = np.cross(y, z)
    A = np.stack([x, y, z], 1)
    al = X @ A
    ctr, hwl = 0.5 * (al.max(0) + al.min(0)), al.max(0) - al.min(0)
    return A, A @ ctr + c, hwl


def fit_primitives(points, max_radius, max_divisions=4, min_points=4, _depth=0):
    """Capsule / sphere fitting of the reference's tools/mesh_primitives.py:347-402
    (get_sphere_or_capsule + subdivide_points_to_geometry), restated: radius = half the larger of the
    two minor box extents, length = major extent - 2 radius (sphere when <= 0); split into octants of
    the box frame while radius > max_radius. The reference keeps octants with more than 100 points
    (dense visual meshes); the hulls have 36..255 vertices, so min_points is a parameter (default 4).
    Returns [(p0, p1, radius)] with p0 == p1 for spheres."""
    pts = np.asarray(points, float)
    A, ctr, hwl = _principal_box(pts)
    r = 0.5 * max(hwl[0], hwl[1])
    L = hwl[2] - 2.0 * r
    if r > max_radius and _depth < max_divisions:
        loc = (pts - ctr) @ A
        out = []
        for sx, sy, sz in _OCTANTS:
            msk = ((loc[:, 0] >= 0) == (sx > 0)) & ((loc[:, 1] >= 0) == (sy > 0)) & ((loc[:, 2] >= 0) == (sz > 0))
            if msk.sum() > min_points:
                out += fit_primitives(pts[msk], max_radius, max_divisions, min_points, _depth + 1)
        return out
    if L > 0:
        return [(ctr - A[:, 2] * (0.5 * L), ctr + A[:, 2] * (0.5 * L), r)]
    return [(ctr, ctr.copy(), r)]


def use_primitive_collision(m, max_radius=0.2, max_divisions=3, min_points=4):
    """Replace every convex hull by its fitted capsules / spheres; the contact generator then sees the
    capsule END SPHERES as points with a radius (a capsule lying flat touches with both ends)."""
    m = dict(m)
    xyz, rad, start = [], [], [0]
    gs = m["hull_group_start"]
    for b in range(m["nb"]):
        for g in range(len(gs) - 1):
            if not (m["hull_start"][b] <= gs[g] < m["hull_start"][b + 1]):
                continue
            for p0, p1, r in fit_primitives(m["hull_xyz"][gs[g]:gs[g + 1]], max_radius, max_divisions, min_points):
                xyz.append(p0); rad.append(r)
                if np.abs(p1 - p0).max() > 0:
                    xyz.append(p1); rad.append(r)
        start.append(len(xyz))
    m["hull_xyz"] = np.array(xyz).reshape(-1, 3)
    m["hull_radius"] = np.array(rad)
    m["hull_start"] = np.array(start, np.int32)
    m["hull_group_start"] = np.array([0, len(xyz)], np.int32)
    sc, sr = np.zeros((m["nb"], 3)), np.zeros(m["nb"])
    for b in range(m["nb"]):
        v, r = m["hull_xyz"][start[b]:start[b + 1]], m["hull_radius"][start[b]:start[b + 1]]
        if len(v):
            lo, hi = (v - r[:, None]).min(0), (v + r[:, None]).max(0)
            sc[b] = 0.5 * (lo + hi)
            sr[b] = (np.sqrt(((v - sc[b]) ** 2).sum(1)) + r).max()
    m["sphere_center"], m["sphere_radius"] = sc, sr
    return m


def default_params():
    """Engine parameters: env-specified ones cite the reference; the rest are pybullet
    defaults as recollected in SURVEY Appendix C (uncheckable offline -> parameters)."""
    return dict(
        dt=0.01 / 5,            # trex_env.py:54,71
        substeps=5,             # trex_env.py:18,73 (action_repeat 1 x NUM_SUBSTEPS)
        iterations=60,          # trex_env.py:57,72,115
        gravity=9.81,           # trex_env.py:20,117
        motor_kp=5e-3,          # trex_robot.py:421
        motor_kd=0.1,           # sqrt(2*1.0*kp), trex_robot.py:401
        motor_max_force=3e5,    # trex_robot.py:260
        floor_z=0.0005,         # floor.urdf:21 box 0.001 thick centred on z=0
        friction=0.25,          # 0.5 x 0.5, Bullet product rule [EXT]
        erp=0.2,                # joint-limit rows [EXT]
        contact_erp=0.2,        # contact rows [EXT]
        contact_margin=0.02,    # contactBreakingThreshold [EXT]
        link_damping=0.04,      # btMultiBody linear = angular damping [EXT]
        max_coordinate_velocity=100.0,  # btMultiBody m_maxCoordinateVelocity [EXT]
        max_contacts=13,   # 25 motor rows + 3 x 13 contact rows = the 64 lanes of a wavefront
    )
