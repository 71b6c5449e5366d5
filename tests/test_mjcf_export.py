"""MJCF export: MuJoCo is not installed, so the file is re-read with ElementTree and checked
against the compiled model: tree, masses, joint axes/ranges, start-pose kinematics, geoms."""
import xml.etree.ElementTree as ET

import numpy as np

from conftest import ASSET_URDF
from oracle import trex_model as tm


def _q2m(q):   # wxyz
    w, x, y, z = q
    return tm.quat_to_matrix([x, y, z, w])


def test_mjcf_export_is_consistent_with_the_compiled_model(model, tmp_path):
    from trex_gym import _capi, mjcf_export
    m = _capi.Model(ASSET_URDF)
    path = tmp_path / "trex.xml"
    text = mjcf_export.export_mjcf(m, str(path))
    root = ET.fromstring(text)
    assert root.tag == "mujoco" and root.find("compiler").get("angle") == "radian" and root.find("compiler").get("coordinate") == "local"
    assert float(root.find("option").get("timestep")) == 0.002
    bodies = list(root.iter("body"))
    assert len(bodies) == 26 and len(list(root.iter("freejoint"))) == 1
    hinges = list(root.iter("joint"))
    assert sorted(h.get("name") for h in hinges) == model["obs_joint_names"]
    assert abs(sum(float(b.find("inertial").get("mass")) for b in bodies) - 5180.2759) < 1e-3
    geoms = [g for g in root.iter("geom") if g.get("type") in ("capsule", "sphere")]
    n_caps = sum(1 for g in geoms if g.get("type") == "capsule")
    assert len(geoms) + n_caps == len(tm.use_primitive_collision(model)["hull_radius"])   # end spheres: 2 per capsule
    # kinematics: compose pos/quat down the tree at q = 0 and compare with the oracle's zero-pose FK
    pose = {}

    def walk(elem, R, p):
        for b in elem.findall("body"):
            Rb = R @ _q2m([float(x) for x in b.get("quat").split()])
            pb = p + R @ np.array([float(x) for x in b.get("pos").split()])
            pose[b.get("name")] = (Rb, pb)
            walk(b, Rb, pb)
    walk(root.find("worldbody"), np.eye(3), np.zeros(3))
    from oracle import oracle as O
    orc = O.Oracle(model)
    s = orc.new_state()
    st = np.zeros(63)
    st[2] = 3.0
    st[6] = 1.0
    orc.set_state(s, st)
    pos, rot = orc.body_poses(s)
    for i, name in enumerate(model["body_names"]):
        R, p = pose[name]
        np.testing.assert_allclose(p, pos[i], atol=1e-7, err_msg=name)
        np.testing.assert_allclose(R, rot[i], atol=1e-7, err_msg=name)
    for h in hinges:
        b = model["joint_names"].index(h.get("name"))
        np.testing.assert_allclose([float(x) for x in h.get("axis").split()], model["joint_axis"][b], atol=1e-9)
        np.testing.assert_allclose([float(x) for x in h.get("range").split()], [model["q_lower"][b], model["q_upper"][b]], atol=1e-9)
        assert float(h.get("damping")) == 1.0
    for body in bodies:   # inertia about the COM in the body frame, MuJoCo fullinertia order xx yy zz xy xz yz
        fi = [float(x) for x in body.find("inertial").get("fullinertia").split()]
        a = model["inertia"][model["body_names"].index(body.get("name"))]
        np.testing.assert_allclose(fi, [a[0], a[3], a[5], a[1], a[2], a[4]], rtol=1e-8)
    assert path.read_text().startswith("<mujoco")
