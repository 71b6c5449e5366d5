"""CPU-side checks of the product library: it loads, exports every symbol include/trex_batch.h
declares, and its C++ model compiler agrees with the oracle's numpy model compiler field by field.
No compute call is made here (no GPU in this suite)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ASSET_URDF, REFERENCE_ROOT, ROOT


@pytest.fixture(scope="module")
def capi():
    from trex_gym import _capi
    return _capi


def test_header_symbols_exported(capi):
    header = open(os.path.join(ROOT, "include", "trex_batch.h")).read()
    declared = set(re.findall(r"\b(trex_[a-z_0-9]+)\s*\(", header))
    assert declared == set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(capi.lib, name), name


def test_model_matches_oracle_model(capi, model):
    m = capi.Model(ASSET_URDF)
    assert m.num_bodies == 26 and m.num_joints == 25 and m.num_urdf_joints == 132
    assert m.joint_names == model["obs_joint_names"]
    assert m.urdf_joint_indices == list(model["revolute_joint_indices"])
    assert abs(m.total_mass(False) - 4834.866376) < 1e-5   # trex_robot.py:318-320 (base link skipped)
    assert abs(m.total_mass(True) - 5180.275861) < 1e-5
    np.testing.assert_allclose(m.lower, model["q_lower"][model["obs_order"]], rtol=0, atol=0)
    np.testing.assert_allclose(m.upper, model["q_upper"][model["obs_order"]], rtol=0, atol=0)
    for name in ["parent", "depth", "joint_axis", "joint_pos", "joint_rot", "q_lower", "q_upper",
                 "joint_damping", "mass", "com", "inertia", "obs_order", "head_point", "hull_xyz",
                 "hull_start", "sphere_center", "sphere_radius", "q_start", "base_start_pos",
                 "base_start_quat", "revolute_joint_indices", "link_body", "link_tf"]:
        got = m.array(name)
        want = np.asarray(model[name], float).reshape(-1)
        assert got.shape == want.shape, name
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, err_msg=name)
    assert int(m.array("head_body")[0]) == model["head_body"]
    assert [n for n, _ in m.links()] == model["link_names"] and len(m.links()) == 133


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="reference only in the authoring container")
def test_reference_urdf_with_dae_hulls_gives_same_model(capi):
    a = capi.Model(ASSET_URDF)
    b = capi.Model(os.path.join(REFERENCE_ROOT, "assets", "trex.urdf"),
                   os.path.join(REFERENCE_ROOT, "assets", "collisions"))
    for name in ["parent", "joint_pos", "joint_rot", "mass", "com", "inertia", "sphere_radius"]:
        np.testing.assert_array_equal(a.array(name), b.array(name), err_msg=name)
    # hull vertices went through a decimal text round trip (%.9g): equal to f32 precision
    np.testing.assert_allclose(a.array("hull_xyz"), b.array("hull_xyz"), atol=1e-7)


def test_error_behaviour(capi, tmp_path):
    with pytest.raises(capi.TrexError) as e:
        capi.Model(str(tmp_path / "missing.urdf"))
    assert e.value.code == -2
    bad = tmp_path / "bad.urdf"
    bad.write_text("<robot name='x'><link name='a'></robot>")
    with pytest.raises(capi.TrexError) as e:
        capi.Model(str(bad))
    assert e.value.code == -3
    pris = tmp_path / "pris.urdf"
    pris.write_text("<robot name='x'><link name='a'><inertial><mass value='1'/><inertia ixx='1' iyy='1' izz='1'/></inertial></link>"
                    "<link name='b'><inertial><mass value='1'/><inertia ixx='1' iyy='1' izz='1'/></inertial></link>"
                    "<joint name='j' type='prismatic'><parent link='a'/><child link='b'/></joint></robot>")
    with pytest.raises(capi.TrexError) as e:
        capi.Model(str(pris))
    assert e.value.code == -4
    m = capi.Model(ASSET_URDF)
    with pytest.raises(capi.TrexError):     # the reference raises KeyError at trex_robot.py:307
        m.set_start_angle("no_such_joint", 0.1)
    m.set_start_angle("femur_L_joint", -0.5)   # pre-rename spelling accepted
    assert m.array("q_start")[17] == -0.5
    with pytest.raises(capi.TrexError):
        m.set_param("no_such_param", 1.0)
    m.set_param("iterations", 30)
    assert m.get_param("iterations") == 30
    assert m.get_param("dt") == pytest.approx(0.002)


def test_small_generic_urdf(capi, tmp_path):
    """A two-link pendulum with no hulls loads: the loader is not T-rex specific."""
    u = tmp_path / "pend.urdf"
    u.write_text("""<robot name='p'>
      <link name='base'><inertial><origin xyz='0 0 0' rpy='0 0 0'/><mass value='2'/><inertia ixx='1' iyy='1' izz='1'/></inertial></link>
      <link name='arm'><inertial><origin xyz='0 0 -0.5' rpy='0 0 0'/><mass value='1'/><inertia ixx='0.1' iyy='0.1' izz='0.01'/></inertial></link>
      <link name='tip'><inertial><origin xyz='0 0 -0.1' rpy='0 0 0'/><mass value='0.5'/><inertia ixx='0.01' iyy='0.01' izz='0.01'/></inertial></link>
      <joint name='hinge' type='revolute'><parent link='base'/><child link='arm'/><origin xyz='0 0 0' rpy='0 0 0'/><axis xyz='0 1 0'/><limit lower='-1' upper='1'/></joint>
      <joint name='weld' type='fixed'><parent link='arm'/><child link='tip'/><origin xyz='0 0 -1' rpy='0 0 0'/></joint>
    </robot>""")
    m = capi.Model(str(u))
    assert m.num_bodies == 2 and m.num_joints == 1 and m.joint_names == ["hinge"]
    np.testing.assert_allclose(m.array("mass"), [2.0, 1.5])
    np.testing.assert_allclose(m.array("com")[3:], [0, 0, (1 * -0.5 + 0.5 * -1.1) / 1.5])


def test_batch_create_without_gpu_fails_loudly(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = capi.Model(ASSET_URDF)
    with pytest.raises(capi.TrexError) as e:
        capi.Batch(m, 4)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_primitive_fitting_known_answers_and_oracle_agreement(capi, model):
    """SURVEY 8f-2. (a) the reference's own known answers (tools/mesh_primitives_test.py:9-23): a 10^3 grid
    splits into 8 octants and, at max_radius 1.0 with the reference's 100-point octant threshold, into 8
    geometries. (b) the C++ fitter (product) and the numpy fitter (oracle) produce the same end spheres."""
    from oracle import trex_model as tm
    sp = np.linspace(-1.0, 1.0, 10)
    x, y, z = np.meshgrid(sp, sp, sp)
    grid = np.stack([x.ravel(), y.ravel(), z.ravel()], 1)
    assert len(tm.fit_primitives(grid, 1.0, 4, 100)) == 8
    assert len(tm.fit_primitives(grid, 1e9, 4, 100)) == 1
    om = tm.use_primitive_collision(model, 0.2, 3, 4)
    m = capi.Model(ASSET_URDF)
    np.testing.assert_array_equal(m.array("hull_group_start"), model["hull_group_start"])
    assert len(m.array("hull_group_start")) == 29          # 28 convex hulls
    m.use_primitive_collision(0.2, 3, 4)
    np.testing.assert_array_equal(m.array("hull_start"), om["hull_start"])
    np.testing.assert_allclose(m.array("hull_xyz").reshape(-1, 3), om["hull_xyz"], atol=1e-9)
    np.testing.assert_allclose(m.array("hull_radius"), om["hull_radius"], atol=1e-9)
    np.testing.assert_allclose(m.array("sphere_radius"), om["sphere_radius"], atol=1e-9)
    assert len(om["hull_xyz"]) == 148 and om["hull_radius"].min() > 0.005
    # the capsules follow their hull: no hull vertex is farther than 12 cm from the primitive surface (the
    # reference's rule - radius from the minor box extents, length = major - 2 r - rounds corners off)
    for b in range(model["nb"]):
        v = model["hull_xyz"][model["hull_start"][b]:model["hull_start"][b + 1]]
        gs = model["hull_group_start"]
        for g in range(len(gs) - 1):
            if not (model["hull_start"][b] <= gs[g] < model["hull_start"][b + 1]):
                continue
            pts = model["hull_xyz"][gs[g]:gs[g + 1]]
            prims = tm.fit_primitives(pts, 0.2, 3, 4)
            def dist(p):
                best = 1e9
                for p0, p1, r in prims:
                    d = p1 - p0
                    t = 0.0 if d @ d == 0 else np.clip((p - p0) @ d / (d @ d), 0, 1)
                    best = min(best, np.linalg.norm(p - (p0 + t * d)) - r)
                return best
            assert max(dist(p) for p in pts) < 0.12
    with pytest.raises(capi.TrexError):
        m.use_primitive_collision(-1.0)


def test_visual_mesh_table_matches_the_reference_parser(capi):
    """trex_model_visual_info: the 252 <visual> meshes as the reference's parser reads them (UrdfLink.visual_shapes,
    tools/urdf_parsing.py:93-120,299-307; fixture written by scripts/make_golden.py importing that parser): file name,
    link, <origin> as position + quaternion. Host-only: no GPU."""
    import json
    from oracle import trex_model as tm
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "urdf_reference.json")))["visuals"]
    m = capi.Model()
    links = [n for n, _ in m.links()]
    table = m.visuals()
    assert len(table) == len(gold) == 252
    for (file, link, xyz, quat), g in zip(table, gold):
        assert file == g["file"] and links[link] == g["link"]
        np.testing.assert_allclose(xyz, g["origin"]["xyz"], atol=1e-12)
        np.testing.assert_allclose(tm.quat_to_matrix(quat), tm.quat_to_matrix(g["origin"]["quat_xyzw"]), atol=1e-12)
    name, link = capi.C.c_char_p(), capi.C.c_int()
    assert capi.lib.trex_model_visual_info(m.h, 252, capi.C.byref(name), capi.C.byref(link), None, None) == capi.E_INVALID


def test_policy_header_symbols_exported(capi):
    """include/trex_policy.h (the trainer-side kernels): every declared function is exported by libtrex_hip.so."""
    import re
    header = open(os.path.join(ROOT, "include", "trex_policy.h")).read()
    declared = set(re.findall(r"\b(trex_policy_\w+)\s*\(", header))
    assert declared == set(capi.POLICY_SYMBOLS)
    for name in declared:
        assert hasattr(capi.lib, name), name
