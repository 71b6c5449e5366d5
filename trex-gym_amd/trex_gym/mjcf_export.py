"""MJCF (MuJoCo XML) export of the compiled model - the wire format the reference's author was moving
to (tools/mujoco_parsing.py:11-45 is an unfinished stub of exactly this: preamble with <option> and
<compiler coordinate="local" angle="radian">, nested <body pos axisangle> with <joint type="hinge" axis
range>, capsule <geom>s from the fitted primitives; Pipfile:15 pins mujoco).

What is written is the REDUCED tree the kernels simulate (26 bodies for trex.urdf: fixed joints merged,
composite inertias), so a MuJoCo run of this file integrates the same rigid-body system:

  * the floating base gets a <freejoint>; its frame is the base INERTIAL frame (as pybullet poses it);
  * every other body: pos + quat of its hinge frame in the parent body frame at q = 0, one hinge joint
    with axis, range (joint limits) and damping, <inertial> with pos, mass and fullinertia about the COM;
  * collision: capsule / sphere geoms fitted per convex hull (trex_model_use_primitive_collision);
  * a <position> actuator per hinge with the env's gains would NOT reproduce Bullet's velocity-level
    motor rows, so none is emitted; `ctrlrange` of the joints is left to the user.

MuJoCo is not installed in this image, so the file cannot be cross-checked against MuJoCo here
(SURVEY 8f-4); tests/test_mjcf_export.py re-reads it and checks kinematics, masses and ranges against
the compiled model.
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

from . import _capi


def _quat_wxyz(R):
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [(R[k, j] - R[j, k]) / s, 0, 0, 0]
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q if q[0] >= 0 else [-x for x in q]


def _v(a):
    return " ".join("%.9g" % x for x in a)


def export_mjcf(model=None, path=None, primitive_max_radius=0.2, name="stan_t_rex"):
    """-> XML string (and writes `path` if given). `model`: a _capi.Model (default: the T-rex asset).
    The model's collision hulls are replaced by fitted primitives for the geoms (on a private copy)."""
    if model is None:
        model = _capi.Model()
    hulls = _capi.Model(model.urdf_path)       # private copy with the original hulls, for the geoms
    groups = hulls.array("hull_group_start").astype(int)
    hstart = hulls.array("hull_start").astype(int)
    nb = model.num_bodies
    parent = model.array("parent").astype(int)
    axis = model.array("joint_axis").reshape(nb, 3)
    jpos = model.array("joint_pos").reshape(nb, 3)
    jrot = model.array("joint_rot").reshape(nb, 3, 3)
    mass = model.array("mass")
    com = model.array("com").reshape(nb, 3)
    ine = model.array("inertia").reshape(nb, 6)
    lo, hi = model.array("q_lower"), model.array("q_upper")
    damp = model.array("joint_damping")
    obs_order = model.array("obs_order").astype(int)
    jname = {int(b): model.joint_names[k] for k, b in enumerate(obs_order)}
    body_name = {}
    for ln, b in model.links():
        body_name.setdefault(b, ln)       # first link in document order of each body... refined below
    # the head link of a body is the child link of its hinge: the link whose body-frame transform is identity
    ltf = model.array("link_tf").reshape(-1, 12)
    for k, (ln, b) in enumerate(model.links()):
        if b > 0 and np.allclose(ltf[k][:9], np.eye(3).ravel()) and np.allclose(ltf[k][9:], 0):
            body_name[b] = ln
    start_pos = model.array("base_start_pos")
    sq = model.array("base_start_quat")   # xyzw

    root = ET.Element("mujoco", {"model": name})
    ET.SubElement(root, "option", {"timestep": "%.9g" % model.get_param("dt"), "gravity": "0 0 -%.9g" % model.get_param("gravity")})
    ET.SubElement(root, "compiler", {"coordinate": "local", "angle": "radian"})
    world = ET.SubElement(root, "worldbody")
    ET.SubElement(world, "geom", {"name": "floor", "type": "plane", "pos": "0 0 %.9g" % model.get_param("floor_z"),
                                  "size": "500 500 0.1", "friction": "%.9g 0 0" % math.sqrt(model.get_param("friction"))})

    def add_geoms(elem, b):
        for g in range(len(groups) - 1):
            if not (hstart[b] <= groups[g] < hstart[b + 1]):
                continue
            for k, (p0, p1, r) in enumerate(hulls.fit_hull_primitives(g, primitive_max_radius)):
                attrs = {"name": "%s_g%02d_%02d" % (body_name[b], g, k), "friction": "%.9g 0 0" % math.sqrt(model.get_param("friction"))}
                if np.abs(p1 - p0).max() > 0:
                    attrs.update(type="capsule", fromto=_v(np.concatenate([p0, p1])), size="%.9g" % r)
                else:
                    attrs.update(type="sphere", pos=_v(p0), size="%.9g" % r)
                ET.SubElement(elem, "geom", attrs)

    elems = {}
    for b in range(nb):
        if b == 0:
            e = ET.SubElement(world, "body", {"name": body_name[0], "pos": _v(start_pos),
                                              "quat": _v([sq[3], sq[0], sq[1], sq[2]])})
            ET.SubElement(e, "freejoint", {"name": "root"})
        else:
            e = ET.SubElement(elems[parent[b]], "body", {"name": body_name[b], "pos": _v(jpos[b]), "quat": _v(_quat_wxyz(jrot[b]))})
            ET.SubElement(e, "joint", {"name": jname[b], "type": "hinge", "axis": _v(axis[b]), "range": "%.9g %.9g" % (lo[b], hi[b]),
                                       "limited": "true", "damping": "%.9g" % damp[b]})
        a = ine[b]
        ET.SubElement(e, "inertial", {"pos": _v(com[b]), "mass": "%.9g" % mass[b],
                                      "fullinertia": _v([a[0], a[3], a[5], a[1], a[2], a[4]])})   # xx yy zz xy xz yz
        add_geoms(e, b)
        elems[b] = e
    ET.indent(root, space="  ")
    text = ET.tostring(root, encoding="unicode")
    if path:
        with open(path, "w") as f:
            f.write(text + "\n")
    return text
