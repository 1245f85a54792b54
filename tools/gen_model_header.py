#!/usr/bin/env python3
"""Generate gym_lowcostrobot_amd/csrc/lcr_model_gen.h : compile-time constants of the arm + scenes.

Inputs are the MJCF NUMBERS of the reference model (typed below with file:line citations, relative to
/root/reference/gym_lowcostrobot/assets/low_cost_robot_6dof/); everything derived from them (unit
quaternions, body-frame inertia tensors, qpos0 inverse weights used by the constraint regulariser) is
computed here in float64 numpy -- an implementation independent of both the oracle (C) and the kernel
(HIP).  tests/test_model_constants.py checks the three against each other.

    python tools/gen_model_header.py            # rewrites the header
"""
import os

import numpy as np

# ---- inputs: tests/golden/model_golden.json, the MJCF numbers extracted from the reference's follower.xml + scene files by
# tests/golden/make_golden.py:mk_model (follower.xml:56-98 body pos / joint axis / inertial pos, quat, mass, diaginertia;
# :91 site; :7 armature; scene files: cube mass / inertia / friction, rails, cameras).  Nothing is typed by hand here.
import json

_G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "model_golden.json")))
_B = {b["name"]: b for b in _G["follower"]["bodies"]}
_LINKS = [_B[f"link_{i}"] for i in range(1, 7)]
LINK_POS = [tuple(b["pos"]) for b in _LINKS]
LINK_AXIS = [tuple(b["joints"][0]["axis"]) for b in _LINKS]
LINK_RANGE = [tuple(b["joints"][0]["range"]) for b in _LINKS]
LINK_IPOS = [tuple(b["inertial"]["pos"]) for b in _LINKS]
LINK_IQUAT = [tuple(b["inertial"]["quat"]) for b in _LINKS]
LINK_MASS = [b["inertial"]["mass"] for b in _LINKS]
LINK_DIAGI = [tuple(b["inertial"]["diaginertia"]) for b in _LINKS]
BASE_QUAT = tuple(_B["base_link"]["quat"])  # follower.xml:51
SITE_POS = tuple(_B["link_5"]["sites"][0]["pos"])  # follower.xml:91 (on link_5)
ARMATURE = _G["follower"]["defaults"]["follower"]["joint"]["armature"]  # follower.xml:7
SCENE_OF_TASK = ["reach_cube", "lift_cube", "push_cube", "pick_place_cube", "stack_two_cubes", "push_cube_loop"]  # lcr_task order


def scene_cube(task):
    """(mass, inertia, (mu_tan, mu_tors, mu_roll), half size) of the (first) cube of a scene; geom friction defaults (1, 0.005, 0.0001) (MJ-DOC)"""
    b = [x for x in _G["scenes"][SCENE_OF_TASK[task]]["bodies"] if x["joints"] and x["joints"][0].get("type") == "free"][0]
    fr = b["geoms"][0].get("friction", 1.0)
    fr = list(fr) if isinstance(fr, list) else [fr]
    fr = fr + [1.0, 0.005, 0.0001][len(fr):]
    return b["inertial"]["mass"], b["inertial"]["diaginertia"][0], (fr[0], fr[1], fr[2]), b["geoms"][0]["size"][0]


# finger proxies (deviation D3): one sphere per finger geom, fitted to the tip of the fixed finger of the
# link_5_collision hull and to the jaw tip of the link_6_collision hull (follower.xml:89,97)
SPH_LINK = [4, 5]
SPH_POS = [(-0.0610, 0.0142, 0.0005), (-0.0490, 0.0072, -0.0140)]
SPH_RAD = [0.0065, 0.0065]
# arm-link proxies (deviation D3): spheres inscribed in the hulls of link_3 (both ends), link_4 (motor), link_5 (motor body),
# link_6 (jaw root); extents: model_golden.json "mesh_slabs_x".  All collide with the floor, the gripper-body ones (LPX_CUBE)
# also with the cube(s); together they yield one contact.  Same table as oracle/lcr_oracle.c LPX_*.
LPX_LINK = [2, 2, 3, 4, 5]
LPX_POS = [(-0.0100, 0.0145, 0.0030), (-0.0950, 0.0145, 0.0030), (-0.0320, 0.0206, 0.0000), (-0.0130, 0.0015, 0.0000),
           (-0.0120, 0.0000, -0.0145)]
LPX_RAD = [0.0120, 0.0120, 0.0105, 0.0150, 0.0078]
LPX_CUBE = [0, 0, 0, 1, 1]


def pad_boxes(golden=None):
    """[(centre, half extents)] of the two finger pad boxes in their link frames, from the hull slabs of the golden model file"""
    import json
    import os

    if golden is None:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "model_golden.json")) as fh:
            golden = json.load(fh)
    out = []
    for name in ("link_5_collision", "link_6_collision"):
        sl = golden["mesh_slabs_x"][name][:2]
        lo = [min(s_[a][0] for s_ in sl) for a in "xyz"]
        hi = [max(s_[a][1] for s_ in sl) for a in "xyz"]
        out.append(([0.5 * (a + b) for a, b in zip(lo, hi)], [0.5 * (b - a) for a, b in zip(lo, hi)]))
    return out


def arm_boxes(golden=None):
    """[(centre, half extents)] of the collision-hull bounding boxes of base_link, link_1 .. link_6, body frames"""
    import json
    import os

    if golden is None:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "model_golden.json")) as fh:
            golden = json.load(fh)
    out = []
    for name in ("base_link_collision", "link_1_collision", "link_2_collision", "link_3_collision", "link_4_collision", "link_5_collision", "link_6_collision"):
        lo, hi = golden["mesh_aabb"][name]["min"], golden["mesh_aabb"][name]["max"]
        out.append(([0.5 * (a + b) for a, b in zip(lo, hi)], [0.5 * (b - a) for a, b in zip(lo, hi)]))
    return out


def quat2mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot(axis, th):
    a = np.asarray(axis, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def fk(q):
    R = quat2mat(BASE_QUAT)  # follower.xml:51
    p = np.zeros(3)
    out = []
    for i in range(6):
        p = p + R @ np.array(LINK_POS[i])
        R = R @ rot(LINK_AXIS[i], q[i])
        out.append((R.copy(), p.copy()))
    return out


def mass_matrix(q):
    frames = fk(q)
    M = np.zeros((6, 6))
    z = [frames[j][0] @ np.array(LINK_AXIS[j], float) for j in range(6)]
    for i in range(6):
        R, p = frames[i]
        c = p + R @ np.array(LINK_IPOS[i])
        Rw = R @ quat2mat(LINK_IQUAT[i])
        Iw = Rw @ np.diag(LINK_DIAGI[i]) @ Rw.T
        Jv = np.zeros((3, 6))
        Jw = np.zeros((3, 6))
        for j in range(i + 1):
            Jv[:, j] = np.cross(z[j], c - frames[j][1])
            Jw[:, j] = z[j]
        M += LINK_MASS[i] * Jv.T @ Jv + Jw.T @ Iw @ Jw
    return M + ARMATURE * np.eye(6), frames, z


def invweight0():
    q = np.zeros(6)
    M, frames, z = mass_matrix(q)
    Mi = np.linalg.inv(M)
    tran, rotw = [], []
    for i in range(6):
        R, p = frames[i]
        c = p + R @ np.array(LINK_IPOS[i])
        Jv = np.zeros((3, 6))
        Jw = np.zeros((3, 6))
        for j in range(i + 1):
            Jv[:, j] = np.cross(z[j], c - frames[j][1])
            Jw[:, j] = z[j]
        tran.append(np.trace(Jv @ Mi @ Jv.T) / 3)
        rotw.append(np.trace(Jw @ Mi @ Jw.T) / 3)
    return np.array(tran), np.array(rotw), np.diag(Mi).copy()


def body_inertia(i):
    Rq = quat2mat(LINK_IQUAT[i])
    return Rq @ np.diag(LINK_DIAGI[i]) @ Rq.T


def f(x):
    return repr(float(np.float32(x))) + "f" if x != 0 else "0.0f"


def main():
    tran, rotw, dof = invweight0()
    L = []
    L.append("// GENERATED by tools/gen_model_header.py -- do not edit.  Numbers: MJCF of the reference model")
    L.append("// (follower.xml:3-118, scene xmls) and quantities derived from them in float64.")
    L.append("#pragma once")
    L.append("namespace lcrm {")
    for i in range(6):
        n = i + 1
        L.append(f"// link_{n}: follower.xml body pos / inertial")
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float P{n}{ax} = {f(LINK_POS[i][k])};")
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float C{n}{ax} = {f(LINK_IPOS[i][k])};")
        L.append(f"constexpr float M{n} = {f(LINK_MASS[i])};")
        I = body_inertia(i)
        for (a, b) in [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]:
            L.append(f"constexpr float I{n}_{'xyz'[a]}{'xyz'[b]} = {f(I[a, b])};")
    for k, ax in enumerate("xyz"):
        L.append(f"constexpr float SITE{ax} = {f(SITE_POS[k])};")
    for s in range(2):
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float SPH{s}{ax} = {f(SPH_POS[s][k])};")
        L.append(f"constexpr float SPH{s}r = {f(SPH_RAD[s])};")
    # the renderers draw the arm as the bounding boxes of its seven collision hulls (model_golden.json "mesh_aabb", body frames) instead of capsules between the link origins
    L.append("// bounding boxes of the collision hulls of base_link (0), link_1 .. link_6 (1 .. 6) in their body frames (centre, half extents): what lcr_render.hip / oracle/render_oracle.py draw for the arm")
    for i, (c, h) in enumerate(arm_boxes()):
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float ARMB{i}c{ax} = {f(c[k])};")
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float ARMB{i}h{ax} = {f(h[k])};")
    # finger pads as boxes (the faithful preset): bounding box of the two outermost slabs of each finger's collision hull, link frame (centre, half extents)
    L.append("// finger pad boxes (preset faithful; oracle PAD_C / PAD_H): the two outermost slabs of link_5_collision / link_6_collision (model_golden.json mesh_slabs_x)")
    for s, (c, h) in enumerate(pad_boxes()):
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float PAD{s}c{ax} = {f(c[k])};")
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float PAD{s}h{ax} = {f(h[k])};")
    for s in range(len(LPX_LINK)):
        for k, ax in enumerate("xyz"):
            L.append(f"constexpr float LPX{s}{ax} = {f(LPX_POS[s][k])};")
        L.append(f"constexpr float LPX{s}r = {f(LPX_RAD[s])};")
    L.append("// qpos0 inverse weights (MuJoCo body_invweight0[.,0] of the links, dof_invweight0)")
    for i in range(6):
        L.append(f"constexpr float INVW_TRAN_L{i + 1} = {f(tran[i])};")
        L.append(f"constexpr float INVW_ROT_L{i + 1} = {f(rotw[i])};")
    for j in range(6):
        L.append(f"constexpr float INVW_DOF{j + 1} = {f(dof[j])};")
    L.append("// joint ranges (follower.xml:58-95) == actuator ctrlrange through inheritrange")
    L.append("constexpr float JNT_LO[6] = {" + ", ".join(f(r[0]) for r in LINK_RANGE) + "};")
    L.append("constexpr float JNT_HI[6] = {" + ", ".join(f(r[1]) for r in LINK_RANGE) + "};")
    L.append("// scene constants by lcr_task (reach, lift, push, pick_place, stack, push_loop): cube mass, inertia, friction (double: host side)")
    cubes = [scene_cube(t) for t in range(6)]
    L.append("constexpr double SCENE_CUBE_MASS[6] = {" + ", ".join(repr(float(c[0])) for c in cubes) + "};")
    L.append("constexpr double SCENE_CUBE_INERTIA[6] = {" + ", ".join(repr(float(c[1])) for c in cubes) + "};")
    L.append("constexpr double SCENE_CUBE_MU[6] = {" + ", ".join(repr(float(c[2][0])) for c in cubes) + "};")
    L.append("constexpr double SCENE_CUBE_MU_TORS[6] = {" + ", ".join(repr(float(c[2][1])) for c in cubes) + "};")
    L.append("constexpr double SCENE_CUBE_MU_ROLL[6] = {" + ", ".join(repr(float(c[2][2])) for c in cubes) + "};")
    L.append(f"constexpr float CUBE_HALF = {f(cubes[0][3])};")
    L.append("}  // namespace lcrm")
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gym_lowcostrobot_amd", "csrc", "lcr_model_gen.h")
    with open(dst, "w") as fh:
        fh.write("\n".join(L) + "\n")
    print("wrote", os.path.normpath(dst))
    print("invweight tran", tran, "\nrot", rotw, "\ndof", dof)


if __name__ == "__main__":
    main()
