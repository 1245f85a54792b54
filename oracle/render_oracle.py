"""CPU ORACLE of the image observations (test infrastructure, numpy fp64): a plain per-pixel ray-caster of the same scene model
as gym_lowcostrobot_amd/csrc/lcr_render.hip, written independently of it (no tiles, no culling, no cached background, no
precomputed per-camera constants).

What it restates: get_observation()'s two 240x320 renders of `camera_front` / `camera_top` (reach_cube_env.py:288-292) and
render()'s 640x640 `camera_vizu` frame (:350-355).  MuJoCo's OpenGL renderer cannot run here and the reference holds no
frames, so -- like the HIP renderer -- this is a model of the scene, not of MuJoCo's rasteriser: **parity unpinned** for the
pixels themselves.  Pinned from the reference: the camera poses (pos / xyaxes / euler / quat of the scene files, read from
tests/golden/model_golden.json), MuJoCo's default fovy of 45 degrees, cube size and colours, target-marker geometry and
alpha, the 0.1 m checker (texrepeat 5 over 1 m... groundplane material), headlight ambient 0.3 / diffuse 0.6.

Scene model: checker floor z = 0, gradient sky, cubes as oriented boxes, target marker as a translucent box (alpha 0.3), the
arm -- since round 5 -- as the bounding boxes of its seven collision hulls (base_link, link_1 .. link_6: mesh_aabb of the golden model file, body frames;
rounds 1-4: seven capsules between the link origins, kept as scene_capsule_arm), Lambert shading with the light at the camera (MuJoCo headlight), no shadows.
"""
import json
import os

import numpy as np

from . import orc

_G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "model_golden.json")))
SCENE_OF_TASK = {"reach": "reach_cube", "lift": "lift_cube", "push": "push_cube", "pick_place": "pick_place_cube", "stack": "stack_two_cubes",
                 "push_loop": "push_cube_loop"}
CAP_RADII = [0.026, 0.022, 0.016, 0.014, 0.013, 0.0075, 0.0070]


def _quat2mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def camera(task, name):
    """(position, X, Y, Z) of a scene camera: MuJoCo cameras look along -Z, X right, Y up (MJ-DOC); orientation from
    xyaxes (Gram-Schmidt), euler (radians, xyz, here all zero) or quat"""
    rec = [c for c in _G["scenes"][SCENE_OF_TASK[task]]["cameras"] if c["name"] == name][0]
    pos = np.array(rec["pos"], float)
    if "xyaxes" in rec:
        X = np.array(rec["xyaxes"][:3], float); Y = np.array(rec["xyaxes"][3:], float)
        X /= np.linalg.norm(X); Y -= (Y @ X) * X; Y /= np.linalg.norm(Y)
    elif "quat" in rec:
        R = _quat2mat(rec["quat"]); X, Y = R[:, 0], R[:, 1]
    else:
        assert not np.any(rec.get("euler", [0, 0, 0]))
        X, Y = np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    return pos, X, Y, np.cross(X, Y)


def scene(task, qpos, target=None):
    """primitives of one env: capsules (a, b, r), boxes (centre, R, half, rgb, alpha)"""
    qpos = np.asarray(qpos, float)
    lp, _, sph = orc.fk(qpos[:6])
    pts = [np.zeros(3)] + [lp[i] for i in range(6)]
    # the arm: the bounding boxes of its seven collision hulls (model_golden.json "mesh_aabb", body frames; follower.xml:54-97) -- round 5; rounds 1-4 drew capsules
    R, p = orc.link_frames(qpos[:6])
    RB = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])     # base_link: quat (-0.707, 0, 0, 0.707) = Rz(-90 deg) at the origin (follower.xml:51)
    frames = [(RB, np.zeros(3))] + [(R[i], p[i]) for i in range(6)]
    caps, boxes = [], []
    for i, name in enumerate(("base_link_collision", "link_1_collision", "link_2_collision", "link_3_collision", "link_4_collision", "link_5_collision",
                              "link_6_collision")):
        lo, hi = np.array(_G["mesh_aabb"][name]["min"]), np.array(_G["mesh_aabb"][name]["max"])
        Rf, pf = frames[i]
        boxes.append((pf + Rf @ (0.5 * (lo + hi)), Rf, 0.5 * (hi - lo), np.full(3, 0.75 if i >= 5 else 0.8), 1.0))
    ncube = 2 if task == "stack" else 1
    for c in range(ncube):
        p = qpos[6 + 7 * c: 9 + 7 * c]
        R = _quat2mat(qpos[9 + 7 * c: 13 + 7 * c])
        boxes.append((p, R, np.full(3, 0.015), np.array([0.5, 0, 0]) if c == 0 else np.array([0, 0, 0.5]), 1.0))   # geom rgba of the scene files
    if task in ("push", "pick_place"):     # push_cube.xml:35 cylinder r 0.035 h 0.01 (drawn as its bounding box) / pick_place_cube.xml:35 box
        half = np.array([0.035, 0.035, 0.01]) if task == "push" else np.full(3, 0.015)
        boxes.append((np.asarray(target, float), np.eye(3), half, np.array([0, 0, 1.0]), 0.3))
    return caps, boxes


def scene_capsule_arm(task, qpos, target=None):
    """the rounds 1-4 scene: the whole arm as 7 capsules between the link origins / finger spheres (kept for tools/arm_boxes_effect.py: how many pixels the boxes move)"""
    qpos = np.asarray(qpos, float)
    lp, _, sph = orc.fk(qpos[:6])
    pts = [np.zeros(3)] + [lp[i] for i in range(6)]
    caps = [(pts[i], pts[i + 1], CAP_RADII[i]) for i in range(5)] + [(lp[4], sph[0], CAP_RADII[5]), (lp[5], sph[1], CAP_RADII[6])]
    _, boxes = scene(task, qpos, target)
    return caps, boxes[7:]


def render(task, qpos, target=None, cam="camera_front", W=320, H=240, prims=None):
    pos, X, Y, Z = camera(task, cam)
    caps, boxes = scene(task, qpos, target) if prims is None else prims
    s = 2.0 * np.tan(np.radians(45.0) / 2) / H
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    sx = (u + 0.5 - 0.5 * W) * s
    sy = -(v + 0.5 - 0.5 * H) * s
    rd = sx[..., None] * X + sy[..., None] * Y - Z
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    ro = pos
    tbest = np.full((H, W), 1e30)
    col = np.zeros((H, W, 3))
    nbest = np.zeros((H, W, 3)); nbest[..., 2] = 1.0
    sky = np.zeros((H, W), bool)
    # background
    down = rd[..., 2] < -1e-6
    with np.errstate(divide="ignore", invalid="ignore"):
        tf = np.where(down, -ro[2] / rd[..., 2], 1e30)
    fx = np.where(down, ro[0] + np.where(down, tf, 0.0) * rd[..., 0], 0.0)
    fy = np.where(down, ro[1] + np.where(down, tf, 0.0) * rd[..., 1], 0.0)
    cell = ((np.floor(fx * 10).astype(np.int64) + np.floor(fy * 10).astype(np.int64)) & 1).astype(bool)
    col[down & cell] = [0.2, 0.3, 0.4]; col[down & ~cell] = [0.1, 0.2, 0.3]
    tbest[down] = tf[down]
    a = np.clip(rd[..., 2] * 2, 0, 1)
    skycol = np.stack([0.15 + a * 0.15, 0.25 + a * 0.25, 0.35 + a * 0.35], -1)
    col[~down] = skycol[~down]; sky[~down] = True
    # capsules (exact ray / capsule intersection: cylinder body, then the end-cap spheres)
    for k, (ca, cb, r) in enumerate(caps):
        ba, oa = cb - ca, ro - ca
        baba, baoa, oaoa = ba @ ba, ba @ oa, oa @ oa
        bard, rdoa = rd @ ba, rd @ oa
        A = baba - bard * bard
        B = baba * rdoa - baoa * bard
        Cc = baba * oaoa - baoa * baoa - r * r * baba
        h = B * B - A * Cc
        ok = (h >= 0) & (A > 1e-12)
        with np.errstate(invalid="ignore", divide="ignore"):
            t = np.where(ok, (-B - np.sqrt(np.maximum(h, 0))) / np.where(ok, A, 1.0), -1.0)
        y = baoa + t * bard
        body = (y > 0) & (y < baba)
        oc = np.where((y <= 0)[..., None], oa, ro - cb)
        Bc = np.einsum("hwk,hwk->hw", rd, oc); C2 = np.einsum("hwk,hwk->hw", oc, oc) - r * r
        h2 = Bc * Bc - C2
        tc = np.where(h2 > 0, -Bc - np.sqrt(np.maximum(h2, 0)), -1.0)
        t = np.where(ok & ~body, tc, t)
        hit = ok & (t > 0) & (t < tbest)
        pa = ro + t[..., None] * rd - ca
        hh = np.clip((pa @ ba) / max(baba, 1e-12), 0, 1)
        n = (pa - hh[..., None] * ba) / r
        tbest = np.where(hit, t, tbest); sky &= ~hit
        nbest[hit] = n[hit]
        col[hit] = [0.75] * 3 if k >= 5 else [0.8] * 3
    # boxes: opaque cubes, translucent target marker blended over whatever is behind it
    talpha = np.zeros((H, W)); tcol = np.zeros((H, W, 3))
    for (bc, R, bh, bcol, alpha) in boxes:
        ol = R.T @ (ro - bc)
        dl = rd @ R
        dls = np.where(np.abs(dl) > 1e-9, dl, 1e-9)
        t1 = (-bh - ol) / dls; t2 = (bh - ol) / dls
        tn = np.minimum(t1, t2); tx = np.maximum(t1, t2)
        tmin = tn.max(-1); tmax = tx.min(-1)
        hit = (tmin <= tmax) & (tmin > 0) & (tmin < tbest)
        ax = np.where(tmin == tn[..., 0], 0, np.where(tmin == tn[..., 1], 1, 2))
        sign = -np.sign(np.take_along_axis(dl, ax[..., None], -1)[..., 0]); sign[sign == 0] = 1.0
        n = R.T[ax] * sign[..., None]
        if alpha < 1.0:
            lam = 0.3 + 0.6 * np.maximum(0, -np.einsum("hwk,hwk->hw", n, rd))
            tcol = np.where(hit[..., None], lam[..., None] * bcol, tcol); talpha = np.where(hit, alpha, talpha)
        else:
            tbest = np.where(hit, tmin, tbest); sky &= ~hit
            nbest[hit] = n[hit]; col[hit] = bcol
            talpha = np.where(hit, 0.0, talpha)
    lam = np.where(sky, 1.0, np.minimum(0.3 + 0.6 * np.maximum(0, -np.einsum("hwk,hwk->hw", nbest, rd)), 1.0))
    out = lam[..., None] * col
    out = np.where((talpha > 0)[..., None], talpha[..., None] * tcol + (1 - talpha[..., None]) * out, out)
    return np.clip(np.rint(out * 255.0), 0, 255).astype(np.uint8)
