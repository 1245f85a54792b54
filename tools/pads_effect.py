#!/usr/bin/env python3
"""Deviation D3 narrowed and quantified: what do the finger pads as BOXES (preset faithful, orc_params.finger_geom = 1: bounding boxes of the two outermost slabs
of the fingers' collision hulls, follower.xml:15,89,97) change against the inscribed SPHERES of rounds 1-4 (finger_geom = 0)?

Both variants are stepped from the SAME state (a random-policy walk made with the boxes) for one control step; reported: how often a finger contact is active under
either geometry, and the difference of the resulting states over the env-steps in which one is.  Then the grasp: a cube pinched in mid-air (tests/util.pinch_setup)
is held for 25 control steps while the arm lifts it 5 cm -- with boxes and with spheres.      python tools/pads_effect.py [n_envs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from tests import util  # noqa: E402

FLOOR_BITS = (1 << 14) | (1 << 15)
CUBE_BITS = (1 << 12) | (1 << 13)
STATE = ("qpos", "qvel", "ee_lag", "target", "elapsed", "rng", "goal", "sim_time", "warm")


def study(task, n, steps, rng, mode=0):
    kw = dict(auto_reset=0, max_episode_steps=0, action_mode=mode)
    box = orc.Oracle(task, n, **kw)
    sph = orc.Oracle(task, n, finger_geom=0, **kw)
    walk = orc.Oracle(task, n, **kw)
    walk.reset(seeds=np.arange(n))
    dq, onb, ons, cub = [], [], [], []
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, walk.action_dim)).astype(np.float32)
        for o in (box, sph):
            for k in STATE:
                getattr(o, k)[...] = getattr(walk, k)
            o.step(act, 0)
        fb, fs = (box.active_mask & (FLOOR_BITS | CUBE_BITS)) != 0, (sph.active_mask & (FLOOR_BITS | CUBE_BITS)) != 0
        onb.append(fb); ons.append(fs); cub.append(((box.active_mask | sph.active_mask) & CUBE_BITS) != 0)
        dq.append(np.abs(box.qpos[:, : box.nq] - sph.qpos[:, : box.nq]).max(1))
        walk.step(act, 0)
    dq, onb, ons, cub = (np.concatenate(x) for x in (dq, onb, ons, cub))
    m = onb | ons
    print(f"{task:10s} finger contact active in {100 * onb.mean():5.2f} % of the env-steps with boxes, {100 * ons.mean():5.2f} % with spheres (on a cube: {100 * cub.mean():.3f} %);"
          f" |dqpos| boxes vs spheres where either is: median {np.median(dq[m]):.1e} p90 {np.percentile(dq[m], 90):.1e} p99 {np.percentile(dq[m], 99):.1e}"
          f" | where neither is: max {dq[~m].max() if (~m).any() else 0:.1e}", flush=True)


def grasp(geom):
    """pinch, squeeze, lift 5 cm (shoulder lift joint back by small steps), hold: returns the cube's final height above its start and its slip against the fingers"""
    o = orc.Oracle("lift", 16, auto_reset=0, max_episode_steps=0, finger_geom=geom)
    o.reset(seeds=np.arange(16))
    util.pinch_setup(o)
    z0 = o.qpos[:, 8].copy()
    rel0 = None
    for t in range(40):
        a = np.zeros((16, 6), np.float32)
        a[:, 5] = 0.2                      # keep squeezing
        if 5 <= t < 30:
            a[:, 1] = 0.02                 # raise the upper arm: the gripper goes up
        o.step(a, 0)
        mid = np.array([0.5 * (orc.fk(q)[2][0] + orc.fk(q)[2][1]) for q in o.qpos[:, :6]])
        rel = o.qpos[:, 6:9] - mid
        if t == 4:
            rel0 = rel.copy()
    return o.qpos[:, 8] - z0, np.abs(rel - rel0).max(1), ((o.active_mask >> 12) & 3) == 3


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    rng = np.random.default_rng(0)
    for task, mode in (("reach", 0), ("lift", 0), ("pick_place", 1), ("stack", 0), ("push_loop", 0)):
        study(task, n, 30, rng, mode)
    for geom, name in ((1, "boxes"), (0, "spheres")):
        dz, slip, both = grasp(geom)
        print(f"grasp, {name:7s}: cube raised by {1e3 * dz.min():.1f} .. {1e3 * dz.max():.1f} mm in 40 control steps, slip against the fingers {1e3 * slip.max():.2f} mm, "
              f"both finger<->cube contacts still active in {int(both.sum())} of 16 envs", flush=True)
