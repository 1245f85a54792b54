#!/usr/bin/env python3
"""Deviation D5 quantified: how much does the number of cube<->cube contact points matter?

The kernel and the default oracle keep at most 4 points of the overlap polygon of two touching cube faces (the extremes along the
face diagonals); MuJoCo's box-box collider keeps up to 8.  The oracle can keep 8 (the extremes along the four diagonals and the four
face axes: orc_params.cc_points = 8, study only).  Blue cubes are placed on red cubes with random offsets and yaw, (a) at rest and
(b) with a sideways push; both variants step from the SAME state; the difference after one control step and the drift after 25
free-running steps are reported.   python tools/ccpoints_effect.py [n_envs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402


def place(o, rng, push):
    n = o.n
    o.reset(seeds=np.arange(n))
    o.qpos[:, 6:9] = [0.25, 0.25, 0.0149]; o.qpos[:, 9:13] = [1, 0, 0, 0]
    off = rng.uniform(-0.012, 0.012, (n, 2))
    o.qpos[:, 13] = 0.25 + off[:, 0]; o.qpos[:, 14] = 0.25 + off[:, 1]; o.qpos[:, 15] = 0.0447
    yaw = rng.uniform(-0.78, 0.78, n)
    o.qpos[:, 16] = np.cos(yaw / 2); o.qpos[:, 17:19] = 0; o.qpos[:, 19] = np.sin(yaw / 2)
    o.qvel[:] = 0
    if push:
        o.qvel[:, 12:14] = rng.normal(0, 0.15, (n, 2))    # blue cube slides / tips
        o.qvel[:, 15:18] = rng.normal(0, 1.0, (n, 3))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for push in (False, True):
        rng = np.random.default_rng(3)
        a = orc.Oracle("stack", n, auto_reset=0, max_episode_steps=0, cc_points=4)
        b = orc.Oracle("stack", n, auto_reset=0, max_episode_steps=0, cc_points=8)
        place(a, rng, push)
        for name in ("qpos", "qvel", "ee_lag", "target", "elapsed", "rng", "goal", "sim_time"):
            getattr(b, name)[...] = getattr(a, name)
        act = np.zeros((n, 6), np.float32)
        a.step(act, threads=os.cpu_count()); b.step(act, threads=os.cpu_count())
        on = (a.active_mask & 0xF00) != 0
        npts4 = np.array([bin(int(m) & 0xF00).count("1") for m in a.active_mask])
        dq = np.abs(a.qpos[:, :20] - b.qpos[:, :20]).max(1)[on]
        dv = np.abs(a.qvel[:, :18] - b.qvel[:, :18]).max(1)[on]
        print(f"{'pushed' if push else 'at rest'}: {int(on.sum())} stacks in contact (4-point manifold: mean {npts4[on].mean():.2f} points); "
              f"the 8-point manifold differs in {100.0 * (dq > 0).mean():.0f} % of them (overlap polygons with more than four vertices)")
        print(f"    one control step, same start:  |dq| median {np.median(dq):.1e} p99 {np.percentile(dq, 99):.1e} max {dq.max():.1e}"
              f"   |dv| median {np.median(dv):.1e} p99 {np.percentile(dv, 99):.1e} max {dv.max():.1e}")
        for _ in range(24):
            a.step(act, threads=os.cpu_count()); b.step(act, threads=os.cpu_count())
        d = np.linalg.norm(a.qpos[:, 13:16] - b.qpos[:, 13:16], axis=1)
        print(f"    25 free-running control steps:  blue-cube position difference median {np.median(d):.1e} p99 {np.percentile(d, 99):.1e} max {d.max():.1e};"
              f"  still stacked: {int((a.qpos[:, 15] > 0.04).sum())} (4 points) vs {int((b.qpos[:, 15] > 0.04).sum())} (8 points) of {n}")


if __name__ == "__main__":
    main()
