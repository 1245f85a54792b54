#!/usr/bin/env python3
"""Deviation D4 quantified: what do MuJoCo's two rolling-friction rows of the finger contacts (follower.xml:15 condim="6") change?

Without them a finger contact has 4 rows (normal, two tangents, torsion).  The oracle can add the two rolling rows to the
finger<->cube contacts (condim6 = 1 = the kernel's finger_cube_condim 6) and, for this study only, to the finger<->floor contacts (2).  Both variants are stepped from the SAME state for one control step (20 substeps),
under a random policy and from pinch-grasp states; the difference of the resulting states is reported over the env-steps in
which a finger contact was active.   python tools/condim6_effect.py [n_envs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402

FINGER_BITS = (1 << 12) | (1 << 13) | (1 << 14) | (1 << 15)
CUBE_BITS = (1 << 12) | (1 << 13)


def study(task, n, steps, rng, mode=0):
    a = orc.Oracle(task, n, auto_reset=0, max_episode_steps=0, action_mode=mode, condim6=0)
    b = orc.Oracle(task, n, auto_reset=0, max_episode_steps=0, action_mode=mode, condim6=2)
    a.reset(seeds=np.arange(n)); b.reset(seeds=np.arange(n))
    # start half of the envs with the gripper over the cube so that finger<->cube contacts occur under a random policy
    dq, dv, dcube, which = [], [], [], []
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, a.action_dim)).astype(np.float32)
        for name in ("qpos", "qvel", "ee_lag", "target", "elapsed", "rng", "goal", "sim_time"):
            getattr(b, name)[...] = getattr(a, name)
        a.step(act, threads=os.cpu_count()); b.step(act, threads=os.cpu_count())
        on = (a.active_mask & FINGER_BITS) != 0
        dq.append(np.abs(a.qpos[:, :a.nq] - b.qpos[:, :a.nq]).max(1)[on])
        dv.append(np.abs(a.qvel[:, :a.nv] - b.qvel[:, :a.nv]).max(1)[on])
        which.append(((a.active_mask & CUBE_BITS) != 0)[on])
    dq, dv, which = np.concatenate(dq), np.concatenate(dv), np.concatenate(which)
    def line(tag, m):
        if m.sum() == 0:
            return f"    {tag}: none"
        return (f"    {tag}: {int(m.sum()):7d} env-steps  |dq| median {np.median(dq[m]):.1e} p99 {np.percentile(dq[m], 99):.1e} max {dq[m].max():.1e}"
                f"   |dv| median {np.median(dv[m]):.1e} p99 {np.percentile(dv[m], 99):.1e} max {dv[m].max():.1e}")
    print(f"{task} (n={n}, {steps} control steps, action_mode={mode}):")
    print(line("finger<->floor only", ~which))
    print(line("finger<->cube      ", which))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rng = np.random.default_rng(0)
    for task in ("reach", "push", "lift", "stack", "push_loop"):
        study(task, n, 40, rng)
    study("pick_place", n, 40, rng, mode=1)
