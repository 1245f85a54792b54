"""Accuracy of the default contact solve (4 warm-started PGS sweeps, deviation D1) against a converged solve (300 cold sweeps)
of the same model, measured with the CPU oracle on contact-rich random rollouts.  Both start every control step from the same
state; the difference after ONE control step (20 substeps) is reported.   python tools/solver_accuracy.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc

n, steps = 512, 50
for task in ("push", "lift", "stack"):
    a = orc.Oracle(task, n, pgs_iters=4, warm_start=1, auto_reset=0, max_episode_steps=0)
    b = orc.Oracle(task, n, pgs_iters=300, warm_start=0, auto_reset=0, max_episode_steps=0)
    seeds = np.arange(n, dtype=np.uint64) + 1000
    a.reset(seeds); b.reset(seeds)
    rng = np.random.default_rng(3)
    dq, dv, dc = [], [], []
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, a.action_dim)).astype(np.float32)
        a.qpos[:] = b.qpos; a.qvel[:] = b.qvel; a.ee_lag[:] = b.ee_lag      # same start state
        a.step(act, 0); b.step(act, 0)
        dq.append(np.abs(a.qpos[:, :6] - b.qpos[:, :6]).max(1)); dv.append(np.abs(a.qvel[:, :6] - b.qvel[:, :6]).max(1))
        dc.append(np.linalg.norm(a.qpos[:, 6:9] - b.qpos[:, 6:9], axis=1))
    dq, dv, dc = np.concatenate(dq), np.concatenate(dv), np.concatenate(dc)
    pct = lambda x: "median %.1e  95%% %.1e  99.9%% %.1e  max %.1e" % (np.median(x), np.percentile(x, 95), np.percentile(x, 99.9), x.max())
    print(f"{task:6s} arm |dq| rad: {pct(dq)}\n       arm |dqvel| rad/s: {pct(dv)}\n       cube |dpos| m: {pct(dc)}")
