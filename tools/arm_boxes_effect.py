#!/usr/bin/env python3
"""What drawing link_3 .. link_6 as the bounding boxes of their collision hulls (round 5) changes in the image observations against the capsules of rounds 1-4:
fraction of the 240 x 320 pixels of camera_front / camera_top that move by more than 2 grey levels, over random arm poses (CPU ray-caster, oracle/render_oracle.py).
    python tools/arm_boxes_effect.py [n_poses]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc, render_oracle as ro  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
o = orc.Oracle("reach", n, auto_reset=0, max_episode_steps=0)
o.reset(seeds=np.arange(n))
rng = np.random.default_rng(0)
for t in range(6):   # a short random walk: poses a policy visits
    o.step(rng.uniform(-1, 1, (n, o.action_dim)).astype(np.float32), 0)
for cam in ("camera_front", "camera_top"):
    moved, arm_px = [], []
    for e in range(n):
        q = o.qpos[e, :13]
        new = ro.render("reach", q, None, cam).astype(int)
        old = ro.render("reach", q, None, cam, prims=ro.scene_capsule_arm("reach", q)).astype(int)
        bare = ro.render("reach", q, None, cam, prims=([], ro.scene("reach", q)[1][7:])).astype(int)     # no arm at all: which pixels the arm covers
        moved.append((np.abs(new - old).max(-1) > 2).mean())
        arm_px.append(((np.abs(new - bare).max(-1) > 2) | (np.abs(old - bare).max(-1) > 2)).mean())
    print(f"{cam}: {100 * np.mean(moved):.2f} % of the pixels move (max over poses {100 * np.max(moved):.2f} %); the arm covers {100 * np.mean(arm_px):.2f} % of a frame, "
          f"so {100 * np.mean(moved) / np.mean(arm_px):.0f} % of the arm's pixels", flush=True)
