"""One-off large check (GPU box): the batched observation renderer (culled 16x4 tiles over a cached background) against the
plain per-pixel ray-caster of lcr_render on 6 144 frames; prints the number of pixels that differ by more than 2 levels."""
import sys, numpy as np
sys.path.insert(0, ".")
from gym_lowcostrobot_amd import VecSim
worst = 0; tot_bad = 0; tot = 0
for task in ("push", "stack", "pick_place", "lift"):
    n = 256
    sim = VecSim(task, n, observation_mode="both", base_seed=123)
    rng = np.random.default_rng(9)
    for rounds in range(3):
        for _ in range(9):
            sim.step(rng.uniform(-1, 1, (n, sim.action_dim)).astype(np.float32))
        obs = sim.observations()
        for e in range(n):
            for name, key in (("camera_front", "image_front"), ("camera_top", "image_top")):
                ref = sim.render(e, name, 320, 240).astype(int)
                d = np.abs(obs[key][e].astype(int) - ref).max(-1)
                bad = int((d > 2).sum()); tot_bad += bad; tot += d.size; worst = max(worst, bad)
    sim.close()
print("frames checked", tot // 76800, "mismatching pixels (>2 levels):", tot_bad, "of", tot, "worst frame", worst)
