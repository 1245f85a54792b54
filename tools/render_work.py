"""Ray-cast work of the frame kernel per env (GPU box): 16 x 4-pixel passes, primitive tests, pixels written -- the counting instantiation of
lcr_render_obs_kernel (LCR_RENDER_COUNT=1) adds them up in the diagnostics buffers.
    LCR_RENDER_COUNT=1 python tools/render_work.py [task]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from gym_lowcostrobot_amd import VecSim
n = 4096
task = sys.argv[1] if len(sys.argv) > 1 else "stack"
sim = VecSim(task, n, observation_mode="both", diagnostics=1)
act = sim.alloc_actions()
for t in range(12):
    sim.fill_random_actions(act, 1, t)
    sim.step_device(act.ptr)
before = [a.numpy().astype(np.int64).copy() for a in (sim.active_count, sim.choice, sim.max_sweeps)]
sim.reset(mask=np.zeros(n, np.uint8))
after = [a.numpy().astype(np.int64) for a in (sim.active_count, sim.choice, sim.max_sweeps)]
it, pt, px = [(b - a) for a, b in zip(before, after)]
print(f"{task}: passes/env {it.mean():.1f}  prim tests/env {pt.mean():.1f} ({pt.mean()/it.mean():.2f} per pass)  pixels written/env {px.mean():.0f} ({px.mean()/it.mean()/64:.2f} of the lanes)")
