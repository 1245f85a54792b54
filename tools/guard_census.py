"""How often does the divergence guard end an episode?  (GPU box)  python tools/guard_census.py [steps]
An episode truncated before the 50-step TimeLimit can only come from the guard (state beyond 2^34 or non-finite)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from gym_lowcostrobot_amd import VecSim

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536     # (32 768: the two-wave kernels; 65 536: the one-wave kernels)
for task, mode in (("reach", "joint"), ("push", "joint"), ("lift", "joint"), ("pick_place", "ee"), ("stack", "joint"), ("push_loop", "joint")):
    sim = VecSim(task, n, action_mode=mode)
    act = sim.alloc_actions()
    age = torch.zeros(n, dtype=torch.int32, device="cuda")
    early = 0
    trunc, term, dres = sim.truncated.torch(), sim.terminated.torch(), sim.did_reset.torch()
    for t in range(steps):
        sim.fill_random_actions(act, 11, t); sim.step_device(act.ptr); sim.sync()
        age += 1
        g = (trunc != 0) & (term == 0) & (age < 50)
        early += int(g.sum())
        age[dres != 0] = 0
    print(f"{task:10s} {mode:5s} {sim.step_kernel_name:17s} env-steps {n * steps:.2e}  guard-ended episodes {early}", flush=True)
    sim.close()
