"""Per-step launch times over an episode (GPU box): python tools/step_time_series.py [task] -- shows that step cost is flat over
the episode phase."""
import sys, numpy as np
sys.path.insert(0, ".")
from gym_lowcostrobot_amd import VecSim
task = sys.argv[1] if len(sys.argv) > 1 else "stack"
n = 65536
sim = VecSim(task, n)
act = sim.alloc_actions()
ms = []
for t in range(160):
    sim.fill_random_actions(act, 0, t); sim.sync()
    sim.timer_begin(); sim.step_device(act.ptr); ms.append(sim.timer_end())
ms = np.array(ms)
print(task, "mean %.3f" % ms[50:150].mean(), "by phase (steps 50..99):", np.round(ms[50:100], 2))
