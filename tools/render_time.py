"""Time of the observation-frame kernel alone (GPU box): masked no-op reset -> only the re-render of all frames runs.
python tools/render_time.py [n_envs] [task]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from gym_lowcostrobot_amd import VecSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
task = sys.argv[2] if len(sys.argv) > 2 else "stack"
sim = VecSim(task, n, observation_mode="both")
act = sim.alloc_actions()
for t in range(12):          # move the arms / cubes away from the reset pose
    sim.fill_random_actions(act, 1, t)
    sim.step_device(act.ptr)
mask = np.zeros(n, np.uint8)
for _ in range(3):
    sim.reset(mask=mask)
ms = []
for r in range(5):
    sim.timer_begin()
    for _ in range(10):
        sim.reset(mask=mask)
    ms.append(sim.timer_end() / 10)
b = 2 * 240 * 320 * 3 * n
print(f"render {task} n={n}: {np.median(ms):.3f} ms per frame pair set = {b / np.median(ms) / 1e9:.2f} TB/s written ({b / 1e9:.2f} GB)")
sim.close()
