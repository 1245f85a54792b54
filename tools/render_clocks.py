"""The frame kernel's time against how long the GPU has been busy (2.5 ms on a fresh box, 2.85-2.95 ms "steady" -- which is it?).  The kernel alone (masked no-op
reset -> only the re-render of all frames runs), timed in windows of ten launches: right after start-up, then again after soaks of 0.5 / 1 / 2 / 4 s of back-to-back launches.
    python tools/render_clocks.py [n_envs] [task]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from gym_lowcostrobot_amd import VecSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
task = sys.argv[2] if len(sys.argv) > 2 else "stack"
sim = VecSim(task, n, observation_mode="both")
act = sim.alloc_actions()
for t in range(12):
    sim.fill_random_actions(act, 1, t)
    sim.step_device(act.ptr)
mask = np.zeros(n, np.uint8)
b = 2 * 240 * 320 * 3 * n


def window(k=10):
    sim.timer_begin()
    for _ in range(k):
        sim.reset(mask=mask)
    return sim.timer_end() / k


sim.reset(mask=mask); sim.sync()
time.sleep(1.0)   # idle: clocks down
print(f"frame kernel, {task}, {n} envs, {b / 1e9:.2f} GB written per launch")
w = [window() for _ in range(3)]
print(f"  after 1 s idle, first three windows of 10 launches: {' '.join(f'{x:.3f}' for x in w)} ms  ({b / w[0] / 1e9:.2f} TB/s in the first)")
busy = 0.0
for soak in (0.5, 1.0, 2.0, 4.0):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < soak - busy:
        for _ in range(20):
            sim.reset(mask=mask)
        sim.sync()
    busy = soak
    w = [window() for _ in range(5)]
    print(f"  after {soak:3.1f} s of back-to-back launches: median {np.median(w):.3f} ms (min {min(w):.3f}, max {max(w):.3f}) = {b / np.median(w) / 1e9:.2f} TB/s")
sim.close()
