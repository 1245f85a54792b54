"""PCIe-inclusive rate of the host-pointer convenience path (lcr_step_host + device->host copies of the step outputs).
Run on the GPU box:  python tools/host_path_rate.py [n_envs] [steps]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from gym_lowcostrobot_amd import VecSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sim = VecSim("reach", n)
rng = np.random.default_rng(0)
a = rng.uniform(-1, 1, (n, sim.action_dim)).astype(np.float32)
for _ in range(10):
    sim.step(a); sim.observations(); sim.outputs()
sim.sync()
t0 = time.perf_counter()
for _ in range(steps):
    sim.step(a)                 # host (N,k) actions: transpose + H2D + kernel
    o = sim.observations()      # D2H of arm_qpos, arm_qvel, cube_pos (numpy copies)
    r = sim.outputs()           # D2H of reward / terminated / truncated / is_success
dt = time.perf_counter() - t0
print(json.dumps({"path": "host pointers (PCIe-inclusive)", "n_envs": n, "steps": steps, "ms_per_step": 1e3 * dt / steps,
                  "env_steps_per_s": n * steps / dt}))
sim.close()
