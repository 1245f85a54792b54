"""How deep do the arm-link proxies get below the floor under a uniformly random policy?  (GPU box)
    python tools/proxy_census.py [steps]
65 536 ReachCube envs on the HIP path; every 5th step the lowest point of the five proxy spheres of 4 096 sampled envs is
evaluated with the oracle's forward kinematics (measurement only).  Compared: arm_collision on (default) and off (round-1 model)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from gym_lowcostrobot_amd import VecSim  # noqa: E402
from oracle import orc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, sample = 65536, 4096
for on in (1, 0):
    sim = VecSim("reach", n, arm_collision=on, base_seed=5)
    act = sim.alloc_actions()
    depth = []
    for t in range(steps):
        sim.fill_random_actions(act, 3, t)
        sim.step_device(act.ptr)
        if t % 5 == 4:
            q = sim.arm_qpos.numpy()[:, :sample].T.astype(np.float64)
            for e in range(sample):
                c, r = orc.proxies(q[e])
                depth.append(float((c[:, 2] - r).min()))
    d = np.array(depth)
    print(f"arm_collision={on}: {len(d)} sampled states; lowest proxy point below the floor by > 1 mm in {100 * (d < -1e-3).mean():.2f} %, "
          f"> 5 mm in {100 * (d < -5e-3).mean():.3f} %, > 15 mm in {100 * (d < -15e-3).mean():.4f} %; deepest {1e3 * -d.min():.1f} mm", flush=True)
    sim.close()
