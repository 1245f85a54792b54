"""How deep do the soft contacts get under the benchmark's random policy?  (GPU box)   python tools/penetration_census.py [steps]
65 536 envs per task on the HIP path; every 5th step 2 048 sampled envs are evaluated on the CPU (oracle forward kinematics, numpy
geometry; measurement only): finger spheres vs floor and vs cube(s), cube vertices vs floor, cube vs cube (Stack), joints beyond their range."""
import sys

import numpy as np

sys.path.insert(0, ".")
from gym_lowcostrobot_amd import VecSim  # noqa: E402
from oracle import orc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n, sample, H = 65536, 2048, 0.015
T = orc.model_table()
srad = [s["rad"] for s in T["spheres"]]
lo = np.array([l["range"][0] for l in T["links"]]); hi = np.array([l["range"][1] for l in T["links"]])


def rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def sphere_box(c, r, p, R):
    d = np.abs(R.T @ (c - p)) - H
    return (np.linalg.norm(np.maximum(d, 0)) if d.max() > 0 else d.max()) - r


def box_box(p0, R0, p1, R1):   # overlap depth along the best face axis (negative = penetration), face axes only
    best = -1e9
    for R, sign in ((R0, 1), (R1, -1)):
        for k in range(3):
            ax = R[:, k]
            ext = H + H * (np.abs(ax @ (R1 if sign > 0 else R0))).sum()
            best = max(best, abs(ax @ (p1 - p0)) - ext)
    return best


def pct(a, thr):
    a = np.asarray(a)
    return " ".join(f"<{1e3 * t:g}mm {100 * (a < t).mean():.3f}%" for t in thr)


for task, mode in (("reach", "joint"), ("push", "joint"), ("lift", "joint"), ("pick_place", "ee"), ("stack", "joint"), ("push_loop", "joint")):
    sim = VecSim(task, n, action_mode=mode, base_seed=11)
    act = sim.alloc_actions()
    nc = 2 if task == "stack" else 1
    ff, fc, cf, cc, jl = [], [], [], [], []
    for t in range(steps):
        sim.fill_random_actions(act, 5, t)
        sim.step_device(act.ptr)
        if t % 5 != 4:
            continue
        st = sim.get_state()
        qp = st["qpos"][:, :sample].T
        for e in range(sample):
            q = qp[e, :6]
            _, _, sph = orc.fk(q)
            jl.append(max((lo - q).max(), (q - hi).max()))
            cubes = [(qp[e, 6 + 7 * c:9 + 7 * c], rot(qp[e, 9 + 7 * c:13 + 7 * c])) for c in range(nc)]
            for s in range(2):
                ff.append(sph[s][2] - srad[s])
                fc.append(min(sphere_box(sph[s], srad[s], p, R) for p, R in cubes))
            for p, R in cubes:
                cf.append(p[2] - H * np.abs(R[2]).sum())
            if nc == 2:
                cc.append(box_box(cubes[0][0], cubes[0][1], cubes[1][0], cubes[1][1]))
    thr = (-0.0005, -0.002, -0.005, -0.015)
    print(f"{task:10s} {mode:5s}: finger<->floor deepest {1e3 * -min(ff):.1f} mm ({pct(ff, thr)}) | finger<->cube deepest {1e3 * -min(fc):.1f} mm ({pct(fc, thr)}) | "
          f"cube<->floor deepest {1e3 * -min(cf):.1f} mm ({pct(cf, thr)})" + (f" | cube<->cube deepest {1e3 * -min(cc):.1f} mm ({pct(cc, thr)})" if cc else "")
          + f" | joint beyond range: worst {max(jl):.3f} rad, > 0.02 rad in {100 * (np.array(jl) > 0.02).mean():.3f} %", flush=True)
    sim.close()
