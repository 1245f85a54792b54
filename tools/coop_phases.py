"""Per-wave cycle accounting of the two-cooperating-waves step kernel (lcr_config.diagnostics = 3), GPU box:
    python tools/coop_phases.py [task] [n] [family]
arm wave: total / waiting at barriers / up to barrier 1 (dynamics + row set-up); cube wave: total / waiting; coupled substeps."""
import os
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "reach"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
os.environ["LCR_STEP_KERNEL"] = sys.argv[3] if len(sys.argv) > 3 else "coop1"
mode = "ee" if task.endswith("_ee") else "joint"
task = task.replace("_ee", "")
from gym_lowcostrobot_amd import VecSim  # noqa: E402

sim = VecSim(task, n, action_mode=mode, profile="phase_cycles", step_kernel="coop", preset="fast")   # (the two-wave kernels are preset fast's)
bufs = [sim.alloc_actions() for _ in range(16)]
for i, b in enumerate(bufs):
    sim.fill_random_actions(b, 0, i)
for i in range(40):
    sim.step_device(bufs[i % 16].ptr)
rows = []
for i in range(20):
    sim.step_device(bufs[i % 16].ptr)
    sim.sync()
    a_tot, a_wait, a_pre, cpl = (x.numpy()[::64].astype(np.float64) for x in (sim.active_mask, sim.active_count, sim.max_sweeps, sim.choice))
    c = sim.ctrl.numpy()
    b_tot, b_wait = c[0][::64].astype(np.float64), c[1][::64].astype(np.float64)
    rows.append((a_tot, a_wait, a_pre, cpl, b_tot, b_wait, c[2][::64].astype(np.float64), c[3][::64].astype(np.float64)))
f = lambda k: np.concatenate([r[k] for r in rows])
a_tot, a_wait, a_pre, cpl, b_tot, b_wait, a_wx, a_we = (f(k) for k in range(8))
print(f"{task} {mode} n={n} {os.environ['LCR_STEP_KERNEL']}: per control step, cycles (mean / max over workgroups)")
print(f"  arm wave : total {a_tot.mean():.0f} / {a_tot.max():.0f}   waiting {a_wait.mean():.0f} / {a_wait.max():.0f}   before barrier 1 {a_pre.mean():.0f} / {a_pre.max():.0f}")
print(f"  arm wave waits: at X (inertia factor) {a_wx.mean():.0f} / {a_wx.max():.0f}   Y..E (joint acceleration) {a_we.mean():.0f} / {a_we.max():.0f}   other (barrier 1, coupled sweeps) {np.mean(a_wait - a_wx - a_we):.0f}")
print(f"  cube wave: total {b_tot.mean():.0f} / {b_tot.max():.0f}   waiting {b_wait.mean():.0f} / {b_wait.max():.0f}   busy {np.mean(b_tot - b_wait):.0f} / {np.max(b_tot - b_wait):.0f}")
print(f"  coupled substeps per step: mean {cpl.mean():.2f}, workgroups with any {np.mean(cpl > 0):.3f}; arm total where uncoupled {a_tot[cpl == 0].mean():.0f} / coupled {a_tot[cpl > 0].mean() if (cpl > 0).any() else 0:.0f}")
# the slowest workgroups of the last step (they set the launch time): where do their cycles go?
a_tot, a_wait, a_pre, cpl, b_tot, b_wait, a_wx, a_we = rows[-1]
order = np.argsort(-a_tot)[:8]
print("  slowest workgroups of one step: arm total / waiting (X, E, other) / X..B1 own / coupled substeps / cube busy")
for i in order:
    print(f"    wg {i:5d}: {a_tot[i]:.0f} / {a_wait[i]:.0f} ({a_wx[i]:.0f}, {a_we[i]:.0f}, {a_wait[i] - a_wx[i] - a_we[i]:.0f}) / {a_pre[i]:.0f} / {cpl[i]:.0f} / {b_tot[i] - b_wait[i]:.0f}")
q = np.percentile(a_tot, [50, 90, 99, 100])
print(f"  arm total percentiles 50/90/99/100: {q[0]:.0f} {q[1]:.0f} {q[2]:.0f} {q[3]:.0f}")
sim.close()
