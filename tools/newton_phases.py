"""Where a wave of the Newton kernels spends its cycles (lcr_config.diagnostics = 2, VecSim profile="wave_cycles"; GPU box): per control step and wave -- total,
inside the Newton solves, inside COUPLED solves (envs with touching bodies solved cooperatively by the wave, or substeps in the coupled SIMT copy), how many of
each, Newton iterations executed; mean wave against the slowest wave of each step (the launch ends with the slowest).     python tools/newton_phases.py [task ...] [--n 65536]"""
import argparse
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from gym_lowcostrobot_amd import VecSim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tasks", nargs="*", default=["reach", "push", "lift", "pick_place_ee", "stack", "push_loop"])
ap.add_argument("--n", type=int, default=65536)
a = ap.parse_args()
for name in a.tasks:
    mode = "ee" if name.endswith("_ee") else "joint"
    task = name.replace("_ee", "")
    sim = VecSim(task, a.n, action_mode=mode, profile="wave_cycles")
    bufs = [sim.alloc_actions() for _ in range(16)]
    for i, b in enumerate(bufs):
        sim.fill_random_actions(b, 0, i)
    for i in range(60):
        sim.step_device(bufs[i % 16].ptr)
    rows = []
    for i in range(20):
        sim.step_device(bufs[i % 16].ptr)
        sim.sync()
        tot, solve, cpl, ch = (x.numpy()[::64].astype(np.float64) for x in (sim.max_sweeps, sim.active_mask, sim.active_count, sim.choice))
        rows.append((tot, solve, cpl, np.mod(np.floor(ch / 65536.0), 256.0), np.mod(ch, 65536.0), np.floor(ch / 16777216.0)))
    tot, solve, cpl, ncpl, its, nslow = (np.stack([r[k] for r in rows]) for k in range(6))   # [step][wave]
    slow = tot.argmax(1)
    pick = lambda x: x[np.arange(len(slow)), slow].mean()
    print(f"{name:14s} n={a.n}: cycles per control step and wave, MEAN wave | SLOWEST wave of the step (mean over 20 steps)")
    print(f"   total            {tot.mean():10.0f} | {pick(tot):10.0f}     mean / slowest = {tot.mean() / pick(tot):.2f}")
    print(f"   Newton solves    {solve.mean():10.0f} | {pick(solve):10.0f}     ({100 * solve.mean() / tot.mean():.0f} % | {100 * pick(solve) / pick(tot):.0f} % of the wave's cycles; the rest: set-up, integration, reward)")
    print(f"   coupled solves   {cpl.mean():10.0f} | {pick(cpl):10.0f}     (cooperative solves + whole solves of substeps in the coupled SIMT copy); waves with any: {100 * (cpl > 0).mean():.0f} %")
    if True:   # envs solved cooperatively (summed over the substeps); substeps the wave spent in the coupled SIMT copy
        print(f"   substeps in the coupled SIMT copy: {nslow.mean():.2f} | {pick(nslow):.2f}; waves with any: {100 * (nslow > 0).mean():.1f} %; cooperative solves per wave and step: {ncpl.mean():.2f} | {pick(ncpl):.2f}, most {ncpl.max():.0f}")
        srt = np.sort(tot.max(0))[::-1]
        print(f"   slowest waves (max over the steps), cycles: {' '.join(f'{v:.0f}' for v in srt[:6])};  p99 {np.percentile(tot, 99):.0f}  p90 {np.percentile(tot, 90):.0f}  p50 {np.percentile(tot, 50):.0f}")
    print(f"   Newton iterations executed by the wave: {its.mean():.1f} | {pick(its):.1f};  cycles per iteration (solves / iterations): {solve.sum() / max(its.sum(), 1):.0f}", flush=True)
    sim.close()
