"""Quick agreement check of the step-kernel families on the GPU box: the same seeded batch stepped by the one-wave kernel and by the
two-cooperating-waves kernels (LCR_STEP_KERNEL=single|coop1|coop2); prints the largest state difference per control step.
    python tools/coop_check.py [task ...]"""
import os
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from gym_lowcostrobot_amd import VecSim  # noqa: E402

tasks = sys.argv[1:] or ["reach", "push", "lift", "pick_place", "stack", "push_loop"]
n = 4096
for task in tasks:
    for mode in ("joint", "ee"):
        sims = {}
        for fam in ("single", "coop1", "coop2"):
            os.environ["LCR_STEP_KERNEL"] = fam
            sims[fam] = VecSim(task, n, action_mode=mode, base_seed=5, diagnostics=True)
        rng = np.random.default_rng(1)
        worst = {"coop1": 0.0, "coop2": 0.0}
        flips = {"coop1": 0, "coop2": 0}
        for t in range(12):
            a = rng.uniform(-1, 1, (n, sims["single"].action_dim)).astype(np.float32)
            ref = sims["single"]
            st0 = ref.get_state()
            for fam in ("coop1", "coop2"):
                sims[fam].set_state(**st0)      # re-synchronise incl. the carried forces
            for s in sims.values():
                s.step(a)
            r = ref.get_state()
            for fam in ("coop1", "coop2"):
                c = sims[fam].get_state()
                dq = np.abs(c["qpos"] - r["qpos"]).max(axis=0)
                same = (sims[fam].choice.numpy() == ref.choice.numpy()) & (sims[fam].active_count.numpy() == ref.active_count.numpy())
                flips[fam] += int((~same).sum())
                worst[fam] = max(worst[fam], float(dq[same].max()) if same.any() else 0.0)
                assert np.isfinite(c["qpos"]).all()
        print(f"{task:10s} {mode:5s} max |dq| vs single (same decisions): coop1 {worst['coop1']:.2e} coop2 {worst['coop2']:.2e}; decision flips {flips}", flush=True)
        for s in sims.values():
            s.close()
