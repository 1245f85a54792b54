"""Property-based GPU parity (hypothesis): random task / action mode / reward type / solver settings / contact rows / thresholds / batch
size / seeds -- the HIP kernel must track the oracle under every configuration, not only the defaults."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from tests import util

pytestmark = pytest.mark.gpu

TASKS = ["reach", "lift", "push", "pick_place", "stack", "push_loop"]


@settings(max_examples=int(os.environ.get("LCR_HYP_EXAMPLES", "100")),   # (soak runs: LCR_HYP_EXAMPLES=800)
           deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(
    task=st.sampled_from(TASKS),
    mode=st.sampled_from(["joint", "ee"]),
    reward=st.sampled_from(["sparse", "dense"]),
    n=st.integers(min_value=1, max_value=200),
    n_substeps=st.integers(min_value=1, max_value=25),
    pgs_iters=st.integers(min_value=1, max_value=8),
    impratio=st.sampled_from([1.0, 10.0, 100.0]),
    thr=st.floats(min_value=0.02, max_value=0.2),
    condim=st.sampled_from([None, 4, 6]),
    solve=st.sampled_from(["carry", "resync_cold", "compat_cold"]),
    arm_collision=st.booleans(),
    family=st.sampled_from(["auto", "single", "coop2"]),
    preset=st.sampled_from(["faithful", "fast"]),
    seed=st.integers(min_value=0, max_value=2**40),
)
def test_random_configuration_parity(hip_lib, task, mode, reward, n, n_substeps, pgs_iters, impratio, thr, condim, solve, arm_collision, family, preset, seed):
    # preset "faithful" (the default): the Newton kernels -- sweep count, contact-row mode and kernel family do not apply; "fast": the sweep kernels with all of them drawn
    sweeps = dict(pgs_iters=pgs_iters, finger_cube_condim=condim) if preset == "fast" else {}
    kw = dict(action_mode=mode, reward_type=reward, n_substeps=n_substeps, impratio=impratio, preset=preset, **sweeps,
              distance_threshold=thr, auto_reset=False, max_episode_steps=0, arm_collision=arm_collision,
              # "carry": the default product mode, each step starts from the oracle's carried forces; "resync_cold": default mode, forces
              # dropped by lcr_set_state; "compat_cold": LCR_COMPAT_COLD_SOLVE_EACH_STEP on both sides (the kernel then has no warm array)
              compat=2 if solve == "compat_cold" else 0)
    # step-kernel family: the shard-size dispatch (two-cooperating-waves kernels at these sizes), the one-wave kernels, or the two-wave
    # variant compiled for two waves per SIMD (lcr_create reads LCR_STEP_KERNEL)
    if family == "auto":
        os.environ.pop("LCR_STEP_KERNEL", None)
    else:
        os.environ["LCR_STEP_KERNEL"] = family
    try:
        sim, o = util.make_pair(task, n, **kw)
    finally:
        os.environ.pop("LCR_STEP_KERNEL", None)
    rng = np.random.default_rng(seed % (2**32))
    seeds = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)) % np.uint64(2**63)
    o.reset(seeds=seeds); sim.reset(seeds=seeds)
    st0 = util.pull_state(sim)
    np.testing.assert_array_equal(st0["qpos"][:, : sim.nq].astype(np.float32), o.qpos[:, : sim.nq].astype(np.float32))
    np.testing.assert_array_equal(st0["rng"], o.rng)
    for t in range(3):
        a = rng.uniform(-1.3, 1.3, (n, sim.action_dim)).astype(np.float32)
        dq, dv, ok, st1 = util.parity_step(sim, o, a, 3e-5, 5e-3, max_dq=5e-2, max_dv=5.0, where=(task, mode, n_substeps, pgs_iters, solve, t), carry=solve == "carry")
        # (every env outside the tolerance has been explained inside parity_step -- decision flip or ill-conditioned for fp32 -- and is bounded by max_dq / max_dv; what
        #  is limited here is how MANY: 1 %, or two envs of a small batch -- StackTwoCubes seed 146: two of 44 cubes land on the other cube in the same step, eight-point
        #  manifold, and the oracle's own fp32 build is 1e-2 rad/s away from its fp64 build on exactly those two)
        #  (soak, 400 examples: StackTwoCubes n = 153, one substep per control step, impratio 1: three cubes land in the same step -> 2 % of a batch)
        #  (soak, 2 x 1 500 examples: StackTwoCubes with FEW substeps per control step -- both cubes of every env are dropped by the reset and land in the same control
        #   step, a first contact of a few micrometres in all envs at once: up to 10 % of a batch, each explained, |dqvel| <= 1e-2 m/s; seen: 3 of 44 with one substep,
        #   3 of 70 with four)
        allowed = max(2, n // 50, max(3, n // 10) if (task == "stack" and n_substeps <= 8) else 0)
        assert ok.mean() >= min(0.99, 1 - 1.5 / n) or ok.sum() >= n - allowed, (t, ok.mean(), np.sort(dq)[-3:], np.sort(dv)[-3:])
        out = sim.outputs()
        same = out["terminated"] == o.terminated.astype(bool)
        assert same.sum() >= n - max(1, n // 100)
    sim.close()
