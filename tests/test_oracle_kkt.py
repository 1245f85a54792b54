"""The contact solve against a SOLVER-INDEPENDENT yard-stick: MuJoCo's published convex constraint problem itself.

  * `orc_io.kkt` -- the natural residual of the KKT conditions of the DUAL problem  min_{f in K} 1/2 f'(A + R) f + f'(J a0 - aref)  (elliptic cones of
    follower.xml:3) at whatever forces a solver returned: zero exactly at the optimum, whichever algorithm produced them;
  * `orc_params.solver = 1` -- Newton's method on the PRIMAL problem (MuJoCo's default solver) to machine precision: the exact optimum;
  * `orc_params.cone` -- 3: the block projected-gradient step the kernels run since round 4; 0: rounds 1-3 (rows + radial projection, deviation D2);
    1: MuJoCo's PGS block update with the exact friction QCQP.
The physics stays "parity unpinned" against MuJoCo itself (not installable here); what these tests pin is that the product's iteration converges to the
optimum of the problem MuJoCo documents, and how far its default four sweeps are from it (tools/kkt_distance.py -> profiles/r04_kkt_distance.json).
"""
import numpy as np
import pytest

from oracle import orc
from tests import util

STATE = ("qpos", "qvel", "ee_lag", "target", "elapsed", "rng", "goal", "sim_time", "warm")


def _rollout(task, n, steps, seed=0, **kw):
    o = orc.Oracle(task, n, kkt=True, auto_reset=0, max_episode_steps=0, **({} if "solver" in kw or "preset" in kw else {"preset": "fast"}), **kw)
    o.reset(seeds=np.arange(n, dtype=np.uint64) + 5)
    rng = np.random.default_rng(seed)
    worst = np.zeros(n)
    for _ in range(steps):
        o.step(rng.uniform(-1, 1, (n, o.action_dim)).astype(np.float32), threads=0)
        worst = np.maximum(worst, o.kkt)
    return o, worst


@pytest.mark.parametrize("task", ["reach", "lift", "push", "pick_place", "stack", "push_loop"])
def test_newton_solve_satisfies_the_kkt_conditions(task):
    """random policy, every substep of 12 control steps: the primal Newton solve leaves a KKT residual at rounding level -- also on the contact sets on
    which 50 PGS sweeps do not converge (10 kg cube on four floor contacts, two cubes, the rails)"""
    o, worst = _rollout(task, 96, 12, solver=1)
    assert np.isfinite(o.qpos).all()
    assert worst.max() < (1e-6 if task == "push_loop" else 1e-9), worst.max()       # (PushCubeLoop: friction coefficients of 1.5 m scale the residual of its torsional rows)
    assert np.median(worst) < 1e-11


def test_newton_solve_on_the_pinched_stiff_cube():
    """the states round 3 could only excuse ('the oracle's PGS hits its 50-sweep cap'): PushCubeLoop's 50 g cube pinched between the fingers with six-row
    contacts and friction 1.5 -- the exact solver reaches the optimum there as well"""
    n = 32
    o = orc.Oracle("push_loop", n, kkt=True, auto_reset=0, max_episode_steps=0, condim6=1, solver=1)
    o.reset(seeds=np.arange(n))
    util.pinch_setup(o)
    rng = np.random.default_rng(3)
    o.qpos[:, 6:9] += rng.normal(0, 3e-4, (n, 3))
    o.qvel[:, 9:12] = rng.normal(0, 0.5, (n, 3))
    worst = 0.0
    for t in range(4):
        o.step(rng.uniform(-0.1, 0.1, (n, o.action_dim)).astype(np.float32), threads=0)
        assert ((o.active_mask >> 12) & 3).astype(bool).mean() > 0.5 or t > 1
        worst = max(worst, o.kkt.max())
    assert worst < 1e-6, worst


def test_two_independent_algorithms_reach_the_same_optimum():
    """the block projected-gradient iteration (dual, first order) swept to convergence and the Newton solve (primal, second order) share no code beyond the problem
    data: from identical states incl. carried forces they land on the same accelerations"""
    n = 48
    walk = orc.Oracle("push", n, auto_reset=0, max_episode_steps=0, preset="fast")
    pg = orc.Oracle("push", n, kkt=True, auto_reset=0, max_episode_steps=0, preset="fast", pgs_iters=-1, pgs_tol=1e-12, pgs_cap=20000)
    qc = orc.Oracle("push", n, kkt=True, auto_reset=0, max_episode_steps=0, preset="fast", pgs_iters=-1, pgs_tol=1e-12, pgs_cap=20000, cone=1)
    nt = orc.Oracle("push", n, kkt=True, auto_reset=0, max_episode_steps=0, preset="fast", solver=1)
    walk.reset(seeds=np.arange(n, dtype=np.uint64) + 9)
    rng = np.random.default_rng(2)
    d_pg, d_qc = [], []
    for t in range(8):
        a = rng.uniform(-1, 1, (n, walk.action_dim)).astype(np.float32)
        if t >= 2:
            for o in (pg, qc, nt):
                for k in STATE:
                    getattr(o, k)[:] = getattr(walk, k)
                o.step(a, threads=0)
            d_pg.append(np.abs(pg.qpos - nt.qpos).max(axis=1)); d_qc.append(np.abs(qc.qpos - nt.qpos).max(axis=1))
            assert nt.kkt.max() < 1e-9
        walk.step(a, threads=0)
    d_pg, d_qc = np.concatenate(d_pg), np.concatenate(d_qc)
    assert np.percentile(d_pg, 95) < 1e-8 and np.percentile(d_qc, 95) < 1e-8, (np.percentile(d_pg, 95), np.percentile(d_qc, 95))


def test_fixed_points_block_step_vs_rows_with_radial_projection():
    """why round 4 changed the iteration: swept to convergence, the rounds 1-3 update (cone = 0) stops at points that violate the KKT conditions wherever a contact's
    optimum needs a friction-supported normal force (mu = 1.5: the sliding fingers) -- the block step (cone = 3) does not"""
    res = {}
    for cone in (0, 3):
        o, worst = _rollout("reach", 128, 10, pgs_iters=-1, pgs_tol=1e-10, pgs_cap=3000, cone=cone)
        res[cone] = worst
    assert np.percentile(res[3], 95) < 1e-4, np.percentile(res[3], 95)       # (its residual at the sweep cap: the torsional rows of the resting cube, scaled by 1 / 0.005)
    assert np.percentile(res[0], 90) > 1e-2, np.percentile(res[0], 90)       # ~15 % of the env-steps have a finger or proxy contact


def test_default_four_sweeps_distance_from_the_optimum():
    """the number DESIGN.md quotes for deviation D1 (tools/kkt_distance.py at larger n): one control step of the default solve against the exact optimum from identical
    states incl. carried forces -- median at rounding level, 90th percentile below 1e-3, and closer than the rounds 1-3 iteration at the 90th / 99th percentile"""
    n = 128
    walk = orc.Oracle("reach", n, auto_reset=0, max_episode_steps=0, preset="fast")
    var = {"default": orc.Oracle("reach", n, auto_reset=0, max_episode_steps=0, preset="fast"), "legacy": orc.Oracle("reach", n, auto_reset=0, max_episode_steps=0, preset="fast", cone=0),
           "exact": orc.Oracle("reach", n, auto_reset=0, max_episode_steps=0, preset="fast", solver=1)}
    walk.reset(seeds=np.arange(n, dtype=np.uint64) + 77)
    rng = np.random.default_rng(5)
    d = {"default": [], "legacy": []}
    for t in range(14):
        a = rng.uniform(-1, 1, (n, walk.action_dim)).astype(np.float32)
        if t >= 3:
            for o in var.values():
                for k in STATE:
                    getattr(o, k)[:] = getattr(walk, k)
                o.step(a, threads=0)
            for k in d:
                d[k].append(np.abs(var[k].qpos - var["exact"].qpos).max(axis=1))
        walk.step(a, threads=0)
    d = {k: np.concatenate(v) for k, v in d.items()}
    assert np.median(d["default"]) < 1e-7 and np.percentile(d["default"], 90) < 1e-3, (np.median(d["default"]), np.percentile(d["default"], 90))
    assert np.percentile(d["default"], 90) < 0.2 * np.percentile(d["legacy"], 90) and np.percentile(d["default"], 99) < np.percentile(d["legacy"], 99)


# ---------------------------------------------------------------- round 5: the product's Newton solve (orc_params.solver = 2, = the Newton kernels)
@pytest.mark.parametrize("task", ["reach", "lift", "stack", "push_loop"])
def test_default_newton_product_reaches_the_exact_optimum(task):
    """the DEFAULT preset's solve (fixed budget, safeguarded line search, exits at the rounding floor) against the exact optimum from identical states under the
    random policy: |dqpos| per control step p90 <= 2e-5 and p99 <= 1e-3 is what VERDICT r4 #1 asked for; measured 3e-9 / 6e-8 -- asserted two decades inside the ask.
    The same C source in float arithmetic (the fp32 twin, what the kernels' arithmetic can do): p99 <= 5e-6."""
    n = 256
    kw = dict(auto_reset=0, max_episode_steps=0)
    walk, prod, ex, f32 = orc.Oracle(task, n, **kw), orc.Oracle(task, n, **kw), orc.Oracle(task, n, solver=1, **kw), orc.Oracle(task, n, f32=True, **kw)
    assert prod.params.solver == 2 and prod.params.condim6 == 2 and prod.params.finger_geom == 1 and prod.params.cc_points == 8
    walk.reset(seeds=np.arange(n, dtype=np.uint64) + 77)
    rng = np.random.default_rng(5)
    dq, dq32 = [], []
    for t in range(12):
        a = rng.uniform(-1, 1, (n, walk.action_dim)).astype(np.float32)
        if t >= 3:
            for o in (prod, ex, f32):
                for k in STATE:
                    if k == "warm" and o is f32:
                        o.warm.view(np.float32)[:, :192] = walk.warm.view(np.float64)[:, :192]
                    else:
                        getattr(o, k)[:] = getattr(walk, k)
                o.step(a, threads=0)
            dq.append(np.abs(prod.qpos[:, : ex.nq] - ex.qpos[:, : ex.nq]).max(1))
            dq32.append(np.abs(f32.qpos[:, : ex.nq] - ex.qpos[:, : ex.nq]).max(1))
        walk.step(a, threads=0)
    dq, dq32 = np.concatenate(dq), np.concatenate(dq32)
    assert np.percentile(dq, 90) <= 2e-7 and np.percentile(dq, 99) <= 1e-5, (np.percentile(dq, [90, 99]), dq.max())
    assert np.percentile(dq32, 99) <= 5e-6, np.percentile(dq32, [50, 90, 99])


def test_newton_line_search_finds_a_joint_limit_that_switches_on_inside_the_bracket():
    """a joint beyond its range limit from a cold start: the limit row's 1e4 x steeper piece of phi' begins inside the first bracket.  The derivative-only Illinois search of
    the first round-5 kernels crept towards it and left the solve 0.08 rad off after the control step; the safeguarded Newton steps from both ends of the bracket take it in
    one evaluation (oracle and kernel run the same rule; GPU: tests/test_gpu_parity.py::test_joint_limit_rows)."""
    rng = np.random.default_rng(22)
    n = 256
    kw = dict(auto_reset=0, max_episode_steps=0)
    prod, ex = orc.Oracle("lift", n, **kw), orc.Oracle("lift", n, solver=1, **kw)
    prod.reset(seeds=np.arange(n))
    prod.qpos[:, 6:9] = [0.5, 0.5, 0.0149]
    prod.qpos[:, 0] = np.where(rng.uniform(size=n) < 0.5, 3.14 + rng.uniform(0, 0.01, n), -3.14 - rng.uniform(0, 0.01, n))
    prod.qpos[:, 5] = 0.032 + rng.uniform(0, 0.01, n)
    prod.qpos[:, 1] = -0.8; prod.qpos[:, 2] = 0.3
    prod.qvel[:, :6] = rng.normal(0, 0.5, (n, 6))
    for k in STATE:
        getattr(ex, k)[:] = getattr(prod, k)
    a = (0.1 * rng.uniform(-1.2, 1.2, (n, prod.action_dim))).astype(np.float32)
    prod.step(a, threads=0); ex.step(a, threads=0)
    d = np.abs(prod.qpos[:, :13] - ex.qpos[:, :13]).max(1)
    assert d.max() < 1e-6, (d.max(), int(prod.max_sweeps.max()))
