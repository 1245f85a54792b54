"""Shared helpers for the parity tests: drive the HIP path (through the C ABI) and the CPU oracle on
identical float32-representable (qpos, qvel, action) triples."""
import numpy as np

from oracle import orc


def f32r(x):
    """round to float32 and back: the values both sides can represent exactly"""
    return np.asarray(x, np.float32).astype(np.float64)


def sync_oracle_to_f32(o):
    o.qpos[:] = f32r(o.qpos)
    o.qvel[:] = f32r(o.qvel)
    o.ee_lag[:] = f32r(o.ee_lag)


def push_state(sim, o):
    """oracle (AoS, env-major) -> HIP sim (SoA)"""
    sim.set_state(
        qpos=np.ascontiguousarray(o.qpos[:, : sim.nq].T),
        qvel=np.ascontiguousarray(o.qvel[:, : sim.nv].T),
        ee_lag=np.ascontiguousarray(o.ee_lag.T),
        target=np.ascontiguousarray(o.target.T),
        elapsed=o.elapsed.copy(),
        rng=np.ascontiguousarray(o.rng.T),
        current_goal=o.goal.copy(),
        sim_time=o.sim_time.copy(),
    )


def pull_state(sim):
    st = sim.get_state()
    return {k: (v.T.copy() if v.ndim == 2 else v.copy()) for k, v in st.items()}


def make_pair(task, n, **kw):
    from gym_lowcostrobot_amd import VecSim

    okw = {}
    for k in ("action_mode", "reward_type", "block_gripper", "n_substeps", "max_episode_steps", "pgs_iters",
              "auto_reset", "compat", "distance_threshold", "impratio"):
        if k in kw:
            v = kw[k]
            if k == "action_mode":
                v = {"joint": 0, "ee": 1}[v]
            if k == "reward_type":
                v = {"sparse": 0, "dense": 1}[v]
            okw[k] = int(v) if isinstance(v, bool) else v
    o = orc.Oracle(task, n, **okw)
    sim = VecSim(task, n, observation_mode="state", **kw)
    assert sim.action_dim == o.action_dim
    return sim, o


def random_arm_state(rng, n, scale_q=1.0, scale_v=2.0):
    lo = np.array([-3.0, -1.5, -1.4, -1.9, -2.9, -2.0])
    hi = np.array([3.0, 1.2, 1.7, 1.9, 2.9, 0.03])
    q = lo + (hi - lo) * rng.uniform(0.5 - 0.5 * scale_q, 0.5 + 0.5 * scale_q, (n, 6))
    qd = rng.normal(0, scale_v, (n, 6))
    return q, qd


def pinch_setup(o, gap=0.0285):
    """place every env's cube between the two finger spheres with the gripper closed to `gap` (surface to surface):
    both finger<->cube contact slots are active.  Returns the gripper angle used."""
    def g(q6):
        q = np.zeros(6); q[5] = q6
        _, _, sph = orc.fk(q)
        return np.linalg.norm(sph[0] - sph[1]) - 0.013, sph
    lo, hi = -1.5, 0.0
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if g(mid)[0] > gap:
            lo = mid
        else:
            hi = mid
    _, sph = g(mid)
    d = sph[0] - sph[1]
    yaw = np.arctan2(d[1], d[0])
    o.qpos[:, :6] = 0
    o.qpos[:, 5] = mid
    o.qpos[:, 6:9] = 0.5 * (sph[0] + sph[1])
    o.qpos[:, 9:13] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
    o.qvel[:] = 0
    return mid
