"""Vectorised consumers: an SB3-style VecEnv over one VecSim (reference call pattern:
examples/gym_manipulation_sb3.py:26-39 make_vec_env / DummyVecEnv).

DummyVecEnv semantics reproduced in the fused kernel: an env that terminates or hits the TimeLimit is reset
inside the same step; the returned observation is the reset observation, the last observation of the
episode is in infos[i]["terminal_observation"], and infos[i]["TimeLimit.truncated"] / ["is_success"] are set.
"""
import numpy as np

from . import spaces as sp
from .vecsim import VecSim

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except Exception:
    _SB3VecEnv = object


class LowCostRobotVecEnv(_SB3VecEnv):
    def __init__(self, task, num_envs, seed=0, device=0, env_id_offset=0, **kw):
        kw.setdefault("observation_mode", "state")
        self.sim = VecSim(task, num_envs, device=device, env_id_offset=env_id_offset, base_seed=seed, auto_reset=True, **kw)
        self.num_envs = int(num_envs)
        self.task = task
        self.action_space = sp.Box(-1.0, 1.0, shape=(self.sim.action_dim,), dtype=np.float32)
        subs = {
            "arm_qpos": sp.Box(-np.pi, np.pi, shape=(6,), dtype=np.float32),
            "arm_qvel": sp.Box(-10.0, 10.0, shape=(6,), dtype=np.float32),
            self.sim.cube_name: sp.Box(-10.0, 10.0, shape=(3,), dtype=np.float32),
        }
        if self.sim.aux_name:
            subs[self.sim.aux_name] = sp.Box(-10.0, 10.0, shape=(3,), dtype=np.float32)
        self.observation_space = sp.Dict(subs)
        self._keys = list(subs)
        self._actions = None
        if _SB3VecEnv is not object:
            super().__init__(self.num_envs, self.observation_space, self.action_space)

    def _obs(self):
        o = self.sim.observations()
        return {k: o[k] for k in self._keys}

    def seed(self, seed=None):
        self._seed = seed
        return [None if seed is None else seed + i for i in range(self.num_envs)]

    def reset(self):
        seed = getattr(self, "_seed", None)
        if seed is not None:  # SB3: env i seeded with seed + i
            self.sim.reset(seeds=np.arange(self.num_envs, dtype=np.uint64) + np.uint64(seed))
            self._seed = None
        else:
            self.sim.reset()
        return self._obs()

    def step_async(self, actions):
        self._actions = np.asarray(actions, np.float32)

    def step_wait(self):
        self.sim.step(self._actions)
        out = self.sim.outputs()
        obs = self._obs()
        dones = out["terminated"] | out["truncated"]
        infos = [{} for _ in range(self.num_envs)]
        tobs = None
        if out["did_reset"].any():
            tobs = self.sim.terminal_obs.numpy().T  # (N, 18): arm_qpos6, arm_qvel6, cube3, aux3
        for i in range(self.num_envs):
            if self.task != "lift":
                infos[i]["is_success"] = bool(out["is_success"][i])
            infos[i]["TimeLimit.truncated"] = bool(out["truncated"][i] and not out["terminated"][i])
            if out["did_reset"][i]:
                t = tobs[i]
                d = {"arm_qpos": t[0:6].copy(), "arm_qvel": t[6:12].copy(), self.sim.cube_name: t[12:15].copy()}
                if self.sim.aux_name:
                    d[self.sim.aux_name] = t[15:18].copy()
                infos[i]["terminal_observation"] = d
        return obs, out["reward"].copy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.sim.close()

    # SB3 VecEnv abstract API
    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name, None)] * self.num_envs

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        raise NotImplementedError("batched simulator: no per-env python objects")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs


class LowCostRobotVectorEnv:
    """gymnasium.vector.VectorEnv-style API (gymnasium >= 1.0 conventions) over one VecSim:

        obs, infos = venv.reset(seed=...)                       # seed: None | int (env i gets seed + i) | sequence of ints
        obs, rewards, terminations, truncations, infos = venv.step(actions)

    Autoreset happens in the SAME step (fused in the kernel): the returned observation of a finished env is its reset
    observation, `infos["final_obs"]` holds the terminal observations and `infos["_final_obs"]` the mask, as with
    gymnasium's AutoresetMode.SAME_STEP.  Arrays are batched numpy arrays, observations a dict of (N, .) float32.
    """

    def __init__(self, task, num_envs, seed=0, device=0, env_id_offset=0, **kw):
        kw.setdefault("observation_mode", "state")
        self._v = LowCostRobotVecEnv(task, num_envs, seed=seed, device=device, env_id_offset=env_id_offset, **kw)
        self.num_envs = self._v.num_envs
        self.single_action_space = self._v.action_space
        self.single_observation_space = self._v.observation_space
        self.task = task

    def reset(self, *, seed=None, options=None):
        sim = self._v.sim
        if seed is None:
            sim.reset()
        elif np.isscalar(seed):
            sim.reset(seeds=np.arange(self.num_envs, dtype=np.uint64) + np.uint64(int(seed)))
        else:
            sim.reset(seeds=np.asarray(seed, np.uint64))
        return self._v._obs(), {}

    def step(self, actions):
        sim = self._v.sim
        sim.step(np.asarray(actions, np.float32))
        out = sim.outputs()
        obs = self._v._obs()
        infos = {}
        if self.task != "lift":
            infos["is_success"] = out["is_success"]
        if out["did_reset"].any():
            t = sim.terminal_obs.numpy().T
            fin = {"arm_qpos": t[:, 0:6], "arm_qvel": t[:, 6:12], sim.cube_name: t[:, 12:15]}
            if sim.aux_name:
                fin[sim.aux_name] = t[:, 15:18]
            infos["final_obs"] = fin
            infos["_final_obs"] = out["did_reset"]
        return obs, out["reward"].copy(), out["terminated"], out["truncated"], infos

    def close(self):
        self._v.close()
