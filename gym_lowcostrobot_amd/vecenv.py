"""Vectorised consumers: an SB3-style VecEnv and a gymnasium.vector-style env over one VecSim (reference call pattern:
examples/gym_manipulation_sb3.py:26-39 make_vec_env / DummyVecEnv; rl_zoo3 with observation_mode "both", README.md:122).

DummyVecEnv semantics reproduced in the fused kernel: an env that terminates or hits the TimeLimit is reset inside the same
step; the returned observation is the reset observation, the last observation of the episode is in
infos[i]["terminal_observation"], and infos[i]["TimeLimit.truncated"] / ["is_success"] are set.

Host cost per step is O(1) python + one packed device-to-host copy (VecSim.fetch_host, 132 B/env) + work proportional to the
number of envs that finished an episode in this step; there is no per-env python loop.
"""
import numpy as np

from . import spaces as sp
from .vecsim import VecSim

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except Exception:
    _SB3VecEnv = object


class _SharedInfo(dict):
    """The info dict every env that did NOT finish an episode in a step refers to (65 536 fresh dicts per step would dominate the step time).
    It is a real dict -- isinstance checks, copy.deepcopy, pickling (subprocess / IPC wrappers) work, and a copy is an ordinary writable dict --
    but writing into the shared instance fails loudly instead of leaking the key into every env."""
    __slots__ = ()

    def _ro(self, *a, **k):
        raise TypeError("this info dict is shared by all envs that did not finish an episode in this step: copy it (dict(info)) before writing")

    __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = __ior__ = _ro

    def __copy__(self):
        return dict(self)

    def __deepcopy__(self, memo):
        import copy

        return {copy.deepcopy(k, memo): copy.deepcopy(v, memo) for k, v in self.items()}

    def __reduce__(self):
        return (dict, (dict(self),))


def _obs_spaces(sim, observation_mode):
    """observation space of the reference env for this task / mode (reach_cube_env.py:105-115, push:111, stack:115-116)"""
    subs = {
        "arm_qpos": sp.Box(-np.pi, np.pi, shape=(6,), dtype=np.float32),
        "arm_qvel": sp.Box(-10.0, 10.0, shape=(6,), dtype=np.float32),
    }
    if sim.task_name in ("push", "pick_place"):
        subs["target_pos"] = sp.Box(-10.0, 10.0, shape=(3,), dtype=np.float32)     # always present (push_cube_env.py:297)
    if observation_mode in ("image", "both"):
        subs["image_front"] = sp.Box(0, 255, shape=(240, 320, 3), dtype=np.uint8)
        subs["image_top"] = sp.Box(0, 255, shape=(240, 320, 3), dtype=np.uint8)
    if observation_mode in ("state", "both"):
        subs[sim.cube_name] = sp.Box(-10.0, 10.0, shape=(3,), dtype=np.float32)
        if sim.task_name == "stack":
            subs["cube_blue_pos"] = sp.Box(-10.0, 10.0, shape=(3,), dtype=np.float32)
    return subs


class LowCostRobotVecEnv(_SB3VecEnv):
    def __init__(self, task, num_envs, seed=0, device=0, env_id_offset=0, **kw):
        kw.setdefault("observation_mode", "state")
        self.observation_mode = kw["observation_mode"]
        self.sim = VecSim(task, num_envs, device=device, env_id_offset=env_id_offset, base_seed=seed, auto_reset=True, **kw)
        self.num_envs = int(num_envs)
        self.task = task
        self.action_space = sp.Box(-1.0, 1.0, shape=(self.sim.action_dim,), dtype=np.float32)
        subs = _obs_spaces(self.sim, self.observation_mode)
        self.observation_space = sp.Dict(subs)
        self._keys = list(subs)
        self._actions = None
        self._last = None
        if _SB3VecEnv is not object:
            super().__init__(self.num_envs, self.observation_space, self.action_space)

    # ---- observations ----
    def _obs_from(self, h):
        sim = self.sim
        o = {"arm_qpos": h["arm_qpos"], "arm_qvel": h["arm_qvel"]}
        if sim.task_name in ("push", "pick_place"):
            o["target_pos"] = h["aux_pos"]
        if self.observation_mode in ("state", "both"):
            o[sim.cube_name] = h["cube_pos"]
            if sim.task_name == "stack":
                o["cube_blue_pos"] = h["aux_pos"]
        if self.observation_mode in ("image", "both"):
            o["image_front"] = sim.image_front.numpy()
            o["image_top"] = sim.image_top.numpy()
        # forced copies: the pinned mirror is overwritten by the next fetch and freed by close(), and SB3 keeps the returned arrays
        # in its rollout buffer (np.ascontiguousarray would hand out a VIEW of the mirror when num_envs == 1, where the transposed
        # (6, 1) view already counts as contiguous); the frames come from a device-to-host copy of their own
        return {k: (o[k] if k.startswith("image_") else np.array(o[k], copy=True, order="C")) for k in self._keys}

    def _obs(self):
        return self._obs_from(self.sim.fetch_host())

    def _terminal(self, tobs_rows, env_ids):
        """terminal observation dicts -- EXACTLY the keys of observation_space (SB3's VecTransposeImage and the TimeLimit bootstrap of
        PPO index every key of infos[i]["terminal_observation"]) -- for the rows of `tobs_rows` (k, 18): arm_qpos6, arm_qvel6, cube3, aux3.
        Image modes: the kernel has already reset these envs (their frame buffers show the reset state), so the last frames of the episodes
        are ray-cast from the terminal poses -- ALL of them in one batched call (lcr_render_terminal: with TimeLimit(50) and a common
        start every env finishes in the same step)."""
        sim = self.sim
        img = self.observation_mode in ("image", "both")
        fr = tp = None
        if img and len(env_ids):
            fr, tp = sim.render_terminal(env_ids)
        out = []
        for j, t in enumerate(tobs_rows):
            d = {"arm_qpos": t[0:6].copy(), "arm_qvel": t[6:12].copy()}
            if sim.task_name in ("push", "pick_place"):
                d["target_pos"] = t[15:18].copy()
            if self.observation_mode in ("state", "both"):
                d[sim.cube_name] = t[12:15].copy()
                if sim.task_name == "stack":
                    d["cube_blue_pos"] = t[15:18].copy()
            if img:
                d["image_front"], d["image_top"] = fr[j], tp[j]
            out.append({k: d[k] for k in self._keys})
        return out

    # ---- SB3 VecEnv API ----
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
        h = self.sim.fetch_host()
        obs = self._obs_from(h)
        term, trunc, dres = h["terminated"], h["truncated"], h["did_reset"]
        dones = term | trunc
        lift = self.task == "lift"
        # envs that did NOT finish an episode all refer to ONE read-only mapping per step (65 536 dicts per step would dominate the step
        # time); a wrapper that tries to write into it fails loudly instead of leaking the key into every env
        shared = _SharedInfo({"TimeLimit.truncated": False} if lift else {"is_success": False, "TimeLimit.truncated": False})
        infos = [shared] * self.num_envs
        idx = np.nonzero(dones | dres)[0]
        if idx.size:
            tl = trunc[idx] & ~term[idx]
            succ = h["is_success"][idx]
            ridx = idx[dres[idx]]                  # envs the kernel has reset: they carry a terminal observation
            tobs = dict(zip(ridx.tolist(), self._terminal(h["terminal_obs"][ridx], ridx))) if (h["terminal_obs"] is not None and ridx.size) else {}
            for j, i in enumerate(idx):
                d = {"TimeLimit.truncated": bool(tl[j])}
                if not lift:
                    d["is_success"] = bool(succ[j])
                if dres[i]:
                    d["terminal_observation"] = tobs.get(int(i))
                infos[i] = d
        return obs, h["reward"].copy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.sim.close()

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [int(indices)] if np.isscalar(indices) else [int(i) for i in indices]

    def get_attr(self, attr_name, indices=None):
        """per-env attribute reads SB3 performs (`render_mode`, `spec`, `reward_range`, wrapper probes): the batch answers for every env"""
        v = {"render_mode": None, "spec": None, "reward_range": (-float("inf"), float("inf")), "metadata": {"render_modes": ["rgb_array"], "render_fps": 25}}.get(attr_name)
        v = getattr(self, attr_name, v)
        return [v for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        """The calls SB3 / rl_zoo3 make through env_method on a DummyVecEnv, answered per env (a list, one entry per index) from the batch:
        `get_wrapper_attr` (attribute probes of Monitor / evaluate_policy), `render` (one ray-cast frame per env), `compute_reward` /
        `is_success` / `goal_distance` (the reference's public reward helpers, reach_cube_env.py:335-348, evaluated with the env's
        parameters), `get_state`, `action_masks`-style probes of methods the envs do not have raise AttributeError as a real env would."""
        ids = self._indices(indices)
        if method_name == "get_wrapper_attr":
            return self.get_attr(args[0] if args else kwargs["name"], ids)
        if method_name == "render":
            return [self.sim.render(i, "camera_vizu", 640, 640) for i in ids]
        if method_name in ("goal_distance", "is_success", "compute_reward") and self.task != "lift" and self.task != "push_loop":
            a, b = (np.asarray(x, np.float64) for x in (args[0], args[1]))
            d = float(np.linalg.norm(a - b))
            thr = float(self.sim.cfg.distance_threshold)
            if method_name == "goal_distance":
                r = d
            elif method_name == "is_success":
                r = np.bool_(d < thr)
            else:   # reach_cube_env.py:343-348
                r = -np.float32(d > thr) if self.sim.cfg.reward_type == 0 else -d
            return [r for _ in ids]
        if method_name == "get_state":
            st = self.sim.get_state()
            return [{k: (v[..., i] if v.ndim > 1 else v[i]) for k, v in st.items()} for i in ids]
        if method_name in ("seed", "close"):
            return [None for _ in ids]
        raise AttributeError(f"the batched simulator's envs have no method {method_name!r}")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs


class LowCostRobotVectorEnv:
    """gymnasium.vector.VectorEnv-style API (gymnasium >= 1.0 conventions) over one VecSim:

        obs, infos = venv.reset(seed=...)                       # seed: None | int (env i gets seed + i) | sequence of ints
        obs, rewards, terminations, truncations, infos = venv.step(actions)

    Autoreset happens in the SAME step (fused in the kernel): the returned observation of a finished env is its reset
    observation, `infos["final_obs"]` holds the terminal observations and `infos["_final_obs"]` the mask, as with
    gymnasium's AutoresetMode.SAME_STEP.  Arrays are batched numpy arrays, observations a dict of (N, .) arrays.
    """

    def __init__(self, task, num_envs, seed=0, device=0, env_id_offset=0, **kw):
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
        v, sim = self._v, self._v.sim
        sim.step(np.asarray(actions, np.float32))
        h = sim.fetch_host()
        obs = v._obs_from(h)
        infos = {}
        if self.task != "lift":
            infos["is_success"] = h["is_success"].copy()
        if h["terminal_obs"] is not None:
            t = h["terminal_obs"]
            fin = {"arm_qpos": t[:, 0:6].copy(), "arm_qvel": t[:, 6:12].copy()}
            if sim.task_name in ("push", "pick_place"):
                fin["target_pos"] = t[:, 15:18].copy()
            if v.observation_mode in ("state", "both"):
                fin[sim.cube_name] = t[:, 12:15].copy()
                if sim.task_name == "stack":
                    fin["cube_blue_pos"] = t[:, 15:18].copy()
            if v.observation_mode in ("image", "both"):   # final frames: ONE batched ray-cast of the terminal poses of the envs that were reset
                ridx = np.nonzero(h["did_reset"])[0]
                fin["image_front"] = np.zeros((self.num_envs, 240, 320, 3), np.uint8)   # (calloc'ed: pages of envs that were not reset are never touched)
                fin["image_top"] = np.zeros((self.num_envs, 240, 320, 3), np.uint8)
                if ridx.size:
                    fin["image_front"][ridx], fin["image_top"][ridx] = sim.render_terminal(ridx)
            fin = {k: fin[k] for k in v._keys}
            infos["final_obs"] = fin
            infos["_final_obs"] = h["did_reset"].copy()
        return obs, h["reward"].copy(), h["terminated"].copy(), h["truncated"].copy(), infos

    def close(self):
        self._v.close()
