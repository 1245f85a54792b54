"""Golden vectors of the reference's WHOLE step() return for all six envs, by running the reference's own code (no source is copied).

Run in the build container only (needs /root/reference):
    python tests/golden/make_step_golden.py
Writes tests/golden/step_golden.json.

What runs unmodified: the env classes' `__init__` (constructor defaults, action / observation spaces, sampling boxes), `reset()`,
`step()` = `apply_action` + `get_observation` + reward / termination / info, and the package's `register()` calls
(gym_lowcostrobot/__init__.py:9-43).  What is served by stand-ins, because `mujoco` / `gymnasium` are absent here:
  * `mujoco.MjModel.from_xml_path` / `MjData`: objects holding qpos / qvel / ctrl / time, joint ranges (follower.xml:58-95 via
    tests/golden/model_golden.json), the goal-region geoms of push_cube_loop.xml:38,41 (same file), `body(name).id`, `site(name).id`,
    `data.body(id).xpos`, `data.site(id).xpos`;
  * `mujoco.mj_forward`: body xpos := qpos slices of the free joints, site xpos := plain-numpy forward kinematics (tests/golden/kin_numpy.py: no oracle
    involved -- tests/test_kin_golden.py asserts that; pinned to the SURVEY.md 8(c) known answers) -- i.e. the kinematics of the PLACED state;
  * `mujoco.mj_step`: advances `data.time` by the model's timestep (0.002, follower.xml:3) and leaves the state alone: the physics is
    exactly what cannot run here ("parity unpinned").  With the state frozen, body / site xpos are those of the placed
    state -- which is also what the reference's reward sees in real MuJoCo up to one substep (xpos lag) -- so the fixtures pin
    the glue: reward value / dtype / sign bit, `terminated`, `truncated`, `info` keys and types, observation keys / dtypes / shapes, `data.ctrl`;
  * `gymnasium`: `Env.reset` seeding (`Generator(PCG64(SeedSequence(seed)))`), `spaces.Box` / `spaces.Dict` value holders, `register` recorder.
The HIP path consumes the fixtures with n_substeps = 1 (tests/test_step_golden.py, -m gpu): its reward is computed from the kinematics at the top of
that substep, i.e. from the placed state too.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import kin_numpy  # noqa: E402

_ARM = kin_numpy.Arm()
MODEL = json.load(open(os.path.join(HERE, "model_golden.json")))
REGISTERED = []


def _install_stubs():
    mj = types.ModuleType("mujoco")
    mj.viewer = types.ModuleType("mujoco.viewer")

    class _Id:
        def __init__(self, i):
            self.id = i

    class _Geom:
        def __init__(self, pos):
            self.pos = np.array(pos, dtype=np.float64)

    class _Opt:
        timestep = MODEL["follower"]["option"]["timestep"]

    class MjModel:
        BODIES = {"cube": 0, "cube_red": 0, "cube_blue": 1}

        def __init__(self, scene):
            self.scene = scene
            rec = MODEL["scenes"][scene]
            rng = []
            for b in MODEL["follower"]["bodies"]:
                for j in b["joints"]:
                    rng.append(j["range"])
            self.jnt_range = np.array(rng, dtype=np.float64)                 # follower.xml:58-95
            self.actuator_ctrlrange = self.jnt_range.copy()                  # inheritrange=1 (follower.xml:8)
            self.nv = 6 + 6 * sum(1 for b in rec["bodies"] if any(j.get("type") == "free" for j in b["joints"]))
            self.nq = 6 + 7 * (self.nv - 6) // 6
            self.opt = _Opt()
            self._geoms = [g for g in rec["world_geoms"]] + [g for b in rec["bodies"] for g in b["geoms"]]
            self._named = {g["name"]: _Geom(g.get("pos", [0, 0, 0])) for g in self._geoms if "name" in g}
            self.geom_pos = np.array([g.get("pos", [0, 0, 0]) for g in self._geoms], dtype=np.float64)
            self.geom_size = np.array([(list(g["size"]) + [0, 0, 0])[:3] if isinstance(g.get("size"), list) else [g.get("size", 0), 0, 0] for g in self._geoms],
                                      dtype=np.float64)

        @classmethod
        def from_xml_path(cls, path):
            return cls(os.path.splitext(os.path.basename(path))[0])

        def body(self, name):
            return _Id(self.BODIES[name])

        def site(self, name):
            assert name == "end_effector_site"
            return _Id(0)

        def geom(self, name):
            return self._named[name]

        def geom_id(self, name):
            return [g.get("name") for g in self._geoms].index(name)

    class _X:
        def __init__(self, xpos):
            self.xpos = xpos

    class MjData:
        def __init__(self, model):
            self.model = model
            self.qpos = np.zeros(model.nq)
            self.qvel = np.zeros(model.nv)
            self.ctrl = np.zeros(6)
            self.time = 0.0
            self._site = np.zeros(3)
            self._body = np.zeros((2, 3))

        def body(self, i):
            return _X(self._body[i])

        def site(self, i):
            return _X(self._site)

    def mj_forward(model, data):
        data._site[:] = _ARM.site(np.array(data.qpos[:6], dtype=np.float64))   # plain-numpy kinematics from the extracted model numbers (kin_numpy.py): no oracle
        for b in range((model.nq - 6) // 7):
            data._body[b] = data.qpos[6 + 7 * b: 9 + 7 * b]

    def mj_step(model, data):
        data.time += model.opt.timestep     # (MuJoCo: mj_step advances mjData.time by opt.timestep; the state itself is frozen here)

    def mj_jacSite(model, data, jacp, jacr, sid):
        jacp[:] = 0
        jacp[:, :6] = _ARM.site_jac(np.array(data.qpos[:6], dtype=np.float64))

    class mjtObj:
        mjOBJ_GEOM = 5

    def mj_name2id(model, objtype, name):
        assert objtype == mjtObj.mjOBJ_GEOM
        return model.geom_id(name)

    mj.MjModel, mj.MjData, mj.mj_forward, mj.mj_step, mj.mj_jacSite, mj.mjtObj, mj.mj_name2id = MjModel, MjData, mj_forward, mj_step, mj_jacSite, mjtObj, mj_name2id
    sys.modules["mujoco"] = mj
    sys.modules["mujoco.viewer"] = mj.viewer

    gym = types.ModuleType("gymnasium")

    class Env:
        def reset(self, seed=None, options=None):
            if seed is not None:   # gymnasium.utils.seeding.np_random
                self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):     # gymnasium's default dtype is float32
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype)
            self.dtype = np.dtype(dtype)

    class Dict(dict):
        @property
        def spaces(self):
            return self

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box, spaces.Dict = Box, Dict
    gym.Env, gym.spaces = Env, spaces
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda **kw: REGISTERED.append(kw)
    envs = types.ModuleType("gymnasium.envs")
    envs.registration = reg
    gym.envs = envs
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs, "gymnasium.envs.registration": reg})


def _tv(x):
    """value + exact python / numpy type of a returned scalar"""
    t = type(x)
    name = t.__name__ if t.__module__ == "builtins" else f"numpy.{t.__name__}"
    if isinstance(x, (bool, np.bool_)):
        return {"value": bool(x), "type": name}
    if isinstance(x, (int, np.integer)):
        return {"value": int(x), "type": name}
    v = float(x)
    return {"value": v, "type": name, "signbit": bool(np.signbit(x))}


def _space(sp):
    return {"shape": list(sp.shape), "dtype": str(sp.dtype), "low": [float(v) for v in np.unique(sp.low)], "high": [float(v) for v in np.unique(sp.high)]}


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import gym_lowcostrobot as ref_pkg
    from gym_lowcostrobot import envs as ref_envs
    assert ref_pkg.__file__.startswith(REF), ref_pkg.__file__
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import mujoco

    out = {"_generated_by": "tests/golden/make_step_golden.py", "registry": [dict(r) for r in REGISTERED], "constructors": [], "steps": []}
    rng = np.random.default_rng(20240)
    TASKS = [("reach", ref_envs.ReachCubeEnv), ("lift", ref_envs.LiftCubeEnv), ("push", ref_envs.PushCubeEnv), ("pick_place", ref_envs.PickPlaceCubeEnv),
             ("stack", ref_envs.StackTwoCubesEnv), ("push_loop", ref_envs.PushCubeLoopEnv)]
    jlo = np.array([-3.0, -1.5, -1.4, -1.9, -2.9, -1.7]), np.array([3.0, 1.2, 1.7, 1.9, 2.9, 0.03])

    for task, cls in TASKS:
        # ---- constructors: defaults and every action mode -------------------------------------------------------
        for kw in ({}, {"observation_mode": "state"}, {"observation_mode": "state", "action_mode": "ee"}):
            if kw.get("observation_mode", "image") != "state":
                # the default observation_mode ("image") constructs a mujoco.Renderer: record the default through the signature instead
                import inspect

                sig = inspect.signature(cls.__init__)
                out["constructors"].append({"task": task, "kwargs": kw, "signature_defaults": {k: (v.default if not isinstance(v.default, (np.generic,)) else v.default.item())
                                                                                           for k, v in sig.parameters.items() if k != "self"}})
                continue
            env = cls(**kw)
            rec = {"task": task, "kwargs": kw, "action_space": _space(env.action_space),
                   "observation_space": {k: _space(v) for k, v in env.observation_space.spaces.items()},
                   "metadata": cls.metadata, "control_decimation": int(env.control_decimation), "block_gripper": bool(env.block_gripper)}
            for a in ("distance_threshold", "height_threshold", "reward_type", "cube_low", "cube_high", "target_low", "target_high", "goal_region_low",
                      "goal_region_high", "goal_region_1_center", "goal_region_2_center", "cube_size", "current_goal"):
                if hasattr(env, a):
                    v = getattr(env, a)
                    rec[a] = v.tolist() if isinstance(v, np.ndarray) else v
            out["constructors"].append(rec)

        # ---- whole step() tuples on placed states --------------------------------------------------------------
        for reward_type in (("sparse", "dense") if task not in ("lift", "push_loop") else ("default",)):
            for action_mode in ("joint", "ee"):
                ncase = 16 if action_mode == "joint" else 6
                kw = {"observation_mode": "state", "action_mode": action_mode}
                if reward_type != "default":
                    kw["reward_type"] = reward_type
                for case in range(ncase):
                    env = cls(**kw)
                    env.reset(seed=1000 + case)                       # seeds np_random, samples cube (and target), calls mj_forward
                    nq = env.model.nq
                    q = jlo[0] + (jlo[1] - jlo[0]) * rng.uniform(0.2, 0.8, 6)
                    env.data.qpos[:6] = q
                    ee = _ARM.site(q)
                    near = case % 2 == 0                               # half of the cases near the success / overlap region
                    spread = 0.03 if near else 0.15
                    if task in ("reach", "lift"):
                        cube = ee + rng.normal(0, spread, 3)
                        cube[2] = abs(cube[2])
                        env.data.qpos[6:9] = cube
                    elif task in ("push", "pick_place"):
                        cube = np.array([rng.uniform(-0.15, 0.15), rng.uniform(0.02, 0.25), 0.015 + (rng.uniform(0, 0.08) if task == "pick_place" else 0.0)])
                        env.data.qpos[6:9] = cube
                        env.target_pos = (cube + rng.normal(0, spread, 3)).astype(np.float32)      # float32 as reset() stores it (push_cube_env.py:320)
                    elif task == "stack":
                        red = np.array([rng.uniform(-0.1, 0.1), rng.uniform(0.05, 0.2), 0.015])
                        blue = red + np.array([0, 0, 0.03]) + rng.normal(0, spread, 3)
                        blue[2] = max(blue[2], 0.015)                  # (not under the floor: the placed state should be one a step can start from)
                        env.data.qpos[6:9] = red
                        env.data.qpos[13:16] = blue
                    else:  # push_loop
                        env.current_goal = case % 2
                        cx = -0.06 if env.current_goal else 0.06
                        cube = np.array([cx + rng.normal(0, 0.004 if near else 0.03), 0.135 + rng.normal(0, 0.004 if near else 0.03), 0.0075])
                        env.data.qpos[6:9] = cube
                    env.data.qvel[:] = 0
                    env.data.time = 0.04 * case
                    mujoco.mj_forward(env.model, env.data)             # xpos of the placed state
                    k = env.action_space.shape[0]
                    act = rng.uniform(-1.3, 1.3, k).astype(np.float32)
                    pre = {"qpos": env.data.qpos.tolist(), "qvel": env.data.qvel.tolist(), "time": float(env.data.time), "site_xpos": env.data._site.tolist()}
                    if hasattr(env, "target_pos"):
                        pre["target_pos"] = [float(v) for v in env.target_pos]
                    if task == "push_loop":
                        pre["current_goal"] = int(env.current_goal)
                    ret = env.step(act)
                    assert isinstance(ret, tuple) and len(ret) == 5
                    obs, reward, terminated, truncated, info = ret
                    rec = {"task": task, "reward_type": reward_type, "action_mode": action_mode, "kwargs": kw, "pre": pre, "action": [float(v) for v in act],
                           "reward": _tv(reward), "terminated": _tv(terminated), "truncated": _tv(truncated),
                           "info": {kk: _tv(vv) for kk, vv in info.items()},
                           "obs": {kk: {"dtype": str(vv.dtype), "shape": list(vv.shape), "value": [float(x) for x in vv]} for kk, vv in obs.items()},
                           "ctrl": [float(v) for v in np.asarray(env.data.ctrl, dtype=np.float64)],
                           "qpos_after": env.data.qpos.tolist()}
                    if task == "push_loop":
                        rec["current_goal_after"] = int(env.current_goal)
                    out["steps"].append(rec)

    dst = os.path.join(HERE, "step_golden.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    byt = {}
    for r in out["steps"]:
        byt[r["task"]] = byt.get(r["task"], 0) + 1
    print("wrote", dst, byt, "registry", [r["id"] for r in out["registry"]])


if __name__ == "__main__":
    main()
