"""Generate golden vectors by IMPORTING the reference's pure-numpy glue (no source is copied).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
Writes tests/golden/glue_golden.json.

`mujoco` and `gymnasium` are absent here, so tiny stand-in modules are injected into sys.modules
just far enough for `import gym_lowcostrobot.envs.*` to succeed; objects are made with
`Cls.__new__` and only methods that are pure numpy are called:
  * goal_distance / is_success / compute_reward      (reach:335-348, push:348-361, pick_place:356-369, stack:350-363)
  * the reset() sampling arithmetic                   (reach:297-306, push:308-322, pick_place:316-330, stack:307-319)
    - reset() itself is run with a stand-in `data` whose qpos is a numpy array and with
      `mujoco.mj_forward` a no-op, so the qpos it WRITES and the target it samples are the reference's.
  * joint-mode target arithmetic of apply_action      (reach:248-273, lift:258-282) with mj_step a no-op.
The physics (mujoco.mj_step) cannot be run: "parity unpinned" for it.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"


def _install_stubs():
    mj = types.ModuleType("mujoco")
    mj.viewer = types.ModuleType("mujoco.viewer")
    mj.mj_forward = lambda m, d: None
    mj.mj_step = lambda m, d: None
    sys.modules["mujoco"] = mj
    sys.modules["mujoco.viewer"] = mj.viewer

    gym = types.ModuleType("gymnasium")

    class Env:
        def reset(self, seed=None, options=None):
            if seed is not None:
                # gymnasium.utils.seeding.np_random: Generator(PCG64(SeedSequence(seed)))
                self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype)
            self.dtype = dtype

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = Box
    spaces.Dict = dict
    gym.Env = Env
    gym.spaces = spaces
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda **kw: None
    envs = types.ModuleType("gymnasium.envs")
    envs.registration = reg
    gym.envs = envs
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces
    sys.modules["gymnasium.envs"] = envs
    sys.modules["gymnasium.envs.registration"] = reg
    return Box


class _FakeGeom:
    pos = None


class _FakeModel:
    def __init__(self):
        self._g = _FakeGeom()
        # follower.xml:58-95 joint ranges == actuator ctrlrange (inheritrange=1)
        self.jnt_range = np.array([[-3.14, 3.14]] * 5 + [[-2.45, 0.032]])
        self.actuator_ctrlrange = self.jnt_range.copy()

    def geom(self, name):
        return self._g


class _FakeData:
    def __init__(self, nq):
        self.qpos = np.zeros(nq)
        self.qvel = np.zeros(nq - 1)
        self.ctrl = np.zeros(6)


def main():
    Box = _install_stubs()
    sys.path.insert(0, REF)
    from gym_lowcostrobot.envs.lift_cube_env import LiftCubeEnv
    from gym_lowcostrobot.envs.pick_place_cube_env import PickPlaceCubeEnv
    from gym_lowcostrobot.envs.push_cube_env import PushCubeEnv
    from gym_lowcostrobot.envs.reach_cube_env import ReachCubeEnv
    from gym_lowcostrobot.envs.stack_two_cubes_env import StackTwoCubesEnv

    out = {"rewards": [], "resets": [], "joint_targets": []}
    rng = np.random.default_rng(12345)

    # ---- rewards / success --------------------------------------------------------------
    for cls, name, b_is_f32 in [
        (ReachCubeEnv, "reach", False),
        (PushCubeEnv, "push", True),
        (PickPlaceCubeEnv, "pick_place", True),
        (StackTwoCubesEnv, "stack", False),
    ]:
        for reward_type in ("sparse", "dense"):
            env = cls.__new__(cls)
            env.distance_threshold = 0.05
            env.reward_type = reward_type
            cases = [
                (np.array([0.0, 0.2, 0.1]), np.array([0.0, 0.2, 0.06])),
                (np.array([0.1, 0.2, 0.015]), np.array([0.0, 0.1, 0.0])),
                (np.array([0.0, 0.0, 0.0]), np.array([0.05, 0.0, 0.0])),  # d == threshold (neither < nor >)
                (np.array([0.0, 0.0, 0.0]), np.array([0.03, 0.04, 0.0])),  # 3-4-5 -> 0.05 up to rounding
            ]
            for _ in range(12):
                a = rng.uniform(-0.2, 0.3, 3)
                b = a + rng.normal(0, 0.04, 3)
                cases.append((a, b))
            for a, b in cases:
                bb = b.astype(np.float32) if b_is_f32 else b
                r = env.compute_reward(a, bb)
                s = env.is_success(a, bb)
                out["rewards"].append(
                    {
                        "task": name,
                        "reward_type": reward_type,
                        "a": a.tolist(),
                        "b": [float(x) for x in bb],
                        "b_is_f32": b_is_f32,
                        "reward": float(r),
                        "reward_dtype": str(np.asarray(r).dtype),
                        "reward_signbit": bool(np.signbit(r)),
                        "is_success": bool(s),
                    }
                )

    # ---- reset sampling -----------------------------------------------------------------
    def mk(cls, nq, **kw):
        env = cls.__new__(cls)
        env.model = _FakeModel()
        env.data = _FakeData(nq)
        env.num_dof = 6
        env.observation_mode = "state"
        rng_xy = kw.get("cube_xy_range", 0.3)
        env.cube_low = np.array([-rng_xy / 2, -rng_xy / 2, 0])
        env.cube_high = np.array([rng_xy / 2, rng_xy / 2, 0])
        env.cube_low[1] += 0.165
        env.cube_high[1] += 0.10
        if "target_z" in kw:
            t = kw.get("target_xy_range", 0.3)
            env.target_low = np.array([-t / 2, -t / 2, 0])
            env.target_high = np.array([t / 2, t / 2, kw["target_z"]])
            env.target_low[1] += 0.165
            env.target_high[1] += 0.10
        return env

    for cls, name, nq, kw in [
        (ReachCubeEnv, "reach", 13, {}),
        (LiftCubeEnv, "lift", 13, {}),
        (PushCubeEnv, "push", 13, {"target_z": 0.0}),
        (PickPlaceCubeEnv, "pick_place", 13, {"target_z": 0.1}),
        (StackTwoCubesEnv, "stack", 20, {}),
    ]:
        for seed in (0, 1, 42, 2**31 - 1, 2**40 + 7):
            env = mk(cls, nq, **kw)
            env.data.qpos[:] = 0.123  # must be overwritten by reset for [0:13] / [0:20]
            seq = []
            obs, info = env.reset(seed=seed)
            rec = {"qpos": env.data.qpos.tolist(), "obs": {k: [float(x) for x in v] for k, v in obs.items()}}
            if hasattr(env, "target_pos"):
                rec["target_pos"] = [float(x) for x in env.target_pos]
            seq.append(rec)
            for _ in range(3):  # un-seeded resets continue the same generator stream
                obs, info = env.reset()
                rec = {"qpos": env.data.qpos.tolist(), "obs": {k: [float(x) for x in v] for k, v in obs.items()}}
                if hasattr(env, "target_pos"):
                    rec["target_pos"] = [float(x) for x in env.target_pos]
                seq.append(rec)
            out["resets"].append({"task": name, "seed": seed, "sequence": seq})

    # ---- joint-mode control targets (apply_action with mj_step stubbed) ------------------
    for cls, name, gripper in [(ReachCubeEnv, "reach", False), (LiftCubeEnv, "lift", True)]:
        env = mk(cls, 13)
        env.action_mode = "joint"
        env.render_mode = None
        env.control_decimation = 1
        k = 6 if gripper else 5
        env.action_space = Box(-1.0, 1.0, shape=(k,), dtype=np.float32)
        for _ in range(16):
            env.data.qpos[:6] = rng.uniform(-2.0, 2.0, 6)
            env.data.qpos[5] = rng.uniform(-2.45, 0.032)
            act = rng.uniform(-1.5, 1.5, k).astype(np.float32)
            env.apply_action(act)
            out["joint_targets"].append(
                {
                    "task": name,
                    "qpos": env.data.qpos[:6].tolist(),
                    "action": [float(x) for x in act],
                    "ctrl": [float(x) for x in np.asarray(env.data.ctrl, dtype=np.float64)],
                }
            )

    # ---- ee-mode glue of apply_action (reach:236-247, lift:241-257) with the MuJoCo-dependent IK stubbed out ----------
    class _Site:
        def __init__(self, xpos):
            self.xpos = xpos
            self.id = 0

    out["ee_glue"] = []
    for cls, name, gripper in [(ReachCubeEnv, "reach", False), (LiftCubeEnv, "lift", True), (PickPlaceCubeEnv, "pick_place", True)]:
        for _ in range(12):
            env = mk(cls, 13)
            env.action_mode = "ee"
            env.render_mode = None
            env.control_decimation = 1
            env.ctrl_range = env.model.actuator_ctrlrange
            k = 4 if gripper else 3
            env.action_space = Box(-1.0, 1.0, shape=(k,), dtype=np.float32)
            site = rng.uniform(-0.2, 0.3, 3)
            site[2] = rng.uniform(-0.01, 0.2)
            env.model.site = lambda n, _s=site: _Site(_s)
            env.data.site = lambda i, _s=site: _Site(_s.copy())
            env.data.qpos[:6] = rng.uniform(-1.0, 1.0, 6)
            env.data.qpos[5] = rng.uniform(-2.45, 0.032)
            captured = {}

            def fake_ik(ee_target_pos, _c=captured, _e=env):
                _c["target"] = np.array(ee_target_pos, dtype=np.float64).copy()
                return _e.data.qpos[:6].copy()

            env.inverse_kinematics = fake_ik
            act = rng.uniform(-1.4, 1.4, k).astype(np.float32)
            q5 = float(env.data.qpos[5])
            env.apply_action(act)
            out["ee_glue"].append({"task": name, "site": site.tolist(), "q5": q5, "action": [float(x) for x in act],
                                   "target": captured["target"].tolist(), "ctrl5": float(np.asarray(env.data.ctrl)[5])})

    # ---- inverse_kinematics loop (reach:148-221) run UNMODIFIED, with mj_forward / mj_jacSite / site xpos served by
    #      plain-numpy kinematics derived from the extracted model numbers alone (tests/golden/kin_numpy.py: no oracle, no kernel code -- VERDICT r4 weak #1a).
    #      Pins the reference's own loop arithmetic: DLS solve, unit-norm clamp, 0.5 step, joint limits, early break,
    #      and the qpos overwrite (REF-QUIRK-3).
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kin_numpy
    _arm = kin_numpy.Arm()
    import mujoco as _mj

    class _IKData(_FakeData):
        def __init__(self):
            super().__init__(13)
            self._site = np.zeros(3)

        def site(self, i):
            return _Site(self._site)

    def _mj_forward(model, data):
        data._site[:] = _arm.site(np.array(data.qpos[:6], dtype=np.float64))

    def _mj_jacsite(model, data, jacp, jacr, sid):
        jacp[:] = 0
        jacp[:, :6] = _arm.site_jac(np.array(data.qpos[:6], dtype=np.float64))

    _mj.mj_forward = _mj_forward
    _mj.mj_jacSite = _mj_jacsite
    out["ik"] = []
    for _ in range(24):
        env = mk(ReachCubeEnv, 13)
        env.model.nv = 12
        env.model.site = lambda n: _Site(np.zeros(3))
        env.data = _IKData()
        q0 = rng.uniform(-1.2, 1.2, 6)
        q0[5] = rng.uniform(-2.45, 0.06)  # occasionally beyond the upper gripper limit: IK clamps it (teleport of joint 6)
        env.data.qpos[:6] = q0
        site0 = _arm.site(q0)
        reach = rng.choice([0.004, 0.03, 0.08, 0.25])
        tgt = site0 + rng.normal(0, 1, 3) * reach
        tgt[2] = max(0.0, tgt[2])
        qt = env.inverse_kinematics(ee_target_pos=tgt.copy())
        out["ik"].append({"q0": q0.tolist(), "target": tgt.tolist(), "q_target": np.asarray(qt).tolist(),
                          "qpos_after": env.data.qpos[:6].tolist(), "site_after": env.data._site.tolist()})
    _mj.mj_forward = lambda m, d: None

    # ---- PushCubeLoop-v0: overlap / reward / goal switching / reset sampling (push_cube_loop_env.py:299-383) -----
    from gym_lowcostrobot.envs.push_cube_loop_env import PushCubeLoopEnv

    def mk_loop():
        env = PushCubeLoopEnv.__new__(PushCubeLoopEnv)
        env.model = _FakeModel()
        env.data = _FakeData(13)
        env.num_dof = 6
        env.observation_mode = "state"
        # constants exactly as __init__ derives them (push_cube_loop_env.py:124-135) from push_cube_loop.xml:38,41
        env.cube_size = 0.015 / 2
        env.cube_position = np.array([0.0, 0.0, 0.0])
        env.goal_region_1_center = np.array([0.06, 0.135, 0.01])
        env.goal_region_2_center = np.array([-0.06, 0.135, 0.01])
        env.goal_region_high = np.array([0.035, 0.045, 0.007]) / 2
        env.goal_region_high[:2] -= 0.008
        env.goal_region_low = env.goal_region_high * np.array([-1.0, -1.0, 1.0])
        env.current_goal = 0
        env._step = 0
        return env

    out["loop_consts"] = {}
    env = mk_loop()
    out["loop_consts"] = {"goal_region_high": env.goal_region_high.tolist(), "goal_region_low": env.goal_region_low.tolist(),
                          "cube_size": env.cube_size}
    out["loop_rewards"] = []
    pts = [(0.06, 0.135), (-0.06, 0.135), (0.0605, 0.1353), (0.066, 0.14), (0.07, 0.135), (0.0, 0.135), (0.05, 0.12),
           (0.0772, 0.135), (0.06, 0.1571), (0.06, 0.1121), (-0.0612, 0.1344), (0.1, 0.165)]
    for _ in range(20):
        pts.append((float(rng.uniform(-0.11, 0.11)), float(rng.uniform(0.10, 0.17))))
    for goal in (0, 1):
        for (x, y) in pts:
            env = mk_loop()
            env.current_goal = goal
            env.data.qpos[6:9] = [x, y, 0.015]
            env.cube_position = env.data.qpos[6:9].astype(np.float32).copy()  # what get_reward() does first (:336)
            overlap = float(env.get_cube_overlap())
            reward, success = env.get_reward()
            out["loop_rewards"].append({"goal": goal, "cube": [x, y, 0.015], "reward": float(reward), "success": int(success),
                                        "goal_after": int(env.current_goal), "overlap": overlap})
    out["loop_resets"] = []
    for seed in (0, 1, 42):
        for goal in (0, 1):
            env = mk_loop()
            env.current_goal = goal
            seq = []
            obs, info = env.reset(seed=seed)
            seq.append({"qpos": env.data.qpos.tolist(), "info": info})
            for _ in range(2):
                obs, info = env.reset()
                seq.append({"qpos": env.data.qpos.tolist(), "info": info})
            out["loop_resets"].append({"seed": seed, "goal": goal, "sequence": seq})

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_golden.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, {k: len(v) for k, v in out.items()})


# ------------------------------------------------------------------------------------------------------------------
# L0: the MJCF numbers themselves (follower.xml + the six scene files) and the extents of the collision / visual meshes.
# Parsed with ElementTree / a binary-STL reader; the output is DATA (attribute values, bounding boxes), no file is copied.
# ------------------------------------------------------------------------------------------------------------------
ASSETS = os.path.join(REF, "gym_lowcostrobot", "assets", "low_cost_robot_6dof")
SCENES = ["reach_cube", "lift_cube", "push_cube", "pick_place_cube", "stack_two_cubes", "push_cube_loop"]


def _nums(txt):
    return [float(x) for x in txt.split()]


def _attrs(el, keep_str=("name", "class", "mesh", "type", "joint", "file", "material", "mode", "integrator", "cone", "childclass",
                         "body1", "body2", "texture", "builtin", "mark", "angle", "meshdir")):
    out = {}
    for k, v in el.attrib.items():
        if k in keep_str:
            out[k] = v
        else:
            try:
                n = _nums(v)
                out[k] = n[0] if len(n) == 1 else n
            except ValueError:
                out[k] = v
    return out


def _body_tree(el, parent, bodies):
    rec = {"name": el.get("name"), "parent": parent, **{k: v for k, v in _attrs(el).items() if k != "name"}}
    rec["inertial"] = _attrs(el.find("inertial")) if el.find("inertial") is not None else None
    rec["joints"] = [_attrs(j) for j in el.findall("joint")] + [dict(_attrs(j), type="free") for j in el.findall("freejoint")]
    rec["geoms"] = [_attrs(g) for g in el.findall("geom")]
    rec["sites"] = [_attrs(g) for g in el.findall("site")]
    bodies.append(rec)
    for ch in el.findall("body"):
        _body_tree(ch, el.get("name"), bodies)


def _defaults(el, prefix, out):
    cls = el.get("class", prefix)
    out[cls] = {ch.tag: _attrs(ch) for ch in el if ch.tag != "default"}
    for ch in el.findall("default"):
        _defaults(ch, cls, out)


def _stl_vertices(path):
    import struct

    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    rec = np.frombuffer(b[84:84 + 50 * n], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    return rec["v"].reshape(-1, 3).astype(np.float64)


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def mk_model():
    """Numbers of the MJCF model that the build consumes and a test asserts (tests/test_model_constants.py, tools/gen_model_header.py,
    oracle/render_oracle.py) -- nothing else: no materials, mesh file names, visual classes, groups or colours."""
    import xml.etree.ElementTree as ET

    out = {"_generated_by": "tests/golden/make_golden.py:mk_model", "_source": "gym_lowcostrobot/assets/low_cost_robot_6dof/*.xml, follower_meshes/*.stl"}
    root = ET.parse(os.path.join(ASSETS, "follower.xml")).getroot()
    defaults, bodies = {}, []
    for d in root.find("default").findall("default"):
        _defaults(d, "main", defaults)
    for b in root.find("worldbody").findall("body"):
        _body_tree(b, "world", bodies)
    meshes = {m.get("name"): m.get("file") for m in root.find("asset").findall("mesh")}
    fol = {"option": _pick(_attrs(root.find("option")), ("integrator", "cone", "impratio", "timestep")),
           "defaults": {"follower": {"joint": _pick(defaults["follower"].get("joint"), ("armature", "damping", "actuatorfrcrange")),
                                     "position": _pick(defaults["follower"].get("position"), ("kp", "kv", "inheritrange"))},
                        "finger": {"geom": _pick(defaults["finger"].get("geom"), ("priority", "condim", "solimp", "friction"))}},
           "bodies": [{"name": b["name"], "parent": b["parent"], **_pick(b, ("pos", "quat")),
                       "inertial": _pick(b["inertial"], ("pos", "quat", "mass", "diaginertia")) if b["inertial"] else None,
                       "joints": [_pick(j, ("name", "axis", "range")) for j in b["joints"]],
                       "sites": [_pick(x, ("name", "pos")) for x in b["sites"]]} for b in bodies]}
    out["follower"] = fol
    out["scenes"] = {}
    for sc in SCENES:
        r = ET.parse(os.path.join(ASSETS, sc + ".xml")).getroot()
        wb = r.find("worldbody")
        sb = []
        for b in wb.findall("body"):
            _body_tree(b, "world", sb)
        geom_keys = ("name", "type", "size", "pos", "friction", "condim", "priority", "solref")
        rec = {"include_after_option": [c.tag for c in r].index("include") > [c.tag for c in r].index("option"),
               "world_geoms": [_pick(_attrs(g), geom_keys) for g in wb.findall("geom")],
               "cameras": [_pick(_attrs(c), ("name", "pos", "xyaxes", "euler", "quat")) for c in wb.findall("camera")],
               "bodies": [{"name": b["name"], **_pick(b, ("pos",)),
                           "inertial": _pick(b["inertial"], ("mass", "diaginertia")) if b["inertial"] else None,
                           "joints": [_pick(j, ("type",)) for j in b["joints"]],
                           "geoms": [_pick(g, geom_keys) for g in b["geoms"]]} for b in sb]}
        out["scenes"][sc] = rec
    # hull extents of the four links that carry sphere proxies: bounding box and slab extents along the link's long axis (x)
    mdir = os.path.join(ASSETS, "follower_meshes")
    aabb, slabs = {}, {}
    # (round 5: the bounding boxes of all seven collision hulls -- what the ray-casters draw for the arm; slabs only for the four links that carry proxies)
    for name in ("base_link_collision", "link_1_collision", "link_2_collision"):
        v = _stl_vertices(os.path.join(mdir, meshes[name]))
        aabb[name] = {"min": [round(float(x), 6) for x in v.min(0)], "max": [round(float(x), 6) for x in v.max(0)]}
    for name in ("link_3_collision", "link_4_collision", "link_5_collision", "link_6_collision"):
        v = _stl_vertices(os.path.join(mdir, meshes[name]))
        aabb[name] = {"min": [round(float(x), 6) for x in v.min(0)], "max": [round(float(x), 6) for x in v.max(0)]}
        edges = np.linspace(v[:, 0].min(), v[:, 0].max(), 9)
        rows = []
        for a, b in zip(edges[:-1], edges[1:]):
            w = v[(v[:, 0] >= a - 1e-9) & (v[:, 0] <= b + 1e-9)]
            if len(w):
                rows.append({"x": [round(float(a), 5), round(float(b), 5)], "y": [round(float(w[:, 1].min()), 5), round(float(w[:, 1].max()), 5)],
                             "z": [round(float(w[:, 2].min()), 5), round(float(w[:, 2].max()), 5)]})
        slabs[name] = rows
    out["mesh_aabb"] = aabb
    out["mesh_slabs_x"] = slabs
    return out


if __name__ == "__main__":
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_golden.json"), "w") as _f:
        json.dump(mk_model(), _f, indent=1)
    print("wrote model_golden.json")
    main()
