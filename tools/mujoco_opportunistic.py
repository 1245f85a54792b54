#!/usr/bin/env python3
"""Opportunistic real-MuJoCo check.  MuJoCo is NOT installed in the build image or on the
GPU boxes of this project, so normally this prints "reference MuJoCo unavailable".  Where `import mujoco` does succeed it
  * writes a from-scratch MJCF of the same model this repo simulates (numbers of follower.xml / reach_cube.xml, but the 20
    STL meshes replaced by the two finger spheres of deviation D3 -- no reference file is read),
  * times 20 x mj_step per control step on one env (the reference's hot loop, reach_cube_env.py:276-279),
  * and, if the CPU oracle is built, reports the state gap after one control step from the same (qpos, qvel, ctrl).
Nothing here is imported by the product path.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.gen_model_header import (LINK_AXIS, LINK_DIAGI, LINK_IPOS, LINK_IQUAT, LINK_MASS, LINK_POS, SITE_POS,  # noqa: E402
                                    SPH_LINK, SPH_POS, SPH_RAD)

JOINT_RANGE = [(-3.14, 3.14)] * 5 + [(-2.45, 0.032)]


def build_mjcf(cube_mass=0.1, cube_inertia=0.00016667, impratio=100):
    def v(t):
        return " ".join(repr(float(x)) for x in t)

    body_open, body_close = [], []
    for i in range(6):
        geoms = ""
        for s, link in enumerate(SPH_LINK):
            if link == i:
                geoms += (f'<geom name="finger_{s}" type="sphere" size="{SPH_RAD[s]}" pos="{v(SPH_POS[s])}" mass="0" priority="1" '
                          f'condim="4" solimp="0.015 1 0.036" friction="1.5"/>')
        site = f'<site name="end_effector_site" pos="{v(SITE_POS)}" size="0.001"/>' if i == 4 else ""
        body_open.append(
            f'<body name="link_{i + 1}" pos="{v(LINK_POS[i])}">'
            f'<inertial pos="{v(LINK_IPOS[i])}" quat="{v(LINK_IQUAT[i])}" mass="{LINK_MASS[i]}" diaginertia="{v(LINK_DIAGI[i])}"/>'
            f'<joint name="joint_{i + 1}" axis="{v(LINK_AXIS[i])}" range="{JOINT_RANGE[i][0]} {JOINT_RANGE[i][1]}" armature="0.1" '
            f'damping="1" actuatorfrcrange="-10 10"/>{geoms}{site}')
        body_close.append("</body>")
    acts = "".join(f'<position name="a{i + 1}" joint="joint_{i + 1}" kp="1000" kv="10" inheritrange="1"/>' for i in range(6))
    return f"""<mujoco model="lcr_restated">
  <compiler angle="radian"/>
  <option integrator="implicitfast" cone="elliptic" impratio="{impratio}" timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 0.05" friction="0.1"/>
    <body name="base_link" quat="-0.707 0 0 0.707">{''.join(body_open)}{''.join(body_close)}</body>
    <body name="cube" pos="0 0.2 0.015">
      <freejoint/>
      <inertial pos="0 0 0" mass="{cube_mass}" diaginertia="{cube_inertia} {cube_inertia} {cube_inertia}"/>
      <geom name="red_box" type="box" size="0.015 0.015 0.015" friction="0.5" condim="4" priority="1"/>
    </body>
  </worldbody>
  <actuator>{acts}</actuator>
</mujoco>"""


def run(steps=2000):
    try:
        import mujoco
    except Exception:
        return {"status": "reference MuJoCo unavailable"}
    import numpy as np

    try:
        model = mujoco.MjModel.from_xml_string(build_mjcf())
        data = mujoco.MjData(model)
        rng = np.random.default_rng(0)
        out = {"status": "MuJoCo CPU (opportunistic)", "mujoco_version": mujoco.__version__}
        # one control step from a common state vs the oracle
        try:
            from oracle import orc

            o = orc.Oracle("reach", 1, auto_reset=0, max_episode_steps=0, pgs_iters=50)
            o.reset(seeds=[0])
            o.qpos[0, 6:9] = [0.05, 0.2, 0.0149]
            data.qpos[:13] = o.qpos[0, :13]
            data.qvel[:] = 0
            a = rng.uniform(-1, 1, (1, 5)).astype(np.float32)
            ctrl = np.clip(a[0] + data.qpos[:5], [-3.14159, -1.5708, -1.48353, -1.91986, -2.96706], [3.14159, 1.22173, 1.74533, 1.91986, 2.96706])
            data.ctrl[:5] = ctrl
            data.ctrl[5] = 0
            for _ in range(20):
                mujoco.mj_step(model, data)
            o.step(a)
            out["max_abs_qpos_gap_after_one_control_step"] = float(np.abs(data.qpos[:13] - o.qpos[0, :13]).max())
        except Exception as e:  # the oracle is optional here
            out["oracle_gap_error"] = repr(e)
        t0 = time.perf_counter()
        for _ in range(steps):
            data.ctrl[:5] = np.clip(rng.uniform(-1, 1, 5) + data.qpos[:5], -3.1, 3.1)
            for _ in range(20):
                mujoco.mj_step(model, data)
        dt = time.perf_counter() - t0
        out.update({"value": steps / dt, "unit": "env-steps/s", "cores": 1, "sample": f"{steps} control steps x 20 mj_step, 1 env"})
        return out
    except Exception as e:
        return {"status": "MuJoCo present but the opportunistic harness failed", "error": repr(e)}


if __name__ == "__main__":
    print(json.dumps(run()))
