"""Kinematics / inertia fixtures that involve neither the oracle nor the kernels (VERDICT r4 next #4a): tests/golden/kin_golden.json holds forward kinematics,
site Jacobian, joint-space inertia and gravity torque of 256 poses, computed by tests/golden/kin_numpy.py in plain numpy from the numbers `mk_model` extracted
from follower.xml.  The oracle (CPU) and the HIP path (GPU) are both held to it; the same numpy code serves as mj_forward / mj_jacSite of the stand-in mujoco
module under which the reference's own IK loop and step() produce the ee-mode fixtures."""
import json
import os
import sys

import numpy as np
import pytest

from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import kin_numpy  # noqa: E402

with open(os.path.join(HERE, "golden", "kin_golden.json")) as f:
    KIN = json.load(f)
POSES = KIN["poses"]


def test_fixture_is_what_the_numpy_code_produces_and_matches_the_survey_table():
    again = kin_numpy.make_kin_golden()
    assert len(again["poses"]) == len(POSES) == 256
    for a, b in zip(again["poses"], POSES):
        for k in a:
            np.testing.assert_allclose(a[k], b[k], rtol=0, atol=1e-15)
    # known answers derived from the XML constants (the first three poses of the fixture)
    np.testing.assert_allclose(POSES[0]["site"], [0.002017, 0.212570, 0.168400], atol=5e-7)
    np.testing.assert_allclose(POSES[1]["site"], [0.050143, 0.142755, -0.037193], atol=5e-7)
    np.testing.assert_allclose(POSES[2]["site"], [0.061692, 0.028102, 0.223741], atol=5e-7)
    np.testing.assert_allclose(POSES[0]["gravity_torque"], [0, -0.147385, 0.129713, -0.034035, -0.000725, 0], atol=5e-7)
    np.testing.assert_allclose(POSES[0]["site_jac"][2], [0, 0.22457, -0.20977, 0.10929, 0.00327, 0], atol=5e-6)
    M0 = np.array(POSES[0]["mass_matrix"]) - 0.1 * np.eye(6)
    np.testing.assert_allclose(np.diag(M0), [2.229e-3, 4.037e-3, 1.776e-3, 2.34e-4, 6e-6, 1.1e-5], rtol=2e-2, atol=1e-6)
    # the Jacobian is the derivative of the forward kinematics (central differences)
    arm = kin_numpy.Arm()
    for p in POSES[:16]:
        q = np.array(p["q"])
        J = np.array(p["site_jac"])
        for j in range(6):
            d = np.zeros(6); d[j] = 1e-6
            np.testing.assert_allclose((arm.site(q + d) - arm.site(q - d)) / 2e-6, J[:, j], atol=2e-9)


def test_oracle_kinematics_and_inertia_vs_numpy_fixture():
    for p in POSES:
        q = np.array(p["q"])
        lp, site, _ = orc.fk(q)
        np.testing.assert_allclose(site, p["site"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(lp, p["link_origins"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(orc.site_jac(q), p["site_jac"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(orc.mass_matrix(q), p["mass_matrix"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(-orc.bias(q, np.zeros(6)), p["gravity_torque"], rtol=0, atol=1e-12)


def test_reference_ik_fixtures_use_the_numpy_kinematics():
    """the 24 golden cases of the reference's IK loop were produced with kin_numpy as mj_forward / mj_jacSite: their recorded site positions are its FK"""
    with open(os.path.join(HERE, "golden", "glue_golden.json")) as f:
        ik = json.load(f)["ik"]
    arm = kin_numpy.Arm()
    for c in ik:
        np.testing.assert_allclose(arm.site(np.array(c["qpos_after"])), c["site_after"], rtol=0, atol=1e-15)
    src = open(os.path.join(HERE, "golden", "make_golden.py")).read() + open(os.path.join(HERE, "golden", "make_step_golden.py")).read()
    assert "from oracle" not in src and "import oracle" not in src      # the generators no longer touch the oracle


@pytest.mark.gpu
def test_hip_kinematics_and_inertia_vs_numpy_fixture(hip_lib):
    """HIP path, no oracle in between: from a placed pose at rest one substep with action 0 -- the lagged site position the kernel reports IS its forward kinematics
    (P8), and, nothing touching, the joint velocities after the substep are h (M + h (damping + kv) I)^-1 (gravity torque + kp (ctrl - q)): mass matrix and gravity
    torque of the fixture, joint mode targets of reach_cube_env.py:248-250"""
    from gym_lowcostrobot_amd import VecSim

    TLO = np.array([-3.14159, -1.5708, -1.48353, -1.91986, -2.96706, -1.74533]); THI = np.array([3.14159, 1.22173, 1.74533, 1.91986, 2.96706, 0.0523599])
    JLO = np.array([-3.14, -3.14, -3.14, -3.14, -3.14, -2.45]); JHI = np.array([3.14, 3.14, 3.14, 3.14, 3.14, 0.032])
    keep = [p for p in POSES if min(np.array(p["link_origins"])[2:, 2].min(), p["site"][2]) > 0.07]    # wrist and fingers well above the floor: no contact
    assert len(keep) >= 40
    n = len(keep)
    sim = VecSim("reach", n, observation_mode="state", n_substeps=1, auto_reset=False, max_episode_steps=0)
    sim.reset(seeds=np.arange(n, dtype=np.uint64))
    q = np.array([p["q"] for p in keep], np.float32).astype(np.float64)         # float32-representable
    st = sim.get_state()
    st["qpos"][:6] = q.T
    st["qpos"][6:9] = np.array([[0.0], [0.6], [0.015]])                           # the cube rests far away
    st["qvel"][:] = 0
    sim.set_state(qpos=st["qpos"], qvel=st["qvel"])
    sim.step(np.zeros((n, sim.action_dim), np.float32))
    out = sim.get_state()
    arm = kin_numpy.Arm()
    h = 0.002
    for i, p in enumerate(keep):
        qi = q[i]
        np.testing.assert_allclose(out["ee_lag"][:, i], arm.site(qi), rtol=0, atol=2e-6)      # (fp32 kernel; q itself was rounded to float32)
        ctrl = np.clip(np.clip(qi, TLO, THI), JLO, JHI); ctrl[5] = 0.0                          # reach: gripper target 0 (reach:255), then MuJoCo's ctrlrange clamp
        tau = np.clip(1000.0 * (ctrl - qi), -10, 10) + arm.gravity_torque(qi)
        qacc = np.linalg.solve(arm.mass_matrix(qi) + h * 11.0 * np.eye(6), tau)
        np.testing.assert_allclose(out["qvel"][:6, i], h * qacc, rtol=2e-4, atol=2e-6)
    sim.close()
