"""Kinematics and inertia of the follower arm in plain numpy, straight from tests/golden/model_golden.json (the numbers `mk_model` extracted from
follower.xml) -- NO oracle, NO kernel code behind it.  Serves (i) as `mj_forward` / `mj_jacSite` of the stand-in `mujoco` module under which the fixture
generators run the reference's own env code (make_golden.py, make_step_golden.py: VERDICT r4 weak #1a -- the ee-mode fixtures were circular in the
kinematics while those stubs called the oracle) and (ii) as the generator of tests/golden/kin_golden.json: 256 random poses with forward kinematics, site
Jacobian, joint-space inertia and gravity torque, against which the oracle AND the HIP path are held (tests/test_kin_golden.py).

Conventions (MuJoCo's, MJ-DOC): a body frame = parent frame * (pos, quat) * joint rotation (hinge about `axis`, given in the body frame, through the body
origin); quaternions (w, x, y, z) are normalised on load; `inertial` = centre of mass, principal-axes orientation and principal moments.
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GRAV = 9.81


def _quat2mat(q):
    w, x, y, z = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _rot(axis, th):
    a = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


class Arm:
    def __init__(self, path=None):
        with open(path or os.path.join(_HERE, "model_golden.json")) as f:
            m = json.load(f)["follower"]
        self.armature = m["defaults"]["follower"]["joint"]["armature"]
        bodies = {b["name"]: b for b in m["bodies"]}
        self.base = bodies["base_link"]
        self.links = [bodies[f"link_{i}"] for i in range(1, 7)]
        assert [b["parent"] for b in self.links] == ["base_link"] + [f"link_{i}" for i in range(1, 6)]   # a serial chain
        self.site_body = next(i for i, b in enumerate(self.links) if b["sites"])
        self.site_pos = np.array(self.links[self.site_body]["sites"][0]["pos"])

    def frames(self, q):
        """world rotation R_i, origin p_i and joint axis a_i of link_1..link_6 at joint angles q (6)"""
        R = _quat2mat(self.base.get("quat", [1, 0, 0, 0]))
        p = np.array(self.base.get("pos", [0.0, 0.0, 0.0]), np.float64)
        out = []
        for b, th in zip(self.links, q):
            p = p + R @ np.array(b.get("pos", [0, 0, 0]), np.float64)
            R = R @ _quat2mat(b.get("quat", [1, 0, 0, 0]))
            axis = np.array(b["joints"][0]["axis"], np.float64)
            a_world = R @ (axis / np.linalg.norm(axis))   # (the axis is invariant under the joint's own rotation)
            R = R @ _rot(axis, th)
            out.append((R, p.copy(), a_world))
        return out

    def site(self, q):
        R, p, _ = self.frames(q)[self.site_body]
        return p + R @ self.site_pos

    def link_origins(self, q):
        return np.array([p for _, p, _ in self.frames(q)])

    def _point_jac(self, fr, body, point):
        """3 x 6 translational and rotational Jacobians of a point fixed to link `body` (0-based)"""
        Jp, Jr = np.zeros((3, 6)), np.zeros((3, 6))
        for j in range(body + 1):
            _, pj, aj = fr[j]
            Jp[:, j] = np.cross(aj, point - pj)
            Jr[:, j] = aj
        return Jp, Jr

    def site_jac(self, q):
        fr = self.frames(q)
        R, p, _ = fr[self.site_body]
        return self._point_jac(fr, self.site_body, p + R @ self.site_pos)[0]

    def mass_matrix(self, q, armature=True):
        fr = self.frames(q)
        M = np.zeros((6, 6))
        for i, b in enumerate(self.links):
            R, p, _ = fr[i]
            ine = b["inertial"]
            c = p + R @ np.array(ine["pos"])
            Ri = R @ _quat2mat(ine["quat"])
            Iw = Ri @ np.diag(ine["diaginertia"]) @ Ri.T
            Jp, Jr = self._point_jac(fr, i, c)
            M += ine["mass"] * Jp.T @ Jp + Jr.T @ Iw @ Jr
        if armature:
            M += self.armature * np.eye(6)
        return M

    def gravity_torque(self, q):
        """generalized force of gravity (= -qfrc_bias at zero velocity)"""
        fr = self.frames(q)
        tau = np.zeros(6)
        for i, b in enumerate(self.links):
            R, p, _ = fr[i]
            c = p + R @ np.array(b["inertial"]["pos"])
            Jp, _ = self._point_jac(fr, i, c)
            tau += -b["inertial"]["mass"] * GRAV * Jp[2]
        return tau


def make_kin_golden(n=256, seed=20260930):
    arm = Arm()
    rng = np.random.default_rng(seed)
    lo = np.array([-3.14, -3.14, -3.14, -3.14, -3.14, -2.45])
    hi = np.array([3.14, 3.14, 3.14, 3.14, 3.14, 0.032])
    poses = []
    for k in range(n):
        q = lo + (hi - lo) * rng.uniform(0, 1, 6) if k >= 3 else [np.zeros(6), np.array([0.3, -0.5, 0.8, 0.4, -0.2, -0.5]), np.array([1.0, 1, 1, 1, 1, 0])][k]   # (the three poses of SURVEY.md 8(c) first)
        poses.append({"q": np.asarray(q).tolist(), "site": arm.site(q).tolist(), "link_origins": arm.link_origins(q).tolist(), "site_jac": arm.site_jac(q).tolist(),
                      "mass_matrix": arm.mass_matrix(q).tolist(), "gravity_torque": arm.gravity_torque(q).tolist()})
    return {"_generated_by": "tests/golden/kin_numpy.py:make_kin_golden (numpy, from model_golden.json only)", "armature": arm.armature, "poses": poses}


if __name__ == "__main__":
    out = os.path.join(_HERE, "kin_golden.json")
    with open(out, "w") as f:
        json.dump(make_kin_golden(), f)
    print("wrote", out)
