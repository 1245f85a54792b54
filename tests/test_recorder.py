"""CPU test of the episode recorder (on-disk layout of the reference's RecordHDF5Wrapper, record_hdf5.py:52-61,111)."""
import os

import numpy as np

from gym_lowcostrobot_amd import recorder


class _FakeEnv:
    """gymnasium-style env with the reference's observation keys; terminates every 5th step"""

    def __init__(self):
        self.t = 0

    def reset(self, seed=None, options=None):
        self.t = 0
        return self._obs(), {}

    def _obs(self):
        return {"arm_qpos": np.full(6, self.t, np.float32), "arm_qvel": np.zeros(6, np.float32),
                "image_front": np.full((240, 320, 3), self.t, np.uint8), "image_top": np.zeros((240, 320, 3), np.uint8)}

    def step(self, a):
        self.t += 1
        return self._obs(), -1.0, self.t % 5 == 0, False, {}

    def close(self):
        pass


def test_record_wrapper_layout_and_episode_split(tmp_path):
    env = recorder.RecordHDF5Wrapper(_FakeEnv(), str(tmp_path), name_prefix="demo", disable_logger=True)
    env.reset(seed=0)
    for t in range(12):
        env.step(np.full(5, 0.1 * t, np.float32))
    env.close()
    assert len(env.files) == 3                                        # 5 + 5 + 2 steps
    names = sorted(os.path.basename(f) for f in env.files)
    assert [n.rsplit(".", 1)[0] for n in names] == ["demo-episode-0", "demo-episode-1", "demo-episode-2"]  # record_hdf5.py:111
    ep0 = recorder.load_episode(sorted(env.files)[0])
    assert set(ep0) == set(recorder.DATASETS)                          # record_hdf5.py:52-61
    assert ep0["observations/images/front"].shape == (5, 240, 320, 3) and ep0["observations/images/front"].dtype == np.uint8
    assert ep0["observations/qpos"].shape == (5, 6) and ep0["action"].shape == (5, 5)
    np.testing.assert_array_equal(ep0["observations/qpos"][:, 0], [1, 2, 3, 4, 5])
    ep2 = recorder.load_episode(sorted(env.files)[2])
    assert ep2["action"].shape == (2, 5)


def test_record_wrapper_fixed_length(tmp_path):
    env = recorder.RecordHDF5Wrapper(_FakeEnv(), str(tmp_path), length=3, disable_logger=True)
    env.reset()
    for t in range(7):
        env.step(np.zeros(5, np.float32))
    env.close()
    assert len(env.files) == 1 and recorder.load_episode(env.files[0])["action"].shape == (3, 5)
