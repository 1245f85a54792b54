"""Episode recorder writing the on-disk layout of the reference's RecordHDF5Wrapper
(gym_lowcostrobot/envs/wrappers/record_hdf5.py:52-61,111):

    one file per episode  "{name_prefix}-episode-{id}.hdf5"  with datasets
        observations/images/front   (T, 240, 320, 3) uint8      <- obs["image_front"]
        observations/images/top     (T, 240, 320, 3) uint8      <- obs["image_top"]
        observations/qpos           (T, 6) float32               <- obs["arm_qpos"]
        observations/qvel           (T, 6) float32               <- obs["arm_qvel"]
        action                      (T, k) float32

Back ends, first one available: h5py (what the reference uses); the HDF5 C library itself through ctypes (`_hdf5c.py`:
h5py is a wrapper of the same library, the files are the same -- this project's build image and GPU boxes have libhdf5 1.10
under /opt/conda but no h5py); if neither loads the same arrays go to "{...}.npz" under the same dataset names, so
downstream code only switches the loader.  `backend()` says which one is in use.

`RecordHDF5Wrapper` wraps ONE gymnasium-style env (the reference's usage, examples/hdf5_record.py:9-21);
`VecRecorder` records a chosen subset of a batched VecSim, one file per (env, episode).
"""
import os
import warnings

import numpy as np

from . import _hdf5c

try:  # pragma: no cover - depends on the environment
    import h5py
except Exception:
    h5py = None

DATASETS = ("observations/images/front", "observations/images/top", "observations/qpos", "observations/qvel", "action")


def backend():
    """"h5py ..." / "libhdf5 ..." (real .hdf5 files) or "npz" (fallback)"""
    if h5py is not None:
        return f"h5py {h5py.__version__}"
    return _hdf5c.version() or "npz"


def write_episode(path_hdf5, observations, actions):
    """observations: list of dicts (arm_qpos, arm_qvel, optional image_front/image_top); actions: list of arrays."""
    data = {
        "observations/qpos": np.stack([o["arm_qpos"] for o in observations]),
        "observations/qvel": np.stack([o["arm_qvel"] for o in observations]),
        "action": np.stack([np.asarray(a, np.float32) for a in actions]),
    }
    if "image_front" in observations[0]:
        data["observations/images/front"] = np.stack([o["image_front"] for o in observations])
        data["observations/images/top"] = np.stack([o["image_top"] for o in observations])
    if h5py is not None:
        with h5py.File(path_hdf5, "w") as f:
            for k, v in data.items():
                f.create_dataset(k, data=v)
        return path_hdf5
    if _hdf5c.available():
        return _hdf5c.write_file(path_hdf5, data)
    path = os.path.splitext(path_hdf5)[0] + ".npz"
    np.savez(path, **data)
    return path


def load_episode(path):
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if h5py is not None:  # pragma: no cover
        with h5py.File(path, "r") as f:
            return {k: f[k][()] for k in DATASETS if k in f}
    return _hdf5c.read_file(path, DATASETS)


def hdf5_selftest():
    """Check of the real file format: with h5py or libhdf5 available, write a two-frame episode as .hdf5, read it back and compare
    dataset by dataset; otherwise say that no .hdf5 byte was written (the .npz fallback carries the same dataset names)."""
    if h5py is None and not _hdf5c.available():
        return {"status": "neither h5py nor libhdf5 on this box: no .hdf5 file written (episodes fall back to .npz with the HDF5 dataset names)"}
    import tempfile

    rng = np.random.default_rng(0)
    obs = [{"arm_qpos": rng.normal(size=6).astype(np.float32), "arm_qvel": rng.normal(size=6).astype(np.float32),
            "image_front": rng.integers(0, 255, (240, 320, 3), dtype=np.uint8), "image_top": rng.integers(0, 255, (240, 320, 3), dtype=np.uint8)}
           for _ in range(2)]
    act = [rng.uniform(-1, 1, 5).astype(np.float32) for _ in range(2)]
    with tempfile.TemporaryDirectory() as d:
        path = write_episode(os.path.join(d, "selftest-episode-0.hdf5"), obs, act)
        back = load_episode(path)
        ok = path.endswith(".hdf5") and set(back) == set(DATASETS)
        ok = ok and np.array_equal(back["observations/qpos"], np.stack([o["arm_qpos"] for o in obs])) and np.array_equal(back["action"], np.stack(act))
        ok = ok and np.array_equal(back["observations/images/front"], np.stack([o["image_front"] for o in obs]))
        size = os.path.getsize(path)
    return {"status": "hdf5 written and read back" if ok else "hdf5 round trip MISMATCH", "bytes": size, "backend": backend()}


class RecordHDF5Wrapper:
    """Single-env recorder with the reference's constructor arguments (record_hdf5.py:66-73).

    `length = 0`: one file per episode, a new one after every terminated / truncated step and after every reset() (record_hdf5.py:131-134).
    `length > 0`: the first `length` frames after a reset() go to that episode's file (the reference writes the file at the same moment,
    record_hdf5.py:127-129, `recorded_frames` starting at 1) and recording then pauses until the next reset().  REF-QUIRK not reproduced: the
    reference keeps capturing after that write and overwrites the SAME file with every following block of `length` frames, and its close()
    raises on `np.stack([])` when no frame is pending; here the file keeps its first `length` frames and close() with nothing pending is a no-op."""

    def __init__(self, env, hdf5_folder, length=0, name_prefix="hdf5_record", disable_logger=False):
        self.env = env
        self.hdf5_folder = os.path.abspath(hdf5_folder)
        if os.path.isdir(self.hdf5_folder) and not disable_logger:
            warnings.warn(f"Overwriting existing recordings at {self.hdf5_folder}")
        os.makedirs(self.hdf5_folder, exist_ok=True)
        self.name_prefix = name_prefix
        self.length = length
        self.episode_id = 0
        self.files = []
        self._obs, self._act, self._path = [], [], None
        if backend() == "npz" and not disable_logger:
            warnings.warn("neither h5py nor libhdf5 found: episodes are written as .npz with the HDF5 dataset names")

    def __getattr__(self, name):
        return getattr(self.env, name)

    def _start(self):
        self._flush()
        self._path = os.path.join(self.hdf5_folder, f"{self.name_prefix}-episode-{self.episode_id}.hdf5")  # record_hdf5.py:111
        self.episode_id += 1

    def _flush(self):
        if self._path is not None and self._obs:
            self.files.append(write_episode(self._path, self._obs, self._act))
        self._obs, self._act, self._path = [], [], None

    def reset(self, **kwargs):
        out = self.env.reset(**kwargs)
        self._start()
        return out

    def step(self, action):
        observations, reward, terminated, truncated, info = self.env.step(action)
        if self._path is not None:
            self._obs.append({k: np.array(v) for k, v in observations.items()})
            self._act.append(np.array(action, np.float32))
            if self.length > 0:
                if len(self._obs) >= self.length:
                    self._flush()
            elif terminated or truncated:
                self._start()  # closes the finished episode and opens the next file (record_hdf5.py:131-134)
        return observations, reward, terminated, truncated, info

    def close(self):
        self._flush()
        if hasattr(self.env, "close"):
            self.env.close()


class VecRecorder:
    """Records env indices `which` of a VecSim while the caller steps it; call after_step(actions) once per step."""

    def __init__(self, sim, folder, which=(0,), name_prefix="hdf5_record"):
        self.sim, self.which = sim, list(which)
        self.folder = os.path.abspath(folder)
        os.makedirs(self.folder, exist_ok=True)
        self.name_prefix = name_prefix
        self.episode_id = {e: 0 for e in self.which}
        self._obs = {e: [] for e in self.which}
        self._act = {e: [] for e in self.which}
        self.files = []

    def after_step(self, actions):
        """actions: (N, k) host array that was just applied.  Only the recorded envs' data crosses PCIe: one packed copy of the
        state outputs (132 B/env) and one 230 400-B copy per recorded env and camera."""
        sim = self.sim
        h = sim.fetch_host()
        has_img = sim.image_front is not None
        if has_img:
            front = sim.read_rows(sim.image_front, self.which)
            top = sim.read_rows(sim.image_top, self.which)
        tq = None
        for j, e in enumerate(self.which):
            if h["did_reset"][e]:
                # the kernel already reset this env: its last observation is the terminal one, and its last frame is ray-cast
                # from the terminal pose (terminal_obs + terminal_quat) -- the frame buffers show the reset state
                t = h["terminal_obs"][e]
                o = {"arm_qpos": t[0:6].copy(), "arm_qvel": t[6:12].copy()}
                if has_img:
                    if tq is None:
                        tq = sim.terminal_quat.numpy()
                    qpos = np.zeros(sim.nq)
                    qpos[0:6] = t[0:6]; qpos[6:9] = t[12:15]; qpos[9:13] = tq[0:4, e]
                    if sim.task_name == "stack":
                        qpos[13:16] = t[15:18]; qpos[16:20] = tq[4:8, e]
                    tgt = t[15:18] if sim.task_name in ("push", "pick_place") else None
                    o["image_front"] = sim.render_state(qpos, tgt, "camera_front")
                    o["image_top"] = sim.render_state(qpos, tgt, "camera_top")
            else:
                o = {"arm_qpos": h["arm_qpos"][e].copy(), "arm_qvel": h["arm_qvel"][e].copy()}
                if has_img:
                    o["image_front"], o["image_top"] = front[j], top[j]
            self._obs[e].append(o)
            self._act[e].append(np.asarray(actions[e], np.float32))
            if h["did_reset"][e]:
                self._flush(e)

    def _flush(self, e):
        if self._obs[e]:
            path = os.path.join(self.folder, f"{self.name_prefix}-env{e}-episode-{self.episode_id[e]}.hdf5")
            self.files.append(write_episode(path, self._obs[e], self._act[e]))
            self.episode_id[e] += 1
        self._obs[e], self._act[e] = [], []

    def close(self):
        for e in self.which:
            self._flush(e)
