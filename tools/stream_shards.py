"""One batch, G wave-aligned shards on G HIP streams of ONE GPU (GPU box).    python tools/stream_shards.py [task] [--n 65536] [--steps 200] [--groups 1 2 4 ...]

A launch ends with its slowest wave (one wave per SIMD at 65 536 envs: mean / slowest ~0.5, profiles/r06_newton_phases.txt), and the next control step of EVERY env waits for
it although env e at step k + 1 depends on env e at step k only.  Cut into G shards with a stream each, a shard's next step starts when ITS slowest wave is done and fills the
SIMDs the other shards' finished waves left idle.  Same bits as one launch (shard invariance, tests/test_gpu_parity.py).  Prints env-steps/s of the whole batch per G."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from gym_lowcostrobot_amd import VecSim  # noqa: E402

W = {"reach": ("reach", "joint"), "push": ("push", "joint"), "lift": ("lift", "joint"), "pick_place_ee": ("pick_place", "ee"), "stack": ("stack", "joint"), "push_loop": ("push_loop", "joint")}
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="*", default=["reach"])
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--groups", type=int, nargs="*", default=[1, 2, 4, 8, 16, 32, 64])
ap.add_argument("--preset", default=None)
a = ap.parse_args()
for name in a.names:
    task, mode = W[name]
    for G in a.groups:
        per = a.n // G
        assert per % 64 == 0 and per * G == a.n
        streams = [torch.cuda.Stream() for _ in range(G)]
        sims = [VecSim(task, per, action_mode=mode, env_id_offset=g * per, global_envs=a.n, preset=a.preset) for g in range(G)]
        for s, st in zip(sims, streams):
            s.set_stream(st.cuda_stream)
        bufs = [[s.alloc_actions() for _ in range(8)] for s in sims]
        for s, bb in zip(sims, bufs):
            for i, b in enumerate(bb):
                s.fill_random_actions(b, 0, i)
        for i in range(60):
            for s, bb in zip(sims, bufs):
                s.step_device(bb[i % 8].ptr)
        torch.cuda.synchronize()
        ms = []
        for r in range(3):
            t0 = time.perf_counter()
            for i in range(a.steps):
                for s, bb in zip(sims, bufs):
                    s.step_device(bb[i % 8].ptr)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3 / a.steps)
        m = sorted(ms)[1]
        print(f"{name:14s} n={a.n} shards={G:3d} x {per:6d} envs: {m:.4f} ms per control step of the batch  ({a.n / m * 1e3:.3e} env-steps/s)  regions {['%.3f' % v for v in ms]}", flush=True)
        for s, bb in zip(sims, bufs):
            for b in bb:
                s.free(b)
            s.close()
