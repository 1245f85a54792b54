"""Multi-GPU = independent env shards, one process per GPU, no data-path collective.

GPU/rank g of G owns the contiguous global env ids [g*n, (g+1)*n) (weak scaling: n per GPU is fixed).  RNG
streams and the synthetic policy are keyed on the GLOBAL env id, so any sharding yields the same trajectories.
The only communication is the timing protocol of the benchmark: a barrier on both sides of the timed region
and a MAX-reduction of the elapsed time (works on the "nccl"(=RCCL) backend with GPU tensors and on "gloo"
with CPU tensors, which is how the CPU tests cover it).
"""
import time


def shard_offset(envs_per_rank, rank):
    """global id of this rank's env 0"""
    return int(rank) * int(envs_per_rank)


def shard_range(envs_per_rank, rank):
    o = shard_offset(envs_per_rank, rank)
    return o, o + int(envs_per_rank)


def timed_region(run_steps, steps, *, dist=None, device_sync=None, tensor_device="cpu"):
    """Time exactly `steps` calls of run_steps(i) bracketed by barrier + device sync on both sides.
    Returns (max elapsed seconds over ranks, this rank's elapsed seconds)."""
    def sync_all():
        if dist is not None:
            dist.barrier()
        if device_sync is not None:
            device_sync()

    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        run_steps(i)
    sync_all()
    local = time.perf_counter() - t0
    worst = local
    if dist is not None:
        import torch

        t = torch.tensor([local], dtype=torch.float64, device=tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        worst = float(t.item())
    return worst, local


def aggregate_throughput(envs_per_rank, world, steps, worst_seconds):
    """whole-job env-steps/s: units all ranks processed / max-over-ranks time"""
    return int(envs_per_rank) * int(world) * int(steps) / worst_seconds


class ShardedVecSim:
    """Single-process convenience over several GPUs: one VecSim (one C-ABI handle, one HIP stream) per device, contiguous
    global env-id ranges, launches enqueued on every device before anything is synchronised.  No collective is involved;
    results are identical to any other sharding of the same global batch."""

    def __init__(self, task, n_envs_total, devices, **kw):
        from .vecsim import VecSim

        devices = list(devices)
        if n_envs_total % len(devices):
            raise ValueError("n_envs_total must be divisible by the number of shards")
        self.per = n_envs_total // len(devices)
        self.n = n_envs_total
        # every shard declares the whole job (lcr_config.global_envs): the same kernel family on every shard, and lcr_create checks that the cut is at wave boundaries
        kw.setdefault("global_envs", n_envs_total)
        self.shards = [VecSim(task, self.per, device=d, env_id_offset=shard_offset(self.per, i), **kw) for i, d in enumerate(devices)]
        self.action_dim = self.shards[0].action_dim
        self._act = [s.alloc_actions() for s in self.shards]

    def fill_random_actions(self, seed, step):
        for s, a in zip(self.shards, self._act):
            s.fill_random_actions(a, seed, step)

    def step_device(self):
        """step every shard on its own action buffer (asynchronous on every device)"""
        for s, a in zip(self.shards, self._act):
            s.step_device(a.ptr)

    def sync(self):
        for s in self.shards:
            s.sync()

    def get_state(self):
        import numpy as np

        sts = [s.get_state() for s in self.shards]
        return {k: np.concatenate([st[k] for st in sts], axis=-1) for k in sts[0]}

    def outputs(self):
        import numpy as np

        outs = [s.outputs() for s in self.shards]
        return {k: np.concatenate([o[k] for o in outs]) for k in outs[0]}

    def close(self):
        for s, a in zip(self.shards, self._act):
            s.free(a)
            s.close()
