"""world_size-2 CPU test (gloo) of the multi-GPU path: env-id sharding + the barrier / max-over-ranks timing
protocol bench.py uses (gym_lowcostrobot_amd/sharding.py).  The data path itself has no collective."""
import os
import socket
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from gym_lowcostrobot_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4096
    lo, hi = sharding.shard_range(n, rank)
    calls = []

    def run(i):
        calls.append(i)
        time.sleep(0.01 * (1 + 2 * rank))  # rank 1 is the slow one

    worst, local = sharding.timed_region(run, 5, dist=dist, device_sync=None, tensor_device="cpu")
    value = sharding.aggregate_throughput(n, world, 5, worst)
    q.put((rank, lo, hi, worst, local, value, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing_protocol():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, w0, l0, v0, c0), (r1, lo1, hi1, w1, l1, v1, c1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4096, 4096, 8192)      # disjoint, contiguous, covering
    assert c0 == c1 == [0, 1, 2, 3, 4]                          # exactly K timed steps on every rank
    assert w0 == pytest.approx(w1)                               # every rank reports the MAX over ranks
    assert w0 >= max(l0, l1) - 1e-3                               # MAX over ranks ...
    assert w0 >= 5 * 0.03 and l0 >= 5 * 0.03 - 5e-3               # ... = the slow rank's 5 x 30 ms: the closing barrier holds the fast rank
    assert v0 == pytest.approx(8192 * 5 / w0)                    # whole-job units / max time


def test_shard_offsets_are_weak_scaling():
    from gym_lowcostrobot_amd import sharding

    n = 65536
    ranges = [sharding.shard_range(n, r) for r in range(8)]
    assert ranges[0] == (0, n) and ranges[7] == (7 * n, 8 * n)
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(7))
