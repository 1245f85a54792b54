"""How the CPU oracle's OpenMP leg scales on this host (the number bench.py prints beside the GPU's): env-steps/s at 1, 2, 4, ... threads, with the
host's CPU count, affinity and cgroup quota beside it.      python tools/cpu_scaling.py [task]"""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from oracle import orc  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "reach"
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", orc.lib().orc_max_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
rng = np.random.default_rng(0)
t = 1
base = None
while t <= orc.lib().orc_max_threads():
    n = 64 * t
    o = orc.Oracle(task, n)
    o.reset(seeds=np.arange(n, dtype=np.uint64))
    a = rng.uniform(-1, 1, (n, o.action_dim)).astype(np.float32)
    o.step(a, threads=t)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 1.5:
        o.step(a, threads=t)
        k += 1
    dt = time.perf_counter() - t0
    rate = n * k / dt
    base = base or rate
    print(f"threads {t:4d}: {rate:10.0f} env-steps/s  per thread {rate / t:8.0f}  efficiency {rate / t / base:5.2f}", flush=True)
    t *= 2
