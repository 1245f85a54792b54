#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused HIP step kernel (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one launch of the fused step kernel over this GPU's batch of envs (apply_action -> 20 physics
substeps -> reward/termination/TimeLimit -> auto-reset).  Workload = BASELINE.json configs[1]:
ReachCube-v0, 65 536 envs per GPU, joint control, state-only observations, sparse reward.  Envs are
independent: ranks own disjoint global env-id ranges, there is no data-path collective (weak scaling).
Synthetic actions (Philox keyed on (seed, global env id, step)) are generated on the device BEFORE the
timed region, so inputs are resident in HBM when timing starts.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (task, action_mode, algorithmic bytes per env-step)
    "ReachCube-v0": ("reach", "joint", 294),
    "PushCube-v0": ("push", "joint", 318),
    "PickPlaceCube-v0": ("pick_place", "ee", 338),
    "LiftCube-v0": ("lift", "joint", 294),
    "StackTwoCubes-v0": ("stack", "joint", 414),
    # Reach layout + current_goal r/w (8) + accumulated sim time f64 r/w (16); obs has no cube/target aux (-0)
    "PushCubeLoop-v0": ("push_loop", "joint", 318),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3   # fp32 vector peak
# BASELINE.json configs[1..4] at their per-GPU size: --config N selects (workload, envs per GPU, observation mode)
CONFIGS = {
    2: ("ReachCube-v0", 65536, "state"),
    3: ("PushCube-v0", 65536, "state"),
    4: ("PickPlaceCube-v0", 32768, "state"),     # 131 072 envs over 4 GPUs, ee-IK action mode
    5: ("StackTwoCubes-v0", 32768, "both"),      # 262 144 envs over 8 GPUs, state + two ray-cast 240x320x3 frames
}
# The step-kernel family is a property of the JOB (lcr_config.global_envs / step_kernel, include/lcr.h), never of the shard size, so that every sharding of a
# job gives identical bits.  BASELINE's config 5 (StackTwoCubes, 262 144 envs) is cut into 32 768-env shards, where the two-cooperating-waves family is the
# faster one while a Stack job of that size as ONE shard would run the one-wave family: that job PINS the two-wave family; a run on
# fewer GPUs than the job names is one (or some) of its shards and makes the same choice.
JOB = {4: {"global_envs": 131072, "step_kernel": "auto"}, 5: {"global_envs": 262144, "step_kernel": "coop"}}   # (PickPlace runs the two-wave family at every job size)


def job_kwargs(config_id, n, world, preset="faithful", pgs_iters=None):
    """(global_envs, step_kernel) of the job a bench sim is a shard of.  A pinned family is a property of preset "fast" with fixed sweeps: the Newton kernels of the
    faithful preset and the converged-sweeps mode (pgs_iters < 0) have one family each, the pin is dropped there."""
    j = JOB.get(config_id)
    if j is None:
        return {"global_envs": n * world, "step_kernel": "auto"}
    pinned = j["step_kernel"] if (preset == "fast" and (pgs_iters is None or pgs_iters >= 0)) else "auto"
    return {"global_envs": max(j["global_envs"], n * world), "step_kernel": pinned}


def kernel_sha16():
    """hash of the HIP sources: committed PMC-derived figures (traffic, VALU counts) are echoed only while it still matches"""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "gym_lowcostrobot_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(d, fn), "rb") as f:
                h.update(f.read())
    # the compiler flags are part of what a profile belongs to (the per-unit scheduling flags live in build.py, not in csrc/)
    from gym_lowcostrobot_amd import build as hipbuild

    h.update(repr((hipbuild.FLAGS, hipbuild.UNITS)).encode())
    return h.hexdigest()[:16]



def cpu_baseline(task, action_mode, budget_s=12.0):
    """Time the CPU oracle (a port: the reference's MuJoCo path cannot run here) on a bounded sample."""
    import numpy as np
    from oracle import orc

    # the cores this process may actually use: the container's cgroup CPU quota, not the host's CPU count (round 5 ran 128 threads on a 16-core quota and printed 8 %
    # parallel efficiency: tools/cpu_scaling.py, profiles/r06_cpu_scaling.txt)
    host_threads = orc.lib().orc_max_threads()
    threads = host_threads
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        threads = max(1, min(threads, int(quota)))
    rng = np.random.default_rng(0)
    n = 64 * threads
    o = orc.Oracle(task, n, action_mode={"joint": 0, "ee": 1}[action_mode])
    o.reset(seeds=np.arange(n, dtype=np.uint64))
    a = rng.uniform(-1, 1, (n, o.action_dim)).astype(np.float32)
    t0 = time.perf_counter()
    o.step(a, threads=threads)
    per_step = time.perf_counter() - t0
    steps = max(2, min(200, int(budget_s / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(steps):
        a = rng.uniform(-1, 1, (n, o.action_dim)).astype(np.float32)
        o.step(a, threads=threads)
    dt = time.perf_counter() - t0
    # config 1 of BASELINE.json: one env on one thread (latency per control step).  The box-to-box spread of this number was 4x in round 3 (a cold
    # core, the thread migrating): pin the thread to one core, warm it up for ~0.3 s, report the MEDIAN of five regions of 200 steps
    o1 = orc.Oracle(task, 1, action_mode={"joint": 0, "ee": 1}[action_mode])
    o1.reset(seeds=np.zeros(1, np.uint64))
    a1 = rng.uniform(-1, 1, (200, 1, o1.action_dim)).astype(np.float32)
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(old_aff)[len(old_aff) // 2]})
    except (AttributeError, OSError):
        old_aff = None
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 0.3:
        for i in range(50):
            o1.step(a1[i], threads=1)
    regions = []
    for r in range(5):
        t1 = time.perf_counter()
        for i in range(200):
            o1.step(a1[i], threads=1)
        regions.append(time.perf_counter() - t1)
    if old_aff is not None:
        try:
            os.sched_setaffinity(0, old_aff)
        except OSError:
            pass
    dt1 = float(np.median(regions))
    return {
        "single_env_single_thread": {"value": 200 / dt1, "unit": "env-steps/s", "ms_per_step": 1e3 * dt1 / 200,
                                     "regions_ms_per_step": [1e3 * x / 200 for x in regions], "note": "one pinned core, 0.3 s warm-up, median of 5 x 200 steps"},
        "value_per_core": n * steps / dt / threads,
        "parallel_efficiency": (n * steps / dt / threads) / (200 / dt1),   # per-thread rate of the threaded leg / the pinned single-thread rate
        "value": n * steps / dt,
        "unit": "env-steps/s",
        "cores": threads,
        "host_cpus": host_threads, "cgroup_cpu_quota": quota,
        "kind": "port",
        "sample": f"{n} envs x {steps} control steps of {task} ({action_mode}), fp64 C oracle, OpenMP over envs, "
                  f"{dt:.1f} s; reference MuJoCo unavailable in this image",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--workload", default="ReachCube-v0", choices=sorted(WORKLOADS))
    ap.add_argument("--config", type=int, default=0, choices=[0] + sorted(CONFIGS), help="BASELINE.json configs[N-1] at its per-GPU size (overrides --workload / --envs-per-gpu / --obs)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions in total (the first is the contract's K steps; the rest give the median SURVEY.md 8(d) asks for)")
    ap.add_argument("--arm-collision", type=int, default=1, help="0: round-1 contact model (finger tips only), for like-for-like comparison")
    ap.add_argument("--preset", default="faithful", choices=["faithful", "fast"],
                    help="faithful (default, the headline): the reference's contact model solved by Newton's method; fast: rounds 1-4 (four block projected-gradient sweeps)")
    ap.add_argument("--pgs-iters", type=int, default=None, help="preset fast only: sweeps per substep (default 4; < 0: swept to a tolerance)")
    ap.add_argument("--no-fast-side", action="store_true", help="skip the side measurement of preset fast that a faithful run reports beside its headline")
    ap.add_argument("--obs", default="state", choices=["state", "both"], help="both: also ray-cast the two 240x320x3 observation frames per env")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-tasks", action="store_true", help="BASELINE.md B2: also time the CPU oracle on every workload (adds ~1 min)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl) even for a single rank (tests the N>1 code path on one GPU)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0, help="untimed spin of the step kernel before the contract's warm-up so that the timed region runs at steady clocks (0 disables)")
    ap.add_argument("--sharded-configs", type=int, default=-1, help="1: also time BASELINE configs 4 and 5 at their per-GPU shard size (extra object, never the headline); default: on when --gpus > 1 and no --config")
    ap.add_argument("--calibrate", type=int, default=0, help="after timing, launch the known-byte-count copy kernel this many times (PMC calibration)")
    args = ap.parse_args()
    if args.config:
        args.workload, args.envs_per_gpu, args.obs = CONFIGS[args.config]

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched directly instead of through torchrun: start one rank per GPU ourselves (same command line the driver uses)
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch

    from gym_lowcostrobot_amd import VecSim, build, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test hooks (tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu): several ranks may share GPU 0 and rendezvous
    # over gloo, which exercises the whole N>1 code path on a 1-GPU box; the driver's runs use neither
    backend = os.environ.get("LCR_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("LCR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        dist.barrier()
    else:
        torch.cuda.set_device(local_rank)
    # the in-tree .so normally travels with the snapshot; only a missing library is built here (by rank 0 alone)
    if not os.path.exists(build.LIB):
        if rank == 0:
            build.build()
    if dist is not None:
        dist.barrier()

    task, action_mode, alg_bytes = WORKLOADS[args.workload]
    if args.obs == "both":
        alg_bytes += 2 * 240 * 320 * 3  # write-once frames
    n = args.envs_per_gpu
    jk = job_kwargs(args.config, n, world, args.preset, args.pgs_iters)
    sim = VecSim(task, n, device=local_rank, env_id_offset=sharding.shard_offset(n, rank), observation_mode=args.obs, action_mode=action_mode,
                 preset=args.preset, pgs_iters=args.pgs_iters, base_seed=0, arm_collision=args.arm_collision, **jk)
    stream = torch.cuda.current_stream()
    sim.set_stream(stream.cuda_stream)

    # inputs resident in HBM before timing: a ring of pre-generated action buffers
    ring = min(args.steps + args.warmup, 64)
    bufs = [sim.alloc_actions() for _ in range(ring)]
    for i, b in enumerate(bufs):
        sim.fill_random_actions(b, seed=0, step=i)
    sim.sync()

    # clock ramp: with the driver's --warmup 5 the contract region used to start ~1.5 ms after the first launch and came out 6 % slower
    # than the regions after it; an untimed spin (not part of W or K) brings the GPU to steady clocks first
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < args.prewarm_ms:
            for i in range(16):
                sim.step_device(bufs[(prewarm_steps + i) % ring].ptr)
            sim.sync()
            prewarm_steps += 16
    for i in range(args.warmup):
        sim.step_device(bufs[i % ring].ptr)
    ev = {}

    def run(i):
        if i == 0:
            sim.timer_begin()  # HIP events on the stream the kernel is launched on
        sim.step_device(bufs[(args.warmup + i) % ring].ptr)
        if i == args.steps - 1:
            ev["ms"] = sim.timer_end()

    dt, _ = sharding.timed_region(run, args.steps, dist=dist, device_sync=torch.cuda.synchronize, tensor_device="cuda" if backend == "nccl" else "cpu")
    ev_ms = ev["ms"]
    # further regions of the same K steps (HIP events on the launch stream, this rank only): median of `repeats`
    rep_ms = [ev_ms / args.steps]
    for r in range(max(args.repeats, 1) - 1):
        sim.timer_begin()
        for i in range(args.steps):
            sim.step_device(bufs[(args.warmup + i + 7 * (r + 1)) % ring].ptr)
        rep_ms.append(sim.timer_end() / args.steps)

    # the same workload under preset "fast" (rounds 1-4: four sweeps, fewer contact rows), reported BESIDE the headline, never as it: same barrier + max-over-ranks timing
    fast_side = None
    if args.preset == "faithful" and not args.no_fast_side:
        jf = job_kwargs(args.config, n, world, "fast", None)
        sim_f = VecSim(task, n, device=local_rank, env_id_offset=sharding.shard_offset(n, rank), observation_mode=args.obs, action_mode=action_mode,
                       preset="fast", base_seed=0, arm_collision=args.arm_collision, **jf)
        sim_f.set_stream(stream.cuda_stream)
        bufs_f = [sim_f.alloc_actions() for _ in range(8)]
        for i, b in enumerate(bufs_f):
            sim_f.fill_random_actions(b, seed=0, step=i)
        k_f = 20 if args.obs == "both" else 200
        for i in range(20):
            sim_f.step_device(bufs_f[i % 8].ptr)
        dt_f, _ = sharding.timed_region(lambda i: sim_f.step_device(bufs_f[i % 8].ptr), k_f, dist=dist, device_sync=torch.cuda.synchronize,
                                        tensor_device="cuda" if backend == "nccl" else "cpu")
        fast_side = {"value": sharding.aggregate_throughput(n, world, k_f, dt_f), "unit": "env-steps/s", "steps": k_f, "ms_per_step": dt_f / k_f * 1e3,
                     "kernel": sim_f.step_kernel_name, "kernel_family": sim_f.step_kernel_family,
                     "what": "same workload, preset fast: four warm-started block projected-gradient sweeps on the dual problem, rolling rows only where they matter, four-point "
                             "box-box -- p90 2e-4 / p99 1e-2 rad per control step away from MuJoCo's optimum (DESIGN.md section 4), the round-4 headline configuration"}
        for b in bufs_f:
            sim_f.free(b)
        sim_f.close()

    # BASELINE.json's sharded shapes (configs 4 and 5: PickPlace-ee 4 x 32 768, Stack + frames 8 x 32 768) at their per-GPU size, so that
    # a multi-GPU run of the default command line also covers them; same barrier + max-over-ranks timing, reported beside the headline
    sharded = None
    want_sharded = args.sharded_configs == 1 or (args.sharded_configs < 0 and world > 1 and not args.config)
    if want_sharded:
        sharded = {}
        for cid in (4, 5):
            wl, n_c, obs_c = CONFIGS[cid]
            t_c, m_c, b_c = WORKLOADS[wl]
            sim_c = VecSim(t_c, n_c, device=local_rank, env_id_offset=sharding.shard_offset(n_c, rank), observation_mode=obs_c, action_mode=m_c, base_seed=0,
                           preset=args.preset, **job_kwargs(cid, n_c, world, args.preset, args.pgs_iters))
            sim_c.set_stream(stream.cuda_stream)
            bufs_c = [sim_c.alloc_actions() for _ in range(8)]
            for i, b in enumerate(bufs_c):
                sim_c.fill_random_actions(b, seed=0, step=i)
            k_c = 20 if (obs_c == "both" or args.preset == "faithful") else 100
            for i in range(5):
                sim_c.step_device(bufs_c[i % 8].ptr)
            dt_c, _ = sharding.timed_region(lambda i: sim_c.step_device(bufs_c[i % 8].ptr), k_c, dist=dist, device_sync=torch.cuda.synchronize,
                                            tensor_device="cuda" if backend == "nccl" else "cpu")
            bytes_c = b_c + (2 * 240 * 320 * 3 if obs_c == "both" else 0)
            sharded[f"config{cid}"] = {"workload": f"{wl}, {n_c} envs/GPU, {m_c} control, {obs_c} obs", "steps": k_c, "ms_per_step": dt_c / k_c * 1e3,
                                       "value": sharding.aggregate_throughput(n_c, world, k_c, dt_c), "unit": "env-steps/s", "n_gpus": world,
                                       "hbm_frac_algorithmic": bytes_c * n_c / (dt_c / k_c) / 1e9 / HBM_PEAK_GBS}
            for b in bufs_c:
                sim_c.free(b)
            sim_c.close()

    calib_bytes = 0
    for _ in range(args.calibrate):
        calib_bytes = sim.calibrate_copy(2 * 1024 * 1024)
    if args.calibrate:
        sim.sync()

    # sanity: the timed work must have produced finite state and episodes must have cycled
    st = sim.get_state()
    import numpy as np

    finite = bool(np.isfinite(st["qpos"]).all() and np.isfinite(st["qvel"]).all())

    if rank == 0:
        kern_ms = ev_ms / args.steps
        achieved = alg_bytes * n / (kern_ms * 1e-3) / 1e9
        # measured HBM bytes per launch: rocprofv3 PMC passes cannot run inside this process, so the figure comes from the
        # committed summary of the same command (profiles/traffic.json) -- and only while the kernel sources are unchanged
        sha = kernel_sha16()
        wl_key = f"{args.workload}|{n}|{args.obs}"
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                with open(prof) as f:
                    rec = json.load(f).get(wl_key, {})
                if rec.get("kernel_sha16") == sha:
                    traffic, traffic_src = rec.get("hbm_bytes_per_launch"), f"profiles/traffic.json[{wl_key}] ({rec.get('round')})"
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec",
            "value": sharding.aggregate_throughput(n, world, args.steps, dt),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}, {n} envs/GPU, {action_mode} control, {args.obs} obs, sparse reward, "
                            f"n_substeps=20, max_episode_steps=50, auto-reset on, preset {args.preset}: "
                            + ("Newton's method on the primal constraint problem to its optimum (newton_tol 1e-6, <= 30 iterations, derivative line search with ls_tol 1e-2), six-row finger contacts on cube and floor, finger pads as boxes"
                               + (", eight-point box-box" if task == "stack" else "") if args.preset == "faithful"
                               else f"block projected-gradient sweeps, pgs_iters={4 if args.pgs_iters is None else args.pgs_iters}"),
                "preset": args.preset,
                "envs_per_gpu": n,
                "global_envs": n * world,
                "job": {"global_envs": jk["global_envs"], "step_kernel": jk["step_kernel"],
                        "note": "kernel family chosen from the task and the JOB size (lcr_config.global_envs), never the shard size: any sharding of the job gives identical bits"},
                "parallelism": f"env-sharded x{world}, no collective",
                "arm_collision": bool(args.arm_collision),
                "kernel_sha16": sha,
            },
            "repeats": {"ms_per_step": rep_ms, "median_ms_per_step": float(np.median(rep_ms)),
                        "median_value": n * world / (float(np.median(rep_ms)) * 1e-3)},
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": sim.step_kernel_name if args.obs == "state" else sim.step_kernel_name + " + lcr_render_obs_kernel",
                "kernel_family": sim.step_kernel_family,
                "kernel_ms": kern_ms,
                "algorithmic_bytes_per_env_step": alg_bytes,
                "traffic_note": "traffic counts solver state the SURVEY.md 8(d) formula does not: the constraint forces carried from one control step to the "
                                "next (warm start as MuJoCo's qacc_warmstart, DESIGN.md section 4 D1) are read and written once per env-step: "
                                "2 x 4 B x 42 (one cube) .. 84 (Stack) floats"
                                + ("; the two-waves-per-SIMD build of the two-wave kernels also spills ~30 registers of its cube wave at the 256-register cap "
                                   "(scratch: written once and read once per step, ~0.25 KB per env-step) and hands 42 floats per env from the cube wave to the arm wave "
                                   "through a global record in substeps with a finger on a cube" if "two waves per SIMD" in sim.step_kernel_family else ""),
                "note": ("state-only step is VALU/latency-bound by construction (SURVEY.md 8(d)); HBM is the mandated yard-stick" if args.obs == "state" else
                         "the frames of step k are ray-cast on a second stream while the step kernel of step k + 1 runs (lcr.h: lcr_step; LCR_RENDER_OVERLAP=0 serialises them): "
                         "kernel_ms is the period of a control step -- HIP events on the launch stream, which waits for the last frames before the closing event --, "
                         "not the sum of the two kernels' durations"),
                "frames_overlap_step_kernel": bool(args.obs != "state" and os.environ.get("LCR_RENDER_OVERLAP", "1") != "0"),
            },
            "state_finite": finite,
            "calibration_bytes_per_launch": calib_bytes or None,
            "prewarm": {"ms": args.prewarm_ms, "steps": prewarm_steps, "note": "untimed spin before the W warm-up steps (steady clocks); not part of W or K"},
        }
        if fast_side is not None:
            out["preset_fast"] = fast_side
        if sharded is not None:
            out["baseline_sharded_configs"] = sharded
        # VALU-side view of the same kernel (the state-only step is VALU-issue-bound): taken from the committed
        # rocprofv3 PMC summary of this command, not measured live
        try:
            import glob

            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))[-1]
            with open(pm) as f:
                pj = json.load(f)
            pmc = pj["step_kernel_pmc_per_launch"]
            if pj.get("workload") == args.workload and args.obs == "state" and pj.get("kernel_sha16") == sha and n == 65536:
                two_wave = sim.step_kernel_name == "lcr_step2_kernel"
                out["valu"] = {
                    "source": os.path.basename(pm),
                    # VALU instructions per 64 envs (one wave-instruction serves 64 lanes = 64 envs; the two-wave kernels spend two waves on them)
                    "valu_insts_per_64_envs_per_launch": pmc["SQ_INSTS_VALU"] / ((n + 63) // 64),
                    "waves_per_64_envs": pmc["SQ_WAVES"] / ((n + 63) // 64),
                    "valu_busy_frac_of_wave_cycles": pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"],
                    "wait_frac_of_wave_cycles": pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"],
                    "note": ("two cooperating waves per 64 envs, two waves per SIMD at 65 536 envs: they interleave where a lone wave cannot issue (tools/ubench/valu_issue.hip)"
                             if two_wave else "one wave per SIMD at 65 536 envs; a single wave issues one VALU instruction per ~5.2 cycles (tools/ubench/valu_issue.hip)"),
                }
                # flops: executed by the kernel (measured VALU instructions x flop weight of its instruction
                # mix) and, for reference, the census of the CPU oracle's dense formulation (tools/count_flops.py)
                with open(os.path.join(ROOT, "profiles", "flops.json")) as f:
                    fj = json.load(f)
                steps_per_s = out["value"] / world
                # flop weight of a VALU instruction: from the SAME round's static instruction mix (profiles/rNN_isa_mix.json)
                with open(pm.replace("_pmc.json", "_isa_mix.json")) as f:
                    mixj = json.load(f)
                want = "lcr_step2_kernel" if two_wave else "lcr_step_kernel"
                mix = [v["flop_per_valu"] for k, v in mixj.items() if want in k and "flop_per_valu" in v][0]
                out["valu"]["flop_per_valu_source"] = os.path.basename(pm.replace("_pmc.json", "_isa_mix.json")) + f" ({want})"
                kflops = out["valu"]["valu_insts_per_64_envs_per_launch"] * mix     # per lane == per env-step
                out["valu"].update({
                    "kernel_flops_per_env_step_est": kflops,
                    "kernel_tflops_est": kflops * steps_per_s / 1e12,
                    "frac_of_fp32_vector_peak": kflops * steps_per_s / 1e12 / VALU_PEAK_TFLOPS,
                    "peak_tflops": VALU_PEAK_TFLOPS,
                    "oracle_census_flops_per_env_step": fj["oracle_census_per_env_step"][args.workload]["flops"],
                })
        except Exception as exc:   # (a missing profile file must not take the bench line down -- but it is said, not swallowed: VERDICT r5 weak #10a)
            out.setdefault("valu", {})["note"] = f"PMC-derived VALU figures not echoed: {exc!r}"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(task, action_mode)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if args.cpu_all_tasks:
                out["cpu_baseline"]["all_tasks"] = {w: {k: v for k, v in cpu_baseline(t, m, budget_s=6.0).items() if k in ("value", "value_per_core", "cores", "sample")}
                                                    for w, (t, m, _) in WORKLOADS.items()}
            try:  # real MuJoCo only if it happens to be importable on this box
                from tools import mujoco_opportunistic

                out["cpu_baseline"]["mujoco_opportunistic"] = mujoco_opportunistic.run(300)
            except Exception as e:
                out["cpu_baseline"]["mujoco_opportunistic"] = {"status": "reference MuJoCo unavailable", "error": repr(e)}
        try:  # a real .hdf5 episode file through h5py or, without it, the HDF5 C library (recorder.backend())
            from gym_lowcostrobot_amd import recorder

            out["hdf5"] = recorder.hdf5_selftest()
        except Exception as e:
            out["hdf5"] = {"status": "selftest failed", "error": repr(e)}
        print(json.dumps(out), flush=True)
    for b in bufs:
        sim.free(b)
    sim.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
