"""Build liblcr_hip.so (the C ABI of include/lcr.h) in-tree with hipcc for gfx950.

    python -m gym_lowcostrobot_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so stays next to this file so that it travels with the
source snapshot to the GPU box; it is git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblcr_hip.so")
SOURCES = ["lcr_capi.hip", "lcr_kernels.hip", "lcr_kernels2.hip", "lcr_render.hip"]   # (bench.kernel_sha16 and the tools hash / compile these)
HEADERS = ["lcr_device.h", "lcr_arm.h", "lcr_model_gen.h", "lcr_step_common.h", "lcr_newton.h", "lcr_newton_coop.h", os.path.join("..", "..", "include", "lcr.h")]
# -ffast-math: the kernels carry no NaN/inf/signed-zero semantics (the -0.0 sparse reward is built from its bit pattern,
# the fp64 reset sampling uses explicitly rounded __dmul_rn/__dadd_rn); -fno-slp-vectorize: packed-f32 formation by the
# SLP vectoriser costs more moves than it saves here (measured on MI355X: 0.340 ms -> 0.286 ms per 65 536-env step).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-ffast-math", "-fno-slp-vectorize"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# (source, object, extra flags): lcr_kernels.hip is compiled three times, LCR_PART selecting the step-kernel instantiations a unit emits
# (0 one-cube kernels + dispatcher + small kernels, 2 / 3 the two StackTwoCubes variants) -- kernels of ~50-100 KB code each: 80 s in one
# unit, ~30 s as units compiled concurrently; lcr_kernels_loop.hip is PushCubeLoop's unit
NO_POST_RA = ["-mllvm", "-enable-post-misched=0"]
ITER_ILP = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
UNITS = [("lcr_capi.hip", "lcr_capi.o", []), ("lcr_render.hip", "lcr_render.o", []),
         ("lcr_kernels.hip", "lcr_kernels.o", ["-DLCR_PART=0"]), ("lcr_kernels_loop.hip", "lcr_kernels_loop.o", ["-DLCR_LOOP_PART=0"]),
         ("lcr_kernels_loop.hip", "lcr_kernels_loop_newton.o", ["-DLCR_LOOP_PART=1"] + ITER_ILP),   # (PushCubeLoop's Newton kernels: 6.60 -> 5.98 ms with it, its sweep kernels 0.652 -> 0.730: two units)
         ("lcr_kernels.hip", "lcr_kernels_stack.o", ["-DLCR_PART=2"]), ("lcr_kernels.hip", "lcr_kernels_stack_big.o", ["-DLCR_PART=3"]),
         # the Newton kernels of the faithful preset: one cube / StackTwoCubes (eight cube<->cube slots).  Iterative-ILP scheduling, measured in round 6 against the default
         # (same box, tools/quick_times.py): ReachCube 65 536 envs 2.209 -> 2.082 ms, PushCube 3.499 -> 3.312, Stack 32 768 envs 7.06 -> 6.88, PushCubeLoop 6.60 -> 5.98;
         # no post-RA scheduler 2.362 (worse), both 2.230, max-ilp 2.179, max-memory-clause 2.188
         ("lcr_kernels.hip", "lcr_kernels_newton.o", ["-DLCR_PART=4"] + ITER_ILP),
         ("lcr_kernels.hip", "lcr_kernels_newton_stack.o", ["-DLCR_PART=5"] + ITER_ILP),
         # the two-cooperating-waves family (lcr_kernels2.hip): 10 / 14 one cube built for one / two waves per SIMD, 12 StackTwoCubes, 13 StackTwoCubes with the eight-point
         # manifold.  Instruction-scheduling flags per unit, each measured on the MI355X against the default (tools/quick_times.py; results are bit-identical, the flags only
         # reorder instructions): no post-RA scheduler for the 256-register build (Reach 65 536 envs 0.2578 -> 0.2554 ms, Push 0.3128 -> 0.3064, PickPlace-ee 0.3106 -> 0.3068);
         # no post-RA scheduler + the iterative-ILP strategy for the one-wave-per-SIMD build (PickPlace-ee 32 768 envs 0.2292 -> 0.2180); iterative-ILP for Stack (0.4307 -> 0.4202).
         # The one-wave kernels lose with each of them (Stack 65 536 envs 0.779 -> 0.90 / 1.16): default scheduling there.
         ("lcr_kernels2.hip", "lcr_kernels2.o", ["-DLCR_PART=10"] + NO_POST_RA + ITER_ILP), ("lcr_kernels2.hip", "lcr_kernels2_occ2.o", ["-DLCR_PART=14"] + NO_POST_RA),
         ("lcr_kernels2.hip", "lcr_kernels2_stack.o", ["-DLCR_PART=12"] + ITER_ILP), ("lcr_kernels2.hip", "lcr_kernels2_stack_cc8.o", ["-DLCR_PART=13"])]


def build(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]   # (the per-unit flags live in this file)
    objs, jobs = [], []
    for src, obj, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, obj)
        if force or _newer(o, [s] + deps):
            jobs.append([hipcc] + FLAGS + extra + ["-c", s, "-o", o])
        objs.append(o)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))   # (re-raises the first failure)
    if force or _newer(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
