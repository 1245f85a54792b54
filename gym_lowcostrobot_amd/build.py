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
SOURCES = ["lcr_capi.hip", "lcr_kernels.hip", "lcr_render.hip"]
HEADERS = ["lcr_device.h", "lcr_arm.h", "lcr_model_gen.h", os.path.join("..", "..", "include", "lcr.h")]
# -ffast-math: the kernels carry no NaN/inf/signed-zero semantics (the -0.0 sparse reward is built from its bit pattern,
# the fp64 reset sampling uses explicitly rounded __dmul_rn/__dadd_rn); -fno-slp-vectorize: packed-f32 formation by the
# SLP vectoriser costs more moves than it saves here (measured on MI355X: 0.340 ms -> 0.286 ms per 65 536-env step).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-ffast-math", "-fno-slp-vectorize"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    deps = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + deps):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
