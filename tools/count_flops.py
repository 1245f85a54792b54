"""Algorithmic flops per env-step: census of the CPU oracle compiled with an instrumented scalar type
(oracle/flopcount.cpp, oracle/orc_counted.hpp), on the bench's synthetic workloads; plus, optionally, the flop weight of
the HIP kernel's VALU instruction mix (static ISA count) that bench.py combines with the measured SQ_INSTS_VALU.

    python tools/count_flops.py [--isa]      -> profiles/flops.json

Measurement infrastructure: imports oracle/ (allowed for tools run by hand; bench.py only reads the JSON)."""
import argparse, collections, ctypes, json, os, re, subprocess, sys, tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

WORKLOADS = {  # bench.py workload -> oracle task / params
    "ReachCube-v0": ("reach", {}),
    "PushCube-v0": ("push", {}),
    "LiftCube-v0": ("lift", {}),
    "PickPlaceCube-v0": ("pick_place", {"action_mode": 1}),
    "StackTwoCubes-v0": ("stack", {}),
}
NAMES = ["add", "mul", "div", "sqrt", "transcendental", "compare", "abs_floor"]


def census(task, kw, n=32, steps=50):
    o = orc.Oracle(task, n, f32="count", auto_reset=1, **kw)
    o.reset(np.arange(n, dtype=np.uint64))
    rng = np.random.default_rng(0)
    o.L.orc_count_reset()
    for _ in range(steps):
        o.step(rng.uniform(-1, 1, (n, o.action_dim)).astype(np.float32), 1)
    out = (ctypes.c_uint64 * 7)()
    o.L.orc_count_get(out)
    c = {k: v / (n * steps) for k, v in zip(NAMES, list(out))}
    c["flops"] = sum(c[k] for k in NAMES[:5])
    return c


FLOP_WEIGHT = [  # VALU mnemonic prefix -> flops per lane
    (r"v_pk_fma_f32", 4), (r"v_pk_(mul|add)_f32", 2), (r"v_(fma|fmac|fmamk|fmaak|mad|mac)_f32", 2),
    (r"v_(mul|add|sub|subrev|max|min|med3|max3|min3)_f32", 1), (r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_f32", 1),
]


def isa_mix(kernel_regex=r"lcr_step_kernelILi1ELb0ELb0"):
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffast-math", "-fno-slp-vectorize", "-save-temps",
                               "-c", os.path.join(ROOT, "gym_lowcostrobot_amd", "csrc", "lcr_kernels.hip"), "-o", "k.o"], cwd=d,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(os.path.join(d, "lcr_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kernel_regex + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    ops = collections.Counter(l.split()[0] for l in (x.strip() for x in lines[start:end]) if l and not l.startswith((".", ";")) and not l.endswith(":"))
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    flops = 0
    for k, v in ops.items():
        for pat, w in FLOP_WEIGHT:
            if re.match(pat, k):
                flops += w * v
                break
    return {"static_valu_instructions": valu, "static_flops": flops, "flops_per_valu_instruction": flops / valu,
            "note": "static instruction mix of lcr_step_kernel<1,false,false>; used as a proxy for the dynamic mix"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--isa", action="store_true")
    a = ap.parse_args()
    path = os.path.join(ROOT, "profiles", "flops.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    res["oracle_census_per_env_step"] = {w: census(t, kw) for w, (t, kw) in WORKLOADS.items()}
    res["census_note"] = ("operations executed by the CPU oracle's dense formulation (Jacobian-sum mass matrix, dense Delassus matrix, "
                          "dense PGS), 32 envs x 50 steps, U(-1,1) actions, auto-reset on; the HIP kernel exploits structure and executes fewer")
    if a.isa:
        res["kernel_isa_mix"] = isa_mix()
    json.dump(res, open(path, "w"), indent=1)
    for w, c in res["oracle_census_per_env_step"].items():
        print(f"{w:20s} {c['flops']:.4g} flops/env-step")
    if "kernel_isa_mix" in res:
        print(res["kernel_isa_mix"])
