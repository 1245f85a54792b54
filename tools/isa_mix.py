"""Static ISA-mix histogram of a gfx950 kernel, weighted by loop depth.

    python tools/isa_mix.py [--kernel SUBSTR] [--out profiles/rNN_isa_mix.json] [file.s]

Without a .s file the step kernels are compiled to assembly first (hipcc -S, same flags as gym_lowcostrobot_amd/build.py).
Weights: instructions inside the substep loop (depth 1) x n_substeps(20), inside the PGS sweep loop (depth 2) x 20 x 4;
branches are not resolved (both sides of a wave-uniform branch are counted), so the figures are an upper bound on
the dynamic count -- compare with SQ_INSTS_VALU from profiles/rNN_pmc.json.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_pk_fma"):
        return "pk_fma"
    if op.startswith("v_pk_"):
        return "pk_other"
    if op.startswith(("v_fma_", "v_fmac_", "v_mac_", "v_mad_")):
        return "fma"
    if op.startswith(("v_mul_f", "v_add_f", "v_sub_f", "v_subrev_f")):
        return "mul_add"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_mov"):
        return "mov"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith(("v_min", "v_max", "v_med3")):
        return "minmax"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


FLOPS = {"pk_fma": 4, "pk_other": 2, "fma": 2, "mul_add": 1, "trans": 1, "minmax": 1}


def analyse(lines, n_substeps=20, sweeps=4):
    depth = 0
    hist = {0: collections.Counter(), 1: collections.Counter(), 2: collections.Counter()}
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^\.LBB\d+_\d+:\s*;(.*)$", s)
        if m:
            c = m.group(1)
            if "Depth=" in c:
                depth = min(int(re.search(r"Depth=(\d+)", c).group(1)), 2)
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            depth = 0
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        hist[depth][classify(op)] += 1
    w = {0: 1, 1: n_substeps, 2: n_substeps * sweeps}
    total = collections.Counter()
    for d, h in hist.items():
        for k, v in h.items():
            total[k] += v * w[d]
    valu_keys = ["pk_fma", "pk_other", "fma", "mul_add", "cndmask", "cmp", "accvgpr", "mov", "trans", "minmax", "valu_other"]
    valu = sum(total[k] for k in valu_keys)
    flops = sum(total[k] * FLOPS.get(k, 0) for k in valu_keys)
    return {
        "static_by_depth": {str(d): dict(h) for d, h in hist.items()},
        "weighted": dict(total),
        "weighted_valu": valu,
        "weighted_flops": flops,
        "flop_per_valu": flops / max(valu, 1),
        "valu_share": {k: total[k] / max(valu, 1) for k in valu_keys},
    }


def kernels(path):
    txt = open(path).read().split("\n")
    out, cur, name = {}, None, None
    for ln in txt:
        m = re.match(r"^(_Z\w*lcr_\w+):\s*;?\s*@?", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(ln)
            if ln.strip() == "s_endpgm":
                out[name] = cur
                cur = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="?")
    ap.add_argument("--kernel", default="lcr_step2_kernelILi1ELb0ELb0ELi2ELb0E", help="substring of the mangled name (default: the bench kernel: two-wave family, one cube, joint, four-row contacts, two waves per SIMD)")
    ap.add_argument("--out")
    ap.add_argument("--newton", action="store_true", help="the headline kernel of the default preset: lcr_step_kernel<1, false, false, true, false, true> (merged into --out if that file exists)")
    a = ap.parse_args()
    path = a.asm
    if not path:
        sys.path.insert(0, ROOT)
        from gym_lowcostrobot_amd import build as B

        if a.newton:   # the one-cube Newton kernel of the default preset (unit LCR_PART = 4, iterative-ILP scheduling: as build.py builds it)
            src, part = "lcr_kernels.hip", ["-DLCR_PART=4"] + B.ITER_ILP
            a.kernel = "lcr_step_kernelILi1ELb0ELb0ELb1ELb0ELb1E"
        else:
            src, part = ("lcr_kernels2.hip", ["-DLCR_PART=14"] + B.NO_POST_RA) if "step2" in a.kernel else ("lcr_kernels.hip", ["-DLCR_PART=0"])   # (as build.py builds the unit)
        path = os.path.join(tempfile.gettempdir(), src.replace(".hip", ".s"))
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + part + ["-S", "--cuda-device-only", "-o", path, os.path.join(B.CSRC, src)],
                              stderr=subprocess.DEVNULL)
    res = {}
    for name, body in kernels(path).items():
        if a.kernel in name:
            res[name] = analyse(body)
    for name, r in res.items():
        print(name)
        print("  weighted VALU %d  flop/VALU %.3f" % (r["weighted_valu"], r["flop_per_valu"]))
        for k, v in sorted(r["valu_share"].items(), key=lambda kv: -kv[1]):
            print("    %-11s %5.1f %%  (%d)" % (k, 100 * v, r["weighted"].get(k, 0)))
        print("    lds %d  vmem %d  salu %d  wait/nop %d" % tuple(r["weighted"].get(k, 0) for k in ("lds", "vmem", "salu", "wait_nop")))
    if a.out:
        if a.newton and os.path.exists(a.out):
            res = dict(json.load(open(a.out)), **res)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
