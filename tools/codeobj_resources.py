"""Register / spill / scratch / LDS figures of every kernel from the code object's own metadata (llvm-readelf --notes),
i.e. what the hardware is told -- rocprofv3's per-dispatch VGPR_Count halves the unified-file figures on gfx950.

    python tools/codeobj_resources.py [--out profiles/rNN_codeobj.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_lowcostrobot_amd import build as B  # noqa: E402

KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
        ".group_segment_fixed_size", ".wavefront_size", ".max_flat_workgroup_size")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    a = ap.parse_args()
    res = {}
    for src, obj, extra in B.UNITS:          # every unit with the flags build.py gives it (scheduling flags differ per unit)
        if src == "lcr_capi.hip":
            continue
        co = os.path.join(tempfile.gettempdir(), obj + ".co")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + extra + ["--cuda-device-only", "--no-gpu-bundle-output", "-c", os.path.join(B.CSRC, src), "-o", co],
                              stderr=subprocess.DEVNULL)
        txt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], text=True)
        cur = None
        for ln in txt.split("\n"):
            m = re.match(r"\s*-?\s*\.name:\s+(\S+)", ln)
            if m and "lcr_" in m.group(1):
                cur = m.group(1)
                res.setdefault(cur, {})
            m = re.match(r"\s*-?\s*(\.\w+):\s+(\d+)\s*$", ln)
            if m and m.group(1) in KEYS:
                # keys of a kernel record may precede its .name line: buffer them
                res.setdefault("_pending", {})[m.group(1)] = int(m.group(2))
            if m is None and cur and "_pending" in res:
                pass
            if re.match(r"\s*-?\s*\.symbol:", ln) or re.match(r"\s*-\s*\.args:", ln):
                pass
        # robust second pass: split the metadata into kernel records
        res.pop("_pending", None)
        recs = re.split(r"\n\s*- \.(?=agpr_count|args)", txt)
        for r in recs:
            nm = re.search(r"\.name:\s+(_Z\S*lcr_\S+|lcr_\S+)", r)
            if not nm:
                continue
            d = {}
            for k in KEYS:
                m = re.search(re.escape(k) + r":\s+(\d+)", r)
                if m:
                    d[k.lstrip(".")] = int(m.group(1))
            if d:
                d["registers_total_per_lane"] = d.get("vgpr_count", 0)   # unified file: .vgpr_count already includes the AGPRs' offset
                res[nm.group(1)] = d
    res = {k: v for k, v in res.items() if v}
    for k, v in sorted(res.items()):
        print(k, v)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
