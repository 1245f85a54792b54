#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile_gpu.sh on the GPU box) into profiles/:

    python tools/summarize_profile.py <tag> <round-name> [workload] [envs] [obs]

writes profiles/<round>_kernel_stats.csv (verbatim rocprofv3 --stats table), profiles/<round>_pmc.json
(per-launch PMC means of the step kernel and of the calibration copy) and updates profiles/traffic.json
with the calibrated HBM bytes per launch that bench.py echoes in roofline.traffic.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_means(path, kernel_substr):
    agg = collections.defaultdict(list)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}, meta


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "ReachCube-v0"
    envs = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
    obs = sys.argv[5] if len(sys.argv) > 5 else "state"
    sys.path.insert(0, ROOT)
    import bench

    sha = bench.kernel_sha16()
    default = workload == "ReachCube-v0" and envs == 65536 and obs == "state"
    stem = rnd if default else f"{rnd}_{workload}_{envs}_{obs}"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, f"{stem}_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "bench_line.json")):
        shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{stem}_bench_line.json"))
    out = {"tag": tag, "workload": workload, "envs_per_gpu": envs, "obs": obs, "kernel_sha16": sha,
           "command": "tools/profile_gpu.sh (bench.py --steps 50 --warmup 5 --calibrate 20)"}
    step, calib = {}, {}
    for d in sorted(os.listdir(src)):
        p = os.path.join(src, d, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        m, n, meta = pmc_means(p, "lcr_step")   # lcr_step_kernel (one wave per 64 envs) or lcr_step2_kernel (two cooperating waves)
        step.update(m)
        if meta:
            out["step_kernel_resources_rocprofv3"] = meta   # NB: rocprofv3 halves the register counts (see <round>_codeobj.json for the real ones)
        mr, _, _ = pmc_means(p, "lcr_render_obs_kernel")
        if mr:
            out.setdefault("render_kernel_pmc_per_launch", {}).update(mr)
        m2, _, _ = pmc_means(p, "lcr_calib_copy_kernel")
        calib.update(m2)
    out["step_kernel_pmc_per_launch"] = step
    out["calib_copy_pmc_per_launch"] = calib
    # kernel duration from the trace
    with open(os.path.join(src, "trace", "t_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            if "lcr_step" in r["Name"]:
                out["step_kernel_avg_ns"] = float(r["AverageNs"])
                out["step_kernel_calls"] = int(r["Calls"])
                out["step_kernel_name"] = r["Name"]
            if "lcr_render_obs_kernel" in r["Name"]:
                out["render_kernel_avg_ns"] = float(r["AverageNs"])
    # calibration: FETCH_SIZE / WRITE_SIZE are reported in KiB-units of the L2<->fabric request counters
    known = 4 * 2 * 1024 * 1024
    if "FETCH_SIZE" in calib and "WRITE_SIZE" in calib and calib["FETCH_SIZE"] > 0:
        kf = known / (calib["FETCH_SIZE"] * 1024.0)
        kw = known / (calib["WRITE_SIZE"] * 1024.0)
        rd = step["FETCH_SIZE"] * 1024.0 * kf
        wr = step["WRITE_SIZE"] * 1024.0 * kw
        if "render_kernel_pmc_per_launch" in out and "WRITE_SIZE" in out["render_kernel_pmc_per_launch"]:
            rr = out["render_kernel_pmc_per_launch"]
            out["render_kernel_hbm_bytes_per_launch"] = {"read": rr["FETCH_SIZE"] * 1024.0 * kf, "write": rr["WRITE_SIZE"] * 1024.0 * kw}
            rd += rr["FETCH_SIZE"] * 1024.0 * kf
            wr += rr["WRITE_SIZE"] * 1024.0 * kw
        out["calibration"] = {"known_bytes_each_way": known, "fetch_factor": kf, "write_factor": kw,
                              "note": "factor = known bytes / (counter*1024) on the dword-per-lane copy kernel; applied to the step kernel"}
        out["step_kernel_hbm_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr}
        tj = os.path.join(dst, "traffic.json")
        traffic = json.load(open(tj)) if os.path.exists(tj) else {}
        traffic[f"{workload}|{envs}|{obs}"] = {"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr, "round": rnd, "kernel_sha16": sha,
                             "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), calibrated on lcr_calib_copy_kernel"}
        json.dump(traffic, open(tj, "w"), indent=1)
    json.dump(out, open(os.path.join(dst, f"{stem}_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
