#!/usr/bin/env python
"""Reduce rocprofv3 --pmc CSVs (gpurun_out/<prefix>*/k_counter_collection.csv, produced over
tools/prof_kernels.py; `python tools/pmc_summary.py <tag> <prefix> "<when>"`) to profiles/<tag>_pmc_summary.json and profiles/hbm_traffic.json.

HBM-side traffic per launch of the segment-reduce kernel = FETCH_SIZE * read_scale + WRITE_SIZE * 1024, with
read_scale calibrated on the calibration launch of the same kernel (known fabric read volume, see
prof_kernels.py); both counters are reported by rocprofv3 in KiB.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prefix = sys.argv[2] if len(sys.argv) > 2 else "pmc_"          # gpurun_out/<prefix>*/ directories of the --pmc passes
collected = sys.argv[3] if len(sys.argv) > 3 else tag
rows = []
for p in glob.glob(os.path.join(ROOT, "gpurun_out", prefix + "*", "*counter_collection.csv")):
    rows += list(csv.DictReader(open(p)))
agg = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "k_segreduce" in k:
        name = "segreduce_calib" if int(r["Grid_Size"]) < 4_000_000 else "segreduce_c3"
    elif "k_posttrans_x3" in k:
        name = "posttrans_bf16x3_c3"
    elif "k_posttrans" in k:
        name = "posttrans_f32_c3"
    elif "k_heavy_finalize" in k:
        name = "heavy_finalize_c3"
    else:
        continue
    agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
out = collections.defaultdict(dict)
for (name, c), v in agg.items():
    out[name][c] = sum(v) / len(v)
out = {k: dict(sorted(v.items())) for k, v in out.items()}
F, Vc, Ec = 75, 500_000, 4_000_000
true_read = Ec * 384 + Ec * 4 + (Vc + 1) * 4
summary = {"counters_mean_per_launch": out}
if "segreduce_calib" in out and "FETCH_SIZE" in out["segreduce_calib"]:
    scale = true_read / out["segreduce_calib"]["FETCH_SIZE"]          # bytes per FETCH_SIZE unit
    c3 = out.get("segreduce_c3", {})
    fin = out.get("heavy_finalize_c3", {})
    rd = (c3.get("FETCH_SIZE", 0) + fin.get("FETCH_SIZE", 0)) * scale
    wr = (c3.get("WRITE_SIZE", 0) + fin.get("WRITE_SIZE", 0)) * 1024
    summary["calibration"] = {"true_fabric_read_bytes": true_read, "FETCH_SIZE": out["segreduce_calib"]["FETCH_SIZE"],
                              "bytes_per_FETCH_SIZE_unit": scale,
                              "note": "1024 would mean exact KiB; ~2048 = the documented gfx950 half-count of 128-B requests"}
    summary["pna_segreduce_c3"] = {"fabric_read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                                   "hbm_bytes_per_launch": rd + wr,
                                   "algorithmic_bytes_per_launch": 10_000_000 * (4 * F + 4) + 4 * 1_000_001 + 1_000_000 * 16 * F}
    traffic = {"collected": collected, "pna_segreduce_c3": summary["pna_segreduce_c3"], "calibration": summary["calibration"]}
    for key, name in (("posttrans_bf16x3_c3", "pna_posttrans_x3_c3"), ("posttrans_f32_c3", "pna_posttrans_f32_c3")):
        k = out.get(key, {})
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:      # same byte scale: the contraction's A loads are 16-byte gathers too
            summary[name] = {"fabric_read_bytes_per_launch": k["FETCH_SIZE"] * scale, "write_bytes_per_launch": k["WRITE_SIZE"] * 1024,
                             "hbm_bytes_per_launch": k["FETCH_SIZE"] * scale + k["WRITE_SIZE"] * 1024,
                             "algorithmic_bytes_per_launch": 1_000_000 * (4 * F * 4 + F * 4) + 2 * 1_000_000 * 4}
            traffic[name] = summary[name]
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary.get("pna_segreduce_c3"), indent=1), json.dumps(summary.get("calibration"), indent=1))
