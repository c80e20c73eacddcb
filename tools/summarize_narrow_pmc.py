#!/usr/bin/env python3
"""gpurun_out/narrow_pmc_<tag>/ (tools/narrow_pmc.sh: rocprofv3 passes over the clean 10 M x 300 batch) -> the committed summaries
profiles/<tag>_hq_kernel_stats.csv, profiles/<tag>_pmc_hq.json and profiles/pmc_traffic_hq.json (what bench.py's
extras.high_quality_300 reports as roofline.traffic / valu_busy_pmc).  Byte counts as tools/summarize_profiles.py derives them
(MI355X_MICROARCH.md, HBM: FETCH_SIZE x 2 on gfx950; exact: TCC_EA0_RDREQ_{32B,64B,128B} x their sizes)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "narrow_pmc_" + tag)
dst = os.path.join(root, "profiles")


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def newest(pattern):
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_hq_kernel_stats.csv"))
out = {}
for sub in ("fetch", "write", "tccrd", "tccwr", "sq", "lds"):
    files = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        if not k.startswith("k_narrow"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_us_" + sub].append(dur)
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
for k, v in out.items():
    if "FETCH_SIZE" in v:
        v["hbm_read_bytes_fetch_size_x2"] = v["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in v:
        v["hbm_write_bytes_write_size"] = v["WRITE_SIZE"] * 1024
    if "TCC_EA0_RDREQ_128B_sum" in v:
        v["hbm_read_bytes_exact"] = (32 * v.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0)
                                     + 128 * v["TCC_EA0_RDREQ_128B_sum"])
    if "TCC_EA0_WRREQ_sum" in v:
        v["hbm_write_bytes_exact"] = 64 * v.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (v["TCC_EA0_WRREQ_sum"] - v.get("TCC_EA0_WRREQ_64B_sum", 0))
    if "GRBM_GUI_ACTIVE" in v and "duration_us_sq" in v:
        v["clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8 / v["duration_us_sq"] / 1e3
        v["valu_busy"] = v["SQ_INSTS_VALU"] * 4 / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
json.dump(out, open(os.path.join(dst, tag + "_pmc_hq.json"), "w"), indent=1, sort_keys=True)
kname = next((x for x in sorted(out, reverse=True) if x.startswith("k_narrow_rs<2") or x.startswith("k_narrow<2")), "")
k = out.get(kname, {})
if "hbm_read_bytes_exact" in k:
    rd, wr = k["hbm_read_bytes_exact"], k.get("hbm_write_bytes_exact", k.get("hbm_write_bytes_write_size", 0))
    json.dump({"source": "profiles/%s_pmc_hq.json (rocprofv3 --pmc, separate passes over tools/narrow_probe.py: TCC_EA0_RDREQ_{32B,64B,128B} "
                         "and TCC_EA0_WRREQ{,_64B} request counts x their sizes; FETCH_SIZE x 2 / WRITE_SIZE agree)" % tag,
               "workload": {"reads": 10000000, "length": 300, "stride": 320, "seed": 2, "profile": 1},
               "kernel": kname, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
               "hbm_bytes_per_launch": rd + wr,
               "valu": {"SQ_INSTS_VALU": k.get("SQ_INSTS_VALU"), "GRBM_GUI_ACTIVE": k.get("GRBM_GUI_ACTIVE"),
                        "duration_us": k.get("duration_us_sq"), "clock_ghz": k.get("clock_ghz"), "valu_busy": k.get("valu_busy")}},
              open(os.path.join(dst, "pmc_traffic_hq.json"), "w"), indent=1)
print(json.dumps(out, indent=1, sort_keys=True))
