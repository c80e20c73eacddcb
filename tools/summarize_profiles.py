#!/usr/bin/env python3
"""(reads also counted exactly: TCC_EA0_RDREQ_32B/64B/128B x 32/64/128 B -- on gfx950 practically every
L2 fill is a 128-B request, for coalesced streams and per-lane gathers alike, which is why FETCH_SIZE,
= requests x 64 B, reads half.)

Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and profiles/pmc_traffic.json (what
bench.py reports as roofline.traffic for the default workload).

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts 128-B fabric requests at 64 B, so reads are FETCH_SIZE x 2 (our own calibration: k_prepass
streams exactly 3.2e9 B and reports 1 562 500 KiB = half; k_synth writes exactly 3.2e9 B and
WRITE_SIZE reports 3 125 000 KiB = all of it)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def newest(pattern):
    """gpurun merges every call's files into the same tree: take the most recent run's."""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
out = {}
for sub in ("fetch", "write", "tccrd", "tccwr", "sq"):
    files = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if dur < 20:
            continue
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_us"].append(dur)
    for k, v in agg.items():
        out.setdefault(k, {}).update({c + ("" if c == "duration_us" else ""): sum(x) / len(x) for c, x in v.items()})
for k, v in out.items():
    if "FETCH_SIZE" in v:
        v["hbm_read_bytes"] = v["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in v:
        v["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024
    if "TCC_EA0_RDREQ_128B_sum" in v:     # exact: every read request by its size
        v["hbm_read_bytes_exact"] = (32 * v.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0)
                                     + 128 * v["TCC_EA0_RDREQ_128B_sum"])
    if "TCC_EA0_WRREQ_sum" in v:          # write requests are 64 B or 32 B
        v["hbm_write_bytes_exact"] = 64 * v.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (v["TCC_EA0_WRREQ_sum"] - v.get("TCC_EA0_WRREQ_64B_sum", 0))
    if "GRBM_GUI_ACTIVE" in v:
        v["clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8 / v["duration_us"] / 1e3
json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
dp = out.get("k_dp<false, false>", {})
if "hbm_read_bytes" in dp and "hbm_write_bytes" in dp:
    rd = dp.get("hbm_read_bytes_exact", dp["hbm_read_bytes"])
    wr = dp.get("hbm_write_bytes_exact", dp["hbm_write_bytes"])
    dp = dict(dp, hbm_read_bytes=rd, hbm_write_bytes=wr)
    json.dump({"source": "profiles/%s_pmc.json (rocprofv3 --pmc, separate passes: TCC_EA0_RDREQ_{32B,64B,128B} and "
                         "TCC_EA0_WRREQ{,_64B} request counts x their sizes; FETCH_SIZE x 2 / WRITE_SIZE agree)" % tag,
               "workload": {"reads": 10000000, "length": 300, "stride": 320, "seed": 2},
               "kernel": "k_dp", "hbm_read_bytes_per_launch": dp["hbm_read_bytes"],
               "hbm_write_bytes_per_launch": dp["hbm_write_bytes"],
               "hbm_bytes_per_launch": dp["hbm_read_bytes"] + dp["hbm_write_bytes"],
               # how busy the vector ALUs were, normalised by the clock the chip actually held: SQ_INSTS_VALU counts
               # wave-instructions, each occupies its SIMD for 4 cycles (64 lanes over 16); GRBM_GUI_ACTIVE is summed over
               # the 8 XCDs; 256 CUs x 4 SIMDs
               "valu": ({"source": "profiles/%s_pmc.json, pass `sq` of tools/collect_profiles.sh" % tag,
                         "SQ_INSTS_VALU": dp["SQ_INSTS_VALU"], "GRBM_GUI_ACTIVE": dp["GRBM_GUI_ACTIVE"],
                         "duration_us": dp["duration_us"], "clock_ghz": dp.get("clock_ghz"),
                         "simd_cycles_issued": dp["SQ_INSTS_VALU"] * 4,
                         "simd_cycles_available": dp["GRBM_GUI_ACTIVE"] / 8 * 1024,
                         "valu_busy": dp["SQ_INSTS_VALU"] * 4 / (dp["GRBM_GUI_ACTIVE"] / 8 * 1024)}
                        if "SQ_INSTS_VALU" in dp and "GRBM_GUI_ACTIVE" in dp else None)},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: {c: round(x, 3) for c, x in v.items()} for k, v in out.items() if k.startswith(("k_dp", "k_prepass"))}, indent=1))
