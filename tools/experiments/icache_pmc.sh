#!/bin/bash
# Run ON THE GPU BOX: instruction-cache counters of the narrow passes (is the straight-line code too big for the 64 KB two CUs share?).
#   tools/experiments/icache_pmc.sh            -> gpurun_out/icache/summary.txt
export TMPDIR=/tmp
D=$PWD/gpurun_out/icache
mkdir -p $D
rocprofv3 -L > $D/counters.txt 2>&1
grep -i -o "SQC_[A-Z_]*ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH\|SQ_INST_LEVEL_[A-Z_]*" $D/counters.txt | sort -u > $D/avail.txt
echo "available: $(tr '\n' ' ' < $D/avail.txt)"
run() { tag=$1; shift; rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/$tag -- python3 "$@" > $D/$tag.log 2>&1 || { echo "$tag failed"; tail -3 $D/$tag.log; }; }
run rg0 tools/ragged_probe.py 0
run rg2 tools/ragged_probe.py 2
run rg3 tools/ragged_probe.py 3
run rs2 tools/narrow_probe.py 2
python3 - "$D" > $D/summary.txt <<'PY'
import sys, glob, csv, collections
D = sys.argv[1]
for tag in ("rg0", "rg2", "rg3", "rs2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(D + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        if k.startswith("k_narrow") or k.startswith("k_dp"):
            v = {c: sum(x) / len(x) for c, x in d.items()}
            req, miss = v.get("SQC_ICACHE_REQ", 0), v.get("SQC_ICACHE_MISSES", 0)
            print(tag, k, {c: round(x) for c, x in sorted(v.items())}, "miss rate %.4f" % (miss / req if req else float("nan")))
PY
cat $D/summary.txt
