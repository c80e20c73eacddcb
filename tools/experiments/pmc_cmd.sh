#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass over any python script of the tree, per-kernel averages printed.
#   tools/experiments/pmc_cmd.sh TAG "COUNTER COUNTER ..." tools/poisson_rate.py 10000000
TAG=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
D=$PWD/gpurun_out/pmc_$TAG
mkdir -p $D
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $D -- python3 "$@" > $D/run.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $D/run.log; }
python3 - "$D" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
