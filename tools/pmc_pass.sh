#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass over a short bench run, counters given as arguments;
# prints per-kernel averages.  tools/pmc_pass.sh TAG COUNTER [COUNTER ...]
TAG=$1; shift
export TMPDIR=/tmp
D=$PWD/gpurun_out/pmc_$TAG
mkdir -p $D
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS > $D/run.log 2>&1
python3 - "$D" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
