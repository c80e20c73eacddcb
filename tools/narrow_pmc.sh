#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 passes over the clean 10 M x 300 batch through the narrow pass (tools/narrow_probe.py):
# kernel trace + stats, then the counters in passes of their own (never combined with sys / hip / hsa tracing).
#   tools/narrow_pmc.sh TAG [R]        -> gpurun_out/narrow_pmc_TAG/{trace,fetch,write,sq,lds}/ + summary.txt
#   PROBE=tools/ragged_probe.py N=5000000 tools/narrow_pmc.sh TAG 3      the same passes over a ragged batch (k_narrow_rg)
TAG=${1:-r05}; R=${2:-2}; PROBE=${PROBE:-tools/narrow_probe.py}; N=${N:-10000000}
export TMPDIR=/tmp
D=$PWD/gpurun_out/narrow_pmc_$TAG
mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $PROBE $R $N > $D/trace.log 2>&1 || exit 1
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $D/$name -- python3 $PROBE $R $N > $D/$name.log 2>&1 || exit 1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tccrd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass tccwr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
python3 - "$D" > $D/summary.txt <<'PY'
import sys, glob, csv, collections
D = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(D + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())})
for f in glob.glob(D + "/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
PY
cat $D/summary.txt
