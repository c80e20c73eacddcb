#!/bin/bash
# Run ON THE GPU BOX: PMC counters of k_dp on single-class batches.  tools/experiments/class_pmc.sh CAP [CAP ...]
export TMPDIR=/tmp
for cap in "$@"; do
  D=$PWD/gpurun_out/pmc_class_$cap; mkdir -p $D
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/sq -- python3 tools/experiments/class_pmc.py $cap > $D/sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $D/sq2 -- python3 tools/experiments/class_pmc.py $cap > $D/sq2.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $D/tcc -- python3 tools/experiments/class_pmc.py $cap > $D/tcc.log 2>&1
  python3 - "$D" $cap <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dp<false, false>" not in r["Kernel_Name"]:
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if dur < 500:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc["dur_us"].append(dur)
print("cap", sys.argv[2], {c: round(sum(v[-2:]) / len(v[-2:]), 1) for c, v in sorted(acc.items())})
PY
done
