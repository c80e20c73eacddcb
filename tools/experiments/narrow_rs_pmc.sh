#!/bin/bash
# Run ON THE GPU BOX: SQ counters of a narrow kernel whole, and of its arithmetic alone / its stream alone (patches of a copy of the
# kernel file, tools/experiments/make_variant.py): clock held, how busy the vector ALU is, where a wave's time goes.
#   VARIANTS="whole: arith_alone:rs_arith_alone stream_alone:rs_stream_alone" R=2 tools/experiments/narrow_rs_pmc.sh
#   PROBE=tools/ragged_probe.py N=5000000 VARIANTS="whole: stream_alone:rg_stream_alone" R=3 tools/experiments/narrow_rs_pmc.sh
export TMPDIR=/tmp
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
PROBE=${PROBE:-tools/narrow_probe.py}; N=${N:-10000000}
mkdir -p /tmp/var
for v in ${VARIANTS:-whole: arith_alone:rs_arith_alone stream_alone:rs_stream_alone}; do
  name=${v%%:*}; patch=${v#*:}; SRC=moira_amd/csrc/mpb_kernels.hip
  if [ "${patch:0:1}" = "@" ]; then python tools/experiments/make_variant.py none /tmp/var/$name.hip ${patch:1} || exit 1; SRC=/tmp/var/$name.hip      # name:@another_copy_of_the_kernels.hip
  elif [ -n "$patch" ]; then python tools/experiments/make_variant.py $patch /tmp/var/$patch.hip || exit 1; SRC=/tmp/var/$patch.hip; fi
  /opt/rocm/bin/hipcc $FL $EXTRA $SRC moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/$name.so 2>/tmp/var/$name.err || { tail -5 /tmp/var/$name.err; exit 1; }
  echo "== $name ($patch $EXTRA)"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" $MORE_SETS; do
    D=/tmp/nrs_$RANDOM
    MOIRA_PB_LIB=/tmp/var/$name.so rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -- python3 $PROBE ${R:-2} $N > $D.log 2>&1
    python3 - $D <<'PY'
import sys, glob, csv
v, dur = {}, []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_narrow" in r["Kernel_Name"]:
            v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
a = {k: sum(x) / len(x) for k, x in v.items()}
d = sum(dur) / len(dur)
print({k: round(x) for k, x in a.items()}, "duration_us %.1f" % d)
cyc = a["GRBM_GUI_ACTIVE"] / 8
if "SQ_INSTS_VALU" in a:
    print("clock %.2f GHz, VALU busy %.3f, wave time: waitcnt %.2f, issue-stall %.2f, active %.2f" % (
        cyc / d / 1e3, a["SQ_INSTS_VALU"] * 4 / (cyc * 1024), a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
        a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
if "SQ_LDS_IDX_ACTIVE" in a:
    print("clock %.2f GHz, LDS busy %.3f of the CUs' cycles, bank-conflict cycles %.3f of LDS-active" % (
        cyc / d / 1e3, a["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)))
PY
  done
done
