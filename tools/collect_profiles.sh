#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel trace + stats, then the HBM
# traffic counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is
# never combined with sys/hip/hsa tracing).  Output under gpurun_out/prof_$1/ ; summarise with
# tools/summarize_profiles.py $1 and commit the summaries under profiles/.
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
D=$PWD/gpurun_out/prof_$TAG
mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > $D/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D/write.log 2>&1
# exact memory-side byte counts: read requests by size (32/64/128 B) and write requests (64 B vs 32 B)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $D/tccrd -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D/tccrd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $D/tccwr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D/tccwr.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D/sq.log 2>&1
echo "profiles collected under $D"
