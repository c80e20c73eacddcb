#!/bin/bash
# Run ON THE GPU BOX: interleaved in-session comparison of compile-time variants of the library.
#   tools/experiments/variants.sh [-n rounds] [-t] [-p "COUNTERS"] name1:"-DX=1 -DY=2" name2:"" name3:""@path/to/other_kernels.hip ...
#     -t            also run the GPU parity tests against every variant first
#     -p "C1 C2"    also one rocprofv3 --pmc pass per variant (k_dp / k_prepass rows)
#   VARIANT_CMD="python tools/config5_rate.py 5000000" VARIANT_TAIL=2 tools/experiments/variants.sh ...   (another workload)
# Variants are built into /tmp/var and selected with MOIRA_PB_LIB (moira_amd/_lib.py): the tree's library and
# its stamp are never touched, so an interrupted run cannot leave an experiment build behind.
N=3; TESTS=0; PMC=""
while [ "${1:0:1}" = "-" ]; do
  case $1 in -n) N=$2; shift 2;; -t) TESTS=1; shift;; -p) PMC=$2; shift 2;; *) echo "unknown flag $1"; exit 2;; esac
done
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
names=()
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; defs=${rest%%@*}; names+=($name)
  SRC=moira_amd/csrc/mpb_kernels.hip; if [ "$rest" != "$defs" ]; then SRC=${rest#*@}; fi     # name:"-Dflags"@other_kernels.hip
  /opt/rocm/bin/hipcc $FL $defs $SRC moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/$name.so 2>/tmp/var/$name.err || { echo "build of $name failed"; tail -5 /tmp/var/$name.err; exit 1; }
done
if [ $TESTS = 1 ]; then for v in "${names[@]}"; do
  echo "== parity $v: $(MOIRA_PB_LIB=/tmp/var/$v.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -1)"
done; fi
for i in $(seq $N); do for v in "${names[@]}"; do
  export MOIRA_PB_LIB=/tmp/var/$v.so
  if [ -n "$VARIANT_CMD" ]; then echo "$v: $($VARIANT_CMD 2>&1 | tail -${VARIANT_TAIL:-1})"; else
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v})"; fi
done; done
if [ -n "$PMC" ]; then for v in "${names[@]}"; do
  export MOIRA_PB_LIB=/tmp/var/$v.so
  echo "== pmc $v"; tools/pmc_pass.sh v_$v $PMC 2>&1 | grep "k_dp<false, false>\|k_prepass"
done; fi
unset MOIRA_PB_LIB
