#!/bin/bash
# Run ON THE GPU BOX: interleaved in-session comparison of compile-time variants of the library.
#   tools/variants.sh [-n rounds] name1:"-DX=1 -DY=2" name2:"" ...
#   VARIANT_CMD="python tools/config5_rate.py 5000000" VARIANT_TAIL=2 tools/variants.sh ...   (another workload)
N=3; if [ "$1" = "-n" ]; then N=$2; shift 2; fi
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -x hip"
mkdir -p /tmp/var; cp moira_amd/libmoira_pb.so /tmp/var/_orig.so
names=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; names+=($name)
  /opt/rocm/bin/hipcc $FL $defs moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp -o /tmp/var/$name.so 2>/tmp/var/$name.err || { echo "build of $name failed"; tail -5 /tmp/var/$name.err; exit 1; }
done
for i in $(seq $N); do for v in "${names[@]}"; do
  cp /tmp/var/$v.so moira_amd/libmoira_pb.so
  if [ -n "$VARIANT_CMD" ]; then echo "$v: $($VARIANT_CMD 2>&1 | tail -${VARIANT_TAIL:-1})"; else
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v})"; fi
done; done
cp /tmp/var/_orig.so moira_amd/libmoira_pb.so
