#!/bin/bash
# (written when the {p}-only table and a three-slot ring were the defaults: -DMPB_NAR_LUT128 is now a no-op, -DMPB_NAR_LUT64 /
# -DMPB_NAR_DEPTH=3 select the old forms)
# Run ON THE GPU BOX: k_narrow with and without the shared-line tails (TAIL), ring depth with tails, table width; time and
# memory-side reads per launch.
export TMPDIR=/tmp
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
build() { /opt/rocm/bin/hipcc $FL -DMPB_TUNING_KNOBS -DMPB_NAR_TAILS $2 moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/$1.so 2>/tmp/var/$1.err || { tail -5 /tmp/var/$1.err; exit 1; }; }
build t2 ""; build t3 "-DMPB_NAR_DEPTH_TAIL=3"; build t2_128 "-DMPB_NAR_LUT128"; build t3_128 "-DMPB_NAR_LUT128 -DMPB_NAR_DEPTH_TAIL=3"
run() { echo "$1 $2: $(env $2 MOIRA_PB_LIB=/tmp/var/$1.so python3 tools/narrow_probe.py ${R:-2} 10000000 2>&1 | tail -1)"; }
for rep in 1 2; do
  run t2 MPB_NAR_NO_TAIL=1; run t2 X=1; run t3 X=1; run t2_128 MPB_NAR_NO_TAIL=1; run t2_128 X=1; run t3_128 X=1
done
traffic() {
  D=/tmp/nt_$1_$3; rm -rf $D
  env $2 MOIRA_PB_LIB=/tmp/var/$1.so rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --kernel-trace --output-format csv -d $D -- python3 tools/narrow_probe.py 2 10000000 > $D.log 2>&1
  python3 - $D "$1 $2" <<'PY'
import sys, glob, csv
v = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_narrow" in r["Kernel_Name"]:
            v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
gb = sum(sum(x) / len(x) * (128 if "128B" in k else 64) for k, x in v.items()) / 1e9
print("%s | k_narrow reads %.3f GB per launch" % (sys.argv[2], gb))
PY
}
traffic t2 MPB_NAR_NO_TAIL=1 a; traffic t2 X=1 b; traffic t3 X=1 c
