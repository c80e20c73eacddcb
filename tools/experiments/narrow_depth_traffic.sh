#!/bin/bash
# Run ON THE GPU BOX: k_narrow (final form: pair table, slot refilled early) by ring depth = resident workgroups per CU (4 / 3 / 2):
# time and memory-side reads per launch
export TMPDIR=/tmp
export MPB_NAR_NO_RS=1          # the ring form (k_narrow): at this stride the library would take k_narrow_rs
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
for d in 2 3 4; do /opt/rocm/bin/hipcc $FL -DMPB_NAR_DEPTH=$d moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/dd$d.so 2>/tmp/var/dd$d.err || { tail -5 /tmp/var/dd$d.err; exit 1; }; done
for rep in 1 2 3; do for d in 2 3 4; do for R in 2 3; do
  echo "depth $d: $(MOIRA_PB_LIB=/tmp/var/dd$d.so python3 tools/narrow_probe.py $R 10000000 2>&1 | tail -1)"
done; done; done
for d in 2 3 4; do
  D=/tmp/ndt_$d; rm -rf $D
  MOIRA_PB_LIB=/tmp/var/dd$d.so rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --kernel-trace --output-format csv -d $D -- python3 tools/narrow_probe.py 2 10000000 > $D.log 2>&1
  python3 - $D "depth $d" <<'PY'
import sys, glob, csv
v = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_narrow" in r["Kernel_Name"]:
            v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
gb = sum(sum(x) / len(x) * (128 if "128B" in k else 64) for k, x in v.items()) / 1e9
print("%s | k_narrow reads %.3f GB per launch" % (sys.argv[2], gb))
PY
done
