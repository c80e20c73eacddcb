#!/bin/bash
# Run ON THE GPU BOX: k_narrow's memory-side read requests and time against the number of resident workgroups (the L2
# footprint of lines waiting for their second 64-byte half) and with non-temporal requests.
export TMPDIR=/tmp
export MPB_NAR_NO_RS=1          # the ring form (k_narrow): at this stride the library would take k_narrow_rs
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
/opt/rocm/bin/hipcc $FL -DMPB_TUNING_KNOBS moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/knobs.so || exit 1

run() {  # lib grid
  export MOIRA_PB_LIB=/tmp/var/$1.so MPB_NAR_GRID=$2
  D=/tmp/ng_$1_$2; rm -rf $D
  rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --kernel-trace --output-format csv -d $D -- python3 tools/narrow_probe.py 2 10000000 > $D.log 2>&1
  python3 - $D "$1 grid=$2: $(tail -1 $D.log)" <<'PY'
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
for g in 256 512 768; do run knobs $g; done

