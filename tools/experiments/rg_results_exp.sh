mkdir -p /tmp/var
for v in rg_coalesced_results rg_no_results; do python tools/experiments/make_variant.py $v /tmp/var/$v.hip || exit 1; done
VARIANT_CMD="python tools/ragged_probe.py 3" tools/experiments/variants.sh -n 3 base:"" coal:""@/tmp/var/rg_coalesced_results.hip nores:""@/tmp/var/rg_no_results.hip
VARIANT_CMD="python tools/ragged_probe.py 2" tools/experiments/variants.sh -n 2 base:"" coal:""@/tmp/var/rg_coalesced_results.hip nores:""@/tmp/var/rg_no_results.hip
