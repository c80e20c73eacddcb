#!/bin/bash
# Run ON THE GPU BOX: k_narrow_rg with the rows' later lines requested when their group is armed (rg_touch_rows), whole and stream alone
# (all four without the forced four waves per SIMD: the eight touch registers would spill, and spills are vector-memory operations).
mkdir -p /tmp/var
for v in rg_no_min_waves rg_no_min_waves+rg_touch_rows rg_no_min_waves+rg_stream_alone rg_no_min_waves+rg_stream_alone+rg_touch_rows; do python tools/experiments/make_variant.py $v /tmp/var/$v.hip || exit 1; done
echo "## ragged 5 M x U{50..600} stride ${PROBE_STRIDE:-640}, R = 3"
VARIANT_CMD="python tools/ragged_probe.py 3" tools/experiments/variants.sh -n ${N:-3} whole:""@/tmp/var/rg_no_min_waves.hip touch:""@/tmp/var/rg_no_min_waves+rg_touch_rows.hip stream_alone:""@/tmp/var/rg_no_min_waves+rg_stream_alone.hip stream_touch:""@/tmp/var/rg_no_min_waves+rg_stream_alone+rg_touch_rows.hip
