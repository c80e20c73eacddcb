#!/bin/bash
# Run ON THE GPU BOX: k_narrow_rg whole, its gather stream alone (no arithmetic: loads, tile writes, epilogues, result stores),
# its arithmetic alone (no row is loaded), and without result stores.   tools/experiments/rg_split.sh
mkdir -p /tmp/var
for v in rg_stream_alone rg_arith_alone rg_no_results; do python tools/experiments/make_variant.py $v /tmp/var/$v.hip || exit 1; done
for R in ${ROWS:-3 2}; do
  echo "## ragged 5 M x U{50..600} stride ${PROBE_STRIDE:-640}, R = $R"
  VARIANT_CMD="python tools/ragged_probe.py $R" tools/experiments/variants.sh -n ${N:-2} whole:"" stream_alone:""@/tmp/var/rg_stream_alone.hip arith_alone:""@/tmp/var/rg_arith_alone.hip no_results:""@/tmp/var/rg_no_results.hip
done
