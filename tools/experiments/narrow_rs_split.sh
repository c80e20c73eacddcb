#!/bin/bash
# Run ON THE GPU BOX: k_narrow_rs against k_narrow (the ring form), and k_narrow_rs' two halves alone: the panel stream (loads + tile
# writes) and the arithmetic (stale panels).  The variants are patches of a copy of the kernel file (make_variant.py), not switches.
mkdir -p /tmp/var
for v in rs_stream_alone rs_arith_alone force_ring; do python tools/experiments/make_variant.py $v /tmp/var/$v.hip || exit 1; done
for R in ${ROWS:-2 3}; do
  echo "## R = $R"
  VARIANT_CMD="python tools/narrow_probe.py $R" tools/experiments/variants.sh -n ${N:-2} rs:"" stream_alone:""@/tmp/var/rs_stream_alone.hip arith_alone:""@/tmp/var/rs_arith_alone.hip ring:""@/tmp/var/force_ring.hip $EXTRA
done
