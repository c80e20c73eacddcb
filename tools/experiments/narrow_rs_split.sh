#!/bin/bash
# Run ON THE GPU BOX: k_narrow_rs against k_narrow (MPB_NAR_NO_RS=1), and k_narrow_rs' two halves alone:
# the panel stream (-DMPB_NAR_NOARITH: loads + tile writes) and the arithmetic (-DMPB_NAR_NODMA: stale panels).
for R in ${ROWS:-2 3}; do
  echo "## R = $R"
  VARIANT_CMD="python tools/narrow_probe.py $R" tools/experiments/variants.sh -n ${N:-2} rs:"" stream_alone:"-DMPB_NAR_NOARITH" arith_alone:"-DMPB_NAR_NODMA" $EXTRA
  echo -n "k_narrow (LDS-DMA): "; MPB_NAR_NO_RS=1 python tools/narrow_probe.py $R
done
