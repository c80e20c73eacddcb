#!/bin/bash
# Run ON THE GPU BOX: the narrow passes' table as {1 - p, p'} pairs (shipped) against {p'} alone (nar_lut64), with and without the
# chained halves of k_narrow_rs, interleaved: fixed-length clean batch (k_narrow_rs) and ragged clean batch (k_narrow_rg).
mkdir -p /tmp/var
for v in nar_lut64 rs_chained_halves nar_lut64+rs_chained_halves; do python tools/experiments/make_variant.py $v /tmp/var/$v.hip || exit 1; done
V='base:"" lut64:""@/tmp/var/nar_lut64.hip chained:""@/tmp/var/rs_chained_halves.hip lut64_chained:""@/tmp/var/nar_lut64+rs_chained_halves.hip'
for R in ${ROWS:-2 3}; do
  echo "## fixed length 10 M x 300, R = $R"
  VARIANT_CMD="python tools/narrow_probe.py $R" eval tools/experiments/variants.sh -n ${N:-2} $V
done
for R in ${ROWS:-2 3}; do
  echo "## ragged 5 M x U{50..600}, R = $R"
  VARIANT_CMD="python tools/ragged_probe.py $R" eval tools/experiments/variants.sh -n ${N:-2} base:\"\" lut64:\"\"@/tmp/var/nar_lut64.hip
done
