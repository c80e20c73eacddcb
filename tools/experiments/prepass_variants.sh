#!/bin/bash
# Run ON THE GPU BOX: what the 0.75 ms of a step that is not k_dp could give back.
#   A  the shipped library                         (3 VALU per byte in the prepass: LUT address, packed add, fma)
#   B  -DMPB_PREPASS_NO_K3                         (2 VALU per byte: kappa3 := sigma^2, conservative classes)
# for each: bench step + kernel times (interleaved), PMC of k_prepass / k_dp, and the step as 1 / 2 / 4 sub-batches on
# contexts of their own (tools/experiments/overlap_experiment.py: each sub-batch's prepass / sort overlaps the DP of the others).
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_prepass_variants.txt
{
echo "== interleaved bench runs (ms per step, kernels)"
tools/experiments/variants.sh -n 3 -p "SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" shipped:"" no_k3:"-DMPB_PREPASS_NO_K3"
for v in shipped no_k3; do
  echo "== sub-batches on contexts of their own, variant $v (contexts = sub-batches in flight)"
  MOIRA_PB_LIB=/tmp/var/$v.so timeout -k 10 300 python tools/experiments/overlap_experiment.py
done
} > $OUT 2>&1
cat $OUT
