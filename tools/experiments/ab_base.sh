#!/bin/bash
# Run ON THE GPU BOX: the tree's kernels against a baseline copy kept aside (tools/experiments/_base_kernels.hip, not tracked),
# interleaved in one session.   CMDS="python tools/narrow_probe.py 2|python tools/narrow_probe.py 3" tools/experiments/ab_base.sh [rounds]
mkdir -p /tmp/var
python tools/experiments/make_variant.py none /tmp/var/base.hip tools/experiments/_base_kernels.hip || exit 1
IFS='|' read -ra LIST <<< "${CMDS:-python tools/narrow_probe.py 2}"
for c in "${LIST[@]}"; do
  echo "## $c"
  VARIANT_CMD="$c" tools/experiments/variants.sh -n ${1:-3} base:""@/tmp/var/base.hip new:""
done
