mkdir -p /tmp/var
python tools/experiments/make_variant.py rg_no_min_waves /tmp/var/rg_no_min_waves.hip || exit 1
for R in 3 2 4; do
  echo "## ragged R = $R"
  VARIANT_CMD="python tools/ragged_probe.py $R" tools/experiments/variants.sh -n 3 base:"" nomin:""@/tmp/var/rg_no_min_waves.hip
done
