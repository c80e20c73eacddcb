set -x
rm -rf gpurun_out/prof_r06 gpurun_out/narrow_pmc_r06 gpurun_out/narrow_pmc_r06rg
timeout -k 10 400 bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1 || echo "collect failed"
timeout -k 10 300 bash tools/narrow_pmc.sh r06 2 > gpurun_out/narrow_pmc_r06.log 2>&1 || echo "narrow pmc failed"
PROBE=tools/ragged_probe.py N=5000000 timeout -k 10 300 bash tools/narrow_pmc.sh r06rg 3 > gpurun_out/narrow_pmc_r06rg.log 2>&1 || echo "ragged pmc failed"
timeout -k 10 300 python tools/ragged_rate.py > gpurun_out/r06_ragged_rate.txt 2>&1
timeout -k 10 600 python bench.py > gpurun_out/r06_bench.out 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.out 2> gpurun_out/r06_bench_steps20.err; echo "bench20 rc=$?"
tail -3 gpurun_out/r06_ragged_rate.txt
