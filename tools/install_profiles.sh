#!/bin/bash
# Copy what tools/profile_bench.sh (run for r03_cfg1, r03_cfg2, r03_cfg2_bf16, r03_cfg3, r03_cfg5, the last one
# last) and tools/pmc_bench_mfma.sh left under gpurun_out/ into profiles/.
set -e
cd "$(dirname "$0")/.."
for t in cfg1 cfg2 cfg2_bf16 cfg3 cfg5; do
  cp gpurun_out/prof_r03_$t/kernel_stats.csv profiles/r03_${t}_kernel_stats.csv
  cp gpurun_out/prof_r03_$t/hbm_traffic.csv profiles/r03_${t}_hbm_traffic.csv
  cp gpurun_out/prof_r03_$t.bench.json profiles/r03_${t}_bench.json
done
cp gpurun_out/prof_r03_cfg5/hbm_traffic_per_launch.json profiles/hbm_traffic_per_launch.json
[ -f gpurun_out/mfma_busy_f32.csv ] && cp gpurun_out/mfma_busy_f32.csv profiles/r03_cfg2_mfma_busy.csv
[ -f gpurun_out/mfma_busy_bf16.csv ] && cp gpurun_out/mfma_busy_bf16.csv profiles/r03_cfg2_bf16_mfma_busy.csv
