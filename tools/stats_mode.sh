#!/bin/bash
# Kernel statistics of a short bench.py run in one matmul mode: tools/stats_mode.sh <mode> -> gpurun_out/stats_<mode>.csv
set -e
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/stats_$1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out -o s --output-format csv -- python3 $repo/bench.py --matmul $1 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-alt-modes > /dev/null
cp $out/s_kernel_stats.csv $repo/gpurun_out/stats_$1.csv
rm -rf $out
