#!/bin/bash
# Kernel statistics and HBM-traffic counters of a bench.py run (three separate runs: trace + stats,
# FETCH_SIZE, WRITE_SIZE).  usage: tools/profile_bench.sh <tag> [bench.py arguments]
# Output under gpurun_out/prof_<tag>/: stats/ (rocprofv3 --stats tables), hbm_traffic.csv, and
# gpurun_out/prof_<tag>.bench.json (the bench line of the traced run).
set -e
tag=${1:-run}
shift || true
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes "$@" > $out.bench.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o f --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o w --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
python3 $repo/profiles/summarize_pmc.py $(find $out/fetch -name 'f_counter_collection.csv') $(find $out/write -name 'w_counter_collection.csv') > $out/hbm_traffic.csv
cp $(find $out/stats -name 's_kernel_stats.csv') $out/kernel_stats.csv
rm -rf $out/fetch $out/write $out/stats
ls $out
