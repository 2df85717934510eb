#!/bin/bash
# Kernel statistics and HBM-traffic counters of the default bench.py run (three separate runs:
# trace + stats, FETCH_SIZE, WRITE_SIZE).  Output under gpurun_out/prof_<tag>/.
set -e
tag=${1:-run}
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out.bench.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o f --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o w --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null
python3 $repo/profiles/summarize_pmc.py $out/fetch/f_counter_collection.csv $out/write/w_counter_collection.csv > $out/hbm_traffic.csv
rm -rf $out/fetch $out/write $out/stats/s_kernel_trace.csv
ls $out $out/stats
