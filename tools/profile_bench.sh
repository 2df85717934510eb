#!/bin/bash
# HBM-traffic counters and kernel statistics of a bench.py run, in the order that keeps the committed
# bench line current: the two counter passes (FETCH_SIZE, WRITE_SIZE; one counter per run) first, then the
# traffic table of this workload is rebuilt and stamped with the hash of the library sources, and only then
# the traced bench run whose JSON line reads that table.
# usage: tools/profile_bench.sh <tag> <workload key of bench.py, or - for none> [bench.py arguments]
# Output under gpurun_out/prof_<tag>/: hbm_traffic.csv, kernel_stats.csv, hbm_traffic_per_launch.json (the
# whole table as it stands after this run), and gpurun_out/prof_<tag>.bench.json (the bench line).
set -e
tag=${1:-run}
key=${2:--}
shift 2 || true
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o f --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o w --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
python3 $repo/profiles/summarize_pmc.py $(find $out/fetch -name 'f_counter_collection.csv') $(find $out/write -name 'w_counter_collection.csv') > $out/hbm_traffic.csv
rm -rf $out/fetch $out/write
if [ "$key" != "-" ]; then
  # the table names the csv by the path it is committed under
  cp $out/hbm_traffic.csv $repo/profiles/${tag}_hbm_traffic.csv
  python3 $repo/profiles/make_traffic_table.py $key $repo/profiles/${tag}_hbm_traffic.csv
  cp $repo/profiles/hbm_traffic_per_launch.json $out/hbm_traffic_per_launch.json
fi
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes "$@" > $out.bench.json
cp $(find $out/stats -name 's_kernel_stats.csv') $out/kernel_stats.csv
rm -rf $out/stats
ls $out
