#!/bin/bash
# Texture-addresser / L1 pressure per kernel of the default bench.py run -> gpurun_out/ta.txt
set -e
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/pmc_ta
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace -d $out -o m --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
python3 $repo/tools/pmc_table.py $(find $out -name m_counter_collection.csv) > $repo/gpurun_out/ta.txt
rm -rf $out
