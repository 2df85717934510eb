#!/bin/bash
# What the waves of each kernel of the default bench.py run wait for -> gpurun_out/waits.csv
set -e
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/pmc_waits
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $out -o m --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
python3 $repo/tools/pmc_table.py $(find $out -name m_counter_collection.csv) > $repo/gpurun_out/waits.txt
rm -rf $out
