#!/bin/bash
# Matrix-core occupancy and clock per kernel of the default bench.py run -> gpurun_out/mfma_busy.csv
set -e
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/pmc_mfma
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out -o m --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
python3 $repo/profiles/summarize_mfma.py $(find $out -name m_counter_collection.csv) $(find $out -name m_kernel_trace.csv) > $repo/gpurun_out/mfma_busy.csv
rm -rf $out
