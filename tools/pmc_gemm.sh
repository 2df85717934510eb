#!/bin/bash
# usage: tools/pmc_gemm.sh <mode> -- counters of the GEMM micro-benchmark into gpurun_out/pmc_<mode>
set -e
cd /tmp && export TMPDIR=/tmp
export GUNET_MODE=$1
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $out -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py gemm
