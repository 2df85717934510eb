#!/bin/bash
# usage: tools/trace_gemm.sh <mode> -- kernel trace (no counters) of the GEMM micro-benchmark
set -e
cd /tmp && export TMPDIR=/tmp
export GUNET_MODE=$1
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$1
timeout -s KILL 120 rocprofv3 --kernel-trace -d $out -o tr --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py gemm
