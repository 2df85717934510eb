#!/bin/bash
# Kernel trace of a short bench run; tools/trace_step.py prints the last forward kernel by kernel.
set -e
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_step
timeout -s KILL 200 rocprofv3 --kernel-trace -d $out -o tr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-alt-modes "$@" > /dev/null
