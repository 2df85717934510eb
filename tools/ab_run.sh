#!/bin/bash
# Same-box A/B timing of two builds of the library (box-to-box variance on the pool is a few per cent,
# larger than most kernel changes).  Put the builds at ab/lib_base.so and ab/lib_new.so (ab/ is not
# tracked; AB_VARIANTS="a b c" names other builds ab/lib_<name>.so; AB_ARGS = extra bench.py
# arguments), then: gpurun -- 'bash tools/ab_run.sh'; results in gpurun_out/ab_<build>_<round>.json.
for i in 1 2; do
  for v in ${AB_VARIANTS:-base new}; do
    GUNET_HIP_LIBRARY=$GRAFT_REPO_ROOT/ab/lib_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-modes --steps ${AB_STEPS:-30} $AB_ARGS > gpurun_out/ab_${v}_$i.json 2>/dev/null || exit 1
  done
done
