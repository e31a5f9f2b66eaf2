#!/bin/bash
# GPU box: rocprofv3 kernel stats of the lock-step path (BASELINE config E) -> gpurun_out/prof_e_<tag>/
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_e_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/bench_configs.py E > $OUT/run.log 2>&1
cat $OUT/*/*_kernel_stats.csv | cut -c1-160 | head -8
