#!/bin/bash
# usage: scratch/prof_forward.sh <tag>   (run on the GPU box through gpurun)
set -x
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fwd -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -2 $OUT/bench.log
find $OUT -name "*stats*" | head
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -25 "$F"
