#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04v; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python scratch/rccl_timeline.py > $OUT/log.txt 2>&1; echo "rc=$?"
tail -2 $OUT/log.txt | cut -c1-300
python scratch/rccl_timeline_summary.py $OUT/tr $OUT/rccl_timeline.json 2>&1 | cut -c1-260
find $OUT/tr -name "*.csv" -size +2M -delete
