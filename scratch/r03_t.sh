#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
date
WGNN_BENCH_DUMP_AFTER=240 WGNN_BENCH_SHARE_GPU=1 timeout 420 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_n8_shared.log 2>&1; echo "n8 rc=$?"
date
grep '^{' $O/bench_n8_shared.log | cut -c1-300
grep -n "File \"/root/repo\|File \".*bench.py" $O/bench_n8_shared.log | head -40
