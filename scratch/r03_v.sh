#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
date
WGNN_BENCH_DUMP_AFTER=150 WGNN_BENCH_SHARE_GPU=1 timeout 330 python bench.py --gpus 4 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_n4_shared.log 2>&1; echo "n4 rc=$?"
date
grep '^{' $O/bench_n4_shared.log | cut -c1-300
grep -c "Timeout (0" $O/bench_n4_shared.log
