#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
timeout 600 python scratch/shard_sizes.py > $O/shard_sizes.log 2>&1; echo "shard rc=$?"
tail -2 $O/shard_sizes.log
timeout 1500 python scratch/configs_record_r03.py > $O/configs.log 2>&1; echo "configs rc=$?"
tail -3 $O/configs.log
WGNN_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_n8_shared.log 2>&1; echo "n8 rc=$?"
grep '^{' $O/bench_n8_shared.log | cut -c1-300
