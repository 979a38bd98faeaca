#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "linear_fwd_ex or wgnn_linear or int64 or large_seed or golden or refcode or fp16" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 600 python scratch/shard_sizes.py > $O/shard_sizes.log 2>&1; echo "shard rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_cfg3.log 2>&1; echo "cfg3 rc=$?" >> $O/rc.txt
WGNN_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_n8_shared.log 2>&1; echo "n8 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.log 2>&1; echo "cfg5 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/pytest_sel.log; tail -2 $O/shard_sizes.log
