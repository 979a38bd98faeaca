#!/bin/bash
# 4- and 8-rank shared-GPU dry runs of the final tree (all ranks time-slice ONE GPU over gloo: plumbing evidence, not a scaling number)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
WGNN_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 4 --steps 3 --warmup 1 --no-secondary > gpurun_out/r04o/bench_n4.log 2> gpurun_out/r04o/bench_n4.err; echo "n4 rc=$?"
grep '^{' gpurun_out/r04o/bench_n4.log | cut -c1-300
WGNN_BENCH_SHARE_GPU=1 WGNN_BENCH_DUMP_AFTER=1500 timeout 1700 python bench.py --gpus 8 --steps 3 --warmup 1 --no-secondary > gpurun_out/r04o/bench_n8.log 2> gpurun_out/r04o/bench_n8.err; echo "n8 rc=$?"
grep '^{' gpurun_out/r04o/bench_n8.log | cut -c1-300
tail -3 gpurun_out/r04o/bench_n8.err | cut -c1-300
