#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile or tiled" > $O/pytest_tiled.log 2>&1; echo "pytest rc=$?"
tail -2 $O/pytest_tiled.log
for i in 1 2; do
for v in base imm; do
WGNN_LIB=$PWD/scratch/variants/libwgnn_$v.so REPS=3 timeout 600 python scratch/pairs_ab.py > $O/ab_$v$i.log 2>&1; echo "$v rc=$?"
grep "rep 2.*pairs" $O/ab_$v$i.log
done
done
