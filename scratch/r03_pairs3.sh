#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pairs; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile or tiled" > $O/pytest_tiled.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_tiled.log
for i in 1 2; do
WGNN_LIB=$PWD/scratch/_ab/libwgnn_old.so REPS=3 timeout 600 python scratch/pairs_ab.py > $O/ab_old$i.log 2>&1; echo "old rc=$?"
grep "rep 2.*pairs" $O/ab_old$i.log
REPS=3 timeout 600 python scratch/pairs_ab.py > $O/ab_new$i.log 2>&1; echo "new rc=$?"
grep "rep 2.*pairs" $O/ab_new$i.log
done
