#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pairs; mkdir -p $O
for i in 1 2; do
WGNN_LIB=$PWD/scratch/_ab/libwgnn_old.so REPS=3 timeout 600 python scratch/pairs_ab.py > $O/ab_old$i.log 2>&1; echo "old rc=$?"
grep "rep 2" $O/ab_old$i.log
REPS=3 timeout 600 python scratch/pairs_ab.py > $O/ab_new$i.log 2>&1; echo "new rc=$?"
grep "rep 2" $O/ab_new$i.log
done
