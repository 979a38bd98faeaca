#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03quads; mkdir -p $O
for i in 1 2; do
for v in base quads; do
WGNN_LIB=$PWD/scratch/variants/libwgnn_$v.so REPS=3 timeout 600 python scratch/pairs_ab.py > $O/ab_$v$i.log 2>&1; echo "$v rc=$?"
grep "rep 2.*pairs" $O/ab_$v$i.log
done
done
