#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
for rep in 1 2; do
for v in base fma2; do
  timeout 300 python scratch/var_time.py scratch/variants/libwgnn_$v.so 78 >> $O/var.log 2>&1
done; done
cat $O/var.log | grep -v amdgpu.ids
