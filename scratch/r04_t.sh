#!/bin/bash
# round 4: cost model of a half-wave pipeline for D <= 128 (one ds_read_b128 and one FMA pair per entry PAIR) vs production
cd /root/repo; mkdir -p gpurun_out/r04t
for rep in 1 2; do for v in abl_base abl_noshared abl_half; do
  timeout 300 python scratch/var_time128.py scratch/variants/libwgnn_$v.so 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r04t/times.txt
done; done
