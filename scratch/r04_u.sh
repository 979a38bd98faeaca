#!/bin/bash
# round 4: LDS reads issued TWO pair steps ahead (WGNN_GEN_DEPTH=2, no shared-pair stream; timing only: the third staging buffer
# sits in accumulator slots 8-10) vs one step ahead, same box, interleaved
cd /root/repo; mkdir -p gpurun_out/r04u
for rep in 1 2 3; do for v in abl_base abl_noshared abl_d2; do
  timeout 300 python scratch/var_time.py scratch/variants/libwgnn_$v.so 78 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r04u/times.txt
done; done
