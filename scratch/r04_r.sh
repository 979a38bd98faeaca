#!/bin/bash
# round 4: timing model of a scalar-cache entry feed (norl,nowt + one drained s_load_dwordx16 per 4 pair steps) against its
# upper bound (norl,nowt) and the production pipeline.  Same box, interleaved repetitions.
cd /root/repo
mkdir -p gpurun_out/r04r
for rep in 1 2 3; do
  for v in abl_base abl_norlwt abl_smem; do
    timeout 300 python scratch/var_time.py scratch/variants/libwgnn_$v.so 78 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r04r/times.txt
  done
done
