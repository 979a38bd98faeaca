#!/bin/bash
# round 4: timing-only ablations of the entry pipeline (norl: no per-pair v_readlane; nowt: no weight ds_read_b64) - upper bounds of
# feeding packed words / weights through the scalar cache.  Same box, interleaved repetitions.
cd /root/repo
mkdir -p gpurun_out/r04q
for rep in 1 2; do
  for v in abl_base abl_norl abl_nowt abl_norlwt; do
    timeout 300 python scratch/var_time.py scratch/variants/libwgnn_$v.so 78 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r04q/times.txt
  done
done
