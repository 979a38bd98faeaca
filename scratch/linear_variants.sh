#!/bin/bash
# builds libwgnn_hip.so with WGNN_LIN_BK = 16 / 32 and times wgnn_linear_fwd (scratch/linear_time.py)
cd $GRAFT_REPO_ROOT
for BK in 16 32; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -Wno-inline-asm -DWGNN_LIN_BK=$BK -Iinclude scdeepsort_amd/csrc/wgnn_kernels.hip scdeepsort_amd/csrc/wgnn_tiled.hip scdeepsort_amd/csrc/wgnn_linear.hip -o scdeepsort_amd/libwgnn_hip.so
  echo "BK=$BK"; python scratch/linear_time.py 2>&1 | grep "wgnn_mfma\|max"
done
