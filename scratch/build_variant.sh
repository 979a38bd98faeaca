#!/bin/bash
# usage: build_variant.sh NAME [ABLATE_LIST]  -> scratch/variants/libwgnn_NAME.so
set -e
cd /root/repo
mkdir -p scratch/variants /tmp/var_$1
cp scdeepsort_amd/csrc/*.h scdeepsort_amd/csrc/*.hip /tmp/var_$1/
WGNN_GEN_ABLATE="$2" python scdeepsort_amd/csrc/gen_flat_asm.py /tmp/var_$1/wgnn_flat_asm.inc >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -Wno-inline-asm -Iinclude /tmp/var_$1/wgnn_kernels.hip /tmp/var_$1/wgnn_tiled.hip -o scratch/variants/libwgnn_$1.so
echo built $1
