#!/bin/bash
# usage: build_variant.sh NAME [ABLATE_LIST] [DEPTH] [SED_EXPR on wgnn_tiled.hip] [ORDER]  -> scratch/variants/libwgnn_NAME.so
set -e
cd /root/repo
mkdir -p scratch/variants /tmp/var_$1
cp scdeepsort_amd/csrc/*.h scdeepsort_amd/csrc/*.hip /tmp/var_$1/
if [ -n "$4" ]; then if [ -f "$4" ]; then sed -i -f "$4" /tmp/var_$1/wgnn_tiled.hip; else sed -i "$4" /tmp/var_$1/wgnn_tiled.hip; fi; fi
WGNN_GEN_ABLATE="$2" WGNN_GEN_DEPTH="${3:-1}" WGNN_GEN_ORDER="${5:-RLAWF}" python scdeepsort_amd/csrc/gen_flat_asm.py /tmp/var_$1/wgnn_flat_asm.inc >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -Wno-inline-asm -Iinclude /tmp/var_$1/wgnn_kernels.hip /tmp/var_$1/wgnn_tiled.hip /tmp/var_$1/wgnn_linear.hip /tmp/var_$1/wgnn_sample.hip /tmp/var_$1/wgnn_train.hip /tmp/var_$1/wgnn_plan.hip /tmp/var_$1/wgnn_transpose.hip -o scratch/variants/libwgnn_$1.so
echo built $1
