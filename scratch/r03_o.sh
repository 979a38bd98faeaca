#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in unr1 unr3; do timeout 300 python scratch/var_time.py scratch/variants/libwgnn_$v.so 78 2>&1 | grep -v amdgpu; done; done
python -m pytest tests -m gpu -x -q -k "loader or flat or tiled" 2>&1 | tail -2
