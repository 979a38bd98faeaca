#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in cur wrl; do timeout 300 python scratch/var_time.py scratch/variants/libwgnn_$v.so 78 2>&1 | grep -v amdgpu; done; done
