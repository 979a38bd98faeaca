#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pairs; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shared_pairs or loader" > $O/pytest_pairs.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest_pairs.log
