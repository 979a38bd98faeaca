#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04e/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r04e/pytest.log
for f in 0 1 0 1; do FUSED=$f timeout 300 python scratch/train_prof_r04.py 2>&1 | tail -1; done
timeout 900 python scratch/tune_gemms.py > gpurun_out/r04e/tune.log 2>&1; tail -1 gpurun_out/r04e/tune.log
