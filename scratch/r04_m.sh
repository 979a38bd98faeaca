#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r04m/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04m/pytest.log
timeout 1200 python scratch/tune_gemms.py > gpurun_out/r04m/tune.log 2>&1; tail -1 gpurun_out/r04m/tune.log
cp scdeepsort_amd/tuned_gemms_gfx950.csv gpurun_out/r04m/
for i in 1 2; do WGNN_BENCH_CONFIG=cfg2 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'], d['value'], d['config']['eager_ms_per_step'])"; done
