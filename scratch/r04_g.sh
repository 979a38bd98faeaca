#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
timeout 1200 python scratch/tune_gemms.py > gpurun_out/r04g/tune.log 2>&1; tail -1 gpurun_out/r04g/tune.log
cp scdeepsort_amd/tuned_gemms_gfx950.csv gpurun_out/r04g/ 2>/dev/null
timeout 600 python bench.py > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r04g/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['outside_agg_kernels_ms_per_step'], d['config']['gemm_selection'][:40], d['train_step'], [ (p['rows'],p['avg_ms']) for p in d['roofline']['passes']])"
WGNN_BENCH_TUNED_GEMMS=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04g/bench_untuned.json 2> gpurun_out/r04g/bench_untuned.err; python -c "
import json; d=json.loads(open('gpurun_out/r04g/bench_untuned.json').read().strip().splitlines()[-1]); print('untuned', d['ms_per_step'], d['roofline']['outside_agg_kernels_ms_per_step'], d['train_step'])"
timeout 900 python -m pytest tests -m gpu -q -x -k "tuned_gemm or linear" > gpurun_out/r04g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04g/pytest.log
