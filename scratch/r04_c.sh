#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04c/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r04c/pytest.log
timeout 900 python scratch/narrow_rows.py > gpurun_out/r04c/narrow.log 2>&1; tail -1 gpurun_out/r04c/narrow.log
WGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-secondary > gpurun_out/r04c/bench2.json 2> gpurun_out/r04c/bench2.err; echo "bench2 rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r04c/bench2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['step_launch'], d['config']['launch_calibration_ms'], d['config']['sharded_vs_unsharded'])"
