#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04d/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r04d/pytest.log
for f in 0 1 0 1; do FUSED=$f timeout 300 python scratch/train_prof_r04.py 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d/prof_train
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FUSED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python scratch/train_prof_r04.py > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04d/train_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
timeout 600 python bench.py > gpurun_out/r04d/bench.json 2> gpurun_out/r04d/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['outside_agg_kernels_ms_per_step'], [ (p['rows'],p['avg_ms']) for p in d['roofline']['passes']])"
WGNN_BENCH_CONFIG=cfg2 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04d/bench_cfg2.json 2> gpurun_out/r04d/bench_cfg2.err; python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench_cfg2.json').read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['step_launch'], d['config']['eager_ms_per_step'], [ (p['kernel'], p['rows'],p['avg_ms']) for p in d['roofline']['passes']])"
