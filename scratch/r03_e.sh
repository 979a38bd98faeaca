#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python scratch/tunable_gemm.py > $O/tunable.log 2>&1; echo "tunable rc=$?" >> $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_auto.log 2>&1
WGNN_LINEAR=never timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_never.log 2>&1
timeout 600 python -m pytest tests -m gpu -x -q -k "linear or nograd or fp16" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; grep -v "^$" $O/tunable.log | head -60; tail -3 $O/pytest_sel.log
for f in auto never; do python - $O/bench_$f.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
print(sys.argv[1], d['ms_per_step'], 'agg', r['agg_kernels_ms_per_step'], 'outside', r['outside_agg_kernels_ms_per_step'], 'avg', r['avg_launch_ms'])
PY
done
