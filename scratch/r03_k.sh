#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
timeout 600 python scratch/shard_sizes.py > $O/shard_sizes.log 2>&1; echo "shard rc=$?" >> $O/rc.txt
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_cfg5.log 2>&1; echo "cfg5 rc=$?" >> $O/rc.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -2 $O/shard_sizes.log; tail -3 $O/pytest_all.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03k/bench_cfg5.log') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
print('cfg5', d['ms_per_step'], d['value'], 'frac', r['frac'], [(p['rows'], p['avg_ms']) for p in r['passes']], r['outside_agg_kernels_ms_per_step'])
PY
