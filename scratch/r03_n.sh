#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$i.log 2>&1; done
timeout 300 python examples/train_sharded.py --config cfg3 --steps 10 > $O/train.log 2>&1
cat $O/rc.txt; tail -3 $O/pytest_all.log; tail -1 $O/train.log
python - <<'PY'
import json
for i in (1,2):
    l=[x for x in open(f'gpurun_out/r03n/bench{i}.log') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
    print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'avg', r['avg_launch_ms'], 'outside', r['outside_agg_kernels_ms_per_step'], d['sustained']['ms_per_step'], [(p['rows'], p['avg_ms']) for p in r['passes']])
PY
