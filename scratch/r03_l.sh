#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "loader or tiled or flat or tile or cfg3 or cfg2 or narrow or golden or refcode or kat or neigh or gradient or training or sharded" > $O/pytest_sel.log 2>&1; echo "pytest_sel rc=$?" >> $O/rc.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$i.log 2>&1; done
cat $O/rc.txt; tail -3 $O/pytest_sel.log
python - <<'PY'
import json
for i in (1,2):
    l=[x for x in open(f'gpurun_out/r03l/bench{i}.log') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
    print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'avg', r['avg_launch_ms'], 'outside', r['outside_agg_kernels_ms_per_step'], d['sustained']['ms_per_step'], [(p['rows'], p['avg_ms']) for p in r['passes']])
PY
