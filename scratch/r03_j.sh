#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "loader or tiled or flat or tile or cfg3 or large_seed or all_cells" > $O/pytest_sel.log 2>&1; echo "pytest_sel rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/pytest_sel.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03j/bench.log') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'avg', r['avg_launch_ms'], 'agg', r['agg_kernels_ms_per_step'], 'outside', r['outside_agg_kernels_ms_per_step'], d['sustained'])
for p in r['passes']: print('  ', p['kernel'], p['rows'], p['src_rows'], p['launches_per_step'], p['avg_ms'])
PY
