#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$i.log 2>&1; done
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_cfg5.log 2>&1
timeout 600 python scratch/shard_sizes.py > $O/shard.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "tile or tiled or flat or loader or cfg3 or dist or large_seed" > $O/pytest_sel.log 2>&1
tail -2 $O/pytest_sel.log; tail -1 $O/shard.log
python - <<'PY'
import json
for f in ('bench1','bench2','bench_cfg5'):
    l=[x for x in open(f'gpurun_out/r03p/{f}.log') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
    print(f, d['ms_per_step'], d['value'], 'frac', r['frac'], 'avg', r['avg_launch_ms'], 'outside', r['outside_agg_kernels_ms_per_step'], [(p['rows'], p['avg_ms']) for p in r['passes']])
PY
