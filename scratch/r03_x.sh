#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"
grep '^{' $O/bench.log | cut -c1-250
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03x/bench.log') if x.startswith('{')][0]; d=json.loads(l)
print(d['ms_per_step'], d['roofline'].get('agg_kernels_ms_per_step'), d['roofline'].get('outside_agg_kernels_ms_per_step'), d.get('sustained'))
PY
