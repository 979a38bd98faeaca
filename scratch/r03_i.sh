#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
for cfgenv in "base:" "tiled:WGNN_TILED_MIN_WORK=100000000" "tiled_dual:WGNN_TILED_MIN_WORK=100000000 WGNN_LINEAR_DUAL=1 WGNN_LINEAR=always" "dual_only:WGNN_LINEAR=always"; do
  name=${cfgenv%%:*}; envs=${cfgenv#*:}
  env $envs timeout 300 python bench.py --config cfg2 --steps 200 --warmup 10 --no-cpu-baseline --no-secondary > $O/cfg2_$name.log 2>&1
  python - $O/cfg2_$name.log $name <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[2], 'FAILED', open(sys.argv[1]).read()[-500:]); sys.exit()
d=json.loads(l[-1]); r=d['roofline']
print(sys.argv[2], d['ms_per_step'], 'eager', d['config']['eager_ms_per_step'], [(p['kernel'],p['rows'],p['avg_ms']) for p in r['passes']])
PY
done
