#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r04l/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04l/pytest.log
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04l/prof_cfg2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c2 -- python bench.py --config cfg2 --steps 50 --warmup 3 --graphed off --no-cpu-baseline --no-secondary > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04l/cfg2_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04l/cfg2_kernel_stats.csv")))
for r in rows:
    if 'agg' in r['Name'] or 'scale_rows' in r['Name']:
        print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
WGNN_BENCH_CONFIG=cfg2 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'], d['value'])"
N=8 timeout 300 python scratch/shard_trace.py 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', d['ms_per_step'], [(p['rows'], p['avg_ms']) for p in d['roofline']['passes']])"
