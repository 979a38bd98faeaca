#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04h/prof_cfg2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c2 -- python bench.py --config cfg2 --steps 50 --warmup 3 --graphed off --no-cpu-baseline --no-secondary > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04h/cfg2_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04h/cfg2_kernel_stats.csv")))
for r in rows:
    c=int(r['Calls'])
    if c>=50 and c % 53 == 0 or 'agg' in r['Name'] or 'scale_rows' in r['Name']:
        print(f"{r['Name'][:110]:110s} {c:>5d} {float(r['AverageNs'])/1e3:9.1f} us  per-fwd {float(r['TotalDurationNs'])/53/1e3:8.1f}")
PY
grep '^{' $OUT/log.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], [(p['kernel'],p['rows'],p['avg_ms']) for p in d['roofline']['passes']])"
timeout 600 python scratch/geom_cfg2.py 2>&1 | tail -12
