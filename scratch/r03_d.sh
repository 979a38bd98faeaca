#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python scratch/cpu_blocked_time.py > $O/cpu_blocked.log 2>&1; echo "cpu rc=$?" >> $O/rc.txt
for M in auto never; do
  (cd /tmp && export TMPDIR=/tmp && WGNN_LINEAR=$M timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$M -o fwd -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/bench_$M.log 2>&1)
  F=$(find $O/prof_$M -name "fwd_kernel_stats.csv" | head -1)
  python - "$F" > $O/kstats_$M.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:80]
    print(f"{n:80s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e6:9.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
timeout 1500 python scratch/configs_record_r03.py > $O/configs.log 2>&1; echo "configs rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/cpu_blocked.log | tail -22; grep -v "mbtopk\|rocprim\|radix\|scatter_gather\|Histogram\|log_kernel\|index_elem\|neg_kernel\|clamp\|distribution\|CatArray\|AUnary\|manual_unroll\|CUDAFunctor" $O/kstats_auto.txt | head -14; echo ----; grep -v "mbtopk\|rocprim\|radix\|scatter_gather\|Histogram\|log_kernel\|index_elem\|neg_kernel\|clamp\|distribution\|CatArray\|AUnary\|manual_unroll\|CUDAFunctor" $O/kstats_never.txt | head -14; tail -30 $O/configs.log
