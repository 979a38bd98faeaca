#!/bin/bash
# kernel breakdown of the full-graph cfg3 training step (BASELINE cfg4 at N = 1): examples/train_sharded.py under rocprofv3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train_r02
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python $GRAFT_REPO_ROOT/examples/train_sharded.py --config cfg3 --steps 10 --batch-size 4096 > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python - "$OUT" <<'PY'
import csv, sys, glob, os
f = glob.glob(os.path.join(sys.argv[1], "**", "tr_kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("kernel,calls,total_ms,avg_us,pct")
for r in rows[:40]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:90]
    print(f"\"{n}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.1f},{r['Percentage']}")
PY
