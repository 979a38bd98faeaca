#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python scratch/train_step_time.py > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - "$OUT" <<'PY'
import csv, sys, os
rows=list(csv.DictReader(open(os.path.join(sys.argv[1],"tr_kernel_stats.csv"))))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e6:9.2f} ms  {float(r['AverageNs'])/1e3:9.1f} us")
PY
