#!/bin/bash
# round 4, first GPU call: parity of the re-budgeted register file, bench line, hub ablation, training-step kernel profile, shard sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04a/pytest.log
timeout 600 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r04a/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['outside_agg_kernels_ms_per_step'], [ (p['rows'],p['avg_ms']) for p in d['roofline']['passes']])"
timeout 600 python scratch/hub_ablation.py > gpurun_out/r04a/hub.log 2>&1; tail -2 gpurun_out/r04a/hub.log
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a/prof_train
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python scratch/train_prof_r04.py > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04a/train_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04a/train_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:32]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e6:9.2f} ms  {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
timeout 600 python scratch/shard_sizes.py > gpurun_out/r04a/shard.log 2>&1; tail -1 gpurun_out/r04a/shard.log
