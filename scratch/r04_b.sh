#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests -m gpu -x -q -k "alpha_folded or folds_alpha or genes_finish or world1_nccl or world2_hip or bench_two or hipgraph or test_abi" > gpurun_out/r04b/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r04b/pytest.log
for N in 8 4; do N=$N timeout 300 python scratch/shard_trace.py 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b/prof_shard8
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
N=8 STEPS=50 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python scratch/shard_trace.py > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04b/shard8_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04b/shard8_kernel_stats.csv")))
# 3 timed loops of (3 + 50) eager/eager/graphed forwards + 2 + 1 + 1 more: kernels of the forward have >= 150 calls
for r in rows:
    c=int(r['Calls'])
    if c>=150:
        print(f"{r['Name'][:120]:120s} {c:>5d} {float(r['AverageNs'])/1e3:9.1f} us  per-fwd {float(r['TotalDurationNs'])/c*round(c/163)/1e3:8.1f}")
PY
