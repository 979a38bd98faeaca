#!/bin/bash
# usage: scratch/pmc.sh <tag> "<counters>"   -> gpurun_out/pmc_<tag>/
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $OUT -o p -- env WGNN_ONE_PASS=1 python scratch/one_kernel.py > $OUT/log.txt 2>&1
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name']
    if 'agg_tiled' in k:
        agg[(k.split('::')[-1] if '::' in k else k)[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in d.items(): print(f'   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}')
PY
