#!/bin/bash
# usage (on the GPU box, via gpurun):  bash scratch/profile_round.sh r01
# 1) rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/
# 2) separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) on the two
#    aggregation launches -> HBM bytes per launch
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fwd -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench_line.json
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$T -o p -- python scratch/one_kernel.py > $OUT/pmc_$T.log 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, sys, json, glob, collections, os
out, tag = sys.argv[1], sys.argv[2]
# ---- compact kernel stats
rows = list(csv.DictReader(open(os.path.join(out, "fwd_kernel_stats.csv"))))
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n[:70]
with open(os.path.join(out, f"kernel_stats_{tag}.csv"), "w") as f:
    f.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows[:25]:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.1f},{r['Percentage']}\n")
# ---- PMC: per-launch means for the aggregation kernels
res = collections.defaultdict(dict)
for d in glob.glob(os.path.join(out, "pmc_*")):
    if not os.path.isdir(d): continue
    fs = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "agg_tiled" in k or "agg_finalize" in k or "scale_rows" in k:
            key = ("agg_tiled_flat" if "agg_tiled_flat" in k else "agg_tiled" if "agg_tiled" in k else "agg_finalize" if "finalize" in k else "scale_rows")
            acc[(key, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, grid), cs in acc.items():
        for c, v in cs.items():
            res[f"{k}[grid={grid}]"][c] = sum(v) / len(v)
json.dump(res, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat $OUT/kernel_stats_$TAG.csv | head -14
cat $OUT/bench_line.json | cut -c1-300
