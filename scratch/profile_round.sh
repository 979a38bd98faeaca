#!/bin/bash
# usage (on the GPU box, via gpurun):  bash scratch/profile_round.sh r02
# 1) rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/
# 2) separate --pmc passes (never combined with other trace domains) over the two aggregation launches of
#    scratch/one_kernel.py: FETCH_SIZE / WRITE_SIZE (HBM bytes), TCC hit/miss, SQ instruction mix + LDS activity
# 3) writes kernel_stats_<tag>.csv, pmc_<tag>.json and hbm_traffic.json (the file bench.py reads for roofline.traffic)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fwd -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench_line.json
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$T -o p -- python scratch/one_kernel.py > $OUT/pmc_$T.log 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, sys, json, glob, collections, os, re
out, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(os.path.join(out, "**", "fwd_kernel_stats.csv"), recursive=True)[0])))
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "")[:70]
with open(os.path.join(out, f"kernel_stats_{tag}.csv"), "w") as f:
    f.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows[:25]:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.1f},{r['Percentage']}\n")
res = collections.defaultdict(dict)
for d in glob.glob(os.path.join(out, "pmc_*")):
    if not os.path.isdir(d): continue
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        m = re.search(r"(agg_tiled_flat4|agg_tiled|agg_finalize|agg_main|scale_rows)", k)
        if m:
            acc[(m.group(1), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, grid), cs in acc.items():
        for c, v in cs.items():
            res[f"{k}[grid={grid}]"][c] = sum(v) / len(v)
json.dump(res, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=1)
# HBM bytes per launch: FETCH_SIZE (KiB) doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM) + WRITE_SIZE (KiB)
def hbm(key):
    r = res.get(key, {})
    return int((2 * r.get("FETCH_SIZE", 0) + r.get("WRITE_SIZE", 0)) * 1024) if r else None
tiled = sorted((k for k in res if k.startswith("agg_tiled_flat4")), key=lambda k: int(k.split("=")[1][:-1]))
fin = sorted((k for k in res if k.startswith("agg_finalize")), key=lambda k: int(k.split("=")[1][:-1]))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from bench import kernel_sources_sha
popularity = os.environ.get("WGNN_SYNTH_POPULARITY", "testis199")
traffic = {"_kernel_sources_sha": kernel_sources_sha(), "_popularity": popularity,
           "_commit": os.environ.get("WGNN_COMMIT", "stamped when copied into profiles/ (scratch/stamp_traffic.py)"),
           "_captured": f"round {tag}, scratch/profile_round.sh: separate rocprofv3 --pmc passes over scratch/one_kernel.py (cfg3 operands, D = 256); "
                        "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads), WRITE_SIZE as reported",
           "_kernels": {k: res[k] for k in tiled + fin}}
# the launch with the bigger grid-x is the cells<-genes pass when no column split is used; identify by WRITE_SIZE instead:
for k in tiled:
    w = res[k].get("WRITE_SIZE", 0) * 1024
    name = "cfg3:100000x20000" if w > 90e6 else "cfg3:20000x100000"      # 102 MB output vs partial sums of the gene side
    extra = sum(hbm(f) or 0 for f in fin) if name.endswith("x100000") else 0
    traffic[name] = {"kernel": "agg_tiled_flat4", "hbm_bytes_per_launch": (hbm(k) or 0) + extra, "grid": k,
                     "includes_finalize": bool(extra), "L2_hit": (res[k].get("TCC_HIT_sum", 0) / max(1.0, res[k].get("TCC_HIT_sum", 0) + res[k].get("TCC_MISS_sum", 0)))}
json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:3000])
PY
cat $OUT/kernel_stats_$TAG.csv | head -14
cat $OUT/bench_line.json | cut -c1-300
