#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\bSQ_[A-Z_0-9]*(MFMA|MOPS)[A-Z_0-9]*\b" | sort -u > $O/mfma_counters.txt
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$T -o p -- python $GRAFT_REPO_ROOT/scratch/lin_one.py > $O/pmc_$T.log 2>&1
done
python - "$O" <<'PY'
import csv, sys, glob, collections, os
res = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_mfma" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(res.items()):
    print(f"{k:34s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
cat $O/mfma_counters.txt
