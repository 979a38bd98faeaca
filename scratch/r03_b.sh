#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 600 python scratch/shift_ab.py 0.30 0.36 0.41 0.46 > $O/shift_ab.log 2>&1; echo "shift rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "linear_fwd_ex or wgnn_linear or int64 or tiled or flat or large_seed" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
bash scratch/pmc.sh icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE > $O/pmc_icache.log 2>&1
bash scratch/pmc.sh ifetch SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY > $O/pmc_ifetch.log 2>&1
timeout 300 python bench.py --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_cfg2.log 2>&1; echo "cfg2 rc=$?" >> $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg3.log 2>&1; echo "cfg3 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -8 $O/shift_ab.log; tail -4 $O/pytest_sel.log; cat $O/pmc_icache.log | tail -8; cat $O/pmc_ifetch.log | tail -8
