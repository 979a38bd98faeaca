#!/bin/bash
# round 3, first GPU visit: new tests, the N=1 bench line, the 2-rank strong-scaling line (shared GPU), I-cache probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "nccl or large_seed or all_cells_as or bench or subplan" > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1; echo "bench1 rc=$?" >> $O/rc.txt
WGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_n2_shared.log 2>&1; echo "bench2 rc=$?" >> $O/rc.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>/dev/null | grep -i -o -E "\b(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_CACHE)[A-Z_0-9]*)\b" | sort -u) > $O/icache_counters.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/rc.txt
tail -5 $O/pytest_new.log; tail -3 $O/pytest_all.log; cat $O/rc.txt; cat $O/icache_counters.txt | head
