#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O; rm -f $O/rc.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/rc.txt
WGNN_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_n2_shared.log 2>&1; echo "bench2 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest_all.log; tail -2 $O/smoke.log; grep '^{' $O/bench.log | cut -c1-200; grep '^{' $O/bench_n2_shared.log | cut -c1-200
