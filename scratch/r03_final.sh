#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest_all.log; tail -2 $O/smoke.log; grep '^{' $O/bench.log | cut -c1-400
