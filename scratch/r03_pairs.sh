#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pairs; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "bench rc=$?"
grep '^{' $O/bench.log | cut -c1-400
