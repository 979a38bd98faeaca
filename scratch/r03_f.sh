#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 1500 python scratch/configs_record_r03.py > $O/configs.log 2>&1; echo "configs rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/pytest_all.log; tail -40 $O/configs.log
