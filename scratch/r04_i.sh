#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in "0 0" "1 0" "0 1" "1 1" "1 1"; do set -- $t; TUNED=$1 LT=$2 N=8 timeout 300 python scratch/shard_trace.py 2>&1 | tail -1; done
