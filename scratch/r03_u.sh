#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
REPS=3 timeout 600 python scratch/loader_ab.py 1 2 3 > gpurun_out/r03u/loader_pairs.log 2>&1; echo rc=$?
grep "^rep" gpurun_out/r03u/loader_pairs.log
