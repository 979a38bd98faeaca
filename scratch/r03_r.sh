#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1; echo "profile rc=$?"
bash scratch/r03_final.sh
