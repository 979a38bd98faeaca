#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
for seed in 11 12 13 14 15 16; do
timeout 400 python scratch/fuzz_tiled.py $seed 80 > gpurun_out/fuzz/f$seed.log 2>&1; echo "seed $seed rc=$? $(tail -1 gpurun_out/fuzz/f$seed.log)"
done
