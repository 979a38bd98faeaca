#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
for seed in 11 12 13 14 15 16; do timeout 600 python scratch/fuzz_tiled.py $seed 80 2>&1 | tail -1; done
for i in 1 2; do timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r04j/pytest_$i.log 2>&1; echo "pytest run $i rc=$?"; tail -1 gpurun_out/r04j/pytest_$i.log; done
timeout 600 python examples/train_sharded.py --config cfg3 --steps 10 2>&1 | tail -1
