#!/bin/bash
# round 4: long parity fuzz of the tile kernels (forward and backward routes) on fresh seeds
cd /root/repo; mkdir -p gpurun_out/r04s
for s in 101 102 103 104 105 106; do timeout 600 python scratch/fuzz_tiled.py $s 150 2>&1 | tail -1 | tee -a gpurun_out/r04s/fuzz_fwd.txt; done
for s in 201 202 203 204; do timeout 600 python scratch/fuzz_bwd.py $s 60 2>&1 | tail -1 | tee -a gpurun_out/r04s/fuzz_bwd.txt; done
