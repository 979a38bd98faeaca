#!/bin/bash
mkdir -p gpurun_out/r05m; O=gpurun_out/r05m
python -m pytest tests/test_gpu_parity.py -x -q -k "plan_walk" > $O/pytest1.txt 2>&1; tail -15 $O/pytest1.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "tile or tiled or tall or golden or refcode or full_size_properties_cfg3 or kat" > $O/pytest2.txt 2>&1; tail -3 $O/pytest2.txt
python scratch/plan_prof.py 2>&1 | grep "rep \|Self C"
timeout 600 python scratch/fuzz_tiled.py 31 80 2>&1 | tail -1
