#!/bin/bash
# round 5, GPU visit a: stream micro-benchmark, GPU tests of the ADVICE fixes, bench baseline of this lease
mkdir -p gpurun_out/r05a
./scratch/variants/stream_bench > gpurun_out/r05a/stream_bench.txt 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.txt 2>&1; tail -3 gpurun_out/r05a/pytest.txt
python bench.py > gpurun_out/r05a/bench.txt 2>&1; tail -1 gpurun_out/r05a/bench.txt | cut -c1-400
cat gpurun_out/r05a/stream_bench.txt
