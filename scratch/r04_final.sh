#!/bin/bash
# round-4 records of the final tree -> gpurun_out/r04final (copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O; rm -f $O/rc.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/rc.txt
WGNN_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --graphed on > $O/bench_n2_shared.log 2>&1; echo "bench2 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest_all.log; tail -2 $O/smoke.log; grep '^{' $O/bench.log | cut -c1-200; grep '^{' $O/bench_n2_shared.log | cut -c1-200
bash scratch/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log | cut -c1-300
timeout 1500 python scratch/configs_record_r04.py > $O/configs.log 2>&1; tail -2 $O/configs.log | cut -c1-600
timeout 900 python scratch/shard_sizes_r04.py > $O/shard.log 2>&1; tail -1 $O/shard.log
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04final/prof_train
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python examples/train_sharded.py --config cfg3 --steps 10 > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/log.txt | cut -c1-300
