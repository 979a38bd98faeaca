#!/bin/bash
# round 5: the records of the final tree (one gpurun call): default bench line, rocprofv3 kernel stats + PMC passes, other configs,
# per-rank shard compute, training-step kernel table
mkdir -p gpurun_out/r05final; O=gpurun_out/r05final
python bench.py > $O/bench.txt 2>&1; grep '^{' $O/bench.txt | tail -1 > $O/r05_bench.json; cut -c1-250 $O/r05_bench.json
bash scratch/profile_round.sh r05 > $O/profile_round.txt 2>&1; tail -20 $O/profile_round.txt
timeout 1500 python scratch/configs_record.py > $O/configs.txt 2>&1; tail -3 $O/configs.txt
timeout 900 python scratch/shard_sizes.py > $O/shard_sizes.txt 2>&1; tail -8 $O/shard_sizes.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05final/train_prof -o t -- python $GRAFT_REPO_ROOT/scratch/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r05final/train_prof.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -2 $O/train_prof.txt
