#!/bin/bash
# round 6: the records of the final tree (one gpurun call): default bench line, rocprofv3 kernel stats + PMC passes, other configs
# (incl. the headline under both popularity laws), per-rank shard compute, training-step kernel table, one-shot plan profile
mkdir -p gpurun_out/r06final; O=gpurun_out/r06final
python bench.py > $O/bench.txt 2>&1; grep '^{' $O/bench.txt | tail -1 > $O/r06_bench.json; cut -c1-250 $O/r06_bench.json
bash scratch/profile_round.sh r06 > $O/profile_round.txt 2>&1; tail -20 $O/profile_round.txt
timeout 1500 python scratch/configs_record.py > $O/configs.txt 2>&1; tail -3 $O/configs.txt
timeout 900 python scratch/shard_sizes.py > $O/shard_sizes.txt 2>&1; tail -8 $O/shard_sizes.txt
timeout 600 python scratch/plan_prof.py > $O/plan_prof.txt 2>&1; grep -E "^rep|tile_plan" $O/plan_prof.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06final/train_prof -o t -- python $GRAFT_REPO_ROOT/scratch/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r06final/train_prof.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -2 $O/train_prof.txt
