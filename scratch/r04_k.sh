#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python scratch/hub_remainder.py 2>&1 | tail -6
timeout 600 python -m pytest tests -m gpu -q -x -k "linear_act or dropout or world2_hip" 2>&1 | tail -2
timeout 600 python examples/train_sharded.py --config cfg3 --steps 10 2>&1 | tail -1
