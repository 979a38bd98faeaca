#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "three_layer or hub_source or device_sampled or graphed_training" 2>&1 | tail -12
