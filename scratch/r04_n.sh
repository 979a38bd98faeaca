#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "executed_reference_code or joint_projection" 2>&1 | tail -3
