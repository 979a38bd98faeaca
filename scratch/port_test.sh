#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== baseline"; timeout 200 python scratch/ablate.py 2>&1 | grep "^full\|^nofill"
for V in SALU VALU; do cp scdeepsort_amd/libwgnn_hip.so /tmp/orig.so; cp scratch/libwgnn_extra_$V.so scdeepsort_amd/libwgnn_hip.so; echo "== +4 $V per pair"; timeout 200 python scratch/ablate.py 2>&1 | grep "^full\|^nofill"; cp /tmp/orig.so scdeepsort_amd/libwgnn_hip.so; done
