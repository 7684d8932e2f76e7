#!/bin/bash
# Round 6, after the group records became one array: update grid and launch threads re-scanned on the new engine (two runs each), us per lockstep step, C3 batch of 64
cd "$GRAFT_REPO_ROOT"
run() { echo "$1 $2: $(for i in 1 2; do env $2 DA4ML_HIP_LIB=ab_libs/lib_$1.so timeout 120 python tests/gpu_profile.py 256 ${B:-64} | head -1 | sed 's/.*us\/iter //'; done | tr '\n' ' ')"; }
for b in 1920 2240 2560 2880 3200 3840; do run aos DA4ML_HIP_UPD_BLOCKS=$b; done
