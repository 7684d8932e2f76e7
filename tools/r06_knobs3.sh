#!/bin/bash
# Round 6, final engine: blocks of a k_iter_update launch re-scanned (five runs each), us per lockstep step of the C3 batch of 64
cd "$GRAFT_REPO_ROOT"
run() { echo "$1 $2: $(for i in 1 2 3 4 5; do env $2 DA4ML_HIP_LIB=ab_libs/lib_$1.so timeout 120 python tests/gpu_profile.py 256 ${B:-64} | head -1 | sed 's/.*us\/iter //'; done | tr '\n' ' ')"; }
for b in 2560 2240 1920 2560 2240 2048; do run cur DA4ML_HIP_UPD_BLOCKS=$b; done
