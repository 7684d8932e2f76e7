#!/bin/bash
# Round 6 A/B: the four chain groups on CU-masked streams (ab_libs/lib_mask.so, -DDA_AB_CU_MASK; the poll stream's work on the first group's stream unless
# DA4ML_HIP_MASK_OWN_POLL is set); us per lockstep step of the C3 batch of 64 (B=1: one chain), two runs per mode
cd "$GRAFT_REPO_ROOT"
run() { echo "$1 $2: $(for i in 1 2; do env $2 DA4ML_HIP_LIB=ab_libs/lib_$1.so timeout 120 python tests/gpu_profile.py 256 ${B:-64} | head -1 | sed 's/.*us\/iter //'; done | tr '\n' ' ')"; }
run cur ""
for m in ${MODES:-full quarter half xcd2 not8}; do run mask DA4ML_HIP_CU_MASK=$m; done
run mask "DA4ML_HIP_CU_MASK=full DA4ML_HIP_MASK_OWN_POLL=1"
run cur ""
