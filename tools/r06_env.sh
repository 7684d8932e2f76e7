#!/bin/bash
# Round 6: runtime environment settings against the loop (batch of 64 / one chain, us per lockstep step, two or three runs each)
cd "$GRAFT_REPO_ROOT"
run() { for b in 64 1; do echo "$1 batch $b: $(for i in 1 2 3; do env $1 timeout 90 python tests/gpu_profile.py 256 $b | head -1 | sed 's/.*us\/iter //'; done | paste -sd' ')"; done; }
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "HSA_NO_SCRATCH_RECLAIM=1"
run "GPU_MAX_HW_QUEUES=5"
run "HIP_LAUNCH_BLOCKING=0 AMD_DIRECT_DISPATCH=1"
run "AMD_DIRECT_DISPATCH=0"
run "HSA_ENABLE_INTERRUPT=0"
