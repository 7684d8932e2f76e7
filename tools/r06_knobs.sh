#!/bin/bash
# Round 6: run-time knobs of the loop on the final engine, batch of 64 (us per lockstep step, two runs each): blocks of one k_iter_update launch, launching threads
cd "$GRAFT_REPO_ROOT"
for b in 1920 2240 2560 2880 3200; do echo "UPD_BLOCKS $b: $(for i in 1 2; do DA4ML_HIP_UPD_BLOCKS=$b timeout 90 python tests/gpu_profile.py 256 64 | head -1 | sed 's/.*us\/iter //'; done | paste -sd' ')"; done
for t in 1 2 4; do echo "LAUNCH_THREADS $t: $(for i in 1 2; do DA4ML_HIP_LAUNCH_THREADS=$t timeout 90 python tests/gpu_profile.py 256 64 | head -1 | sed 's/.*us\/iter //'; done | paste -sd' ')"; done
