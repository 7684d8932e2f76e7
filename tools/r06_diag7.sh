#!/bin/bash
# Round 6, seventh diagnostic call: the tail of k_iter_update -- pass durations of its wavefronts by kind over windows of the chain (-DDA_UPD_TAIL builds)
cd "$GRAFT_REPO_ROOT"
for w in A B D; do for b in 64 1; do echo "== window $w batch $b: $(DA4ML_HIP_LIB=ab_libs/lib_tail$w.so timeout 120 python tools/gpu_tail.py $b 2>&1 | tail -1)"; done; done
