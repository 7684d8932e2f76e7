#!/bin/bash
# greedy-loop time per lockstep step against the number of 256x256 chains in the call, for the libraries named
cd "$GRAFT_REPO_ROOT"
for n in "$@"; do for b in 1 8 16 32 48 64 96 128; do echo "$n chains $b: $(DA4ML_HIP_LIB=ab_libs/lib_$n.so timeout 120 python tests/gpu_profile.py 256 $b | grep -h 'batch\|sampled' | tr '\n' ' ' | sed 's/256x256 batch [0-9]*: //; s/lockstep iters [0-9]*, //')"; done; done
