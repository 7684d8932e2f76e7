#!/bin/bash
# Round 4: the step engine (k_steps, DA4ML_HIP_FUSE=K steps per launch) against the kernel pair (DA4ML_HIP_FUSE=0) on smaller problems.
# usage: gpurun -- 'bash tools/r04_sizes.sh'
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_sizes; mkdir -p $O; rm -f $O/*
for cfg in "16 64" "32 64" "64 64" "64 1" "128 64" "128 1"; do
  for e in DA4ML_HIP_FUSE=0 DA4ML_HIP_FUSE=8 DA4ML_HIP_FUSE=64 "DA4ML_HIP_FUSE=64 DA4ML_HIP_FUSE_NH=100000 DA4ML_HIP_FUSE_M=16"; do
    echo "== [$cfg] [$e]"; env $e timeout 200 python tests/gpu_profile.py $cfg 2>&1 | sed -n '1p;7p'
  done
done 2>&1 | tee $O/sizes.txt
