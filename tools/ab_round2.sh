#!/bin/bash
# round-2 experiment driver: parity suite, the kernel variants in ab_libs/ (if any), then a timing of the default library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
ls ab_libs/lib_*.so > /dev/null 2>&1 && bash tools/ab_run.sh 50
unset DA4ML_HIP_LIB
for rep in 1 2 3; do DA4ML_HIP_VERBOSE=1 timeout 60 python tests/gpu_profile.py 256 64 2>&1 | grep -E "256x256 batch|extract|upload|adder trees" | tail -4 | tr '\n' ' '; echo; done
