#!/bin/bash
# round-2 experiment driver: parity suite, the kernel variants in ab_libs/ (if any), then timings of the default library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
ls ab_libs/lib_*.so > /dev/null 2>&1 && bash tools/ab_run.sh 50
unset DA4ML_HIP_LIB
for ub in 1280 1536 2048 2560; do echo "== UPD_BLOCKS=$ub $(DA4ML_HIP_UPD_BLOCKS=$ub timeout 60 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1p;4p;5p' | tr '\n' ' ')"; done
timeout 60 python tests/gpu_profile.py 256 1 2>&1 | sed -n '1p;5p'
