#!/bin/bash
# round-2 experiment driver: parity suite, the kernel variants in ab_libs/ (if any), then timings of the default library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
ls ab_libs/lib_*.so > /dev/null 2>&1 && bash tools/ab_run.sh 50
unset DA4ML_HIP_LIB
for rep in 1 2; do timeout 60 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1p;5p' | tr '\n' ' '; echo; done
for n in timers; do [ -f gpurun_out/ab/$n.perf.log ] && { echo "--- $n"; sed -n 3,5p gpurun_out/ab/$n.perf.log; sed -n 3,5p gpurun_out/ab/$n.perf1.log; }; done
