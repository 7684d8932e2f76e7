#!/bin/bash
# Run on the GPU box (via gpurun): for every library in ab_libs/ a determinism stress against the oracle, the most
# sensitive parity tests, and the 64-chain C3 batch timing.  ~45 s per variant.
# usage: gpurun --timeout 600 -- 'bash tools/ab_run.sh [reps]'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
REPS=${1:-300}
for lib in ab_libs/lib_*.so; do
  n=$(basename $lib .so); n=${n#lib_}
  export DA4ML_HIP_LIB=$lib
  timeout 120 python tools/gpu_stress_small.py $REPS > gpurun_out/ab/$n.stress.log 2>&1
  timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -k "random_small or c3_256 or c2_64" > gpurun_out/ab/$n.parity.log 2>&1
  timeout 60 python tests/gpu_profile.py 256 64 > gpurun_out/ab/$n.perf.log 2>&1
  timeout 30 python tests/gpu_profile.py 256 1 > gpurun_out/ab/$n.perf1.log 2>&1
  echo "[$n] $(tail -1 gpurun_out/ab/$n.stress.log) | $(tail -1 gpurun_out/ab/$n.parity.log) | $(head -1 gpurun_out/ab/$n.perf.log) | $(tail -1 gpurun_out/ab/$n.perf.log) | single: $(head -1 gpurun_out/ab/$n.perf1.log)"
done
