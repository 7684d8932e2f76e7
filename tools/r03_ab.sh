#!/bin/bash
# A/B timing on the GPU box: working tree vs every ab_libs/lib_*.so (tools/ab_ref.sh), 64-chain C3 batch and one chain alone.
# usage: gpurun -- 'bash tools/r03_ab.sh [reps] [pytest-args...]'   (PARITY=0 skips the GPU parity suite; ENVS="A=1;B=2" adds
# runs of the working tree under the given environment settings)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*
REPS=${1:-2}
if [ "${PARITY:-1}" != "0" ]; then timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/ab/parity_head.log; fi
run() { # name, env...
  local n=$1; shift
  a=$(env "$@" timeout 120 python tests/gpu_profile.py 256 64 2>&1 | tee gpurun_out/ab/$n.perf.log | sed -n '1p;6p' | tr '\n' ' ')
  b=$(env "$@" timeout 60 python tests/gpu_profile.py 256 1 2>&1 | tee gpurun_out/ab/$n.single.log | sed -n '1p')
  echo "[$n] $a | single: $b"
}
for rep in $(seq 1 $REPS); do
  run head$rep DA4ML_X=0
  for lib in ab_libs/lib_*.so; do [ -f "$lib" ] || continue; n=$(basename $lib .so); run ${n#lib_}$rep DA4ML_HIP_LIB=$lib; done
done 2>&1 | tee gpurun_out/ab/summary.txt
IFS=';' read -ra ES <<< "${ENVS:-}"
for e in "${ES[@]}"; do [ -n "$e" ] && run "$(echo $e | tr ' =' '__')" $e; done 2>&1 | tee -a gpurun_out/ab/summary.txt
DA4ML_HIP_VERBOSE=1 timeout 120 python tests/gpu_profile.py 256 64 2>&1 | grep "leaderboard\|greedy loop" | tail -3 | tee -a gpurun_out/ab/summary.txt
