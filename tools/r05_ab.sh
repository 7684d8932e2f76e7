#!/bin/bash
# Round 5 A/B: every library in ab_libs/ -- determinism stress, the most sensitive parity tests, the 64-chain and the one-chain C3 timing, the 64x64 batch.
# usage: gpurun -- 'bash tools/r05_ab.sh [name ...]'   (default: all of ab_libs/lib_*.so; names containing "timers" skip stress and parity)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_ab; mkdir -p $O
names="$@"; [ -z "$names" ] && names=$(ls ab_libs/lib_*.so | sed 's|ab_libs/lib_||; s|\.so||')
for n in $names; do
  export DA4ML_HIP_LIB=ab_libs/lib_$n.so
  if [[ $n != *timers* && -z "${SKIP_CHECKS:-}" ]]; then
    timeout 120 python tools/gpu_stress_small.py 100 > $O/$n.stress.log 2>&1
    timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -k "random_small or c3_256 or c2_64 or capacity" > $O/$n.parity.log 2>&1
    echo "[$n] $(tail -1 $O/$n.stress.log) | $(tail -1 $O/$n.parity.log)"
  else
    echo "[$n]"
  fi
  timeout 90 python tests/gpu_profile.py 256 64 > $O/$n.perf64.log 2>&1
  timeout 60 python tests/gpu_profile.py 256 1 > $O/$n.perf1.log 2>&1
  timeout 60 python tests/gpu_profile.py 64 64 > $O/$n.perf64x64.log 2>&1
  grep -h "batch\|picks known\|sampled\|search block\|select cycles" $O/$n.perf64.log $O/$n.perf1.log; head -1 $O/$n.perf64x64.log
done
