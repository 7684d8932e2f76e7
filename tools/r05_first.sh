#!/bin/bash
# Round 5, first GPU call: k_iter_select2 (pick known a step ahead) against the round-4 engine (ab_libs/lib_base.so, built from 4b226e3),
# same box, same call: determinism stress, the most sensitive parity tests, the 64-chain and the one-chain C3 timing; phase timers of the new pair.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_first; mkdir -p $O
for n in spec base; do
  export DA4ML_HIP_LIB=ab_libs/lib_$n.so
  timeout 120 python tools/gpu_stress_small.py 150 > $O/$n.stress.log 2>&1
  timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -k "random_small or c3_256 or c2_64 or capacity" > $O/$n.parity.log 2>&1
  timeout 90 python tests/gpu_profile.py 256 64 > $O/$n.perf64.log 2>&1
  timeout 60 python tests/gpu_profile.py 256 1 > $O/$n.perf1.log 2>&1
  timeout 60 python tests/gpu_profile.py 64 64 > $O/$n.perf64x64.log 2>&1
  echo "[$n] $(tail -1 $O/$n.stress.log) | $(tail -1 $O/$n.parity.log)"; cat $O/$n.perf64.log; head -1 $O/$n.perf1.log; grep -h "picks known\|sampled" $O/$n.perf1.log; head -1 $O/$n.perf64x64.log
done
export DA4ML_HIP_LIB=ab_libs/lib_spec_timers.so
timeout 90 python tests/gpu_profile.py 256 64 > $O/timers.perf64.log 2>&1
timeout 60 python tests/gpu_profile.py 256 1 > $O/timers.perf1.log 2>&1
echo "[timers]"; cat $O/timers.perf64.log $O/timers.perf1.log
