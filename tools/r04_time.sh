#!/bin/bash
# Round 4 timing on the GPU box: the C3 batch and one chain under the settings in ENVS (semicolon-separated, "-" = defaults),
# then the phase-timer build (ab_libs/lib_timers.so, tools/ab_ref.sh timers=WORKTREE:"PHASE_TIMERS=1") if it is there.
# usage: gpurun -- 'ENVS="-;DA4ML_HIP_FUSE=0" bash tools/r04_time.sh'
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_time; mkdir -p $O; rm -f $O/*
IFS=';' read -ra ES <<< "${ENVS:--}"
for e in "${ES[@]}"; do
  [ "$e" = "-" ] && e="DA4ML_X=0"
  echo "== [$e] batch 64"; env $e timeout 200 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1,2p;6,7p'
  echo "== [$e] one chain"; env $e timeout 100 python tests/gpu_profile.py 256 1 2>&1 | sed -n '1p;6,7p'
done 2>&1 | tee $O/timings.txt
if [ -f ab_libs/lib_timers.so ]; then
  for e in ${TIMER_ENVS:-DA4ML_X=0}; do
    echo "== timers [$e] one chain"; env $e DA4ML_HIP_LIB=ab_libs/lib_timers.so timeout 100 python tests/gpu_profile.py 256 1 2>&1 | sed -n '1p;3,7p'
    echo "== timers [$e] batch 64"; env $e DA4ML_HIP_LIB=ab_libs/lib_timers.so timeout 200 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1p;3,7p'
  done 2>&1 | tee $O/timers.txt
fi
