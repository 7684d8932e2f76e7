#!/bin/bash
# Round 4 quick check on the GPU box: step-engine parity (k_steps under all settings), then timings of the C3 batch and one chain
# under the settings given in ENVS (semicolon-separated environment settings, "-" = defaults).
# usage: gpurun -- 'ENVS="-;DA4ML_HIP_FUSE=0" bash tools/r04_check.sh [pytest -k expression]'
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_check; mkdir -p $O; rm -f $O/*
K=${1:-step_engine_settings or deterministic or c3_256 or random_small or c2_64}
if [ "${PARITY:-1}" != "0" ]; then timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -6 | tee $O/parity.log; fi
IFS=';' read -ra ES <<< "${ENVS:--}"
for e in "${ES[@]}"; do
  [ "$e" = "-" ] && e="DA4ML_X=0"
  for rep in 1 2; do
    echo "== [$e] batch 64 (run $rep)"; env $e timeout 200 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1p;6,7p'
  done
  echo "== [$e] one chain"; env $e timeout 100 python tests/gpu_profile.py 256 1 2>&1 | sed -n '1p;6,7p'
done 2>&1 | tee $O/timings.txt
