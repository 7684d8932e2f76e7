#!/bin/bash
# Round 5: the full GPU parity suite, then bench line + kernel stats + PMC passes of the same sources (tools/collect_profiles.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_suite.txt 2>&1; tail -3 gpurun_out/r05/gpu_suite.txt
PMC_ONLY=1 DAIS=0 bash tools/collect_profiles.sh r05 > gpurun_out/r05/collect.log 2>&1; tail -5 gpurun_out/r05/collect.log
