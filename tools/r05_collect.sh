#!/bin/bash
# Round 5, final sources: the full GPU parity suite + smoke, then kernel stats + PMC passes (tools/collect_profiles.sh, PMC_ONLY), then -- once the
# summaries have been copied into profiles/ -- tools/r05_bench.sh writes the lines with the counters of the sources they measure.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_suite.txt 2>&1; tail -3 gpurun_out/r05/gpu_suite.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r05/gpu_suite.txt
PMC_ONLY=1 DAIS=0 bash tools/collect_profiles.sh r05 > gpurun_out/r05/collect.log 2>&1; tail -3 gpurun_out/r05/collect.log
