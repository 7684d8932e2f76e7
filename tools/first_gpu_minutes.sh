#!/bin/bash
# What the round-1 sessions could not run any more (GPU budget spent), in the order it should be run next round:
#   gpurun --timeout 900 -- 'bash tools/first_gpu_minutes.sh'
# 1. the whole GPU suite (includes the cases added after the budget ended: default-search digests at 64/128/256, seeds
#    1-3 of the 256x256 chain, the DAIS device executor);  2. DAIS device executor throughput;  3. the kernel variants
# built HERE beforehand with tools/ab_build.sh (skipped when ab_libs/ is empty).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/first
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/first/pytest_gpu.log 2>&1; echo "pytest -m gpu: $(tail -1 gpurun_out/first/pytest_gpu.log)"
for n in 64 128; do
  timeout 120 python tools/dais_bench.py $n 1048576 > gpurun_out/first/dais_$n.log 2>&1; tail -3 gpurun_out/first/dais_$n.log
done
for tile in 65536 262144; do
  DA4ML_DAIS_TILE=$tile timeout 120 python tools/dais_bench.py 64 1048576 2>&1 | grep '^device:' | sed "s/^/tile $tile /"
done
ls ab_libs/lib_*.so > /dev/null 2>&1 && bash tools/ab_run.sh 200
