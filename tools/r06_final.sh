#!/bin/bash
# Round 6, final collection on the final sources: GPU parity suite + smoke + soak, kernel stats + PMC passes (tools/collect_profiles.sh, PMC_ONLY),
# then -- once the summaries have been copied into profiles/ (second call: BENCH=1) -- the contract line and the secondary workloads.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_final; mkdir -p $O
if [ -z "${BENCH:-}" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -rsx > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/gpu_suite.txt
  timeout 900 python tools/gpu_soak.py 600 > $O/gpu_soak.txt 2>&1; tail -3 $O/gpu_soak.txt
  PMC_ONLY=1 DAIS=0 bash tools/collect_profiles.sh r06 > $O/collect.log 2>&1; tail -3 $O/collect.log
else
  timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
  timeout 200 python bench.py --workload c2_64x64_int8_batch64_single_chain --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_c2.json 2> $O/bench_c2.err
  timeout 300 python bench.py --workload c5_model_batch --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
  timeout 300 python bench.py --workload c4_256x256_int8_column_sharded --steps 2 --warmup 1 > $O/bench_c4_column.json 2> $O/bench_c4_column.err
  timeout 300 python bench.py --workload c4_256x256_int8_candidate_sharded --steps 2 --warmup 1 > $O/bench_c4_candidates.json 2> $O/bench_c4_candidates.err
  timeout 300 python bench.py --workload c3_256x256_int8_batch8_default_search --steps 2 --warmup 1 --cpu-seconds 0 > $O/bench_default_search.json 2> $O/bench_default_search.err
  bash tools/r05_batch_scan.sh cur > $O/batch_scan.txt 2>&1; tail -9 $O/batch_scan.txt
  timeout 300 python tools/dais_bench.py 64 1048576 > $O/dais_bench.txt 2>&1
  for f in bench_c2 bench_c5 bench_c4_column bench_c4_candidates bench_default_search; do echo "$f: $(tail -1 $O/$f.json | head -c 240)"; done
fi
