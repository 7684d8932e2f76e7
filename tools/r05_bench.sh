#!/bin/bash
# Round 5: the contract line (with the CPU leg) and the secondary workloads, after the PMC summaries of the same sources were copied into profiles/.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_bench; mkdir -p $O
timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
timeout 300 python bench.py --workload c2_64x64_int8_batch64_single_chain --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --workload c5_model_batch --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python tools/shard_bench.py 256 > $O/column_sharded.txt 2>&1
timeout 300 python tools/dais_bench.py 64 1048576 > $O/dais_bench.txt 2>&1
bash tools/r05_batch_scan.sh cur > $O/batch_scan.txt 2>&1
ls -la $O; head -c 400 $O/bench.json; tail -3 $O/column_sharded.txt; tail -2 $O/dais_bench.txt
