#!/bin/bash
# Round 6, fifth GPU call: kernel-boundary shape probe; the column-sharded chain with its step status published into pinned host memory; C4 bench lines (one rank)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_fifth; mkdir -p $O
timeout 200 tools/micro/gap_probe > $O/gap_probe.txt 2>&1; cat $O/gap_probe.txt
timeout 300 python tools/shard_bench.py 256 > $O/column_sharded.txt 2>&1; cat $O/column_sharded.txt | cut -c1-300
timeout 300 python bench.py --workload c4_256x256_int8_column_sharded --steps 2 --warmup 1 > $O/bench_c4_column.json 2> $O/bench_c4_column.err; cat $O/bench_c4_column.json | cut -c1-1500; tail -2 $O/bench_c4_column.err
timeout 300 python bench.py --workload c4_256x256_int8_candidate_sharded --steps 2 --warmup 1 > $O/bench_c4_cand.json 2> $O/bench_c4_cand.err; cat $O/bench_c4_cand.json | cut -c1-1200; tail -2 $O/bench_c4_cand.err
N=3 B=64 bash tools/r05_repeat.sh base cur > $O/ab_b64.txt 2>&1; cat $O/ab_b64.txt
timeout 300 python -m pytest tests/test_zy_shard_gpu.py -q -m gpu -rsx 2>&1 | tail -3
