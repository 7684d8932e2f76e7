#!/bin/bash
# Round 6, ninth GPU call: the column-sharded chain on the two-block selection (one selection implementation in the tree): GPU suite, step time, C4 line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_ninth; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
timeout 300 python tools/shard_bench.py 256 > $O/column_sharded.txt 2>&1; cat $O/column_sharded.txt | grep "single chain" | cut -c1-260
timeout 300 python bench.py --workload c4_256x256_int8_column_sharded --steps 2 --warmup 1 > $O/bench_c4_column.json 2> $O/bench_c4_column.err; python -c "
import json; l=json.loads(open('$O/bench_c4_column.json').read().strip().splitlines()[-1]); print(l['value'], l['engine'], l['check'])"
N=2 B=64 bash tools/r05_repeat.sh base cur 2>&1 | tail -2
N=2 B=1 bash tools/r05_repeat.sh base cur 2>&1 | tail -2
