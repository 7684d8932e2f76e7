#!/bin/bash
# Round 5, last call: GPU parity suite on the final sources (the two one-rank RCCL scripts last and bounded: see tests/test_zy_shard_gpu.py::_run_bounded),
# the contract line and the C2 / C5 lines with the PMC summaries of the same sources in profiles/.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_final; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q -k "not rccl_transport_one_rank and not torch_nccl_paths_one_rank" > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/gpu_suite.txt
timeout 600 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
timeout 120 python bench.py --workload c2_64x64_int8_batch64_single_chain --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 200 python bench.py --workload c5_model_batch --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
DA4ML_TEST_RCCL_SECONDS=55 timeout 260 python -m pytest tests/test_zy_shard_gpu.py -q -m gpu -rs -k "rccl_transport_one_rank or torch_nccl_paths_one_rank" > $O/rccl_one_rank.txt 2>&1; tail -4 $O/rccl_one_rank.txt
