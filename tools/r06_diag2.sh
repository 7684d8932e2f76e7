#!/bin/bash
# Round 6, second GPU call: the engine without the in-launch wait (pick found by the substitution block itself), 2048 group records: GPU suite,
# step clocks under several geometries, A/B timing against the round-5 library, the two one-rank RCCL scripts five times from inside a pytest
# process that holds the GPU (the condition under which round 5 saw the set-up hang).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_diag2; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "not rccl_transport_one_rank and not torch_nccl_paths_one_rank" > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt
clk() { echo "== $1 batch $2 $3" >> $O/step_clocks.txt; env $3 STEP_CLOCKS=1 DA4ML_HIP_LIB=ab_libs/lib_$1.so timeout 120 python tests/gpu_profile.py 256 $2 2>&1 | grep "step clocks\|us/iter" >> $O/step_clocks.txt; }
clk clk 64; clk clk 1; clk clk512 64; clk clk512 1
clk clk 64 DA4ML_HIP_UPD_BLOCKS=1280; clk clk 64 DA4ML_HIP_UPD_BLOCKS=5120; clk clk 64 DA4ML_HIP_LANES=2; clk clk 64 DA4ML_HIP_LANES=3
cat $O/step_clocks.txt
N=3 B=64 bash tools/r05_repeat.sh base new t512 > $O/ab_b64.txt 2>&1; cat $O/ab_b64.txt
N=2 B=1 bash tools/r05_repeat.sh base new t512 > $O/ab_b1.txt 2>&1; cat $O/ab_b1.txt
for i in 1 2 3 4 5; do
  DA4ML_TEST_RCCL_SECONDS=100 timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_zy_shard_gpu.py -q -m gpu -rsx -k "test_c1 or test_capacity_retry or rccl_transport_one_rank or torch_nccl_paths_one_rank" 2>&1 | tail -4 >> $O/rccl_in_suite.txt
done
cat $O/rccl_in_suite.txt | grep -c passed; grep -h "xfail\|XFAIL\|failed" $O/rccl_in_suite.txt | head -5
