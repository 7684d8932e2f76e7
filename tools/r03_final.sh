#!/bin/bash
# Last GPU call of round 3 on the final tree: the driver-contract bench line (fresh PMC profiles -> traffic / valu on the line, digests
# of every seed that has a record), the GPU parity suite, smoke, and three cheap measurements for DESIGN.md:
#   * the 64-chain C3 call on a 32-core slice of the host (what one rank of eight gets) next to the whole host,
#   * batches of 128 and 256 chains (how the throughput follows the number of restarts),
#   * more chain groups with more hardware queues (environment knobs only).
# usage: gpurun --timeout 1500 -- 'bash tools/r03_final.sh'
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O; rm -f $O/*
python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/parity.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
prof() { # name, env/prefix...
  local n=$1; shift
  env "$@" > $O/$n.txt 2>&1; echo "[$n] $(grep -m1 'solves/s' $O/$n.txt)"
}
prof host_all DA4ML_HIP_VERBOSE=1 timeout 120 python tests/gpu_profile.py 256 64
prof host_slice32 DA4ML_HIP_VERBOSE=1 timeout 120 taskset -c 0-31 python tests/gpu_profile.py 256 64
prof host_slice16 DA4ML_HIP_VERBOSE=1 timeout 120 taskset -c 0-15 python tests/gpu_profile.py 256 64
prof batch128 X=0 timeout 200 python tests/gpu_profile.py 256 128
prof batch256 X=0 timeout 300 python tests/gpu_profile.py 256 256
prof lanes8_q8 GPU_MAX_HW_QUEUES=8 DA4ML_HIP_LANES=8 timeout 120 python tests/gpu_profile.py 256 64
prof lanes6_q8 GPU_MAX_HW_QUEUES=8 DA4ML_HIP_LANES=6 timeout 120 python tests/gpu_profile.py 256 64
prof lanes4_q8 GPU_MAX_HW_QUEUES=8 timeout 120 python tests/gpu_profile.py 256 64
