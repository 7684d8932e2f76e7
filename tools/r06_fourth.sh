#!/bin/bash
# Round 6, fourth GPU call: the live-block counter tallied once per workgroup (was: one device atomic per created / deleted block on one descriptor line)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_fourth; mkdir -p $O
N=3 B=64 bash tools/r05_repeat.sh base new live > $O/ab_b64.txt 2>&1; cat $O/ab_b64.txt
N=2 B=1 bash tools/r05_repeat.sh base new live > $O/ab_b1.txt 2>&1; cat $O/ab_b1.txt
for w in A B; do for b in 64 1; do
  echo "== window $w batch $b" >> $O/step_clock_windows.txt
  STEP_CLOCKS=1 DA4ML_HIP_LIB=ab_libs/lib_clk$w.so timeout 120 python tests/gpu_profile.py 256 $b 2>&1 | grep "step clocks\|us/iter" >> $O/step_clock_windows.txt
done; done
echo "== all batch 64" >> $O/step_clock_windows.txt; STEP_CLOCKS=1 DA4ML_HIP_LIB=ab_libs/lib_clk.so timeout 120 python tests/gpu_profile.py 256 64 2>&1 | grep "step clocks\|us/iter" >> $O/step_clock_windows.txt
cat $O/step_clock_windows.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "not rccl_transport_one_rank and not torch_nccl_paths_one_rank" > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt
