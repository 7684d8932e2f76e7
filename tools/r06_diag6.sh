#!/bin/bash
# Round 6, sixth diagnostic call: phase timers of the loop kernels over windows of the chain on the current engine (ab_libs/lib_tm{A,B,D}.so =
# -DDA_PHASE_TIMERS -DDA_TIMER_STEP_LO/HI), batch 64 and one chain; instruction mix of both kernels (PMC).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_diag6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in A:2000 B:4000 D:7000; do for b in 64 1; do
  echo "== window ${w%%:*} batch $b" >> $O/phase_windows.txt
  TIMER_WINDOW_STEPS=${w##*:} DA4ML_HIP_LIB=ab_libs/lib_tm${w%%:*}.so timeout 120 python tests/gpu_profile.py 256 $b 2>&1 | grep "window:\|us/iter\|update cycles" >> $O/phase_windows.txt
done; done
cat $O/phase_windows.txt
bash tools/pmc_quick.sh 64 2>&1 | tail -30
