#!/bin/bash
# Round 6, sixth GPU call: k_iter_update at 6 / 8 wavefronts per SIMD (spilling) in the slot-saturated early windows of the chain; two processes per GPU
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_diag4; mkdir -p $O
for l in clkA clkA_o6 clkA_o8 clkB clkB_o6 clkB_o8; do
  echo "== $l batch 64" >> $O/occ_windows.txt
  STEP_CLOCKS=1 DA4ML_HIP_LIB=ab_libs/lib_$l.so timeout 120 python tests/gpu_profile.py 256 64 2>&1 | grep "step clocks\|us/iter" >> $O/occ_windows.txt
done
cat $O/occ_windows.txt
N=2 B=64 bash tools/r05_repeat.sh cur o6 o7 o8 > $O/ab_occ_b64.txt 2>&1; cat $O/ab_occ_b64.txt
timeout 400 python tools/two_proc_probe.py 2 32 2 > $O/two_proc.txt 2>&1; tail -2 $O/two_proc.txt | cut -c1-600
timeout 400 python tools/two_proc_probe.py 4 16 2 > $O/four_proc.txt 2>&1; tail -1 $O/four_proc.txt | cut -c1-600
