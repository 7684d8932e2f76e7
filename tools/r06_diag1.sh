#!/bin/bash
# Round 6, first GPU call: sanity (GPU suite without the two RCCL scripts), where a chain's step goes (-DDA_STEP_CLOCKS builds), the group-count A/B,
# dependent-load latency at cache-sized footprints, and the RCCL set-up probe.  Everything bounded.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_diag1; mkdir -p $O
nproc > $O/host.txt; /opt/rocm/bin/rocm-smi --showproductname 2>/dev/null | head -12 >> $O/host.txt
timeout 500 python -m pytest tests -m gpu -x -q -k "not rccl_transport_one_rank and not torch_nccl_paths_one_rank" > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt
for spec in clk:64 clk:32 clk:8 clk:1 clk2048:64 clk2048:1; do
  l=${spec%%:*}; b=${spec#*:}
  echo "== $l batch $b" >> $O/step_clocks.txt
  STEP_CLOCKS=1 DA4ML_HIP_LIB=ab_libs/lib_$l.so timeout 120 python tests/gpu_profile.py 256 $b 2>&1 | grep -v "^{" >> $O/step_clocks.txt
done
cat $O/step_clocks.txt | grep "==\|step clocks\|us/iter"
N=3 B=64 bash tools/r05_repeat.sh base g2048r8 g1024r8 g2048 g1024 > $O/ab_groups_b64.txt 2>&1; cat $O/ab_groups_b64.txt
N=2 B=1 bash tools/r05_repeat.sh base g2048r8 g1024r8 g2048 g1024 > $O/ab_groups_b1.txt 2>&1; cat $O/ab_groups_b1.txt
timeout 300 tools/micro/latency_probe small > $O/latency_small.txt 2>&1; cat $O/latency_small.txt
timeout 900 python tools/rccl_init_probe.py 3 90 > $O/rccl_probe.txt 2>&1; tail -3 $O/rccl_probe.txt
