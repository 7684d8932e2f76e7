#!/bin/bash
# FIRST gpurun call of round 3.  The last GPU-measured kernels are those of revision 70c8f1a (profiles/r02_*); everything after
# it -- the round-trip cuts of k_iter_update / k_iter_select made by ISA reading, the kernel-argument preload -- was verified on
# the emulated device only (tests/emu: oracle parity, ASan, race detector), never timed.  This script settles them:
#   HERE first (no GPU):   tools/ab_ref.sh measured=70c8f1a kernels=a9a2b25 nopreload=WORKTREE:"KERNARG=" timers=WORKTREE:"PHASE_TIMERS=1"
#                          (measured -> kernels: the round-trip cuts + the preload; kernels -> HEAD: the host-path changes --
#                          thread pool, parallel centring, k_init_state, k_gather, in-place adder trees)
#                          (timers: per-phase shader-clock cycles of both kernels, printed at the end; it is slower by design)
#   then:                  gpurun --timeout 1500 -- 'bash tools/r03_first.sh'
# 1. the whole GPU parity suite on the working tree; 2. alternating timings (64-chain C3 batch and one chain alone) of the
# working tree and of every library in ab_libs/; 3. QUICK profile collection of the working tree -> gpurun_out/profiles_r03/
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/ab/parity_head.log
for rep in 1 2; do
  for lib in HEAD ab_libs/lib_*.so; do
    if [ "$lib" = HEAD ]; then unset DA4ML_HIP_LIB; n=head; else export DA4ML_HIP_LIB=$lib; n=$(basename $lib .so); n=${n#lib_}; fi
    a=$(timeout 90 python tests/gpu_profile.py 256 64 2>&1 | tee gpurun_out/ab/$n.perf$rep.log | sed -n '1p;6p' | tr '\n' ' ')
    b=$(timeout 60 python tests/gpu_profile.py 256 1 2>&1 | tee gpurun_out/ab/$n.single$rep.log | sed -n '1p')
    echo "[$n #$rep] $a | single: $b"
  done
done | tee gpurun_out/ab/summary.txt
for n in timers; do [ -f gpurun_out/ab/$n.perf1.log ] && { echo "--- phase cycles ($n): batch 64, then one chain"; sed -n 3,4p gpurun_out/ab/$n.perf1.log; sed -n 3,4p gpurun_out/ab/$n.single1.log; }; done
unset DA4ML_HIP_LIB
# host phases of one 64-chain call of HEAD (upload, init, loop, download, set-up, adder trees): where the non-loop time goes
DA4ML_HIP_VERBOSE=1 timeout 90 python tests/gpu_profile.py 256 64 2>&1 | grep "da4ml_hip" | tail -24 | tee gpurun_out/ab/host_phases.txt
QUICK=1 bash tools/collect_profiles.sh r03 2>&1 | tail -3
tail -c 600 gpurun_out/profiles_r03/bench.json
