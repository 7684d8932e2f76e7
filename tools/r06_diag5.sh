#!/bin/bash
# Round 6, seventh GPU call: what the end of a kernel costs as a function of what it wrote (gap probe, second part); kernel-level gaps of the loop
# from the profiler's own timestamps (one chain and the batch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_diag5; mkdir -p $O
timeout 300 tools/micro/gap_probe > $O/gap_probe.txt 2>&1; tail -6 $O/gap_probe.txt
export DA4ML_HIP_LIB=ab_libs/lib_cur.so
for B in 1 64; do
  rm -rf /tmp/tr_$B; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$B -o t -- python tests/gpu_profile.py 256 $B > $O/trace.$B.log 2>&1
  echo "== batch $B (traced)"; python tools/trace_gaps.py /tmp/tr_$B | tee $O/trace.$B.gaps.txt
done
