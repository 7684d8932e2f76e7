#!/bin/bash
# kernel-trace of the 64-chain C3 batch for the libraries named: durations of the loop kernels and the gaps between them (tools/trace_gaps.py);
# then the untraced timing of the same libraries (64 chains, one chain)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_trace; mkdir -p $O
for n in "$@"; do
  export DA4ML_HIP_LIB=ab_libs/lib_$n.so
  for B in ${BATCHES:-64}; do
    rm -rf /tmp/tr_$n; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o t -- python tests/gpu_profile.py 256 $B > $O/$n.$B.log 2>&1
    echo "== $n batch $B (traced)"; python tools/trace_gaps.py /tmp/tr_$n | grep -v "^gap" | tee $O/$n.$B.gaps.txt
  done
  timeout 90 python tests/gpu_profile.py 256 64 > $O/$n.perf64.log 2>&1; timeout 60 python tests/gpu_profile.py 256 1 > $O/$n.perf1.log 2>&1
  head -1 $O/$n.perf64.log; grep sampled $O/$n.perf64.log; head -1 $O/$n.perf1.log
done
