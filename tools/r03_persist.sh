#!/bin/bash
# Staged first runs of the persistent greedy kernel on the GPU box: smallest first, every stage under its own timeout, stop at
# the first failure.  usage: gpurun -- 'bash tools/r03_persist.sh [pytest-args]'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/persist; rm -f gpurun_out/persist/*
O=gpurun_out/persist
stage() { # name, timeout, command...
  local n=$1 t=$2; shift 2
  echo "== $n"; timeout $t "$@" > $O/$n.log 2>&1; local rc=$?
  tail -${TAIL:-6} $O/$n.log; echo "== $n rc=$rc"
  [ $rc -eq 0 ] || { echo "STOP at $n"; exit 1; }
}
export DA4ML_HIP_VERBOSE=1 DA4ML_HIP_ENGINE=persistent   # the engine under test (the launch engine is the default)
stage smoke 120 python __graft_entry__.py smoke
stage c2 120 python tests/gpu_profile.py 64 8
stage single 120 python tests/gpu_profile.py 256 1
stage batch64 180 python tests/gpu_profile.py 256 64
unset DA4ML_HIP_VERBOSE
TAIL=4 stage parity 1200 python -m pytest tests -m gpu -x -q "$@"
for mc in 16 32 64 128 256; do DA4ML_HIP_MAX_CHUNKS=$mc timeout 120 python tests/gpu_profile.py 256 64 2>&1 | sed -n 1p | sed "s/^/[max_chunks $mc] /"; done | tee $O/chunks.txt
DA4ML_HIP_ENGINE=launch timeout 120 python tests/gpu_profile.py 256 64 2>&1 | sed -n 1p | sed "s/^/[launch engine] /" | tee -a $O/chunks.txt
DA4ML_HIP_ENGINE=launch timeout 120 python tests/gpu_profile.py 256 1 2>&1 | sed -n 1p | sed "s/^/[launch engine] /" | tee -a $O/chunks.txt
