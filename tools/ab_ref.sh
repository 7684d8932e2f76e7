#!/bin/bash
# Build libda4ml_hip.so of a GIT REVISION into ab_libs/lib_<name>.so (git-ignored, travels with gpurun) for an A/B run
# against the working tree on the GPU box (tools/ab_run.sh / tools/r03_first.sh).  Run HERE (no GPU needed).
# usage: tools/ab_ref.sh name=<git-ref> [name2=<git-ref2> ...] [name3=WORKTREE:"make variables"]
#   e.g. tools/ab_ref.sh measured=70c8f1a nopreload=WORKTREE:"KERNARG="
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p ab_libs
for spec in "$@"; do
  name=${spec%%=*}; ref=${spec#*=}
  if [[ $ref == WORKTREE:* ]]; then
    make -s -C da4ml_amd/csrc variant OUT="$ROOT/ab_libs/lib_$name.so" ${ref#WORKTREE:} 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *| *\^" || true
  else
    tmp=$(mktemp -d /tmp/ab_ref.XXXXXX)
    git archive "$ref" da4ml_amd/csrc include | tar -x -C "$tmp"
    make -s -C "$tmp/da4ml_amd/csrc" variant OUT="$ROOT/ab_libs/lib_$name.so" 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *| *\^" || true
    rm -rf "$tmp"
  fi
  [ -f ab_libs/lib_$name.so ] && echo "built ab_libs/lib_$name.so  [$ref]"
done
