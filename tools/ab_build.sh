#!/bin/bash
# Build experimental variants of libda4ml_hip.so into ab_libs/ (git-ignored, travels with gpurun).  Run HERE (no GPU
# needed), then run tools/ab_run.sh on the GPU box.  usage: tools/ab_build.sh name1="-DDEF1" name2="-DDEF2 -DDEF3" ...
# With no arguments the variants currently waiting for a measurement are built.
set -e
cd "$(dirname "$0")/.."
mkdir -p ab_libs
if [ $# -eq 0 ]; then
  set -- base="" uncond="-DDA_UNCOND_PREFETCH" nortn="-DDA_NORTN_ATOMICS" selfast="-DDA_SELECT_FAST" \
         occ6="-DDA_UPD_OCC=6" upd6="-DDA_UPD_OCC=6 -DDA_UNCOND_PREFETCH -DDA_NORTN_ATOMICS" \
         upd7="-DDA_UPD_OCC=7 -DDA_UNCOND_PREFETCH -DDA_NORTN_ATOMICS" \
         all8="-DDA_UNCOND_PREFETCH -DDA_NORTN_ATOMICS -DDA_SELECT_FAST" \
         all6="-DDA_UPD_OCC=6 -DDA_UNCOND_PREFETCH -DDA_NORTN_ATOMICS -DDA_SELECT_FAST"
fi
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  make -s -C da4ml_amd/csrc variant OUT="$PWD/ab_libs/lib_$name.so" VARIANT_DEFS="$defs" 2>&1 | grep -v "warning: argument unused" || true
  echo "built ab_libs/lib_$name.so  [$defs]"
done
