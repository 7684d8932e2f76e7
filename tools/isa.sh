#!/bin/bash
# ISA of the greedy-loop kernels (gfx950) into /tmp/{upd,sel}.s + the memory / wait / branch skeleton of one of them.
# usage: tools/isa.sh [upd|sel] [first_line last_line]
set -e
cd "$(dirname "$0")/../da4ml_amd/csrc"
make asm 2>&1 | grep -v "remark:" | grep -v "^/opt/rocm/bin/hipcc" | grep -v "c0_\|^ *[0-9]* |\|^ *| *\^\|1 warning generated" || true
S=/tmp/cmvm_engine-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN2da3gpu13k_iter_updateIjEEvPNS0_8ChainDevEi:/{p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' $S > /tmp/upd.s
awk '/^_ZN2da3gpu14k_iter_select2IjEEvPNS0_8ChainDevEiPji:/{p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' $S > /tmp/sel.s
wc -l /tmp/upd.s /tmp/sel.s | head -2
grep -A9 "k_iter_updateIjEE\|k_iter_select2IjEE" <(make asm 2>&1) | grep "VGPRs\|SGPRs\|Scratch\|Occupancy" | tr -s ' ' | cut -d: -f4- | paste -sd' ' || true
k=${1:-upd}
grep -n "s_load\|s_waitcnt\|s_barrier\|global_load\|s_cbranch\|s_endpgm\|^.LBB\|global_store\|global_atomic\|s_memtime\|ds_add\|ds_read\|ds_write\|ds_max\|ds_or\|flat_\|scratch_" /tmp/$k.s | sed -n "${2:-1},${3:-80}p"
