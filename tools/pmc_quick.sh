#!/bin/bash
# Quick instruction-mix profile of the greedy-loop kernels (one PMC pass + kernel stats) -> gpurun_out/pmcq/
#   gpurun --timeout 600 -- 'bash tools/pmc_quick.sh [batch]'
B=${1:-64}
OUT=gpurun_out/pmcq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -o pmc -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --batch $B > $OUT/pmc_$NAME.log 2>&1
  python tools/summarise_pmc.py $OUT/pmc_$NAME > $OUT/pmc_$NAME.summary.txt 2>&1
  rm -rf $OUT/pmc_$NAME
  grep -A6 "k_iter_update\|k_iter_select" $OUT/pmc_$NAME.summary.txt | head -20
done
