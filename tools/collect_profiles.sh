#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC counter passes -> gpurun_out/profiles_rNN/
# usage: tools/collect_profiles.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --cpu-seconds 20 > $OUT/bench.json 2> $OUT/bench.err
# kernel trace + stats of the same command (one step, no CPU leg)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c3 -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/trace.log 2>&1
cp $OUT/trace/c3_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -f $OUT/trace/c3_kernel_trace.csv   # large per-dispatch trace: the stats summary is what gets committed
# PMC passes (counters only, own runs; smaller batch to bound the time)
# QUICK=1: only the two HBM-traffic passes (the roofline's `traffic` field needs them)
if [ -n "${QUICK:-}" ]; then SETS=("FETCH_SIZE" "WRITE_SIZE"); else SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"); fi
for SET in "${SETS[@]}"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -o pmc -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --batch 16 > $OUT/pmc_$NAME.log 2>&1
  python tools/summarise_pmc.py $OUT/pmc_$NAME > $OUT/pmc_$NAME.summary.txt 2>&1
  rm -rf $OUT/pmc_$NAME
done
# DAIS device executor (k_dais_run): throughput line + kernel stats + its HBM traffic (DAIS=0 skips)
if [ "${DAIS:-1}" != "0" ]; then
  timeout 300 python tools/dais_bench.py 64 1048576 > $OUT/dais_bench.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dais_trace -o dais -- python tools/dais_bench.py 64 1048576 > $OUT/dais_trace.log 2>&1
  cp $OUT/dais_trace/dais_kernel_stats.csv $OUT/dais_kernel_stats.csv 2>/dev/null; rm -rf $OUT/dais_trace
  for SET in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/dais_pmc_$SET -o pmc -- python tools/dais_bench.py 64 262144 > $OUT/dais_pmc_$SET.log 2>&1
    python tools/summarise_pmc.py $OUT/dais_pmc_$SET > $OUT/dais_pmc_$SET.summary.txt 2>&1
    rm -rf $OUT/dais_pmc_$SET
  done
fi
ls -la $OUT
