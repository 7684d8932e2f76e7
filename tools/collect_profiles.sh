#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC counter passes -> gpurun_out/profiles_rNN/
# usage: tools/collect_profiles.sh r02        (QUICK=1: bench line, kernel stats and the two HBM-traffic passes only; SKIP_BENCH=1: reuse gpurun_out/profiles_rNN/bench.json;
#        PMC_ONLY=1: kernel stats and all PMC passes, nothing else -- then copy the summaries into profiles/ and run bench.py, so that the line is
#        written with counters of the sources it measures)
set -u
TAG=${1:-r02}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. the driver's contract line (incl. the all-core CPU baseline and the replay of all 64 results)
[ -n "${PMC_ONLY:-}" ] || { [ -n "${SKIP_BENCH:-}" ] && [ -s $OUT/bench.json ]; } || python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err   # SKIP_BENCH=1: keep the line of an earlier call
# 2. kernel trace + stats of the same workload (one step, no CPU leg, no verification)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c3 -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify > $OUT/trace.log 2>&1
cp $OUT/trace/c3_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace   # large per-dispatch trace: the stats summary is what gets committed
# 3. PMC passes (counters only, own runs) at the BENCHED batch of 64 = 4 chain groups of 16 chains per dispatch
python - > $OUT/pmc_meta.json <<'PY'
import json, subprocess, sys
sys.path.insert(0, '.')
import bench
rev = subprocess.run(['git', 'rev-parse', 'HEAD'], capture_output=True, text=True).stdout.strip() or None  # (no .git on the GPU box)
print(json.dumps({'batch': 64, 'groups': 4, 'chains_per_dispatch': 16.0, 'workload': 'c3_256x256_int8_batch64_single_chain', 'src_sha256': bench.source_digest(), 'git': rev}))
PY
if [ -n "${QUICK:-}" ]; then SETS=("FETCH_SIZE" "WRITE_SIZE"); else SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"); fi
for SET in "${SETS[@]}"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -o pmc -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify > $OUT/pmc_$NAME.log 2>&1
  python tools/summarise_pmc.py $OUT/pmc_$NAME > $OUT/pmc_$NAME.summary.txt 2>&1
  rm -rf $OUT/pmc_$NAME
done
[ -n "${QUICK:-}${PMC_ONLY:-}" ] && { ls -la $OUT; exit 0; }
# 4. the other workloads: 64x64 batch, end-to-end model compile (C5), column-sharded chain (one GPU: phases forced, exchanges no-ops)
timeout 300 python bench.py --workload c2_64x64_int8_batch64_single_chain --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 600 python bench.py --workload c5_model_batch --steps 3 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
timeout 600 python tools/shard_bench.py 256 > $OUT/column_sharded.txt 2>&1
# 5. DAIS device executor (k_dais_run): throughput line + kernel stats
if [ "${DAIS:-1}" != "0" ]; then
  timeout 300 python tools/dais_bench.py 64 1048576 > $OUT/dais_bench.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dais_trace -o dais -- python tools/dais_bench.py 64 1048576 > $OUT/dais_trace.log 2>&1
  cp $OUT/dais_trace/dais_kernel_stats.csv $OUT/dais_kernel_stats.csv 2>/dev/null; rm -rf $OUT/dais_trace
fi
ls -la $OUT
