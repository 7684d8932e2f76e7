#!/bin/bash
# Round 6, third GPU call: where the batch's step time goes ALONG the chain (step clocks over four windows, batch 64 and one chain), the instruction
# cache (PMC pass + the alternating-launch probe), hipExtAnyOrderLaunch on gfx950.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_diag3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in A B C D; do for b in 64 1; do
  echo "== window $w batch $b" >> $O/step_clock_windows.txt
  STEP_CLOCKS=1 DA4ML_HIP_LIB=ab_libs/lib_clk$w.so timeout 120 python tests/gpu_profile.py 256 $b 2>&1 | grep "step clocks\|us/iter" >> $O/step_clock_windows.txt
done; done
cat $O/step_clock_windows.txt
timeout 300 tools/micro/launch_probe > $O/launch_probe.txt 2>&1; cat $O/launch_probe.txt
rocprofv3 -L 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_WAIT_ANY\|SQ_WAIT_INST_ANY" | sort -u > $O/icache_counters.txt; cat $O/icache_counters.txt | tr '\n' ' '; echo
SET="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
timeout 600 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_icache -o pmc -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify > $O/pmc_icache.log 2>&1
python tools/summarise_pmc.py $O/pmc_icache > $O/pmc_icache.summary.txt 2>&1; rm -rf $O/pmc_icache; head -12 $O/pmc_icache.summary.txt
SET="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH"
timeout 600 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_wait -o pmc -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify > $O/pmc_wait.log 2>&1
python tools/summarise_pmc.py $O/pmc_wait > $O/pmc_wait.summary.txt 2>&1; rm -rf $O/pmc_wait; head -14 $O/pmc_wait.summary.txt; tail -3 $O/pmc_wait.log
