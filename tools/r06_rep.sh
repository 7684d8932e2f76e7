cd "$GRAFT_REPO_ROOT"; N=4 B=64 bash tools/r05_repeat.sh cur par 2>&1 | tail -2; N=3 B=1 bash tools/r05_repeat.sh cur par 2>&1 | tail -2
for l in clkA clkApar; do for b in 64 1; do echo "== $l batch $b"; STEP_CLOCKS=1 DA4ML_HIP_LIB=ab_libs/lib_$l.so timeout 120 python tests/gpu_profile.py 256 $b 2>&1 | grep "step clocks"; done; done
