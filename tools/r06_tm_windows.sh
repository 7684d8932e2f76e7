cd "$GRAFT_REPO_ROOT"
for w in A:2000 D:7000; do for b in 64 1; do
  echo "== window ${w%%:*} batch $b"
  DA4ML_HIP_STATS=1 TIMER_WINDOW_STEPS=${w##*:} DA4ML_HIP_LIB=ab_libs/lib_tm${w%%:*}.so timeout 120 python tests/gpu_profile.py 256 $b 2>&1 | grep "window:\|us/iter"
done; done
