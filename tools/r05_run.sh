cd "$GRAFT_REPO_ROOT"
for w in "early 2000" "late 9000"; do set -- $w; for B in 64 1; do echo "== $1 batch $B: $(TIMER_WINDOW_STEPS=$2 DA4ML_HIP_LIB=ab_libs/lib_timers_$1.so timeout 120 python tests/gpu_profile.py 256 $B | grep 'window\|search work\|search block')"; done; done
