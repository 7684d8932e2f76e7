cd "$GRAFT_REPO_ROOT"
for t in 1 2 1 2 1 2; do echo "threads $t: C3x64 $(DA4ML_HIP_LAUNCH_THREADS=$t DA4ML_HIP_LIB=ab_libs/lib_cur.so timeout 90 python tests/gpu_profile.py 256 64 | head -1 | sed 's/.*loop/loop/') | 64x64 $(DA4ML_HIP_LAUNCH_THREADS=$t DA4ML_HIP_LIB=ab_libs/lib_cur.so timeout 90 python tests/gpu_profile.py 64 64 | head -1) | 128 $(DA4ML_HIP_LAUNCH_THREADS=$t DA4ML_HIP_LIB=ab_libs/lib_cur.so timeout 90 python tests/gpu_profile.py 128 64 | head -1)"; done
DA4ML_HIP_LIB=ab_libs/lib_cur.so timeout 200 python tools/gpu_stress_small.py 100 | tail -1
DA4ML_HIP_LIB=ab_libs/lib_cur.so timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -k "random_small or c3_256 or c2_64 or capacity or batch" | tail -1
