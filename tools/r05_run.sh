cd "$GRAFT_REPO_ROOT"
export DA4ML_HIP_LIB=ab_libs/lib_young.so
for rep in 1 2; do for y in -1 512 128 32 8 0; do echo "young_min $y: C3x64 $(DA4ML_HIP_YOUNG_MIN=$y timeout 90 python tests/gpu_profile.py 256 64 | head -1 | sed 's/.*loop/loop/') | one chain $(DA4ML_HIP_YOUNG_MIN=$y timeout 90 python tests/gpu_profile.py 256 1 | head -1 | sed 's/.*us.iter/us\/iter/') | 64x64 $(DA4ML_HIP_YOUNG_MIN=$y timeout 90 python tests/gpu_profile.py 64 64 | head -1 | sed 's/.*us.iter/us\/iter/')"; done; done
echo "orsplit2: $(DA4ML_HIP_LIB=ab_libs/lib_orsplit2.so timeout 90 python tests/gpu_profile.py 256 64 | head -1 | sed 's/.*loop/loop/')"
timeout 200 python tools/gpu_stress_small.py 100 | tail -1
DA4ML_HIP_YOUNG_MIN=0 timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_small or c3_256 or c2_64 or capacity or batch" | tail -1
