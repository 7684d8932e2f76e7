cd "$GRAFT_REPO_ROOT"
for b in 1 4 8 12 16 24 32 48 64 96 128; do echo "batch $b: $(timeout 100 python tests/gpu_profile.py 256 $b 2>&1 | sed -n '1p;5p' | tr '\n' ' ')"; done
for l in 1 2; do echo "LANES=$l $(DA4ML_HIP_LANES=$l timeout 60 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1p;5p' | tr '\n' ' ')"; done
