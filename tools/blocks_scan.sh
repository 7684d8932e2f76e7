cd "$GRAFT_REPO_ROOT"
for ub in 2560 3072 3584 4096 5120; do echo "== UPD_BLOCKS=$ub $(DA4ML_HIP_UPD_BLOCKS=$ub timeout 60 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1p;5p' | tr '\n' ' ')"; done
