cd "$GRAFT_REPO_ROOT"
run() { echo "$1 $3 B=$2: $(env $3 DA4ML_HIP_LIB=ab_libs/lib_$1.so timeout 90 python tests/gpu_profile.py 256 $2 | head -1 | sed 's/.*us\/iter //')"; }
for rep in 1 2 3; do run side 64; run q16 64; run q8 64; run q8 64 DA4ML_HIP_UPD_BLOCKS=1280; run q8 64 DA4ML_HIP_UPD_BLOCKS=1920; done
for rep in 1 2; do run side 1; run q16 1; run q8 1; done
