cd "$GRAFT_REPO_ROOT"
export DA4ML_HIP_LIB=ab_libs/lib_cur.so
for b in 2048 2560 3072 3584; do echo "upd_blocks $b: $(DA4ML_HIP_UPD_BLOCKS=$b timeout 90 python tests/gpu_profile.py 256 64 | head -1 | sed 's/.*loop/loop/')"; done
for l in 2 3 4 5; do echo "lanes $l: $(DA4ML_HIP_LANES=$l timeout 90 python tests/gpu_profile.py 256 64 | head -1| sed 's/.*loop/loop/')"; done
for n in cur w5 ch2 cur w5 ch2; do echo "$n: $(DA4ML_HIP_LIB=ab_libs/lib_$n.so timeout 90 python tests/gpu_profile.py 256 64 | head -1| sed 's/.*loop/loop/')"; done
bash tools/r05_batch_scan.sh cur
