cd "$GRAFT_REPO_ROOT"; export DA_ROOT=$PWD
for n in pre head; do
  echo "== $n"; DA4ML_HIP_LIB=ab_libs/lib_$n.so timeout 90 python -c "
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import test_zy_shard_gpu as t
exec(t.RCCL_ONE)
" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3; echo "rc $?"
done
