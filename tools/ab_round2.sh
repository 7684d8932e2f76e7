bash tools/ab_run.sh 100 2>&1 | tee gpurun_out/ab/summary.txt
unset DA4ML_HIP_LIB
for q in 4 8; do for l in 3 4 5 6; do
  echo "== QUEUES=$q LANES=$l $(GPU_MAX_HW_QUEUES=$q DA4ML_HIP_LANES=$l timeout 60 python tests/gpu_profile.py 256 64 2>&1 | head -1)"
done; done 2>&1 | tee gpurun_out/ab/lanes.txt
echo "== TABLE_SCALE=2 $(DA4ML_HIP_TABLE_SCALE=2 timeout 60 python tests/gpu_profile.py 256 64 2>&1 | head -1)" | tee -a gpurun_out/ab/lanes.txt
for ub in 512 1024 2048 4096; do echo "== UPD_BLOCKS=$ub $(DA4ML_HIP_UPD_BLOCKS=$ub timeout 60 python tests/gpu_profile.py 256 64 2>&1 | head -1)"; done | tee -a gpurun_out/ab/lanes.txt
