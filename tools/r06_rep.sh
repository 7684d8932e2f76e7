cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_par; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -rsx > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
timeout 900 python tools/gpu_soak.py 600 > $O/gpu_soak.txt 2>&1; tail -4 $O/gpu_soak.txt
timeout 300 python tools/gpu_stress.py 9 2>&1 | tail -2
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 | python -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['engine']['us_per_lockstep_iteration'], l['check']['verify']['all_ok'], l['check']['verify']['digests_checked'])"
