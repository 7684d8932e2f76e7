cd "$GRAFT_REPO_ROOT"
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -20
(for i in 1 2 3 4 5 6; do sleep 0.7; rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | tr '\n' ' '; echo; done) &
python tests/gpu_profile.py 256 64 | head -1
wait
rocm-smi --showpower 2>&1 | grep -i power | head -3
