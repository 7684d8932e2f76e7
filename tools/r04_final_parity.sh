#!/bin/bash
# Round 4, final sources: the whole GPU parity suite, smoke, then the broad soak (random option sets, odd steps, methods, a 512x512 chain).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/parity.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
timeout 900 python tools/gpu_soak.py 600 2>&1 | tail -12 | tee $O/soak.log
