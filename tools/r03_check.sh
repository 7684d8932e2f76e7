#!/bin/bash
# GPU parity suite + one short bench line (no CPU leg) + timing of both engines: the quick check after a change
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/check; rm -f gpurun_out/check/*
timeout 1500 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -6 | tee gpurun_out/check/parity.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err; tail -3 gpurun_out/check/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/check/bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 1), 'dominant', r['kernel'], 'profiles', r['profiles'])
for n, k in r['kernels'].items():
    print(' ', n, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in k.items() if a in ('achieved', 'frac', 'avg_launch_us', 'alg_bytes_per_chain_step', 'traffic')})
print(' engine', {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d['engine'].items() if a != 'source_digest'})
v = d['check']['verify']
print(' verify', v['kernel_reproduced'], '/', v['of'], 'digests', v['digests_checked'], 'ref', v['digests_from_reference_build'], 'all_ok', v['all_ok'])
PY
for i in 1 2; do timeout 120 python tests/gpu_profile.py 256 64 2>&1 | sed -n '1p;6p'; done
timeout 120 python tests/gpu_profile.py 256 1 2>&1 | sed -n 1p
