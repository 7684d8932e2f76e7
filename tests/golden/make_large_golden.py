"""Golden record of a BASELINE C3 matrix (256x256 int8, seed 0 by default) from the CPU oracle -- a ~70 minute single-thread run
(usage: make_large_golden.py [n] [seed] [path] [kind]; concurrent runs must not share the output file: pass a third argument as
path; kind = 'port' (the restatement, default) or 'ref' (oracle/_ref/libref.so, the reference's own sources -> key suffix '_ref').
Writes tests/golden/large_chain_golden.json (digest of the full result, cost, adders, ops per stage, wall time)."""
import hashlib, json, sys, time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
import numpy as np  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
k = np.random.default_rng(seed).integers(-128, 128, (n, n)).astype(np.float32)
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
t = time.time()
kind = sys.argv[4] if len(sys.argv) > 4 else 'port'
p = Oracle(kind).solve(k, **opts)
dt = time.time() - t
dump = json.loads(json.dumps(p, default=lambda o: o.to_dict()))
rec = {'sha256': hashlib.sha256(json.dumps(dump, separators=(',', ':')).encode()).hexdigest(), 'cost': p.cost, 'adders': p.n_adders,
       'n_ops': [len(s.ops) for s in p.solutions], 'oracle_seconds': dt, 'opts': opts, 'oracle': 'oracle/_ref/libref.so' if kind == 'ref' else 'oracle/liboracle.so'}
path = Path(sys.argv[3]) if len(sys.argv) > 3 else HERE / 'large_chain_golden.json'
data = json.loads(path.read_text()) if path.exists() else {}
data[f'{n}x{n}_seed{seed}_single_chain' + ('_ref' if kind == 'ref' else '')] = rec
path.write_text(json.dumps(data, indent=1))
print(rec)
