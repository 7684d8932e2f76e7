"""Golden record of the BASELINE C3 (i) workload -- one full default solve() (all decompose_dc candidates, both stages)
of the 256x256 int8 seed-0 matrix -- from the CPU oracle with all host cores over the candidates (hours).
Writes tests/golden/large_default_golden.json.  usage: python tests/golden/make_default_golden.py [n] [seed] [kind] [path]
(kind 'port' = the restatement, default; 'ref' = oracle/_ref/libref.so, the reference's own sources -> key suffix '_ref')."""
import hashlib, json, os, sys, time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
import numpy as np  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
k = np.random.default_rng(seed).integers(-128, 128, (n, n)).astype(np.float32)
t = time.time()
kind = sys.argv[3] if len(sys.argv) > 3 else 'port'
p, _, picked = Oracle(kind).solve(k, stats=True)
dt = time.time() - t
dump = json.loads(json.dumps(p, default=lambda o: o.to_dict()))
rec = {'sha256': hashlib.sha256(json.dumps(dump, separators=(',', ':')).encode()).hexdigest(), 'cost': p.cost, 'adders': p.n_adders,
       'n_ops': [len(s.ops) for s in p.solutions], 'picked_candidate': picked, 'oracle_seconds': dt,
       'threads': int(os.environ.get('OMP_NUM_THREADS', os.cpu_count())), 'opts': {}, 'oracle': 'oracle/_ref/libref.so' if kind == 'ref' else 'oracle/liboracle.so'}
path = Path(sys.argv[4]) if len(sys.argv) > 4 else HERE / 'large_default_golden.json'
data = json.loads(path.read_text()) if path.exists() else {}
data[f'{n}x{n}_seed{seed}_default' + ('_ref' if kind == 'ref' else '')] = rec
path.write_text(json.dumps(data, indent=1))
print(rec)
