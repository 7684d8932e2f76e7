"""Time curve of one full greedy chain of the REFERENCE build (oracle/_ref/libref.so, 1 thread) -> tests/golden/cpu_calibration.json.

bench.py's cpu_baseline leg times a bounded prefix of the same chain on the GPU box's host and scales it to a full chain
with the ratio (total time) / (time of the same prefix) taken from this curve.  The curve must come from the same code
(the reference's own sources, not the restated port) -- the host differs, the ratio of iteration costs along the chain
does not (it is set by the table size F(t) and the regenerated pairs per iteration, both properties of the chain).
usage: python tests/golden/make_cpu_calibration.py [n=256] [seed=0]     (~70-80 minutes for 256)"""
import json, os, sys, tempfile, time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
import numpy as np  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ['OMP_NUM_THREADS'] = '1'
trace = Path(tempfile.mkdtemp()) / 'trace.txt'
os.environ['REF_TRACE'] = str(trace)
from oracle.oracle import Oracle, sample_chain  # noqa: E402

k = np.random.default_rng(seed).integers(-128, 128, (n, n)).astype(np.float32)
t = time.time()
s = sample_chain(Oracle('ref'), k, 'wmc', 1e9)
assert s['finished'], s
marks = [line.split() for line in trace.read_text().splitlines()]
rec = {
    'source': f'oracle/_ref/libref.so (the reference sources, 1 thread) on the build container, seed-{seed} {n}x{n} int8, method wmc; REF_TRACE time marks',
    'kind': 'reference', 'iterations': s['iterations'], 'total_s': s['create_s'] + s['iter_s'], 'create_s': s['create_s'],
    'iter_marks': [int(a) for a, _ in marks], 'time_marks_s': [float(b) for _, b in marks], 'wall_s': time.time() - t,
}  # fmt: skip
path = HERE / 'cpu_calibration.json'
data = json.loads(path.read_text()) if path.exists() else {}
data[f'{n}x{n}'] = rec
path.write_text(json.dumps(data))
print({k: v for k, v in rec.items() if not isinstance(v, list)})
