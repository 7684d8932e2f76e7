"""Golden digests for the MEDIUM-size method / hard_dc / cost-model grid (tests/test_gpu_methods.py), produced by the
reference's own sources (oracle/_ref/libref.so).  Run in the build container only:

    python tests/golden/make_methods_golden.py

Every selector of reference indexers.cc:6-90 (mc, mc-dc, mc-pdc, wmc, wmc-dc, wmc-pdc) x hard_dc in {-1, 0, 2} x
(adder_size, carry_size) in {(-1,-1), (1,-1), (4,8)} on 32x32 and 64x64 int8 matrices, search_all_decompose_dc=False (so the
`decompose_dc--` retry of api.cc:117-139 runs), plus every method as a single 64x64 chain (decompose_dc=-1).  The GPU test
compares sha256 digests of the complete result (every op's ids, opcode, shift, interval, latency, cost; all output
indices / shifts / signs)."""
import gzip, hashlib, json, sys, time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))

from cases import METHOD_GRID, int_matrix  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def digest(p):
    dump = json.loads(json.dumps(p, default=lambda o: o.to_dict()))
    return hashlib.sha256(json.dumps(dump, separators=(',', ':')).encode()).hexdigest()


def main():
    R = Oracle('ref')
    out = {'source': 'oracle/_ref/libref.so = /root/reference/src/da4ml/_binary/cmvm/*.cc', 'digests': []}
    t0 = time.time()
    for name, kspec, opts in METHOD_GRID:
        p = R.solve(int_matrix(*kspec), **opts)
        out['digests'].append({'case': name, 'kernel': list(kspec), 'opts': opts, 'sha256': digest(p), 'cost': p.cost,
                               'n_ops': [len(s.ops) for s in p.solutions]})  # fmt: skip
    with gzip.open(HERE / 'methods_golden.json.gz', 'wt') as f:
        json.dump(out, f, separators=(',', ':'))
    print('digests', len(out['digests']), f'{time.time() - t0:.0f} s')


if __name__ == '__main__':
    main()
