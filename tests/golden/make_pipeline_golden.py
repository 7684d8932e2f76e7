"""Golden vectors for da4ml_amd.trace (to_pipeline / retime_pipeline / dead_statement_elimination) from the reference's
OWN Python (src/da4ml/trace/pipeline.py, tracer.py, fixed_variable.py, types.py), imported in the build container with
its native module supplied by oracle/_ref/libref.so (tests/golden/ref_py.py).  Run in the build container only:

    python tests/golden/make_pipeline_golden.py
"""
import contextlib, gzip, io, json, sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE))

import ref_py  # noqa: E402
from pipeline_cases import BIG, CUTOFFS, REPLAY, SOLVES, handmade, handmade_tables, replay_inputs, solve_inputs  # noqa: E402


def dump(p):
    return json.loads(json.dumps(p, default=lambda o: o.to_dict()))


def main():
    m = ref_py.load()
    R, rt, rp, rtr = m['oracle'], m['types'], m['pipeline'], m['tracer']
    out = {'source': 'reference src/da4ml/trace/{pipeline,tracer,fixed_variable}.py + types.py run over oracle/_ref/libref.so', 'split': [], 'retime': [], 'dce': [], 'replay': []}
    for spec in SOLVES:
        k, opts = solve_inputs(spec)
        pipe = ref_py.to_ref_pipeline(rt, R.solve(k, **opts))
        for si, comb in enumerate(pipe.solutions):
            if not comb.ops:
                continue
            for cut in CUTOFFS:
                try:
                    res = dump(rp.to_pipeline(comb, cut, retiming=False))
                except Exception as e:  # the reference's own failure mode is part of the contract
                    res = {'raises': type(e).__name__}
                out['split'].append({'solve': spec[0], 'stage': si, 'cutoff': cut, 'result': res})
            for cut in CUTOFFS[1:5]:
                buf = io.StringIO()
                try:
                    with contextlib.redirect_stdout(buf):
                        res = dump(rp.to_pipeline(comb, cut, retiming=True))
                except Exception as e:
                    res = {'raises': type(e).__name__}
                out['retime'].append({'solve': spec[0], 'stage': si, 'cutoff': cut, 'result': res, 'stdout': buf.getvalue()})
            # dead statements: keep every other output only
            half = rt.CombLogic(comb.shape, comb.inp_shifts, [i if j % 2 == 0 else -1 for j, i in enumerate(comb.out_idxs)], comb.out_shifts,
                                comb.out_negs, comb.ops, comb.carry_size, comb.adder_size, None)  # fmt: skip
            for keep in (False, True):
                out['dce'].append({'solve': spec[0], 'stage': si, 'keep_dead_inputs': keep, 'result': dump(rtr.dead_statement_elimination(half, keep))})
        # whole two-stage result through retime_pipeline directly
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                res = dump(rp.retime_pipeline(pipe))
        except Exception as e:
            res = {'raises': type(e).__name__}
        out['retime'].append({'solve': spec[0], 'stage': 'pipeline', 'cutoff': None, 'result': res, 'stdout': buf.getvalue()})
    hm = ref_py.to_ref_comb(rt, handmade())
    for keep in (False, True):
        out['dce'].append({'solve': 'handmade', 'stage': 0, 'keep_dead_inputs': keep, 'result': dump(rtr.dead_statement_elimination(hm, keep))})
    for cut in (0.0, 1.0, 2.0):
        out['split'].append({'solve': 'handmade', 'stage': 0, 'cutoff': cut, 'result': dump(rp.to_pipeline(hm, cut, retiming=False))})
    ht = handmade_tables()
    ht_ref = ref_py.to_ref_comb(rt, ht)._replace(lookup_tables=ht.lookup_tables)  # table objects are opaque to these passes
    for cut in (0.0, 1.0, 1.5, 3.0):
        out['split'].append({'solve': 'handmade_tables', 'stage': 0, 'cutoff': cut, 'result': dump(rp.to_pipeline(ht_ref, cut, retiming=False))})
    for keep in (False, True):
        out['dce'].append({'solve': 'handmade_tables', 'stage': 0, 'keep_dead_inputs': keep, 'result': dump(rtr.dead_statement_elimination(ht_ref, keep))})
    # numeric replay (reference types.py CombLogic.__call__, one sample at a time) of graphs with tracer statements
    for spec in REPLAY:
        comb, x = replay_inputs(spec)
        rc = ref_py.to_ref_comb(rt, comb)
        out['replay'].append({'graph': spec[0], 'outputs': [[float(v) for v in rc(list(row))] for row in x],
                              'buffer0': [float(v) for v in rc(list(x[0]), dump=True)]})  # fmt: skip
    # quantize=True: inputs truncated + wrapped into the declared input formats first (types.py:247-249), every stage
    import numpy as np

    out['quantized'] = []
    for spec in SOLVES[:7]:
        k, opts = solve_inputs(spec)
        pipe = ref_py.to_ref_pipeline(rt, R.solve(k, **opts))
        x = np.random.default_rng(3).uniform(-300, 300, (16, k.shape[0]))
        out['quantized'].append({'solve': spec[0], 'outputs': [[float(v) for v in pipe(list(row), quantize=True)] for row in x]})
    # larger solver outputs by digest
    import hashlib
    import signal

    class _Hang(Exception):
        pass

    def _alarm(*a):
        raise _Hang()

    signal.signal(signal.SIGALRM, _alarm)
    out['big'] = []
    for name, recipe, opts, cuts in BIG:
        k, _ = solve_inputs((name, recipe, opts))
        comb = ref_py.to_ref_pipeline(rt, R.solve(k, **opts)).solutions[0]
        for cut in cuts:
            for retiming in (False, True):
                signal.alarm(60)  # the reference's retiming bisection does not always terminate
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        res = rp.to_pipeline(comb, cut, retiming=retiming)
                except _Hang:
                    out['big'].append({'solve': name, 'cutoff': cut, 'retiming': retiming, 'reference_hangs': True})
                    continue
                finally:
                    signal.alarm(0)
                out['big'].append({'solve': name, 'cutoff': cut, 'retiming': retiming, 'n_ops': [len(s.ops) for s in res[0]],
                                   'sha256': hashlib.sha256(json.dumps(dump(res), separators=(',', ':')).encode()).hexdigest()})  # fmt: skip
    path = HERE / 'pipeline_golden.json.gz'
    with gzip.open(path, 'wt', compresslevel=9) as f:
        json.dump(out, f, separators=(',', ':'))
    print(path, path.stat().st_size, 'bytes;', {k: len(v) for k, v in out.items() if isinstance(v, list)})
    print('raises:', [(c['solve'], c['stage'], c['cutoff'], c['result']['raises']) for sec in ('split', 'retime') for c in out[sec] if isinstance(c['result'], dict)])


if __name__ == '__main__':
    main()
