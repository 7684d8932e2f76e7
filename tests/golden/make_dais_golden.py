"""Golden vectors for the DAIS executor from the REFERENCE's own interpreter (oracle/_ref/libdais_ref.so, built from
/root/reference/src/da4ml/_binary/dais/DAISInterpreter.cc by oracle/Makefile).  Writes tests/golden/dais_golden.json.gz:
for each seed the program words, the inputs and the reference's outputs (as float64 hex strings, bit-exact)."""
import ctypes as C, gzip, json, sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent)); sys.path.insert(0, str(HERE.parent))
import numpy as np  # noqa: E402
from dais_cases import narrow_condition_program, random_program  # noqa: E402

R = C.CDLL(str(HERE.parent.parent / 'oracle' / '_ref' / 'libdais_ref.so'))
R.dref_run.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
R.dref_last_error.restype = C.c_char_p

cases = []
for seed in range(49):
    prog, x = narrow_condition_program() if seed == 48 else random_program(seed, n_samples=8)
    x = np.ascontiguousarray(x)
    out = np.zeros((x.shape[0], int(prog[3])))
    if R.dref_run(prog.ctypes.data, prog.size, x.ctypes.data, x.shape[0], out.ctypes.data) != 0:
        raise SystemExit(f'reference interpreter failed on seed {seed}: {R.dref_last_error().decode()}')
    cases.append({'seed': seed, 'program': prog.tolist(), 'inputs': [v.hex() for v in x.ravel().tolist()], 'n_samples': x.shape[0],
                  'outputs': [v.hex() for v in out.ravel().tolist()]})
with gzip.open(HERE / 'dais_golden.json.gz', 'wt') as f:
    json.dump({'generator': 'tests/golden/make_dais_golden.py', 'source': 'reference DAISInterpreter.cc via oracle/_ref/libdais_ref.so', 'cases': cases}, f, separators=(',', ':'))
print(len(cases), 'cases', (HERE / 'dais_golden.json.gz').stat().st_size, 'bytes')
