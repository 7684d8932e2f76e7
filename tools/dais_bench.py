"""Throughput of the DAIS executors on a solver result (run through gpurun):  python tools/dais_bench.py [n] [samples]

Prints samples/s of the host executor (all threads) and of the device executor (k_dais_run), and the device executor's
algorithmic register traffic (8 B written + 8 B per operand read, per step and sample) over its wall time -- wall time
includes the H2D/D2H copies of inputs and outputs; the kernel-only figure comes from rocprofv3 --kernel-trace --stats."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'tests'))
from cases import int_matrix  # noqa: E402

from da4ml_amd.cmvm import solve  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
stage = solve(int_matrix(0, n, n, -128, 128), method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False).solutions[0]
x = np.random.default_rng(0).integers(-128, 128, (samples, n)).astype(np.float64)
adders = sum(op.opcode in (0, 1) for op in stage.ops)
bytes_per_sample = 8 * len(stage.ops) + 16 * adders + 8 * (n + stage.shape[1])
stage.predict(x[:4096], executor='device')  # warm-up (module load)
for name, kw, count in (('host', dict(n_threads=0), min(samples, 1 << 17)), ('device', dict(executor='device'), samples)):
    t0 = time.perf_counter()
    y = stage.predict(x[:count], **kw)
    dt = time.perf_counter() - t0
    print(f'{name}: {n}x{n}, {len(stage.ops)} steps, {count} samples in {dt:.3f} s = {count / dt:.3e} samples/s, '
          f'{bytes_per_sample * count / dt / 1e9:.1f} GB/s algorithmic register traffic')
assert np.array_equal(y, x @ stage.kernel.astype(np.float64))
print('device result == matrix product')
