"""One-rank exercise, on a GPU, of the torch.distributed / `nccl` (= RCCL) code paths that the N-rank bench and the callback
transport of the column-sharded chain use -- the paths a one-GPU box can reach: process group over RCCL, max / sum over
ranks and barrier on device tensors, the DEVICE-buffer branch of the all-reduce callback (exchanges forced although there is
one rank), shutdown.  Prints one JSON line.  usage (GPU box): python tools/nccl_world1_check.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1')
if 'MASTER_PORT' not in os.environ:  # a free port of this host
    import socket

    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
os.environ['DA4ML_SHARD_FORCE'] = '1'  # sharded phases although there is one rank
os.environ['DA4ML_SHARD_FORCE_COMM'] = '1'  # ... and every exchange really calls the collective

import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)

from da4ml_amd import _binary as hip
from da4ml_amd import multi_gpu as mg

dev = torch.device('cuda:0')
out = {'backend': dist.get_backend(), 'max_over_ranks': mg.max_over_ranks(1.5, dev), 'sum_over_ranks': mg.sum_over_ranks(64, dev)}
mg.barrier()
k = np.random.default_rng(0).integers(-128, 128, (48, 48)).astype(np.float32)
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
ref = hip.solve(k, **opts)
for transport in ('callback', 'rccl'):
    t = time.perf_counter()
    pipe, stats = mg.solve_column_sharded(k, transport=transport, return_stats=True, **opts)
    out[transport] = {'equals_unsharded_solve': pipe == ref, 'seconds': round(time.perf_counter() - t, 3), 'stats': stats}
mg.shutdown()
out['ok'] = bool(out['callback']['equals_unsharded_solve'] and out['rccl']['equals_unsharded_solve'] and out['max_over_ranks'] == 1.5 and out['sum_over_ranks'] == 64)
print(json.dumps(out), flush=True)
