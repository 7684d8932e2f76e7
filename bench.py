#!/usr/bin/env python
"""Benchmark of the CMVM solve path on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[2], the one the metric is quoted on): a batch of 64 independent 256x256 int8 constant
matrices ("64 restarts, one per CU block"), each solved as one greedy CSE chain
    solve(k, method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
(stage 2a on the centred matrix + the identity stage 2b + adder trees).  One "step" = one
pass over one such batch through the C ABI (da_solve_batch), results left on the C side.  With N GPUs every rank
solves its own 64 matrices (weak scaling, no data-path collective); value = matrices solved by all ranks / max-over-
ranks time.  Synthetic data: default_rng(seed).integers(-128, 128).

Extra objects on the JSON line:
  roofline      dominant kernel k_iter_update: algorithmic bytes per launch / its average
                duration measured with HIP events on the launch stream inside the library (every 16th launch sampled)
  cpu_baseline  the CPU oracle (restated reference, 1 thread) on a bounded sample of the same workload
"""

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (n_in, n_out, batch, solve options)
    'c3_256x256_int8_batch64_single_chain': (256, 256, 64, dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)),
    'c3_256x256_int8_batch8_default_search': (256, 256, 8, dict()),
    'c2_64x64_int8_batch64_single_chain': (64, 64, 64, dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def pmc_traffic_per_chain():
    """HBM bytes per chain and k_iter_update launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_FETCH_SIZE.txt,
    ..._WRITE_SIZE.txt; collected with --batch 16, i.e. 8 chains per launch; KiB units, uncorrected).  None if absent."""
    import re

    total = 0.0
    for name in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = sorted((ROOT / 'profiles').glob(f'r*_pmc_{name}.txt'))
        if not files:
            return None
        m = re.search(r'k_iter_update.*?per_dispatch=([0-9.eE+-]+)', files[-1].read_text(), re.S)
        if not m:
            return None
        total += float(m.group(1)) * 1024.0
    return total / 8.0


def make_batch(n_in, n_out, batch, first_seed):
    return [np.random.default_rng(first_seed + i).integers(-128, 128, (n_in, n_out)).astype(np.float32) for i in range(batch)]


def cpu_baseline(n_in, n_out, opts, budget_s):
    """Oracle (restated reference, single thread) on a bounded sample: state + pair-table creation and the first greedy
    iterations of the seed-0 chain, scaled to a full chain with the calibration recorded in tests/golden/."""
    from oracle.oracle import HERE, Oracle, sample_chain

    # the reference's own sources (built against the container shim, oracle/README.md) when the prebuilt library travelled
    # with the snapshot, else the restated port (same algorithm, op-for-op identical results)
    kind = 'reference' if (HERE / '_ref' / 'libref.so').exists() else 'port'
    k = make_batch(n_in, n_out, 1, 0)[0]
    s = sample_chain(Oracle('ref' if kind == 'reference' else 'port'), k, opts.get('method0', 'wmc'), budget_s)
    cal_path = ROOT / 'tests' / 'golden' / 'cpu_calibration.json'
    cal = json.loads(cal_path.read_text()) if cal_path.exists() else {}
    key = f'{n_in}x{n_out}'
    sample = f'seed-0 {key} chain: state+table build ({s["create_s"]:.1f} s) + first {s["iterations"]} greedy iterations ({s["iter_s"]:.1f} s), 1 thread'
    if s['finished']:
        total = s['create_s'] + s['iter_s']
        sample += '; chain finished inside the budget'
    elif key in cal:
        # same-prefix scaling: full-run time / time of the same first iterations, both measured once on the build container
        c = cal[key]
        its = np.array(c['iter_marks'], dtype=np.float64)
        tms = np.array(c['time_marks_s'], dtype=np.float64)
        t_prefix_cal = float(np.interp(s['iterations'], its, tms))
        scale = (c['total_s'] - c['create_s']) / max(t_prefix_cal - c['create_s'], 1e-9)
        total = s['create_s'] + s['iter_s'] * scale
        sample += f'; iteration time scaled x{scale:.1f} to all {c["iterations"]} iterations with tests/golden/cpu_calibration.json'
    else:
        total = float('nan')
        sample += '; no calibration available to scale to a full chain'
    return {'value': (1.0 / total) if total == total else None, 'unit': 'solves/s', 'cores': 1, 'kind': kind, 'sample': sample,
            'est_seconds_per_solve': total if total == total else None}  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c3_256x256_int8_batch64_single_chain', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='override the per-GPU batch size')
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the CPU baseline sample (0 = skip)')
    args = ap.parse_args()

    import torch

    from da4ml_amd import _binary as hip
    from da4ml_amd import multi_gpu as mg

    rank, world, local, device = mg.init()
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if hip.device_count() < 1:
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    hip.set_device(local % hip.device_count())

    n_in, n_out, batch, opts = WORKLOADS[args.workload]
    batch = args.batch or batch
    kernels = make_batch(n_in, n_out, batch, first_seed=rank * batch)

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        mg.barrier()

    for _ in range(args.warmup):
        hip.solve_many_raw(kernels, **opts).free()
    hip.timings(reset=True)
    sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        if last is not None:
            last.free()
        last = hip.solve_many_raw(kernels, **opts)
    sync()
    elapsed = time.perf_counter() - t0
    tm = hip.timings(reset=True)
    elapsed = mg.max_over_ranks(elapsed, device if device.type == 'cuda' else None)
    total_solves = mg.sum_over_ranks(batch * args.steps, device if device.type == 'cuda' else None)
    mg.shutdown()  # last collective is done: every rank leaves the group here, rank 0 goes on alone

    # ---- correctness anchor outside the timed region: seed-0 result against the committed oracle digest
    check = None
    if rank == 0:
        summ = last.summary(0)
        records = {}
        for name in ('large_chain_golden.json', 'large_default_golden.json'):  # oracle digests of the hours-long CPU runs
            gold_path = ROOT / 'tests' / 'golden' / name
            if gold_path.exists():
                records.update(json.loads(gold_path.read_text()))
        gold = records.get(f'{n_in}x{n_out}_seed0_{"default" if not opts else "single_chain"}')
        check = {'seed0': summ, 'oracle': gold, 'adders_match_oracle': (gold is not None and gold['adders'] == summ['adders'] and gold['n_ops'] == summ['n_ops']) if gold else None}
    last.free()

    if rank != 0:
        return
    samples = max(tm['samples'], 1.0)
    upd_avg_us = 1e3 * tm['update_ms_sampled'] / samples
    sel_avg_us = 1e3 * tm['select_ms_sampled'] / samples
    chains_per_launch = tm['sampled_chain_launches'] / samples  # the batch runs as a few chain groups on separate streams
    # algorithmic bytes of k_iter_update: per partner row its cells in the substituted columns
    # and two 8-byte pair keys; per touched count block its interval record, rank and K u16 counts read + written;
    # per created block key + record + rank + index + K counts + the partner's interval.  Counted on the device,
    # summed over all chains and iterations; one launch covers `chains_per_launch` chains for one iteration.
    K = 2 * (2 * 8 - 1)
    alg_bytes = tm['cell_bytes'] + 16.0 * tm['partners'] + tm['found'] * (12.0 + 4.0 * K) + tm['inserts'] * (37.0 + 2.0 * K)
    alg_per_chain_iter = alg_bytes / max(tm['iterations'], 1.0)
    alg_per_launch = alg_per_chain_iter * chains_per_launch
    achieved = alg_per_launch / (upd_avg_us * 1e-6) / 1e9 if upd_avg_us > 0 else 0.0
    per_chain = pmc_traffic_per_chain()
    traffic = per_chain * chains_per_launch if per_chain else None
    launches = tm['lockstep_iters'] * max(1.0, round(batch / max(chains_per_launch, 1.0)))
    line = {
        'metric': 'CMVM solves/sec, 256x256 int8 matrix' if n_in == 256 else f'CMVM solves/sec, {n_in}x{n_out} int8 matrix',
        'value': total_solves / elapsed,
        'unit': 'solves/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u32',
        'data': 'synthetic',
        'config': {'workload': args.workload, 'matrix': f'{n_in}x{n_out} int8 (default_rng(seed).integers(-128,128))', 'batch_per_gpu': batch,
                   'solve_options': opts, 'parallelism': f'{world} x independent-instance shard, no data-path collective'},  # fmt: skip
        'roofline': {'kernel': 'k_iter_update', 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic, 'traffic_source': 'profiles/ PMC passes (FETCH_SIZE + WRITE_SIZE per chain) x chains per launch' if traffic else None,
                     'alg_bytes_per_launch': alg_per_launch, 'chains_per_launch': chains_per_launch, 'avg_launch_us': upd_avg_us, 'launches': launches,
                     'select_avg_launch_us': sel_avg_us, 'note': 'latency-bound random access into an HBM/L2-resident pair table; see DESIGN.md section 5'},  # fmt: skip
        'engine': {'greedy_loop_ms_per_step': tm['loop_ms'] / args.steps, 'library_ms_per_step': tm['total_ms'] / args.steps,
                   'greedy_iterations_per_step': tm['iterations'] / args.steps, 'lockstep_iterations_per_step': tm['lockstep_iters'] / args.steps,
                   'partner_rows_per_step': tm['partners'] / args.steps, 'arena_GB': tm['arena_bytes'] / 1e9},  # fmt: skip
        'check': check,
    }
    if args.cpu_seconds > 0 and world == 1:
        try:
            line['cpu_baseline'] = cpu_baseline(n_in, n_out, opts, args.cpu_seconds)
        except Exception as e:  # the baseline leg must never break the benchmark line
            line['cpu_baseline'] = {'value': None, 'unit': 'solves/s', 'cores': 1, 'kind': 'port', 'sample': f'failed: {e}'}
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
