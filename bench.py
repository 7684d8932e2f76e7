#!/usr/bin/env python
"""Benchmark of the CMVM solve path on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[2], the one the metric is quoted on): a batch of 64 independent 256x256 int8 constant
matrices ("64 restarts, one per CU block"), each solved as one greedy CSE chain
    solve(k, method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
(stage 2a on the centred matrix + the identity stage 2b + adder trees).  One "step" = one
pass over one such batch through the C ABI (da_solve_batch), results left on the C side.  With N GPUs every rank
solves its own 64 matrices (weak scaling, no data-path collective); value = matrices solved by all ranks / max-over-
ranks time.  Synthetic data: default_rng(seed).integers(-128, 128).

`python bench.py --gpus N` run directly (no torchrun environment) starts the N ranks itself, one process per GPU, and
relays rank 0's line; under `python -m torch.distributed.run` it is one of the ranks.

Extra objects on the JSON line:
  roofline      the greedy loop's two kernels, dominant one (by live launch time) on top and both under `kernels`: algorithmic
                bytes per launch (device-counted) / average launch duration measured with HIP events on
                the launch stream inside the library; `traffic` (FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for gfx950 +
                WRITE_SIZE) and `valu` from the committed rocprofv3 PMC passes -- ONLY when those passes were taken on the very
                sources that are loaded now (profiles/rNN_pmc_meta.json carries their digest), otherwise null + `stale_profiles`.
                The update kernel's share of the byte model (blocks it finds / creates) is counted by ONE pass of the same matrices
                outside the timed region, run under DA4ML_HIP_STATS=1: those tallies are instrumentation of this benchmark, touch no
                result, and cost 4 % of a step -- the product, and therefore the timed steps, run the kernel without them
  cpu_baseline  the reference's own sources (oracle/_ref/libref.so) on the host cores, one single-threaded solver process per
                core on its own matrix: a bounded prefix of every 256x256 chain scaled to full chains with the time curve of a
                complete run (labelled extrapolated), plus two fully MEASURED pairs through the same pool with the GPU timed on
                the same matrices in the same run: 64x64 and 128x128 complete chains
  check.verify  outside the timed region: every result of the last step replayed (own numpy replay of the C arrays) and
                compared with its matrix; digests of every seed that has a committed record (reference build preferred)

Other workloads (`--workload`): the 8-matrix default search, the 64x64 batch, `c5_model_batch` = the end-to-end
compile of a synthetic layer stack through `solve_many_sharded` (BASELINE configs[4]) with a CPU process-pool baseline, and
BASELINE configs[3] -- ONE 256x256 solve over the N GPUs, strong scaling -- in its two layouts: `c4_256x256_int8_column_sharded`
(one chain over the output columns, two RCCL all-reduces per greedy step) and `c4_256x256_int8_candidate_sharded` (the
default search's ten candidates over the ranks).
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SINGLE_CHAIN = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
WORKLOADS = {
    # name: (n_in, n_out, batch, solve options)
    'c3_256x256_int8_batch64_single_chain': (256, 256, 64, SINGLE_CHAIN),
    'c3_256x256_int8_batch8_default_search': (256, 256, 8, dict()),
    'c2_64x64_int8_batch64_single_chain': (64, 64, 64, SINGLE_CHAIN),
    'c5_model_batch': None,  # see run_c5
    # BASELINE configs[3]: ONE 256x256 chain sharded over the ranks' GPUs (see run_c4) -- by output columns with two all-reduce(sum) per greedy
    # step (the layout the config names; latency-bound), or the ten decompose_dc candidates of one searching solve over the ranks
    'c4_256x256_int8_column_sharded': (256, 256, 1, SINGLE_CHAIN),
    'c4_256x256_int8_candidate_sharded': (256, 256, 1, dict()),
    # plumbing check of the multi-rank path (tests/test_bench_contract.py runs it at 8 ranks on the emulated device): not a benchmark
    't0_plumbing_12x10_batch3_single_chain': (12, 10, 3, SINGLE_CHAIN),
    't0_plumbing_12x10_column_sharded': (12, 10, 1, SINGLE_CHAIN),
    't0_plumbing_12x10_candidate_sharded': (12, 10, 1, dict()),
}
# BASELINE configs[4] stand-in (JEDI-linear's weights are not in the reference tree and there is no network): a documented
# synthetic layer stack; every layer is applied to C5_ROWS row vectors with their own input intervals / latencies, which is
# the tracer's loop over the rows of the left operand (reference trace/fixed_variable_array.py:366-371) -- one solve per row
C5_LAYERS = [(16, 64), (64, 64), (64, 64), (64, 32), (32, 8)]
C5_ROWS = 8
C5_OPTS = dict(adder_size=1, carry_size=-1)  # the tracer's cost model (reference trace/fixed_variable.py:35)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12  # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T 32-bit lane-ops/s


# ------------------------------------------------------------------------------------------------ launching the ranks
def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """`bench.py --gpus N` without a torchrun environment: start N ranks of this script (one process per GPU, rendezvous on
    127.0.0.1), relay rank 0's output, return the first non-zero exit code."""
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))  # fmt: skip
    rc = 0
    for r, p in enumerate(procs):  # stderr of every rank is inherited (visible); name the rank that failed
        code = p.wait()
        if code != 0:
            print(f'bench.py: rank {r} of {n} exited with code {code}', file=sys.stderr, flush=True)
        rc = rc or code
    return rc


# ------------------------------------------------------------------------------------------------ PMC-derived figures
def source_digest() -> str:
    """sha256 over what the greedy loop's KERNELS are compiled from -- the kernel source, the scalar code it shares with the host
    and the compile flags: what a set of profiles must have been taken on to describe the loaded kernels
    (tools/collect_profiles.sh stores it in profiles/rNN_pmc_meta.json).  Host-side files (C ABI, orchestration, transports) do
    not enter: they do not change a kernel."""
    import hashlib

    h = hashlib.sha256()
    csrc = ROOT / 'da4ml_amd' / 'csrc'
    for name in ('cmvm_engine.hip', 'cmvm_core.h'):
        h.update(name.encode())
        h.update((csrc / name).read_bytes())
    for line in (csrc / 'Makefile').read_text().splitlines():
        if line.split('?=')[0].strip() in ('ARCH', 'KERNARG', 'CXXFLAGS'):
            h.update(line.encode())
    return h.hexdigest()


def newest_pmc_meta():
    metas = sorted((ROOT / 'profiles').glob('r*_pmc_meta.json'), reverse=True)
    if not metas:
        return None, None
    return metas[0].name.split('_pmc_meta')[0], json.loads(metas[0].read_text())


def pmc_per_dispatch(counter: str, kernel: str, tag: str):
    """per-dispatch mean of a counter of `kernel` from the committed rocprofv3 PMC summaries profiles/<tag>_pmc_*.txt
    (tools/collect_profiles.sh + tools/summarise_pmc.py), or None"""
    import re

    for path in sorted((ROOT / 'profiles').glob(f'{tag}_pmc_*.txt')):
        text = path.read_text()
        m = re.search(re.escape(kernel) + r'[^\n]*\n((?:    [^\n]*\n)+)', text)
        if not m:
            continue
        c = re.search(r'^\s+' + re.escape(counter) + r'\s+sum=\S+\s+per_dispatch=([0-9.eE+-]+)', m.group(1), re.M)
        if c:
            return float(c.group(1))
    return None


def pmc_traffic(kernel: str, tag: str):
    """HBM bytes per dispatch of `kernel`: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
    requests of the L2's fabric side at 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as
    reported (uncalibrated there).  Returns (corrected bytes, raw bytes) or None."""
    f, w = pmc_per_dispatch('FETCH_SIZE', kernel, tag), pmc_per_dispatch('WRITE_SIZE', kernel, tag)
    if f is None or w is None:
        return None
    return (2.0 * f + w) * 1024.0, (f + w) * 1024.0


def make_batch(n_in, n_out, batch, first_seed):
    return [np.random.default_rng(first_seed + i).integers(-128, 128, (n_in, n_out)).astype(np.float32) for i in range(batch)]


# ------------------------------------------------------------------------------------------------ verification
def replay_stage(n_in, inp_shifts, out_idxs, out_shifts, out_negs, ops_i):
    """Matrix implemented by one stage, from the C arrays alone: own level-by-level numpy replay of
    buf[i] = buf[id0] +- 2**data * buf[id1] on the n_in unit vectors (the reference's functional criterion
    `sol.kernel == kernel`, tests/test_cmvm.py:55).  float64 is exact here: all values are integers below 2**53."""
    n_ops = len(ops_i)
    id0, id1, code, data = (ops_i[:, c] for c in range(4))
    level = [0] * n_ops
    l0, l1, cd = id0.tolist(), id1.tolist(), code.tolist()
    for i in range(n_ops):
        level[i] = 0 if cd[i] == -1 else 1 + max(level[l0[i]], level[l1[i]])
    level = np.asarray(level)
    buf = np.zeros((n_ops, n_in))
    inputs = np.nonzero(code == -1)[0]
    buf[inputs, id0[inputs]] = 2.0 ** inp_shifts[id0[inputs]].astype(np.float64)
    for lv in range(1, int(level.max(initial=0)) + 1):
        idx = np.nonzero(level == lv)[0]
        assert np.all((code[idx] == 0) | (code[idx] == 1)), 'solver output holds only input / add / subtract statements'
        sign = np.where(code[idx] == 1, -1.0, 1.0) * 2.0 ** data[idx].astype(np.float64)
        buf[idx] = buf[id0[idx]] + sign[:, None] * buf[id1[idx]]
    scale = 2.0 ** out_shifts.astype(np.float64) * np.where(out_negs != 0, -1.0, 1.0) * (out_idxs >= 0)
    return (buf[np.where(out_idxs < 0, 0, out_idxs)] * scale[:, None]).T  # [n_in, n_out]


def result_matrix(hip, raw, i):
    L, h = hip.lib(), raw.handles[i]
    mat = None
    for s in range(L.da_n_stages(h)):
        info = np.zeros(5, np.int64)
        L.da_stage_info(h, s, info)
        n_in, n_out, n_ops = int(info[0]), int(info[1]), int(info[2])
        a = [np.zeros(k, np.int64) for k in (n_in, n_out, n_out, n_out)]
        oi, of = np.zeros((n_ops, 4), np.int64), np.zeros((n_ops, 5), np.float32)
        L.da_stage_copy(h, s, *a, oi, of)
        stage = replay_stage(n_in, *a, oi)
        mat = stage if mat is None else mat @ stage
    return mat


def verify(hip, raw, kernels, n_in, n_out, opts, first_seed):
    """all results of the last timed step against their matrices; digests of the first seeds against the oracle records"""
    import hashlib

    t = time.perf_counter()
    bad = [i for i, k in enumerate(kernels) if not np.array_equal(result_matrix(hip, raw, i), k.astype(np.float64))]
    out = {'kernel_reproduced': len(kernels) - len(bad), 'of': len(kernels), 'failed_seeds': [first_seed + i for i in bad]}
    records = {}
    for name in ('large_chain_golden.json', 'large_default_golden.json'):  # oracle digests of the hours-long CPU runs
        path = ROOT / 'tests' / 'golden' / name
        if path.exists():
            records.update(json.loads(path.read_text()))
    kind = 'default' if not opts else 'single_chain'
    digests, marshal_s = {}, []
    for i in range(len(kernels)):
        key = f'{n_in}x{n_out}_seed{first_seed + i}_{kind}'
        gold = records.get(key + '_ref') or records.get(key)  # the reference build's record where there is one
        if gold is None:
            continue
        t_obj = time.perf_counter()
        p = raw.pipeline(i)  # consumes the handle; builds the Python Pipeline / CombLogic / Op objects (reference: bindings.cc:106-151)
        marshal_s.append(time.perf_counter() - t_obj)
        dump = json.loads(json.dumps(p, default=lambda o: o.to_dict()))
        sha = hashlib.sha256(json.dumps(dump, separators=(',', ':')).encode()).hexdigest()
        digests[f'seed{first_seed + i}'] = {'match': sha == gold['sha256'], 'cost': p.cost, 'adders': p.n_adders, 'oracle_adders': gold['adders'],
                                            'oracle': gold.get('oracle', 'oracle/liboracle.so')}  # fmt: skip
    out['digests_vs_oracle'] = digests
    out['digests_checked'], out['digests_of'] = len(digests), len(kernels)
    out['digests_from_reference_build'] = sum(1 for d in digests.values() if d['oracle'].endswith('libref.so'))
    # Python object construction is reported separately from the timed C-ABI call
    out['python_objects_seconds_per_result'] = float(np.mean(marshal_s)) if marshal_s else None
    out['all_ok'] = not bad and all(d['match'] for d in digests.values())
    out['seconds'] = time.perf_counter() - t
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(n_in, n_out, opts, budget_s, batch, gpu_time_batch=None):
    """The reference CPU path on the host cores (oracle/cpu_pool.py: one single-threaded solver process per core, each on its
    own matrix -- the CPU's best case for independent matrices; the reference itself only threads over <= 10 candidates).
    `gpu_time_batch(kernels, opts)` -> seconds of one GPU batch call on the given matrices (for the measured pairs)."""
    from oracle import cpu_pool
    from oracle.oracle import HERE

    kind = 'reference' if (HERE / '_ref' / 'libref.so').exists() else 'port'  # the reference's own sources when the build travelled
    okind = 'ref' if kind == 'reference' else 'port'
    cores = cpu_pool.host_cores()
    # one 256x256 chain of the reference holds its 64.7 M initial pairs (24 bytes each) plus the sort buffer: ~4 GB per process
    need_gb = 4.0 * (n_in * n_out / 65536.0) ** 2 + 0.5
    # one process per chain of the batch, never more than half of the available memory allows at 1.5 x the estimate
    # Capped at the batch size (64).  Round 4 tried "every host core the memory allows" again (230 single-threaded processes on the
    # 256-core box, 1.5 x 4.5 GB each = half of the available memory by the estimate above): the box was lost within the first
    # 100 seconds of the call, as in round 2 -- so the pool stays at one process per chain of the batch, and the line carries the
    # complete runs of the build container beside the extrapolated figure (measured_full_runs_build_container).
    workers = max(1, min(cores, batch, int(0.5 * cpu_pool.mem_available_gb() / (1.5 * need_gb))))
    method = opts.get('method0', 'wmc')
    # (1) a time-bounded prefix of a chain of the batch on every core, all cores busy at once
    samples, wall = cpu_pool.run_pool(cpu_pool.sample_worker, [(okind, n_in, n_out, i % batch, method, budget_s) for i in range(workers)], workers)
    cal_path = ROOT / 'tests' / 'golden' / 'cpu_calibration.json'
    cal = (json.loads(cal_path.read_text()) if cal_path.exists() else {}).get(f'{n_in}x{n_out}')
    est = []
    for smp in samples:
        if smp['finished']:
            est.append(smp['create_s'] + smp['iter_s'])
        elif cal:
            # same-prefix scaling: (full-chain time) / (time of the same first iterations) from the complete run of the same code
            its, tms = np.asarray(cal['iter_marks'], np.float64), np.asarray(cal['time_marks_s'], np.float64)
            t_prefix = float(np.interp(smp['iterations'], its, tms)) - cal['create_s']
            est.append(smp['create_s'] + smp['iter_s'] * (cal['total_s'] - cal['create_s']) / max(t_prefix, 1e-9))
    finished = all(smp['finished'] for smp in samples)
    out = {'value': float(sum(1.0 / e for e in est)) if est else None, 'unit': 'solves/s', 'cores': workers, 'host_cores': cores, 'kind': kind,
           'extrapolated': not finished,
           'sample': f'{workers} processes x 1 thread, each a seed-i {n_in}x{n_out} chain: state + pair table build (mean {np.mean([smp["create_s"] for smp in samples]):.1f} s) + '
                     f'first {int(np.mean([smp["iterations"] for smp in samples]))} greedy iterations in {budget_s:.0f} s (wall {wall:.1f} s)'
                     + ('' if finished else f'; scaled to full chains with the time curve of a complete run of the same code ({cal["source"] if cal else "no calibration"})'),
           'est_seconds_per_solve_one_core': float(np.mean(est)) if est else None}  # fmt: skip
    # (1b) complete runs of the same code at this very size, not extrapolated: the records of the benchmark matrices were made by
    # oracle/_ref/libref.so in the build container (tests/golden/large_chain_golden.json keeps their wall times: one chain per core,
    # eight at a time on 8 cores).  A per-core figure, and what `cores` perfectly scaling cores would reach at that rate.
    try:
        recs = json.loads((ROOT / 'tests' / 'golden' / 'large_chain_golden.json').read_text())
        secs = [r['oracle_seconds'] for k, r in recs.items() if k.startswith(f'{n_in}x{n_out}_seed') and k.endswith('_single_chain_ref') and 'oracle_seconds' in r]
        if secs:
            out['measured_full_runs_build_container'] = {
                'runs': len(secs), 'seconds_per_solve_one_core': {'mean': float(np.mean(secs)), 'min': float(np.min(secs)), 'max': float(np.max(secs))},
                'solves_per_s_per_core': float(1.0 / np.mean(secs)), 'solves_per_s_if_all_host_cores_scaled_perfectly': float(cores / np.mean(secs)),
                'sample': f'{len(secs)} complete {n_in}x{n_out} single-chain runs of the reference build (seeds of the benchmark batch), one core each, measured in the build container -- not on this box, not extrapolated'}  # fmt: skip
    except (OSError, ValueError, KeyError):
        pass
    # (2) fully measured, no scaling, GPU and CPU on the SAME matrices in the same run: complete 64x64 and 128x128 chains
    for n, key in ((64, 'measured_64x64'), (128, 'measured_128x128')):
        count = batch if n == 64 or cores >= batch else max(1, min(batch, cores))  # 128x128: ~60-90 s per chain and core -- one wave of the pool
        w = max(1, min(cores, count))
        res, wall_n = cpu_pool.run_pool(cpu_pool.solve_worker, [(okind, n, n, seed, SINGLE_CHAIN) for seed in range(count)], w)
        rec = {'value': count / wall_n, 'unit': 'solves/s', 'cores': w, 'wall_s': wall_n, 'one_core_seconds_per_solve': float(np.mean([r[0] for r in res])),
               'sample': f'{count} complete {n}x{n} int8 single-chain solves (seeds 0..{count - 1}), process pool of {w}, {kind} build'}  # fmt: skip
        if gpu_time_batch is not None:
            ks = make_batch(n, n, count, 0)
            gpu_time_batch(ks, SINGLE_CHAIN)  # warm-up (arena growth)
            g = min(gpu_time_batch(ks, SINGLE_CHAIN) for _ in range(3))
            rec['gpu'] = {'value': count / g, 'unit': 'solves/s', 'seconds_per_batch': g, 'sample': f'the same {count} matrices in one da_solve_batch call, best of 3'}
            rec['gpu_over_cpu'] = rec['gpu']['value'] / rec['value']
        out[key] = rec
    return out


# ------------------------------------------------------------------------------------------------ C5: model compile
def c5_problems():
    """(kernels, qintervals, latencies) of the synthetic model: layer l is solved once per row vector r"""
    ks, qs, ls = [], [], []
    for li, (a, b) in enumerate(C5_LAYERS):
        k = np.random.default_rng(100 + li).integers(-128, 128, (a, b)).astype(np.float32)
        rng = np.random.default_rng(200 + li)
        for r in range(C5_ROWS):
            bits = rng.integers(4, 9, a)  # per-input fixed-point widths 4..8, fractional bits 0..3
            frac = rng.integers(0, 4, a)
            step = 2.0 ** -frac.astype(np.float64)
            ks.append(k)
            qs.append([(float(-(2.0 ** (w - 1)) * s), float((2.0 ** (w - 1) - 1) * s), float(s)) for w, s in zip(bits, step)])
            ls.append([float(v) for v in rng.integers(0, 3, a)])
    return ks, qs, ls


def run_c5(args):
    """End-to-end compile time of the synthetic layer stack: every (layer, row vector) solve through solve_many_sharded
    (cost-balanced over the ranks), Python Pipeline objects included; CPU process pool beside it."""
    import torch

    from da4ml_amd import _binary as hip
    from da4ml_amd import multi_gpu as mg

    rank, world, local, device = mg.init()
    if hip.device_count() < 1:
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    hip.set_device(local % hip.device_count())
    ks, qs, ls = c5_problems()

    def step():
        return mg.solve_many_sharded(ks, qintervals=qs, latencies=ls, **C5_OPTS)

    for _ in range(args.warmup):
        step()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    mg.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    mg.barrier()
    elapsed = mg.max_over_ranks(time.perf_counter() - t0, device if device.type == 'cuda' else None)
    mg.shutdown()
    if rank != 0:
        return
    ok = all(np.array_equal(p.kernel, k) for p, k in zip(res, ks))
    line = {'metric': 'CMVM model compile time, synthetic 5-layer stack x 8 row vectors (BASELINE configs[4] stand-in)', 'value': elapsed / args.steps, 'unit': 's',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': False, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': 'c5_model_batch', 'layers': C5_LAYERS, 'row_vectors_per_layer': C5_ROWS, 'solves': len(ks), 'solve_options': C5_OPTS,
                       'parallelism': f'{world} x cost-balanced instance shard (solve_many_sharded), one result gather'},
            'includes': 'per-solve qintervals/latencies, all decompose_dc candidates, Python Pipeline construction and the gather to rank 0',
            'check': {'kernel_reproduced': ok, 'total_cost': float(sum(p.cost for p in res))}}  # fmt: skip
    if args.cpu_seconds > 0:
        from oracle import cpu_pool
        from oracle.oracle import HERE

        okind = 'ref' if (HERE / '_ref' / 'libref.so').exists() else 'port'
        workers = max(1, min(cpu_pool.host_cores(), len(ks)))
        jobs = [(okind, k, dict(C5_OPTS, qintervals=q, latencies=l)) for k, q, l in zip(ks, qs, ls)]
        out, wall = cpu_pool.run_pool(cpu_pool.problem_worker, jobs, workers)
        line['cpu_baseline'] = {'value': wall, 'unit': 's', 'cores': workers, 'kind': 'reference' if okind == 'ref' else 'port',
                                'sample': f'the same {len(ks)} solves, one single-threaded solver process per core (pool of {workers}), C call only, complete (not scaled)',
                                'sum_of_solve_seconds': float(sum(o[0] for o in out)), 'total_cost': float(sum(o[1] for o in out))}  # fmt: skip
        line['check']['cost_equals_cpu'] = line['check']['total_cost'] == line['cpu_baseline']['total_cost']
    print(json.dumps(line), flush=True)



# ------------------------------------------------------------------------------------------------ C4: one solve over N GPUs
def run_c4(args):
    """BASELINE configs[3]: ONE matrix (seed 0), one solve per step, sharded over the ranks -- strong scaling.
    `*_column_sharded`: the greedy chain sharded over the output columns (mg.solve_column_sharded; the library's RCCL transport on GPUs:
    ncclAllReduce stream-ordered with the kernels, two per greedy step; with one rank the sharded phases and the collectives run all the same).
    `*_candidate_sharded`: the decompose_dc candidates of the default search over the ranks (mg.solve_candidates_sharded: one all-reduce(MIN) +
    one broadcast per solve).  Every rank returns the same Pipeline; rank 0 compares it with its own unsharded solve and with the
    reference build's record of the matrix, outside the timed region."""
    import hashlib

    import torch

    from da4ml_amd import _binary as hip
    from da4ml_amd import multi_gpu as mg

    rank, world, local, device = mg.init()
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if hip.device_count() < 1:
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    hip.set_device(local % hip.device_count())
    n_in, n_out, _, opts = WORKLOADS[args.workload]
    k = make_batch(n_in, n_out, 1, 0)[0]
    column = args.workload.endswith('column_sharded')
    on_gpus = device.type == 'cuda'
    if column and world == 1:  # the sharded phases and their collectives also with a single rank: what every rank pays before any xGMI latency
        os.environ['DA4ML_SHARD_FORCE'] = '1'
        os.environ['DA4ML_SHARD_FORCE_COMM'] = '1'
    transport = 'rccl' if on_gpus else 'callback'  # (gloo ranks of the CPU tests: torch.distributed called back from the library)

    def step():
        if column:
            return mg.solve_column_sharded(k, transport=transport, return_stats=True, **opts)
        return mg.solve_candidates_sharded(k), {}

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        mg.barrier()

    for _ in range(args.warmup):  # (communicator set-up, arena growth)
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe, stats = step()
    sync()
    elapsed = mg.max_over_ranks(time.perf_counter() - t0, device if on_gpus else None)
    exchanged = hip.shard_exchanged_elements() if column else 0
    mg.shutdown()
    if rank != 0:
        return
    check = {'kernel_reproduced': bool(np.array_equal(pipe.kernel, k)), 'cost': float(pipe.cost), 'adders': int(pipe.n_adders)}
    if not args.no_verify:
        os.environ.pop('DA4ML_SHARD_FORCE', None)
        os.environ.pop('DA4ML_SHARD_FORCE_COMM', None)
        check['equals_unsharded_solve_on_rank0'] = bool(pipe == hip.solve(k, **opts))
        rec_path = ROOT / 'tests' / 'golden' / ('large_chain_golden.json' if opts else 'large_default_golden.json')
        key = f'{n_in}x{n_out}_seed0_' + ('single_chain' if opts else 'default')
        recs = json.loads(rec_path.read_text()) if rec_path.exists() else {}
        gold = recs.get(key + '_ref') or recs.get(key)
        if gold:
            dump = json.loads(json.dumps(pipe, default=lambda o: o.to_dict()))
            check['digest_matches_record'] = hashlib.sha256(json.dumps(dump, separators=(',', ':')).encode()).hexdigest() == gold['sha256']
            check['record_from'] = gold.get('oracle', 'oracle/liboracle.so')
    per = elapsed / args.steps
    line = {'metric': f'CMVM solves/sec, {n_in}x{n_out} int8 matrix' + (', one chain column-sharded over the GPUs' if column else ', default search with its candidates over the GPUs'),
            'value': args.steps / elapsed, 'unit': 'solves/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * per,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': args.workload, 'matrix': f'{n_in}x{n_out} int8 (default_rng(0).integers(-128,128))', 'solve_options': opts,
                       'parallelism': (f'{world} x column shard of one greedy chain, pair table replicated, two all-reduce(sum) per greedy step over '
                                       + ('RCCL (library transport, stream-ordered)' if on_gpus else 'the process group (callback transport)')) if column
                                      else f'{world} x candidate shard (candidate i on rank i mod {world}), one all-reduce(MIN) and one broadcast per solve'},
            'check': check}  # fmt: skip
    if column:
        gs = max(int(stats.get('greedy_steps', 0)), 1)
        line['engine'] = {'greedy_steps': int(stats.get('greedy_steps', 0)), 'us_per_greedy_step': 1e6 * per / gs, 'allreduce_calls_per_solve': int(stats.get('allreduce_calls', 0)),
                          'exchanged_bytes_per_greedy_step_and_rank': 4.0 * exchanged / gs, 'transport': transport,
                          'note': 'latency-bound by construction: two collectives per greedy step with the union size read by the host in between (DESIGN.md section 7); the layout that scales is the instance shard of the default workload'}
    try:  # RCCL writes a version banner through the C library's buffered stdout at communicator set-up: out first, so that the JSON line is the LAST line
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------------------ main
def selftest_launcher():
    """CPU-runnable check of the rank launcher and the collectives bench.py uses (no GPU work): every rank joins the group
    (nccl on a GPU host, gloo otherwise), reduces a fake time and count, rank 0 prints the contract's n_gpus."""
    from da4ml_amd import multi_gpu as mg

    rank, world, local, device = mg.init()
    dev = device if device.type == 'cuda' else None
    mg.barrier()
    elapsed = mg.max_over_ranks(1.0 + rank, dev)
    total = mg.sum_over_ranks(64, dev)
    mg.shutdown()
    if rank == 0:
        print(json.dumps({'selftest': 'launcher', 'n_gpus': world, 'max_elapsed': elapsed, 'total_solves': total, 'backend_device': device.type}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c3_256x256_int8_batch64_single_chain', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='override the per-GPU batch size')
    ap.add_argument('--cpu-seconds', type=float, default=30.0, help='budget of the CPU baseline sample (0 = skip)')
    ap.add_argument('--no-verify', action='store_true', help='skip the replay of all results after the timed region')
    ap.add_argument('--selftest-launcher', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:  # started directly: become the launcher of the N ranks
        if not args.selftest_launcher:
            from da4ml_amd import _binary as hip

            if hip.device_count() < args.gpus:
                raise SystemExit(f'--gpus {args.gpus} but only {hip.device_count()} HIP device(s) are visible (one process per GPU, no oversubscription)')
        raise SystemExit(launch_ranks(args.gpus))
    if args.selftest_launcher:
        return selftest_launcher()
    if args.workload == 'c5_model_batch':
        return run_c5(args)
    if args.workload.endswith(('column_sharded', 'candidate_sharded')):
        return run_c4(args)

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

    # One COUNTED pass outside the timed region (DA4ML_HIP_STATS=1: k_iter_update tallies the blocks it finds / creates -- instrumentation that costs
    # 1.8 % of a step, so the product and the timed steps run without it): the same matrices, hence the same counts as every timed step.
    # Skipped with --no-verify (the profiler runs of tools/collect_profiles.sh: only the product's kernels in their traces).
    counted = None
    if not args.no_verify:
        os.environ['DA4ML_HIP_STATS'] = '1'
        hip.timings(reset=True)
        hip.solve_many_raw(kernels, **opts).free()
        counted = hip.timings(reset=True)
    os.environ['DA4ML_HIP_STATS'] = '0'
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

    if rank != 0:
        last.free()
        return
    # ---- correctness outside the timed region
    summ = last.summary(0)
    check = {'seed0': summ}
    if not args.no_verify:
        check['verify'] = verify(hip, last, kernels, n_in, n_out, opts, first_seed=0)
    last.free()

    samples = max(tm['samples'], 1.0)
    upd_avg_us = 1e3 * tm['update_ms_sampled'] / samples
    sel_avg_us = 1e3 * tm['select_ms_sampled'] / samples
    chains_per_launch = tm['sampled_chain_launches'] / samples  # the batch runs as a few chain groups on separate streams
    iters = max(tm['iterations'], 1.0)
    # Algorithmic bytes, counted on the device and summed over all chains and steps; one launch covers
    # `chains_per_launch` chains for one step.
    #   k_iter_update: per partner row its cells in the substituted columns and two 8-byte pair keys; per touched count block its
    #     interval record, rank and K u16 counts read + written; per created block key + record + rank + index + K counts + the
    #     partner's interval
    #   k_iter_select2 (two workgroups per chain: search + substitution): bound, flags, tie word and lowered-value mark of all groups,
    #     every re-read group, the pick (64 B), both row lists read and written back, the row bitmaps of the substituted columns,
    #     partner ids / references, the hand-off stores, the six special-pair count vectors (st_sel_bytes in the kernel + the host's
    #     pricing of the search, HipBackend::run_chains)
    K = 2 * (2 * 8 - 1)
    if counted and counted['iterations'] > 0:  # blocks found / created per greedy iteration, from the counted pass on the same matrices
        tm['found'] = counted['found'] / counted['iterations'] * tm['iterations']
        tm['inserts'] = counted['inserts'] / counted['iterations'] * tm['iterations']
    alg = {'k_iter_update': tm['cell_bytes'] + 16.0 * tm['partners'] + tm['found'] * (12.0 + 4.0 * K) + tm['inserts'] * (37.0 + 2.0 * K),
           'k_iter_select2': tm['select_bytes']}
    avg_us = {'k_iter_update': upd_avg_us, 'k_iter_select2': sel_avg_us}
    tag, meta = newest_pmc_meta()
    same_sources = bool(meta) and meta.get('src_sha256') == source_digest()
    # the PMC passes are taken on the headline workload (tools/collect_profiles.sh): their per-launch traffic says nothing about another one
    same_workload = bool(meta) and meta.get('workload', 'c3_256x256_int8_batch64_single_chain') == args.workload
    fresh = same_sources and same_workload
    launches = tm['lockstep_iters'] * max(1.0, round(batch / max(chains_per_launch, 1.0)))
    kernels = {}
    for name in ('k_iter_select2', 'k_iter_update'):
        per_launch = alg[name] / iters * chains_per_launch
        ach = per_launch / (avg_us[name] * 1e-6) / 1e9 if avg_us[name] > 0 else 0.0
        k = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'alg_bytes_per_launch': per_launch,
             'alg_bytes_per_chain_step': alg[name] / iters, 'avg_launch_us': avg_us[name], 'launches': launches, 'traffic': None}
        if fresh:
            tr = pmc_traffic(name, tag)
            scale = chains_per_launch / float(meta.get('chains_per_dispatch', chains_per_launch) or chains_per_launch)
            if tr:
                k['traffic'], k['traffic_uncorrected'] = tr[0] * scale, tr[1] * scale
                k['traffic_over_algorithmic'] = k['traffic'] / per_launch if per_launch else None
            insts = pmc_per_dispatch('SQ_INSTS_VALU', name, tag)
            if insts and avg_us[name] > 0:
                lane_ops = insts * scale * 64.0  # wave instructions x 64 lanes (upper bound: full exec mask)
                act, cyc = pmc_per_dispatch('SQ_ACTIVE_INST_VALU', name, tag), pmc_per_dispatch('SQ_WAVE_CYCLES', name, tag)
                k['valu'] = {'achieved': lane_ops / (avg_us[name] * 1e-6) / 1e12, 'peak': VALU_PEAK_TOPS, 'unit': 'T lane-ops/s',
                             'frac': lane_ops / (avg_us[name] * 1e-6) / 1e12 / VALU_PEAK_TOPS, 'valu_busy_of_wave_cycles': act / cyc if act and cyc else None}
        kernels[name] = k
    dominant = max(kernels, key=lambda n: avg_us[n])
    loop_s = max(tm['loop_ms'] * 1e-3, 1e-9)
    roofline = {'kernel': dominant, **{k: v for k, v in kernels[dominant].items() if k != 'valu'}, 'chains_per_launch': chains_per_launch,
                'kernels': kernels,
                'profiles': {'tag': tag, 'fresh': fresh, 'note': None if fresh else ('stale_profiles: the committed PMC passes were taken on other sources than the loaded library; traffic / valu withheld' if not same_sources else 'stale_profiles: the committed PMC passes were taken on another workload (' + str(meta.get('workload', 'c3_256x256_int8_batch64_single_chain')) + '); traffic / valu withheld')},
                'traffic_source': f'profiles/{tag}_pmc_*: (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch (gfx950 correction of MI355X_MICROARCH.md), scaled to the live chains per launch' if fresh else None,
                # the chain groups' launches overlap (4 streams): the algorithmic bytes of both kernels over the whole greedy loop
                'whole_loop': {'achieved': sum(alg.values()) / loop_s / 1e9, 'unit': 'GB/s', 'frac': sum(alg.values()) / loop_s / 1e9 / HBM_PEAK_GBS,
                               'what': 'algorithmic bytes of all k_iter_select2 + k_iter_update launches of the timed steps / their greedy-loop time (concurrent chain groups included)'},
                'note': 'bound by dependent memory round trips into an HBM-resident pair table and by wave slots, not by bytes or VALU; see DESIGN.md section 5'}  # fmt: skip
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
        'roofline': roofline,
        'engine': {'greedy_loop_ms_per_step': tm['loop_ms'] / args.steps, 'library_ms_per_step': tm['total_ms'] / args.steps,
                   'greedy_iterations_per_step': tm['iterations'] / args.steps, 'lockstep_iterations_per_step': tm['lockstep_iters'] / args.steps,
                   'partner_rows_per_step': tm['partners'] / args.steps, 'arena_GB': tm['arena_bytes'] / 1e9,
                   'us_per_lockstep_iteration': 1e3 * tm['loop_ms'] / max(tm['lockstep_iters'], 1.0),
                   # the launch thread: host time spent queueing the loop's launches vs the time the device needed for them
                   'host_launch_us_per_iter': 1e3 * tm['host_launch_ms'] / max(tm['lockstep_iters'], 1.0),
                   'host_launch_share_of_loop': tm['host_launch_ms'] / max(tm['loop_ms'], 1e-9),
                   # k_iter_select2: steps whose pick was known before the step began (the best entry the previous step left untouched, or one of
                   # the few entries its update listed): no bounds, no arg-max in front of the substitution
                   'picks_known_a_step_ahead': tm['fast_steps'] / max(tm['iterations'], 1.0),
                   'group_rereads_per_chain_step': tm['rescans'] / max(tm['iterations'], 1.0),
                   'blocks_found_created_from_counted_pass': bool(counted), 'source_digest': source_digest()},  # fmt: skip
        'check': check,
    }
    if args.cpu_seconds > 0 and world == 1:
        try:
            def gpu_time_batch(ks, o):
                t = time.perf_counter()
                hip.solve_many_raw(ks, **o).free()
                return time.perf_counter() - t

            line['cpu_baseline'] = cpu_baseline(n_in, n_out, opts, args.cpu_seconds, batch, gpu_time_batch)
        except Exception as e:  # the baseline leg must never break the benchmark line
            line['cpu_baseline'] = {'value': None, 'unit': 'solves/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {type(e).__name__}: {e}'}
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
