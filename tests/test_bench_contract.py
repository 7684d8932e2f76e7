"""bench.py's driver contract, as far as it can be checked without a GPU: the committed bench line of the last profiled
round carries every field the contract names, the roofline figures are self-consistent, the helper that reads the PMC
passes agrees with the committed summaries, and bench.py fails loudly without a HIP device."""

import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _last_bench_line():
    files = sorted((ROOT / 'profiles').glob('r*_bench.json'))
    assert files, 'no committed bench line under profiles/'
    return json.loads(files[-1].read_text().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    line = _last_bench_line()
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['unit'] == 'solves/s' and line['higher_is_better'] is True and line['scaling'] == 'weak' and line['data'] == 'synthetic'
    assert line['vs_baseline'] is None  # BASELINE.md holds no published number for this metric
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert abs(line['value'] - line['config']['batch_per_gpu'] * line['n_gpus'] / (line['ms_per_step'] * 1e-3)) < 1e-6 * line['value']
    r = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['achieved'] - r['alg_bytes_per_launch'] / (r['avg_launch_us'] * 1e-6) / 1e9) < 1e-6 * r['achieved']
    c = line['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in c, key
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['unit'] == line['unit']
    chk = line['check']
    if 'verify' in chk:  # round 2 on: every result of the last step replayed, digests of the first seeds against the oracle records
        v = chk['verify']
        assert v['all_ok'] is True and v['kernel_reproduced'] == v['of'] == line['config']['batch_per_gpu']
        assert len(v['digests_vs_oracle']) >= 4 and all(d['match'] for d in v['digests_vs_oracle'].values())
        assert c['cores'] > 1 or c.get('host_cores', 1) == 1  # all host cores, not one
        assert 'extrapolated' in c and c['measured_64x64']['value'] > 0
        assert r.get('valu') is None or 0 < r['valu']['frac'] < 1
    else:
        assert chk['adders_match_oracle'] is True


def test_pmc_traffic_helper_reads_the_committed_passes():
    """the committed PMC summaries are readable per kernel; a bench line of the two-kernel format (round 3 on) agrees with them
    when it says its profiles were fresh, and withholds traffic when it says they were stale"""
    sys.path.insert(0, str(ROOT))
    import bench

    tag, meta = bench.newest_pmc_meta()
    assert tag and meta and meta['chains_per_dispatch'] >= 1
    for kernel in ('k_iter_select2', 'k_iter_update'):
        tr = bench.pmc_traffic(kernel, tag)
        assert tr and tr[0] > tr[1] > 0  # corrected (2 x FETCH_SIZE + WRITE_SIZE) above the raw sum
    r = _last_bench_line()['roofline']
    if 'kernels' in r:
        assert set(r['kernels']) == {'k_iter_select2', 'k_iter_update'} and r['kernel'] in r['kernels']
        for name, k in r['kernels'].items():
            assert abs(k['frac'] - k['achieved'] / k['peak']) < 1e-12
            assert abs(k['achieved'] - k['alg_bytes_per_launch'] / (k['avg_launch_us'] * 1e-6) / 1e9) < 1e-6 * max(k['achieved'], 1e-9)
            if r['profiles']['fresh']:
                # within 1 %: same passes, scaled to the line's chains per launch
                want = bench.pmc_traffic(name, r['profiles']['tag'])[0] * r['chains_per_launch'] / meta['chains_per_dispatch']
                assert abs(want - k['traffic']) < 1e-2 * k['traffic']
            else:
                assert k['traffic'] is None and 'stale_profiles' in r['profiles']['note']
    assert len(bench.source_digest()) == 64
    assert 'c3_256x256_int8_batch64_single_chain' in bench.WORKLOADS and bench.WORKLOADS['c3_256x256_int8_batch64_single_chain'][:3] == (256, 256, 64)
    ks = bench.make_batch(4, 3, 2, first_seed=5)
    assert len(ks) == 2 and ks[0].shape == (4, 3) and ks[0].dtype.name == 'float32' and abs(ks[0]).max() <= 128


def test_bench_needs_a_gpu():
    from da4ml_amd import _binary

    if _binary.device_count() > 0:
        import pytest

        pytest.skip('a GPU is present')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'needs a HIP device' in (r.stdout + r.stderr)


def test_replay_used_by_the_bench_verification(oracle):
    """bench.py verifies all results of the timed batch with its own replay of the C arrays: it reproduces the matrix of
    oracle results (both stages), and notices a corrupted statement"""
    sys.path.insert(0, str(ROOT))
    import numpy as np

    import bench
    from cases import int_matrix

    def arrays(sol):
        ops_i = np.asarray([[op.id0, op.id1, op.opcode, op.data] for op in sol.ops], dtype=np.int64).reshape(-1, 4)
        return sol.shape[0], np.asarray(sol.inp_shifts, np.int64), np.asarray(sol.out_idxs, np.int64), np.asarray(sol.out_shifts, np.int64), np.asarray(sol.out_negs, np.int64), ops_i

    for seed, opts in ((3, {}), (4, dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)), (5, dict(decompose_dc=1))):
        k = int_matrix(seed, 14, 11, -64, 64)
        p = oracle.solve(k, **opts)
        m0, m1 = (bench.replay_stage(*arrays(s)) for s in p.solutions)
        assert np.array_equal(m0 @ m1, k.astype(np.float64))
        a = list(arrays(p.solutions[0]))
        adders = np.nonzero(a[5][:, 2] >= 0)[0]
        if len(adders):
            a[5][adders[-1], 3] += 1  # one wrong shift
            assert not np.array_equal(bench.replay_stage(*a) @ m1, k.astype(np.float64))


import pytest  # noqa: E402


@pytest.mark.parametrize('world', [2, 8])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` started directly (no torchrun environment) spawns the N ranks itself; here with the
    launcher self-test (gloo on this CPU host, nccl on a GPU host): rank 0 reports n_gpus N, the max over ranks of the
    per-rank times and the sum of the per-rank solves -- at 2 ranks and at the 8 of a full node (SCALE / BASELINE configs[3])"""
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', str(world), '--selftest-launcher'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == world and line['max_elapsed'] == float(world) and line['total_solves'] == 64.0 * world


def test_bench_runs_end_to_end_at_eight_ranks_on_the_emulated_device():
    """The REAL bench path at world 8 (not only the launcher self-test): `bench.py --gpus 8` spawns its ranks, every rank pins itself to
    its core slice, solves its own shard of a tiny workload through the product's Python layer, C ABI and kernels (the latter on the
    emulated device, tests/emu), the ranks reduce time and count over gloo and rank 0 verifies its results and prints the contract
    line.  What it cannot show is RCCL: ncclCommInitRank with more than one rank has never run (no multi-GPU box)."""
    import os

    emu = ROOT / 'tests' / 'emu' / 'libda4ml_emu.so'
    if not emu.exists():
        pytest.skip('tests/emu/libda4ml_emu.so is not built')
    env = dict(os.environ, DA4ML_HIP_LIB=str(emu), HIPEMU_DEVICES='8')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0', '--cpu-seconds', '0', '--workload', 't0_plumbing_12x10_batch3_single_chain'],
                       capture_output=True, text=True, timeout=900, env=env)  # fmt: skip
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['scaling'] == 'weak' and line['config']['batch_per_gpu'] == 3
    assert abs(line['value'] - 8 * 3 * line['steps'] / (line['ms_per_step'] * 1e-3 * line['steps'])) < 1e-6 * line['value']  # whole-job aggregate: all ranks' solves
    v = line['check']['verify']
    assert v['all_ok'] is True and v['kernel_reproduced'] == v['of'] == 3
    assert line['engine']['picks_known_a_step_ahead'] > 0.5


@pytest.mark.parametrize('workload,world', [('t0_plumbing_12x10_column_sharded', 8), ('t0_plumbing_12x10_column_sharded', 1), ('t0_plumbing_12x10_candidate_sharded', 8)])
def test_bench_c4_layouts_on_the_emulated_device(workload, world):
    """BASELINE configs[3] as bench workloads: ONE solve per step over the ranks, strong scaling -- the column-sharded chain
    (every rank its slice of the columns, two all-reduces per greedy step; with one rank the sharded phases and collectives forced) and the
    candidate-sharded default search -- at the config's own 8 ranks, kernels on the emulated device, exchanges over gloo.  Rank 0 checks the
    result against its own unsharded solve.  (What this cannot show: RCCL with more than one rank.)"""
    import os

    emu = ROOT / 'tests' / 'emu' / 'libda4ml_emu.so'
    if not emu.exists():
        pytest.skip('tests/emu/libda4ml_emu.so is not built')
    env = dict(os.environ, DA4ML_HIP_LIB=str(emu), HIPEMU_DEVICES='8', DA4ML_HIP_UPD_BLOCKS='8')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', str(world), '--steps', '1', '--warmup', '1', '--workload', workload], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == world and line['scaling'] == 'strong' and line['unit'] == 'solves/s' and line['config']['workload'] == workload
    assert abs(line['value'] - line['steps'] / (line['ms_per_step'] * 1e-3 * line['steps'])) < 1e-6 * line['value']  # one solve per step, whole job
    assert line['check']['kernel_reproduced'] is True and line['check']['equals_unsharded_solve_on_rank0'] is True
    if workload.endswith('column_sharded'):
        e = line['engine']
        assert e['greedy_steps'] > 0 and e['allreduce_calls_per_solve'] >= 2 * e['greedy_steps'] and e['exchanged_bytes_per_greedy_step_and_rank'] > 0


def test_bench_refuses_more_ranks_than_gpus():
    from da4ml_amd import _binary

    if _binary.device_count() >= 2:
        import pytest

        pytest.skip('two GPUs are present')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'HIP device' in (r.stdout + r.stderr)


def test_cpu_pool_times_complete_solves():
    """the process pool behind cpu_baseline: spawned single-threaded workers, results in job order, identical to a direct call"""
    from oracle import cpu_pool
    from oracle.oracle import Oracle

    jobs = [('port', 12, 9, seed, dict(adder_size=1, carry_size=-1)) for seed in range(3)]
    out, wall = cpu_pool.run_pool(cpu_pool.solve_worker, jobs, 2)
    assert wall > 0 and len(out) == 3
    for (dt, cost, n_ops), job in zip(out, jobs):
        p = Oracle('port').solve(cpu_pool._matrix(12, 9, job[3]), **job[4])
        assert dt > 0 and cost == p.cost and n_ops == [len(s.ops) for s in p.solutions]
    s, _ = cpu_pool.run_pool(cpu_pool.sample_worker, [('port', 16, 16, 0, 'wmc', 5.0)], 1)
    assert s[0]['finished'] and s[0]['iterations'] > 0
