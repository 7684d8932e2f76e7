"""The product's KERNELS on the CPU.  tests/emu compiles the unmodified sources of da4ml_amd/csrc -- kernels included -- as plain
C++ against a stand-in for the HIP runtime (every GPU thread a fiber, wave-level operations as rendezvous of the lanes,
blocks and launches one after the other) into tests/emu/libda4ml_emu.so.  The worker processes load it through the
product's own loader (DA4ML_HIP_LIB), so that Python layer, C ABI, host logic and kernel code are the shipped ones; only
the machine underneath is emulated.  What this checks: indexing and logic of every kernel against the oracle, without a
GPU.  What it cannot check: races between wavefronts and blocks (nothing runs concurrently) and speed -- the `-m gpu`
tests remain the parity tests proper."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EMU_DIR = ROOT / 'tests' / 'emu'
EMU_LIB = EMU_DIR / 'libda4ml_emu.so'


@pytest.fixture(scope='module')
def emu():
    r = subprocess.run(['make', '-s', '-C', str(EMU_DIR)], capture_output=True, text=True)
    assert r.returncode == 0 and EMU_LIB.exists(), r.stdout[-2000:] + r.stderr[-2000:]

    def run(*args, env=None, timeout=900):
        e = dict(os.environ, DA4ML_HIP_LIB=str(EMU_LIB), DA4ML_HIP_UPD_BLOCKS='8', **(env or {}))
        out = subprocess.run([sys.executable, str(EMU_DIR / 'worker.py'), *map(str, args)], env=e, capture_output=True, text=True, cwd=str(ROOT), timeout=timeout)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    return run


@pytest.mark.parametrize('block', range(3))
def test_random_option_sets(emu, block):
    """every method / cost model / decomposition / search combination of cases.random_case, kernels emulated"""
    r = emu('random', block * 40, block * 40 + 40)
    assert r == {'bad': [], 'n': 40}


def test_non_power_of_two_steps(emu):
    """quantisation steps m * 2^k (m not a power of two): the selection's latency model reads -log2f(step) from the table the host
    built with its libm (StepLog2, cmvm_core.h) -- reference state_opr.cc:57 takes log2 of any step"""
    assert emu('oddsteps', 0, 30) == {'bad': [], 'n': 30}


def test_entry_layouts_and_their_boundaries(emu):
    """narrow and wide row-list entries: 12 / 13 digits, 256 / 257 columns, fractional weights, degenerate shapes"""
    assert emu('layouts')['bad'] == []


def test_batched_chains(emu):
    r = emu('batch')
    assert r['bad'] == [] and r['chains'] >= 13


def test_wide_host_pool(emu):
    """the host thread pool as on the 256-core GPU box: 200 threads, of which those beyond the first 63 are woken for wide loops
    only (here: 14 chains x 8 column ranges of the adder trees)"""
    r = emu('batch', env=dict(DA4ML_HOST_THREADS='200', DA4ML_HIP_TREE_THREADS='2'))
    assert r['bad'] == [] and r['chains'] >= 13


def test_lds_budget_of_the_selection_kernel(emu):
    """static + dynamic LDS of k_iter_select2 against the device's 160 KB per workgroup: with static arrays that leave room for
    everything the chain runs with its claim area; with less the area is dropped and the results stay the same; with no room for the fixed part
    the call fails with a message that names the sizes, not with a raw HIP launch error"""
    roomy = emu('lds_budget', env=dict(HIPEMU_STATIC_LDS='75000'))
    tight = emu('lds_budget', env=dict(HIPEMU_STATIC_LDS=str(160 * 1024 - 256 - 1000)))  # fixed part of these 12-column chains: 976 bytes; with the claim words and the alignment reserve > 1000
    none = emu('lds_budget', env=dict(HIPEMU_STATIC_LDS=str(160 * 1024 - 256 - 64)))
    assert roomy == {'ok': True, 'bad': [], 'message': ''} and tight == {'ok': True, 'bad': [], 'message': ''}
    assert not none['ok'] and 'bytes of dynamic LDS' in none['message'] and 'n_out too large' in none['message']


def test_fork_after_use(emu):
    r = emu('fork', timeout=600)
    assert (r['child'], r['parent'], r['fork_during_solve_ok'], r['child_of_busy_fork']) == (0, True, True, 0), r


def test_capacity_retry(emu):
    """arena heuristics far too small: capacity error on the device, rerun with larger arenas"""
    r = emu('retry', env=dict(DA4ML_HIP_TABLE_SCALE='0.02', DA4ML_HIP_ROW_SCALE='0.05'))
    assert r['equal'] and r['retries'] >= 1


def test_table_geometry_of_a_large_chain(emu):
    """DA4ML_HIP_TABLE_SCALE inflates the pair table of small problems to 2 M slots = 4096 groups of 512: all four bound
    registers of every lane and all eight slots per lane and group of the arg-max are in use, as in a 256x256 chain"""
    assert emu('big_table', env=dict(DA4ML_HIP_TABLE_SCALE='6000'))['bad'] == []


@pytest.mark.skipif(not os.environ.get('DA4ML_EMU_SLOW'), reason='minutes to half an hour per record: set DA4ML_EMU_SLOW=1 (run before every hand-over of kernel changes that no GPU has seen)')
@pytest.mark.parametrize('name,fname', [('128x128_seed0_single_chain_ref', 'large_chain_golden.json'), ('64x64_seed0_default', 'large_default_golden.json'),
                                        ('256x256_seed0_single_chain', 'large_chain_golden.json')])  # fmt: skip
def test_large_records_on_the_emulated_device(emu, name, fname):
    """the benchmark's own chain (256x256 int8, seed 0: 19 810 greedy steps, 2 M-slot table, thousands of partner rows per
    step) and the 128x128 record of the reference build through the kernels on the emulated device: 3 / 4 / 21 minutes"""
    assert emu('record', name, fname, timeout=7200)['equal']


def test_column_sharded_engine_single_rank(emu):
    """HipShardEngine (k_cs_init_counts, k_cs_init_table, k_iter_select<SHARDED>, k_cs_union, k_cs_partial, k_cs_apply)"""
    assert emu('shard_single', env=dict(DA4ML_SHARD_FORCE='1'))['bad'] == []


def test_column_sharded_capacity_retry(emu):
    """the sharded path retries on a capacity error like the ordinary one (it used to fail where da_solve succeeds)"""
    ok = emu('shard_retry', env=dict(DA4ML_SHARD_FORCE='1'))
    r = emu('shard_retry', env=dict(DA4ML_SHARD_FORCE='1', DA4ML_HIP_TABLE_SCALE='0.02', DA4ML_HIP_ROW_SCALE='0.05'))
    assert ok['equal'] and r['equal'] and r['chains'] == ok['chains']
    assert r['steps'] > ok['steps'] and r['calls'] > ok['calls']  # the aborted attempt's steps and exchanges are counted too


def test_column_sharded_engine_two_ranks_gloo(emu, tmp_path):
    """two processes, each with its own emulated device, exchanging the slabs over gloo: the product's sharded engine and
    orchestration end to end, bit-identical to the single-process result on every rank"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    res = tmp_path / 'rank0.json'
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   DA4ML_HIP_LIB=str(EMU_LIB), DA4ML_HIP_UPD_BLOCKS='8', EMU_OUT=str(res))  # fmt: skip
        procs.append(subprocess.Popen([sys.executable, str(EMU_DIR / 'worker.py'), 'shard_rank'], env=env, cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
    assert json.loads(res.read_text()) == {'equal': True}


def test_dais_device_executor(emu):
    assert emu('dais')['bad'] == []


def test_kernels_under_address_sanitizer():
    """the same library built with AddressSanitizer: no out-of-bounds access of static or dynamic LDS, of device allocations
    or of per-thread arrays in any kernel on the layout, batch, sharded-engine and large-table cases"""
    r = subprocess.run(['make', '-s', '-C', str(EMU_DIR), 'asan'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    runtime = subprocess.run(['make', '-s', '-C', str(EMU_DIR), 'asan-runtime'], capture_output=True, text=True).stdout.strip()
    if not Path(runtime).exists():
        pytest.skip('no AddressSanitizer runtime in this toolchain')
    base = dict(os.environ, DA4ML_HIP_LIB=str(EMU_DIR / 'build' / 'libda4ml_emu_asan.so'), DA4ML_HIP_UPD_BLOCKS='8', LD_PRELOAD=runtime,
                ASAN_OPTIONS='detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1')  # fmt: skip
    for what, extra in (('layouts', {}), ('batch', {}), ('shard_single', {'DA4ML_SHARD_FORCE': '1'}), ('big_table', {'DA4ML_HIP_TABLE_SCALE': '6000'})):
        out = subprocess.run([sys.executable, str(EMU_DIR / 'worker.py'), what], env=dict(base, **extra), capture_output=True, text=True, cwd=str(ROOT), timeout=900)
        assert out.returncode == 0 and 'AddressSanitizer' not in out.stderr, (what, out.stderr[-3000:])
        assert json.loads(out.stdout.strip().splitlines()[-1])['bad'] == [], what


def _tsan_runtime():
    r = subprocess.run(['make', '-s', '-C', str(EMU_DIR), 'tsan'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    runtime = subprocess.run(['make', '-s', '-C', str(EMU_DIR), 'tsan-runtime'], capture_output=True, text=True).stdout.strip()
    if not Path(runtime).exists():
        pytest.skip('no ThreadSanitizer runtime in this toolchain')
    return runtime


def test_race_detector_sees_what_it_should():
    """the race-detecting build of the emulated device (GPU threads as ThreadSanitizer fibers; happens-before only through launch
    boundaries, block barriers and wave-level rendezvous) on kernels with known races and on their repaired versions"""
    _tsan_runtime()
    exe = EMU_DIR / 'build' / 'race_selftest'
    for case in ('clean_barrier', 'clean_wave', 'clean_atomic', 'clean_lds_blocks', 'clean_launches', 'race_blocks', 'race_blocks_after_barrier', 'race_waves', 'race_lanes'):
        out = subprocess.run([str(exe), case], env=dict(os.environ, TSAN_OPTIONS='halt_on_error=0 exitcode=0'), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (case, out.stderr[-2000:])
        reports = out.stderr.count('WARNING: ThreadSanitizer: data race')
        assert (reports >= 1) if case.startswith('race_') else (reports == 0), (case, reports, out.stderr[-3000:])


def test_kernels_have_no_unexpected_data_race(oracle):
    """the product under that detector: a batch of two chains (several partner rows per update block, both entry layouts) and a
    default search.  The only races allowed are the two by-design patterns of the pair table listed, with their reasons, in
    tests/emu/tsan_benign.supp; the results must still be the oracle's"""
    import hashlib

    import numpy as np  # noqa: F401
    from cases import int_matrix

    runtime = _tsan_runtime()
    env = dict(os.environ, DA4ML_HIP_LIB=str(EMU_DIR / 'build' / 'libda4ml_emu_tsan.so'), DA4ML_HIP_UPD_BLOCKS='8', LD_PRELOAD=runtime,
               TSAN_OPTIONS=f'halt_on_error=0 exitcode=0 history_size=2 suppressions={EMU_DIR / "tsan_benign.supp"}')  # fmt: skip
    out = subprocess.run([sys.executable, str(EMU_DIR / 'worker.py'), 'race_cases'], env=env, capture_output=True, text=True, cwd=str(ROOT), timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'WARNING: ThreadSanitizer' not in out.stderr, out.stderr[:6000]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r['partners_per_step'] > 16  # more than one update block had work

    def digest(p):
        return hashlib.sha256(json.dumps(json.loads(json.dumps(p, default=lambda x: x.to_dict())), separators=(',', ':')).encode()).hexdigest()

    single = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
    want = [digest(oracle.solve(int_matrix(1, 28, 5, -128, 128), **single)), digest(oracle.solve(int_matrix(3, 5, 4, -4096, 4096), **single)),
            digest(oracle.solve(int_matrix(4, 8, 8, -32, 32)))]  # fmt: skip
    assert r['digests'] == want
