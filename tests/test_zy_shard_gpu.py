"""Column-sharded chains on the GPU (csrc/cmvm_engine.hip HipShardEngine + k_cs_* kernels behind da_solve_sharded).
(1) one process, sharded phases forced (the exchanges are no-ops): every kernel of the sharded path runs and the result must
equal the ordinary solve / the oracle; (2) two ranks -- both on GPU 0 of this one-GPU box, exchange staged through the host
over gloo (RCCL refuses two ranks on one device; with one GPU per rank the same code takes the nccl branch) -- each holding
half of the columns.  Bodies run in child processes (file sorted late: a device fault here cannot mask the solver suite)."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

SINGLE = r'''
import os, sys, json
os.environ["DA4ML_SHARD_FORCE"] = "1"
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DA_ROOT"], "tests"))
from da4ml_amd import _binary as hip
from oracle.oracle import Oracle
from cases import int_matrix, random_case
O = Oracle("ref" if os.path.exists(os.path.join(os.environ["DA_ROOT"], "oracle", "_ref", "libref.so")) else "port")  # the reference's own sources, live, where the build travelled
cases = [(int_matrix(0, 16, 16, -128, 128), {}), (int_matrix(1, 48, 40, -128, 128), dict(method0="wmc", method1="wmc", decompose_dc=-1, search_all_decompose_dc=False)),
         (int_matrix(2, 9, 20, -8, 8), dict(adder_size=1, carry_size=-1)), (int_matrix(3, 33, 7, -64, 64), dict(hard_dc=1))]
cases += [random_case(s)[:2] for s in (1003, 1006, 1012, 1021, 1030, 1033)]
same, chains = [], 0
for k, kw in cases:
    p, st = hip.solve_sharded(k, rank=0, world=1, **kw)
    same.append(bool(p == O.solve(k, **kw)))
    chains += st["sharded_chains"]
print(json.dumps({"same": same, "chains": chains}), flush=True)
'''

TWO = r'''
import os, sys, json
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DA_ROOT"], "tests"))
os.environ["LOCAL_RANK"] = "0"   # both ranks on GPU 0
from da4ml_amd import multi_gpu as mg
from da4ml_amd import _binary as hip
from oracle.oracle import Oracle
from cases import int_matrix
rank, world, local, device = mg.init("gloo")
O = Oracle("ref" if os.path.exists(os.path.join(os.environ["DA_ROOT"], "oracle", "_ref", "libref.so")) else "port")  # the reference's own sources, live, where the build travelled
same, steps = [], 0
for k, kw in [(int_matrix(0, 16, 16, -128, 128), {}), (int_matrix(5, 64, 64, -128, 128), dict(method0="wmc", method1="wmc", decompose_dc=-1, search_all_decompose_dc=False)),
              (int_matrix(2, 9, 21, -8, 8), dict(adder_size=1, carry_size=-1))]:
    p, st = mg.solve_column_sharded(k, return_stats=True, **kw)
    same.append(bool(p == O.solve(k, **kw)))
    steps += st["greedy_steps"]
print(json.dumps({"rank": rank, "same": same, "steps": steps}), flush=True)
mg.shutdown()
'''


def test_sharded_phases_single_process():
    env = dict(os.environ, DA_ROOT=str(ROOT))
    r = subprocess.run([sys.executable, '-c', SINGLE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(out['same']) and len(out['same']) == 10, out
    assert out['chains'] >= 10


def test_two_ranks_share_the_gpu_over_gloo():
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DA_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, '-c', TWO], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    res = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs]
    for r in res:
        assert all(r['same']) and len(r['same']) == 3, r
    assert res[0]['steps'] == res[1]['steps'] > 500


RCCL_ONE = r'''
import os, sys, json
os.environ["DA4ML_SHARD_FORCE"] = "1"        # run the sharded phases with one rank ...
os.environ["DA4ML_SHARD_FORCE_COMM"] = "1"   # ... and call the collective all the same: an all-reduce over one rank
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DA_ROOT"], "tests"))
from da4ml_amd import _binary as hip
from da4ml_amd import multi_gpu as mg
from oracle.oracle import Oracle
from cases import int_matrix, random_case
O = Oracle("ref" if os.path.exists(os.path.join(os.environ["DA_ROOT"], "oracle", "_ref", "libref.so")) else "port")  # the reference's own sources, live, where the build travelled
uid = hip.rccl_unique_id()
cases = [(int_matrix(0, 16, 16, -128, 128), {}), (int_matrix(1, 48, 40, -128, 128), dict(method0="wmc", method1="wmc", decompose_dc=-1, search_all_decompose_dc=False)),
         (int_matrix(2, 9, 20, -8, 8), dict(adder_size=1, carry_size=-1))]
cases += [random_case(s)[:2] for s in (1003, 1012, 1030)]
same, calls = [], 0
for k, kw in cases:
    p, st = hip.solve_sharded_rccl(k, uid, rank=0, world=1, **kw)
    same.append(bool(p == O.solve(k, **kw)))
    calls += st["allreduce_calls"]
k, kw = cases[1]
n_ids, fresh_id = [0], hip.rccl_unique_id
def counted_id():
    n_ids[0] += 1
    return fresh_id()
hip.rccl_unique_id = counted_id
p = mg.solve_column_sharded(k, transport="rccl", **kw)   # the user-facing entry (one rank: no process group needed)
same.append(bool(p == O.solve(k, **kw)))
p = mg.solve_column_sharded(k, transport="rccl", **kw)   # again: the group's unique id -- and with it the communicator -- is re-used
same.append(bool(p == O.solve(k, **kw)))
ids_before = n_ids[0]
mg._rccl_ids.clear()
closed = hip.rccl_shutdown()                             # the communicator of `uid` and the one of the group
p = mg.solve_column_sharded(k, transport="rccl", **kw)   # a new id, a new communicator
same.append(bool(p == O.solve(k, **kw)))
mg.shutdown()                                            # destroys it (no process group to tear down here)
print(json.dumps({"same": same, "calls": calls, "ids_before_shutdown": ids_before, "ids": n_ids[0], "closed": closed, "left": hip.rccl_shutdown()}), flush=True)
'''


def _run_bounded(cmd, env, seconds=int(os.environ.get('DA4ML_TEST_RCCL_SECONDS', '200'))):
    """The one-rank RCCL scripts take 30 - 50 s.  On two of five boxes of round 5 the same script, same library, did not return from RCCL's
    communicator set-up when started from inside the test suite (and ran through in 45 s when started alone on a third box); round 6 could not
    reproduce it (profiles/r06_rccl_probe.txt: every set-up in ~2 s under each suspected condition).  So the run is bounded, tried twice, and a time-out is reported as an
    XFAIL -- not a skip -- that carries what a debugger-less box can tell about the stuck process: RCCL's own log (NCCL_DEBUG=INFO), the
    Python frames (faulthandler) and every thread's kernel stack / wait channel from /proc (tools/rccl_init_probe.py), also written
    to gpurun_out/rccl_hang/.  No multi-rank RCCL run has ever been possible here (one-GPU boxes); the exchange protocol itself is
    covered by the gloo tests."""
    import tempfile

    sys.path.insert(0, str(ROOT / 'tools'))
    from rccl_init_probe import proc_snapshot

    notes = []
    for attempt in range(2):
        with tempfile.TemporaryDirectory() as tmp:
            log = Path(tmp) / 'nccl.log'
            e = dict(env, NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,ENV,NET,BOOTSTRAP', NCCL_DEBUG_FILE=str(log), PYTHONFAULTHANDLER='1')
            p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            try:
                so, se = p.communicate(timeout=seconds)
                return subprocess.CompletedProcess(cmd, p.returncode, so, se)
            except subprocess.TimeoutExpired:
                snap = proc_snapshot(p.pid)
                import signal

                p.send_signal(signal.SIGABRT)  # faulthandler prints every thread's Python frames
                try:
                    so, se = p.communicate(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    so, se = p.communicate()
                nccl = log.read_text()[-3000:] if log.exists() else '<no RCCL log>'
                note = f'--- attempt {attempt}: no return within {seconds} s\n--- RCCL log (tail)\n{nccl}\n--- stderr (tail)\n{se[-3000:]}\n--- threads\n{snap[-6000:]}'
                notes.append(note)
                out_dir = ROOT / 'gpurun_out' / 'rccl_hang'
                out_dir.mkdir(parents=True, exist_ok=True)
                (out_dir / f'hang_{os.getpid()}_{attempt}.txt').write_text(note)
    pytest.xfail(f'RCCL set-up did not return within {seconds} s (twice) on this box\n' + '\n'.join(notes)[-8000:])


def test_rccl_transport_one_rank():
    """The library's own RCCL transport (csrc/cmvm_rccl.*: librccl.so opened at run time, ncclCommInitRank, ncclAllReduce in place on
    the library's stream): one rank on this one-GPU box, the collective called for every exchange all the same -- communicator
    set-up, the stream-ordered device path and the host-staged path all run, results equal the oracle.  (Several ranks need one
    GPU each: RCCL refuses two ranks on one device; the exchange protocol itself is covered by the gloo tests.)"""
    env = dict(os.environ, DA_ROOT=str(ROOT))
    out = _run_bounded([sys.executable, '-c', RCCL_ONE], env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])  # (RCCL prints a version banner on stdout)
    assert r['same'] == [True] * 9 and r['calls'] > 100
    # one unique id (and one communicator) per process group however many solves; da_rccl_shutdown destroys what the library kept
    assert r['ids_before_shutdown'] == 1 and r['ids'] == 2 and r['closed'] == 2 and r['left'] == 0


def test_torch_nccl_paths_one_rank():
    """tools/nccl_world1_check.py: the torch.distributed `nccl` (= RCCL) paths of the N-rank bench and of the callback transport, as far
    as one GPU reaches -- process group, max / sum over ranks and barrier on device tensors, the DEVICE-buffer branch of the all-reduce
    callback (every exchange of a 48x48 chain forced through dist.all_reduce on the library's buffers), the library's own transport
    beside it; both results equal the unsharded solve."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = _run_bounded([sys.executable, str(ROOT / 'tools' / 'nccl_world1_check.py')], env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert r['ok'] and r['backend'] == 'nccl' and r['callback']['stats']['allreduce_calls'] > 100, r
