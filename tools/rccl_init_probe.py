"""Why does RCCL's communicator set-up sometimes not return on a one-GPU box?  (round 5: `ncclCommInitRank` of the library's transport hung
for > 600 s on 2 of 5 boxes when the one-rank script was started from inside the GPU suite; the same script ran in 45 s started alone.)

Runs the smallest thing that reaches `ncclCommInitRank` through the library -- `rccl_unique_id` + one tiny `solve_sharded_rccl` -- in a CHILD
process, N times per condition, with `NCCL_DEBUG=INFO` into a file per run, a bounded wait, and on a time-out everything that can be read
without a debugger: Python frames (faulthandler), every thread's kernel stack / wait channel / system call from /proc, the tail of RCCL's log.
Conditions (the suspects named in VERDICT r05):
  alone        nothing else in the process tree holds the GPU
  parent_ctx   THIS process holds a HIP context and a 64-chain arena (as the pytest process does when it starts the script)
  parent_busy  ... and is running solves on the GPU while the child initialises RCCL
  lo           child with NCCL_SOCKET_IFNAME=lo (bootstrap over loopback only)
  mutex_free   child that calls ncclCommInitRank directly through ctypes, outside the library's cache mutex (control)
usage (GPU box): python tools/rccl_init_probe.py [runs per condition = 4] [seconds per run = 120] [conditions, comma separated]
Writes gpurun_out/rccl_probe/{summary.json, <condition>_<i>.{log,nccl,proc}}."""
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / 'gpurun_out' / 'rccl_probe'

CHILD = r'''
import faulthandler, json, os, sys, time
faulthandler.dump_traceback_later(float(os.environ["PROBE_DUMP_AFTER"]), exit=False)
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.environ["DA_ROOT"] + "/tests")
t0 = time.perf_counter()
import numpy as np
from da4ml_amd import _binary as hip
t1 = time.perf_counter()
print("PROBE imported", round(t1 - t0, 2), flush=True)
if os.environ.get("PROBE_DIRECT"):
    import ctypes
    lib = ctypes.CDLL("/opt/rocm/lib/librccl.so", mode=ctypes.RTLD_LOCAL)
    uid = (ctypes.c_char * 128)()
    rc = lib.ncclGetUniqueId(uid)
    t2 = time.perf_counter()
    print("PROBE unique id", rc, round(t2 - t1, 2), flush=True)
    class Uid(ctypes.Structure):
        _fields_ = [("b", ctypes.c_char * 128)]
    u = Uid(); ctypes.memmove(ctypes.byref(u), uid, 128)
    comm = ctypes.c_void_p()
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
    rc = lib.ncclCommInitRank(ctypes.byref(comm), 1, u, 0)
    t3 = time.perf_counter()
    print(json.dumps({"ok": rc == 0, "import_s": round(t1 - t0, 2), "id_s": round(t2 - t1, 2), "init_and_solve_s": round(t3 - t2, 2)}), flush=True)
    os._exit(0)
uid = hip.rccl_unique_id()
t2 = time.perf_counter()
print("PROBE unique id", round(t2 - t1, 2), flush=True)
k = np.random.default_rng(0).integers(-8, 8, (8, 8)).astype(np.float32)
p, st = hip.solve_sharded_rccl(k, uid, rank=0, world=1)
t3 = time.perf_counter()
print(json.dumps({"ok": bool(np.all(p.kernel == k)), "import_s": round(t1 - t0, 2), "id_s": round(t2 - t1, 2), "init_and_solve_s": round(t3 - t2, 2), "allreduce_calls": st["allreduce_calls"]}), flush=True)
os._exit(0)   # (no interpreter teardown: the probe is about set-up)
'''


def proc_snapshot(pid: int) -> str:
    """every thread of the process: name, state, wait channel, system call, kernel stack (root only) -- what a debugger-less box offers"""
    lines = []
    base = Path(f'/proc/{pid}/task')
    try:
        tids = sorted(int(p.name) for p in base.iterdir())
    except OSError as e:
        return f'no /proc/{pid}: {e}'
    for tid in tids:
        def rd(name):
            try:
                return (base / str(tid) / name).read_text().strip()
            except OSError as e:
                return f'<{e.__class__.__name__}>'
        state = [ln for ln in rd('status').splitlines() if ln.startswith(('Name:', 'State:'))]
        lines.append(f'--- tid {tid} {" ".join(state)} wchan={rd("wchan")} syscall={rd("syscall")}')
        st = rd('stack')
        if st:
            lines.append(st)
    try:
        fds = sorted(os.listdir(f'/proc/{pid}/fd'), key=int)
        lines.append('--- fds: ' + ', '.join(f'{fd}->{os.readlink(f"/proc/{pid}/fd/{fd}")}' for fd in fds[:200]))
    except OSError:
        pass
    return '\n'.join(lines)


def run_child(tag: str, seconds: float, extra_env: dict) -> dict:
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(DA_ROOT=str(ROOT), NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,ENV,NET,BOOTSTRAP', NCCL_DEBUG_FILE=str(OUT / f'{tag}.nccl'), PROBE_DUMP_AFTER=str(max(5.0, seconds - 15)))
    env.update(extra_env)
    t0 = time.perf_counter()
    with open(OUT / f'{tag}.log', 'w') as log:
        p = subprocess.Popen([sys.executable, '-c', CHILD], env=env, stdout=log, stderr=subprocess.STDOUT, cwd=str(ROOT))
        try:
            rc = p.wait(timeout=seconds)
            hung = False
        except subprocess.TimeoutExpired:
            hung = True
            (OUT / f'{tag}.proc').write_text(proc_snapshot(p.pid))
            p.kill()
            rc = p.wait()
    dt = time.perf_counter() - t0
    text = (OUT / f'{tag}.log').read_text()
    last = [ln for ln in text.splitlines() if ln.startswith('{')]
    res = json.loads(last[-1]) if last else {}
    nccl_tail = ''
    try:
        nccl_tail = '\n'.join((OUT / f'{tag}.nccl').read_text().splitlines()[-12:])
    except OSError:
        pass
    return {'tag': tag, 'hung': hung, 'rc': rc, 'seconds': round(dt, 2), 'result': res, 'reached': [ln for ln in text.splitlines() if ln.startswith('PROBE')], 'nccl_tail': nccl_tail if hung else ''}


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    conds = sys.argv[3].split(',') if len(sys.argv) > 3 else ['alone', 'parent_ctx', 'parent_busy', 'lo', 'mutex_free']
    OUT.mkdir(parents=True, exist_ok=True)
    host = socket.gethostname()
    t = time.perf_counter()
    try:
        resolved = socket.gethostbyname(host)
    except OSError as e:
        resolved = f'unresolvable ({e})'
    ifaces = [ln.split(':')[0].strip() for ln in Path('/proc/net/dev').read_text().splitlines()[2:]]
    summary = {'hostname': host, 'hostname_resolves_to': resolved, 'resolve_seconds': round(time.perf_counter() - t, 3), 'interfaces': ifaces, 'env': {k: v for k, v in os.environ.items() if k.startswith(('NCCL', 'RCCL', 'HSA', 'HIP', 'ROCR', 'GPU_'))}, 'runs': []}
    hip = None
    busy_stop = threading.Event()
    busy_thread = None
    for cond in conds:
        if cond in ('parent_ctx', 'parent_busy') and hip is None:  # this process takes a HIP context and the arena of a 64-chain batch
            sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
            import numpy as np
            from cases import int_matrix
            from da4ml_amd import _binary as hip_mod
            hip = hip_mod
            ks = [int_matrix(s, 128, 128, -128, 128) for s in range(64)]
            opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
            hip.solve_many_raw(ks, **opts).free()
        if cond == 'parent_busy':
            def busy():
                while not busy_stop.is_set():
                    hip.solve_many_raw(ks[:16], **opts).free()
            busy_thread = threading.Thread(target=busy, daemon=True)
            busy_thread.start()
        extra = {'NCCL_SOCKET_IFNAME': 'lo'} if cond == 'lo' else {'PROBE_DIRECT': '1'} if cond == 'mutex_free' else {}
        for i in range(runs):
            r = run_child(f'{cond}_{i}', seconds, extra)
            r['condition'] = cond
            summary['runs'].append(r)
            print(json.dumps({k: r[k] for k in ('tag', 'hung', 'rc', 'seconds', 'result')}), flush=True)
            (OUT / 'summary.json').write_text(json.dumps(summary, indent=1))
        if cond == 'parent_busy':
            busy_stop.set()
            busy_thread.join()
    by = {}
    for r in summary['runs']:
        b = by.setdefault(r['condition'], {'runs': 0, 'hung': 0, 'seconds': []})
        b['runs'] += 1
        b['hung'] += int(r['hung'])
        b['seconds'].append(r['seconds'])
    summary['by_condition'] = by
    (OUT / 'summary.json').write_text(json.dumps(summary, indent=1))
    print(json.dumps(by), flush=True)


if __name__ == '__main__':
    main()
