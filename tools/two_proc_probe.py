"""Do TWO processes on one MI355X, each driving half of the 64-chain batch on its own four streams, finish the batch sooner than one process with
all 64 chains?  (More than four busy hardware queues inside ONE process slow the greedy loop 2.6 x -- GPU_MAX_HW_QUEUES=8, round 3; CU-masked streams,
round 4 -- but whether queues of different processes share that fate was never measured.)  Each worker warms up, meets the other at a file barrier and
times `reps` solves of its `chains` 256x256 chains.  usage (GPU box): python tools/two_proc_probe.py [procs=2] [chains per process=32] [reps=2]"""
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.environ["DA_ROOT"] + "/tests")
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
me, n, chains, reps, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
ks = [int_matrix(me * chains + s, 256, 256, -128, 128) for s in range(chains)]
opts = dict(method0="wmc", method1="wmc", decompose_dc=-1, search_all_decompose_dc=False)
hip.solve_many_raw(ks[:8], **opts).free()          # arena, streams, code objects
hip.solve_many_raw(ks, **opts).free()
open(f"{d}/ready{me}", "w").close()
while len([f for f in os.listdir(d) if f.startswith("ready")]) < n:
    time.sleep(0.001)
hip.timings(reset=True)
t = time.perf_counter()
for _ in range(reps):
    hip.solve_many_raw(ks, **opts).free()
dt = time.perf_counter() - t
tm = hip.timings(reset=True)
print(json.dumps({"proc": me, "chains": chains, "seconds_per_solve_call": dt / reps, "loop_ms": tm["loop_ms"] / reps, "us_per_lockstep_step": 1e3 * tm["loop_ms"] / max(tm["lockstep_iters"], 1)}), flush=True)
'''


def run(procs, chains, reps):
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, DA_ROOT=str(ROOT))
        ps = [subprocess.Popen([sys.executable, '-c', WORKER, str(i), str(procs), str(chains), str(reps), d], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(procs)]
        outs = [p.communicate(timeout=600) for p in ps]
    res = []
    for p, (o, e) in zip(ps, outs):
        if p.returncode != 0:
            print(e[-1500:], file=sys.stderr)
            continue
        res.append(json.loads([ln for ln in o.splitlines() if ln.startswith('{')][-1]))
    return res


if __name__ == '__main__':
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    chains = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    one = run(1, procs * chains, reps)
    many = run(procs, chains, reps)
    print(json.dumps({'one_process': one, 'processes': many}))
    if one and len(many) == procs:
        t1, tn = one[0]['seconds_per_solve_call'], max(r['seconds_per_solve_call'] for r in many)
        print(f'{procs * chains} chains: one process {t1:.3f} s per call ({procs * chains / t1:.1f} solves/s); {procs} processes x {chains} chains {tn:.3f} s ({procs * chains / tn:.1f} solves/s): {t1 / tn:.2f} x')
