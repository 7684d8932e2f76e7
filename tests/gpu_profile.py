"""Profiling workload: a batch of single greedy chains.  Usage: python tests/gpu_profile.py N BATCH"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
n = int(sys.argv[1]); B = int(sys.argv[2])
ks = [int_matrix(s, n, n, -128, 128) for s in range(B)]
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
hip.solve(ks[0][:8, :8].copy(), **opts)
hip.timings(reset=True)
t = time.time(); raw = hip.solve_many_raw(ks, **opts); dt = time.time() - t; res = [(None, raw.summary(0))]; raw.free()
tm = hip.timings(reset=True)
print(f'{n}x{n} batch {B}: {dt:.3f}s  -> {B/dt:.2f} solves/s; loop {tm["loop_ms"]:.1f} ms, lockstep iters {tm["lockstep_iters"]:.0f}, us/iter {1e3*tm["loop_ms"]/max(tm["lockstep_iters"],1):.1f}')
print(res[0][1])
ph = {k: v for k, v in tm.items() if k.startswith('sel_') or k.startswith('upd_')}
its = max(tm['iterations'], 1); pa = max(tm['partners'], 1)
import os
if os.environ.get('TIMER_WINDOW_STEPS'):  # a phase-timer build restricted to a window of steps (DA_TIMER_STEP_LO / HI): per step of the window
    w = float(os.environ['TIMER_WINDOW_STEPS']) * B
    print('window: select cycles/step', {k: round(v / w) for k, v in ph.items() if k.startswith('sel_')}, '| search', {k: round(tm[k] / max(tm['search_steps_timed'], 1)) for k in ('search_bounds', 'search_argmax', 'search_excluded')},
          '| update wave-cycles/step', {k: round(v / w) for k, v in ph.items() if k.startswith('upd_')})
print('select cycles/iteration:', {k: round(v / its) for k, v in ph.items() if k.startswith('sel_')})
print('update cycles/partner  :', {k: round(v / pa) for k, v in ph.items() if k.startswith('upd_')}, 'partners/iter', round(pa / its), 'found/partner %.2f' % (tm['found'] / pa), 'inserts/partner %.3f' % (tm['inserts'] / pa),
      '' if os.environ.get('DA4ML_HIP_STATS', '0') not in ('', '0') else '(blocks found / created are tallied only under DA4ML_HIP_STATS=1)')
sm = max(tm['samples'], 1)
print('picks known a step ahead: %.1f %% of %d steps; group re-reads per step %.2f' % (100.0 * tm['fast_steps'] / its, its, tm['rescans'] / its))
if tm['search_steps_timed']:
    q = tm['search_steps_timed']
    print('search block cycles/step: bounds %.0f, arg-max (per step of ALL steps) %.0f, excluded search %.0f; re-reads/step: stale below the floor %.2f, clean with an excluded best entry %.2f; rounds of the longest wave %.2f' % (tm['search_bounds'] / q, tm['search_argmax'] / q, tm['search_excluded'] / q, tm['search_stale_rereads'] / q, tm['search_touch_rereads'] / q, tm['search_rounds'] / q))
if tm['search_steps_timed']:
    print('search work lists: %.2f groups per step, %d steps with more than 32, longest %d; passes over the whole table %d; steps with a known pick %d of %d' % (tm['search_list_entries'] / tm['search_steps_timed'], tm['search_long_lists'], tm['search_longest_list'], tm['search_full_passes'], tm['fast_steps'], its))
print('sampled per-launch: select %.1f us, update %.1f us, chains/launch %.1f' % (1e3 * tm['select_ms_sampled'] / sm, 1e3 * tm['update_ms_sampled'] / sm, tm['sampled_chain_launches'] / sm))
if os.environ.get('STEP_CLOCKS'):  # a -DDA_STEP_CLOCKS build: the four sums ride in the search diagnostics (ticks of the 100 MHz real-time counter)
    q = max(tm['search_full_passes'], 1)
    print('step clocks (us per chain-step, %d chain-steps): select %.2f | boundary + placement %.2f | update %.2f | boundary + placement %.2f | sum %.2f' % (
        q, 0.01 * tm['search_stale_rereads'] / q, 0.01 * tm['search_touch_rereads'] / q, 0.01 * tm['search_rounds'] / q, 0.01 * tm['search_long_lists'] / q,
        0.01 * (tm['search_stale_rereads'] + tm['search_touch_rereads'] + tm['search_rounds'] + tm['search_long_lists']) / q))
