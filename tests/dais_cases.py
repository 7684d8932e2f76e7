"""Random, VALID DAIS programs covering every opcode of the format (reference docs/dais.md:25-105), for pinning
da4ml_amd's executor (csrc/dais_interp.cc) against the reference's interpreter.  "Valid" = what the reference's tracer can
emit: causal operand ids, non-negative alignment shifts, quantisation only drops fractional bits, lookup inputs are
quantised values, magnitudes stay far below 2**62."""

import numpy as np

WRAPPED = {'input', 'quant', 'relu', 'mux'}  # kinds whose result is guaranteed to lie inside its own format


def _imm_words(v):
    v &= 0xFFFFFFFFFFFFFFFF
    lo, hi = v & 0xFFFFFFFF, v >> 32
    s32 = lambda w: w - (1 << 32) if w >= (1 << 31) else w  # noqa: E731
    return s32(lo), s32(hi)


def random_program(seed, n_in=4, n_ops=64, n_out=6, n_samples=24):
    rng = np.random.default_rng(seed)
    ops, fmt, kind, bound = [], [], [], []  # op words, (sgn, ints, frac), kind tag, bound on |value|
    tables = []
    inp_shifts = rng.integers(-2, 3, n_in)

    def fmt_bound(f):
        return 1 << max(f[1] + f[2] + 1, 1)

    def push(opcode, id0, id1, lo, hi, f, k, b):
        ops.append([opcode, id0, id1, lo, hi, *f])
        fmt.append(tuple(int(v) for v in f))
        kind.append(k)
        bound.append(int(b))

    for j in range(n_in):
        f = (int(rng.integers(0, 2)), int(rng.integers(1, 6)), int(rng.integers(0, 5)))
        push(-1, j, -1, 0, 0, f, 'input', fmt_bound(f))
    n_mul = 0
    while len(ops) < n_ops:
        i = len(ops)
        small = [q for q in range(i) if bound[q] < (1 << 40)]
        a, b = (int(rng.choice(small)) for _ in range(2))
        fa, fb = fmt[a], fmt[b]
        what = rng.choice(['add', 'add', 'add', 'relu', 'quant', 'cadd', 'const', 'mux', 'mul', 'lut', 'bitu', 'bitb'])
        if what == 'add':
            sh = int(rng.integers(-3, 4))
            base = max(fa[2], fb[2] - sh)
            f = (1, 12, base - int(rng.integers(0, 2)))
            push(int(rng.integers(0, 2)), a, b, sh, 0, f, 'add', (bound[a] << max(0, -(sh + fa[2] - fb[2]))) + (bound[b] << max(0, sh + fa[2] - fb[2])))
        elif what in ('relu', 'quant'):
            f = (int(rng.integers(0, 2)), int(rng.integers(0, 6)), fa[2] - int(rng.integers(0, 3)))
            if sum(f) < 1:
                f = (f[0], 1 - f[0] - f[2], f[2])
            code = 2 if what == 'relu' else 3
            push(code * (1 if rng.integers(0, 2) else -1), a, -1, 0, 0, f, what, fmt_bound(f))
        elif what == 'cadd':
            f = (1, 14, fa[2] + int(rng.integers(0, 2)))
            lo, hi = _imm_words(int(rng.integers(-1000, 1000)))
            push(4, a, -1, lo, hi, f, 'cadd', (bound[a] << 1) + 1000)
        elif what == 'const':
            v = int(rng.integers(-500, 500))
            lo, hi = _imm_words(v)
            push(5, -1, -1, lo, hi, (1, 10, int(rng.integers(0, 3))), 'const', abs(v) + 1)
        elif what == 'mux':
            cands = list(range(i))  # incl. unsigned 1-bit conditions (reduce-or outputs, np.where's to_bool)
            c = int(rng.choice(cands))
            out_frac = fa[2]
            hi = (fb[2] - out_frac) + int(rng.integers(0, 3))
            f = (int(rng.integers(0, 2)), int(rng.integers(1, 7)), out_frac)
            if sum(f) < 1:
                f = (f[0], 1 - f[0] - f[2], f[2])
            push(6 * (1 if rng.integers(0, 2) else -1), a, b, c, hi, f, 'mux', fmt_bound(f))
        elif what == 'mul':
            if n_mul >= 3 or bound[a] * bound[b] >= (1 << 44):
                continue
            n_mul += 1
            push(7, a, b, 0, 0, (1, 20, fa[2] + fb[2]), 'mul', bound[a] * bound[b])
        elif what == 'lut':
            src = [q for q in range(i) if kind[q] in WRAPPED and 1 <= sum(fmt[q]) <= 6]
            if not src:
                continue
            a = int(rng.choice(src))
            size = 1 << sum(fmt[a])
            tables.append(rng.integers(-200, 200, size).astype(np.int32))
            push(8, a, -1, len(tables) - 1, 0, (1, 9, int(rng.integers(0, 3))), 'lut', 256)
        elif what == 'bitu':
            if sum(fa) > 40:
                continue
            f = (int(rng.integers(0, 2)), sum(fa), 0)
            push(9 * (1 if rng.integers(0, 2) else -1), a, -1, int(rng.integers(0, 3)), 0, f, 'bitu', 1 << (sum(fa) + 1))
        else:  # bitb
            sh = int(rng.integers(-2, 3))
            hi = (int(rng.integers(0, 3)) << 24) | int(rng.integers(0, 4))
            push(10, a, b, sh, hi, (1, 30, max(fa[2], fb[2] - sh)), 'bitb', (bound[a] + bound[b]) << 4)
    out_idxs = rng.integers(-1, n_ops, n_out)
    out_idxs[0] = n_ops - 1
    out_shifts = rng.integers(-2, 3, n_out)
    out_negs = rng.integers(0, 2, n_out)
    head = [1, 0, n_in, n_out, n_ops, len(tables)]
    words = np.concatenate(
        [np.asarray(head), inp_shifts, out_idxs, out_shifts, out_negs, np.asarray(ops).ravel(), np.asarray([len(t) for t in tables], dtype=np.int64)]
        + [t.astype(np.int64) for t in tables]
    ).astype(np.int32)
    x = rng.uniform(-40.0, 40.0, (n_samples, n_in)) * rng.choice([1.0, 0.25, 0.03125], (1, n_in))
    x[0] = 0.0
    return words, x


def narrow_condition_program():
    """msb_mux on UNSIGNED 1-BIT conditions (what np.where(cond.to_bool(), a, b) traces to): an input declared (0,1,0) and a
    reduce-or result.  The reference tests `value > max(1LL << (width-2), 0)`, i.e. `value > 0` for width 1."""
    ops = [[-1, 0, -1, 0, 0, 1, 4, 0], [-1, 1, -1, 0, 0, 0, 1, 0], [9, 0, -1, 1, 0, 0, 1, 0],
           [-6, 0, 0, 1, 0, 1, 5, 0], [-6, 0, 0, 2, 0, 1, 5, 0], [6, 0, 1, 1, 0, 1, 5, 0]]  # fmt: skip
    words = np.asarray([1, 0, 2, 3, len(ops), 0] + [0, 0] + [3, 4, 5] + [0, 0, 0] + [0, 0, 0] + [w for op in ops for w in op], dtype=np.int32)
    x = np.asarray([[-3, 0], [-3, 1], [0, 0], [0, 1], [2, 0], [2, 1], [5, 1], [7, 0]], dtype=np.float64)
    return words, x
