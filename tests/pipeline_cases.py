"""Seeded inputs of the pipelining / dead-statement tests, shared by tests/golden/make_pipeline_golden.py (which runs
the reference's own Python on them) and tests/test_pipeline.py."""

from cases import int_matrix, random_case

from da4ml_amd.types import CombLogic, Op, QInterval

# (name, kernel recipe, solve options): the solver output is what gets pipelined
SOLVES = [
    ('c1_16x16_int4_default', ('int_matrix', 0, 16, 16, -8, 8), {}),
    ('c1_16x16_int4_tracer_cost', ('int_matrix', 1, 16, 16, -8, 8), dict(adder_size=1, carry_size=-1)),
    ('16x16_int8_tracer_cost', ('int_matrix', 0, 16, 16, -128, 128), dict(adder_size=1, carry_size=-1)),
    ('16x16_int8_carry8', ('int_matrix', 2, 16, 16, -128, 128), dict(adder_size=4, carry_size=8)),
    ('12x20_int8_dc0', ('int_matrix', 3, 12, 20, -128, 128), dict(adder_size=1, carry_size=-1, decompose_dc=0, search_all_decompose_dc=False)),
    ('32x32_int8_tracer_cost', ('int_matrix', 4, 32, 32, -128, 128), dict(adder_size=1, carry_size=-1, decompose_dc=-1, search_all_decompose_dc=False)),
    ('random_case_3', ('random_case', 3), None),  # custom intervals + input latencies
    ('random_case_9', ('random_case', 9), None),
    ('random_case_14', ('random_case', 14), None),  # zero row
    ('random_case_22', ('random_case', 22), None),  # zero column -> absent output
] + [(f'const_inputs_{s}', ('const_input_case', s), None) for s in range(12)]
CUTOFFS = [0.0, 1.0, 2.0, 3.5, 5.0, 1000.0]
# larger solver outputs, compared by digest: (name, kernel recipe, options, cutoffs)
BIG = [
    ('64x64_int8_tracer_cost', ('int_matrix', 0, 64, 64, -128, 128), dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False, adder_size=1, carry_size=-1), (2.0, 4.0, 7.0)),
    ('64x64_int8_carry8', ('int_matrix', 1, 64, 64, -128, 128), dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False, adder_size=4, carry_size=8), (6.0, 15.0)),
    ('48x40_int8_dc2', ('int_matrix', 2, 48, 40, -128, 128), dict(adder_size=2, carry_size=4, decompose_dc=2, search_all_decompose_dc=False), (5.0, 12.5)),
]


def const_input_case(seed):
    """A small matrix whose inputs are partly constants (zero-width intervals): retiming turns them into constant adds."""
    import numpy as np

    rng = np.random.default_rng(seed)
    n_in, n_out = int(rng.integers(3, 14)), int(rng.integers(2, 14))
    k = int_matrix(seed, n_in, n_out, -64, 64)
    q = []
    for _ in range(n_in):
        st = float(2.0 ** rng.integers(-2, 2))
        if rng.random() < 0.35:
            c = float(rng.integers(-9, 10)) * st
            q.append((c, c, st))
        else:
            lo = float(rng.integers(-40, 1)) * st
            q.append((lo, lo + float(rng.integers(1, 90)) * st, st))
    return k, dict(qintervals=q, adder_size=int(rng.choice([-1, 1, 4])), carry_size=int(rng.choice([-1, 4, 8])))


def solve_inputs(spec):
    name, recipe, opts = spec
    if recipe[0] == 'const_input_case':
        return const_input_case(recipe[1])
    if recipe[0] == 'random_case':
        k, o, _ = random_case(recipe[1])
        return k, o
    return int_matrix(*recipe[1:]), opts


def handmade() -> CombLogic:
    """A graph with tracer-only statements (relu, msb-mux, constant) and dead code, for the passes that do not replay."""
    q = QInterval(-8.0, 7.0, 1.0)
    ops = [
        Op(0, -1, -1, 0, q, 0.0, 0.0),  # 0: x0
        Op(1, -1, -1, 0, q, 0.0, 0.0),  # 1: x1
        Op(2, -1, -1, 0, q, 0.0, 0.0),  # 2: x2 (dead)
        Op(0, 1, 0, 1, QInterval(-24.0, 21.0, 1.0), 1.0, 5.0),  # 3: x0 + 2 x1
        Op(0, 1, 1, 0, QInterval(-15.0, 15.0, 1.0), 1.0, 5.0),  # 4: x0 - x1
        Op(3, -1, 2, 0, QInterval(0.0, 21.0, 1.0), 1.0, 2.5),  # 5: relu(3)
        Op(2, 0, 0, 0, QInterval(-16.0, 14.0, 1.0), 1.0, 5.0),  # 6: dead
        Op(5, 4, 6, (1 << 32) | 4, QInterval(-30.0, 30.0, 1.0), 2.5, 3.0),  # 7: msb(4) ? 5 : 4 << 1
        Op(7, 3, 0, 0, QInterval(-54.0, 51.0, 1.0), 4.0, 6.0),  # 8
        Op(6, 6, 0, 0, QInterval(-32.0, 28.0, 1.0), 2.0, 5.0),  # 9: dead
    ]
    return CombLogic((3, 3), [0, 0, 0], [8, 5, -1], [0, 1, 0], [False, True, False], ops, -1, -1, None)


def handmade2() -> CombLogic:
    """quantize / constant add / constant / multiply / relu of a negated value / msb-mux on an unsigned condition"""
    q = QInterval(-8.0, 7.75, 0.25)
    ops = [
        Op(0, -1, -1, 0, q, 0.0, 0.0),
        Op(1, -1, -1, 0, q, 0.0, 0.0),
        Op(0, 1, 0, -1, QInterval(-12.0, 11.625, 0.125), 1.0, 4.0),  # 2: x0 + x1/2
        Op(2, -1, 3, 0, QInterval(-4.0, 3.5, 0.5), 1.0, 0.0),  # 3: quantize(2), wraps
        Op(2, -1, -3, 0, QInterval(-16.0, 15.0, 1.0), 1.0, 0.0),  # 4: quantize(-2)
        Op(3, -1, 4, 5, QInterval(-1.5, 6.0, 0.5), 1.0, 1.0),  # 5: (3) + 5 * 0.5
        Op(-1, -1, 5, -3, QInterval(-0.75, -0.75, 0.25), 0.0, 0.0),  # 6: constant -0.75
        Op(5, 6, 7, 0, QInterval(-4.5, 1.125, 0.125), 2.0, 9.0),  # 7: (5) * (6)
        Op(1, -1, -2, 0, QInterval(0.0, 8.0, 0.25), 1.0, 1.0),  # 8: relu(-x1)
        Op(8, 7, -6, (0xFFFFFFFF << 32) | 8, QInterval(-8.0, 8.0, 0.125), 3.0, 2.0),  # 9: msb(8) ? (8) : -(7) >> 1
        Op(9, 4, 1, 1, QInterval(-40.0, 40.0, 0.125), 4.0, 5.0),  # 10: (9) - 2 (4)
    ]
    return CombLogic((2, 3), [0, 0], [10, 7, 3], [0, -1, 2], [False, True, False], ops, -1, -1, None)


REPLAY = [('handmade', handmade, -8, 8, 1.0), ('handmade2', handmade2, -32, 32, 0.25)]


def replay_inputs(spec):
    import numpy as np

    name, make, lo, hi, step = spec
    comb = make()
    return comb, np.random.default_rng(11).integers(lo, hi, (64, comb.shape[0])) * step


def handmade_tables() -> CombLogic:
    """lookup statements (opcode 8) spread over stages: each stage keeps and renumbers only its own tables"""
    q = QInterval(0.0, 7.0, 1.0)
    ops = [
        Op(0, -1, -1, 0, q, 0.0, 0.0),
        Op(1, -1, -1, 0, q, 0.0, 0.0),
        Op(0, -1, 8, 2, QInterval(0.0, 15.0, 1.0), 1.0, 4.0),  # 2: table 2
        Op(1, -1, 8, 0, QInterval(0.0, 15.0, 1.0), 1.0, 4.0),  # 3: table 0
        Op(2, 3, 0, 0, QInterval(0.0, 30.0, 1.0), 2.0, 5.0),  # 4
        Op(4, -1, 8, 3, QInterval(0.0, 3.0, 1.0), 3.5, 4.0),  # 5: table 3
        Op(2, -1, 8, 2, QInterval(0.0, 3.0, 1.0), 3.0, 4.0),  # 6: table 2 again, later stage
        Op(5, 6, 0, 0, QInterval(0.0, 6.0, 1.0), 4.5, 3.0),  # 7
    ]
    return CombLogic((2, 2), [0, 0], [7, 4], [0, 0], [False, False], ops, -1, -1, ('t0', 't1', 't2', 't3'))
