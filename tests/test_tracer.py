"""CPU: the recording number type (host logic of mkSystem / mkSystem')."""
import math

import numpy as np
import pytest

from hamilton_amd import examples as E
from hamilton_amd import tracer as T


@pytest.mark.parametrize("name", list(E.REGISTRY) + ["chain4"])
def test_tape_replays_the_function(name):
    spec = E.get(name)
    tf, tu = spec.trace()
    rng = np.random.default_rng(1)
    for _ in range(5):
        q = [lo + (hi - lo) * rng.random() for lo, hi in spec.q_box]
        x = tf.evaluate(q)
        np.testing.assert_allclose(x, [float(v) for v in spec.coords(q)], rtol=1e-15, atol=1e-15)
        z = x if spec.u_space == E.U_CARTESIAN else q
        assert tu.evaluate(z)[0] == pytest.approx(float(spec.potential_of_q(q)), rel=1e-15, abs=1e-15)


def test_constant_folding_and_identities():
    t = T.Tape(1)
    x = t.input(0)
    assert (x + 0).idx == x.idx and (0 + x).idx == x.idx and (x * 1).idx == x.idx and (x / 1).idx == x.idx
    assert (x - 0).idx == x.idx
    c = t.const(2.0) * 3 + 1
    assert t.const_value(c.idx) == 7.0
    assert (-(-x)).idx == x.idx
    assert (x + 1).idx == (x + 1).idx           # identical subexpressions are hash-consed ...
    assert (x + 1).idx != (1 + x).idx           # ... operands stay in the order written (the tape is a function of the expression)


def test_power_rules():
    t = T.Tape(1)
    x = t.input(0)
    sq = x ** 2                                 # Haskell `x ** 2` with literal exponent
    assert t.ops[sq.idx][0] == T.OP_POWI and t.ops[sq.idx][2] == 2
    sq2 = x ** 2.0
    assert sq2.idx == sq.idx
    r = x ** 0.5
    assert t.ops[r.idx][0] == T.OP_POWC and t.ops[r.idx][3] == 0.5
    y = t.input(0) + 1
    g = x ** y
    assert t.ops[g.idx][0] == T.OP_POW
    assert (x ** 1).idx == x.idx
    assert t.const_value((x ** 0).idx) == 1.0
    t.outs = [sq.idx, r.idx, g.idx]
    a, b, c = t.evaluate([1.7])
    assert a == pytest.approx(1.7 * 1.7) and b == pytest.approx(math.sqrt(1.7)) and c == pytest.approx(1.7 ** 2.7)
    t2 = T.trace(lambda q: q[0] ** 2, 1, None)
    assert t2.evaluate([-0.3])[0] == pytest.approx(0.09)   # negative base stays valid


def test_comparisons_cannot_be_traced():
    t = T.Tape(1)
    x = t.input(0)
    with pytest.raises(TypeError):
        _ = x < 1.0
    with pytest.raises(TypeError):
        bool(x)
    with pytest.raises(TypeError):
        float(x)


def test_mixing_recordings_is_an_error():
    a, b = T.Tape(1), T.Tape(1)
    with pytest.raises(ValueError):
        _ = a.input(0) + b.input(0)


def test_logistic_constants_are_fp64_as_written():
    """beta = log(0.9/(1-0.9))/width evaluated in fp64 like the Haskell source (Examples.hs:601-605)."""
    beta = math.log(0.9 / (1 - 0.9)) / 0.1
    t = T.trace(lambda q: E.logistic(1.5, 25, 0.1, q[0]), 1, None)
    consts = [c for (op, _, _, c) in t.ops if op == T.OP_CONST]
    assert beta in consts and 25.0 in consts


def test_sampler_is_shard_invariant():
    spec = E.get("doublePendulum")
    q, qd = E.sample_config(spec, 0, 1000)
    q2, qd2 = E.sample_config(spec, 400, 100)
    np.testing.assert_array_equal(q[:, 400:500], q2)
    np.testing.assert_array_equal(qd[:, 400:500], qd2)
    assert np.all(q >= -math.pi) and np.all(q < math.pi) and abs(q.mean()) < 0.2


def test_abs_and_signum_are_recorded():
    """Num's abs / signum are tape opcodes (27, 28): recorded, folded on constants, evaluated; what a
    traced function still cannot do is BRANCH on a value."""
    t = T.trace(lambda q: [abs(q[0]) * T.signum(q[1]) + abs(-2.5), T.signum(q[0] * 0 + 3.0)], 2, 2)
    ops = [o[0] for o in t.ops]
    assert T.OP_ABS in ops and T.OP_SIGNUM in ops
    assert t.evaluate([-1.5, -0.2]) == [-1.5 + 2.5, 1.0] and t.evaluate([0.7, 4.0]) == [0.7 + 2.5, 1.0]
    with pytest.raises(TypeError):
        T.trace(lambda q: [q[0] if q[0] > 0 else -q[0]], 1, 1)
