"""Recording number type: turns a polymorphic user function into an expression tape.

The reference takes its coordinate map and potential as rank-2 polymorphic
functions, ``forall a. RealFloat a => V.Vector n a -> V.Vector m a``
(/root/reference/src/Numeric/Hamilton.hs:212,215,249,252) and instantiates them
at `ad`'s number types (:221-224).  The replacement instantiates them ONCE at
`Var`, whose arithmetic records an SSA tape (`hamk_op[]`, include/hamk.h); the
tape crosses the C ABI and the device library does the differentiation.

A Python function written against `Var` uses the same vocabulary a Haskell one
uses against `RealFloat a`: + - * / negate, `**`, `^` (integer power is
`powi`), and the `Floating` methods exported here as module functions (sin,
cos, exp, log, sqrt, ...).  Comparisons raise: `Ord`-dependent functions
cannot be traced (hard parts).
"""
from __future__ import annotations

import ctypes
import math
from typing import Callable, List, Sequence, Tuple

# opcodes: keep in lock-step with enum hamk_opcode in include/hamk.h
OP_CONST, OP_INPUT, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_RECIP = range(8)
OP_SIN, OP_COS, OP_TAN, OP_ASIN, OP_ACOS, OP_ATAN = range(8, 14)
OP_SINH, OP_COSH, OP_TANH, OP_EXP, OP_LOG, OP_SQRT = range(14, 20)
OP_POWC, OP_POWI, OP_POW, OP_ATAN2, OP_ASINH, OP_ACOSH, OP_ATANH = range(20, 27)
OP_ABS, OP_SIGNUM = 27, 28

OP_NAMES = {
    OP_CONST: "const", OP_INPUT: "input", OP_ADD: "add", OP_SUB: "sub", OP_MUL: "mul",
    OP_DIV: "div", OP_NEG: "neg", OP_RECIP: "recip", OP_SIN: "sin", OP_COS: "cos",
    OP_TAN: "tan", OP_ASIN: "asin", OP_ACOS: "acos", OP_ATAN: "atan", OP_SINH: "sinh",
    OP_COSH: "cosh", OP_TANH: "tanh", OP_EXP: "exp", OP_LOG: "log", OP_SQRT: "sqrt",
    OP_POWC: "powc", OP_POWI: "powi", OP_POW: "pow", OP_ATAN2: "atan2",
    OP_ASINH: "asinh", OP_ACOSH: "acosh", OP_ATANH: "atanh", OP_ABS: "abs", OP_SIGNUM: "signum",
}


class HamkOp(ctypes.Structure):
    """ctypes image of `struct hamk_op` (include/hamk.h)."""
    _fields_ = [("op", ctypes.c_int32), ("a", ctypes.c_int32), ("b", ctypes.c_int32),
                ("_pad", ctypes.c_int32), ("c", ctypes.c_double)]


class Tape:
    """An SSA recording: ops[i] = (opcode, a, b, c) defines value i."""

    def __init__(self, n_in: int):
        self.n_in = n_in
        self.ops: List[Tuple[int, int, int, float]] = []
        self.outs: List[int] = []
        self._memo = {}

    def emit(self, op: int, a: int = 0, b: int = 0, c: float = 0.0) -> int:
        key = (op, a, b, float(c).hex())
        idx = self._memo.get(key)
        if idx is None:           # hash-consing: identical subexpressions share one value
            idx = len(self.ops)
            self.ops.append((op, a, b, float(c)))
            self._memo[key] = idx
        return idx

    def const(self, c: float) -> "Var":
        return Var(self, self.emit(OP_CONST, 0, 0, float(c)))

    def input(self, j: int) -> "Var":
        return Var(self, self.emit(OP_INPUT, j))

    def const_value(self, idx: int):
        op, _, _, c = self.ops[idx]
        return c if op == OP_CONST else None

    def as_ctypes(self):
        arr = (HamkOp * max(1, len(self.ops)))()
        for i, (op, a, b, c) in enumerate(self.ops):
            arr[i].op, arr[i].a, arr[i].b, arr[i]._pad, arr[i].c = op, a, b, 0, c
        outs = (ctypes.c_int32 * max(1, len(self.outs)))(*self.outs)
        return arr, len(self.ops), outs

    def canonical(self) -> "Tape":
        """The same recording in CANONICAL form: only the values the outputs depend on, numbered in
        depth-first post-order from the outputs (first operand before second, outputs in order).
        A recorder is free to emit constants early or late and to leave folded-away operands behind;
        the canonical form depends on the expression alone, so every host shim (this one,
        include/hamilton.hpp, bindings/haskell) ships byte-identical tapes for the same function
        (tests/test_recorders.py) -- and the code-object cache keys on it."""
        binary = (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_POW, OP_ATAN2)
        new_id = {}
        out = Tape(self.n_in)
        for root in self.outs:
            stack = [(root, 0)]
            while stack:
                node, phase = stack.pop()
                if node in new_id:
                    continue
                op, a, b, c = self.ops[node]
                kids = [] if op in (OP_CONST, OP_INPUT) else ([a, b] if op in binary else [a])
                if phase == 0:
                    stack.append((node, 1))
                    for k in reversed(kids):
                        if k not in new_id:
                            stack.append((k, 0))
                else:
                    na = new_id[a] if kids else a
                    nb = new_id[b] if len(kids) == 2 else b
                    new_id[node] = len(out.ops)
                    out.ops.append((op, na, nb, c))
        out.outs = [new_id[r] for r in self.outs]
        return out

    def evaluate(self, xs: Sequence[float]) -> List[float]:
        """Plain-float interpretation (host-side sanity check of a recording)."""
        v: List[float] = []
        for op, a, b, c in self.ops:
            v.append(_eval_ieee(op, v, xs, a, b, c))
        return [v[o] for o in self.outs]

    def __len__(self):
        return len(self.ops)

    def __repr__(self):
        lines = []
        for i, (op, a, b, c) in enumerate(self.ops):
            lines.append(f"v{i} = {OP_NAMES[op]} a={a} b={b} c={c!r}")
        lines.append(f"outs = {self.outs}")
        return "\n".join(lines)


def _powi(x: float, k: int) -> float:
    if k < 0:
        return 1.0 / _powi(x, -k)
    r, base = 1.0, x
    while k:
        if k & 1:
            r *= base
        base *= base
        k >>= 1
    return r


_EVAL = {
    OP_CONST: lambda v, xs, a, b, c: c,
    OP_INPUT: lambda v, xs, a, b, c: float(xs[a]),
    OP_ADD: lambda v, xs, a, b, c: v[a] + v[b],
    OP_SUB: lambda v, xs, a, b, c: v[a] - v[b],
    OP_MUL: lambda v, xs, a, b, c: v[a] * v[b],
    OP_DIV: lambda v, xs, a, b, c: v[a] / v[b],
    OP_NEG: lambda v, xs, a, b, c: -v[a],
    OP_RECIP: lambda v, xs, a, b, c: 1.0 / v[a],
    OP_SIN: lambda v, xs, a, b, c: math.sin(v[a]),
    OP_COS: lambda v, xs, a, b, c: math.cos(v[a]),
    OP_TAN: lambda v, xs, a, b, c: math.tan(v[a]),
    OP_ASIN: lambda v, xs, a, b, c: math.asin(v[a]),
    OP_ACOS: lambda v, xs, a, b, c: math.acos(v[a]),
    OP_ATAN: lambda v, xs, a, b, c: math.atan(v[a]),
    OP_SINH: lambda v, xs, a, b, c: math.sinh(v[a]),
    OP_COSH: lambda v, xs, a, b, c: math.cosh(v[a]),
    OP_TANH: lambda v, xs, a, b, c: math.tanh(v[a]),
    OP_EXP: lambda v, xs, a, b, c: math.exp(v[a]),
    OP_LOG: lambda v, xs, a, b, c: math.log(v[a]),
    OP_SQRT: lambda v, xs, a, b, c: math.sqrt(v[a]),
    OP_POWC: lambda v, xs, a, b, c: math.pow(v[a], c),
    OP_POWI: lambda v, xs, a, b, c: _powi(v[a], b),
    OP_POW: lambda v, xs, a, b, c: math.pow(v[a], v[b]),
    OP_ATAN2: lambda v, xs, a, b, c: math.atan2(v[a], v[b]),
    OP_ASINH: lambda v, xs, a, b, c: math.asinh(v[a]),
    OP_ACOSH: lambda v, xs, a, b, c: math.acosh(v[a]),
    OP_ATANH: lambda v, xs, a, b, c: math.atanh(v[a]),
    OP_ABS: lambda v, xs, a, b, c: abs(v[a]),
    OP_SIGNUM: lambda v, xs, a, b, c: float((v[a] > 0) - (v[a] < 0)),
}


def _eval_ieee(op: int, v, xs, a, b, c) -> float:
    """One op on fp64 values with IEEE semantics: where Python's math raises (1/0, log 0, sqrt -1,
    exp 1000) Haskell's Double yields Inf / NaN -- and so must a folded constant."""
    try:
        return _EVAL[op](v, xs, a, b, c)
    except (ZeroDivisionError, ValueError, OverflowError):
        import numpy as np
        f8 = np.float64
        va = f8(v[a]) if op != OP_CONST and op != OP_INPUT else f8(0)
        with np.errstate(all="ignore"):
            if op == OP_DIV:
                r = va / f8(v[b])
            elif op == OP_RECIP:
                r = f8(1.0) / va
            elif op == OP_POWC:
                r = np.power(va, f8(c))
            elif op == OP_POW:
                r = np.power(va, f8(v[b]))
            elif op == OP_POWI:
                r = f8(1.0) / f8(_powi_ieee(float(va), -b)) if b < 0 else f8(_powi_ieee(float(va), b))
            else:
                r = getattr(np, {OP_SIN: "sin", OP_COS: "cos", OP_TAN: "tan", OP_ASIN: "arcsin", OP_ACOS: "arccos",
                                 OP_ATAN: "arctan", OP_SINH: "sinh", OP_COSH: "cosh", OP_TANH: "tanh", OP_EXP: "exp",
                                 OP_LOG: "log", OP_SQRT: "sqrt", OP_ASINH: "arcsinh", OP_ACOSH: "arccosh",
                                 OP_ATANH: "arctanh"}[op])(va)
        return float(r)


def _powi_ieee(x: float, k: int) -> float:
    import numpy as np
    with np.errstate(all="ignore"):
        r, base = np.float64(1.0), np.float64(x)
        while k:
            if k & 1:
                r = r * base
            base = base * base
            k >>= 1
    return float(r)


class Var:
    """A traced fp64 value (the `a` of `forall a. RealFloat a`)."""
    __slots__ = ("tape", "idx")
    __array_priority__ = 1000

    def __init__(self, tape: Tape, idx: int):
        self.tape = tape
        self.idx = idx

    # -- lifting ---------------------------------------------------------------
    def _lift(self, other) -> "Var":
        if isinstance(other, Var):
            if other.tape is not self.tape:
                raise ValueError("mixing values of two different recordings")
            return other
        if isinstance(other, (int, float)):
            return self.tape.const(float(other))   # realToFrac / fromInteger
        try:
            return self.tape.const(float(other))
        except Exception:
            return NotImplemented

    def _cv(self):
        return self.tape.const_value(self.idx)

    def _bin(self, op, other, swap=False):
        o = self._lift(other)
        if o is NotImplemented:
            return NotImplemented
        a, b = (o, self) if swap else (self, o)
        ca, cb = a._cv(), b._cv()
        t = self.tape
        if ca is not None and cb is not None:      # constant folding
            return t.const(_eval_ieee(op, [ca, cb], None, 0, 1, 0.0))
        # exact identities only (x+0, 0+x, x-0, x*1, 1*x, x/1): never change a result bit
        if op == OP_ADD:
            if ca == 0.0:
                return b
            if cb == 0.0:
                return a
        elif op == OP_SUB:
            if cb == 0.0:
                return a
            if ca == 0.0:
                return -b
        elif op == OP_MUL:
            if ca == 1.0:
                return b
            if cb == 1.0:
                return a
            if ca == -1.0:
                return -b
            if cb == -1.0:
                return -a
        elif op == OP_DIV:
            if cb == 1.0:
                return a
            if ca == 1.0:
                return Var(t, t.emit(OP_RECIP, b.idx))
        return Var(t, t.emit(op, a.idx, b.idx))      # operands in the order written (no commutative reordering:
        #                                              the tape must be a function of the expression alone, see Tape.canonical)

    def __add__(self, o): return self._bin(OP_ADD, o)
    def __radd__(self, o): return self._bin(OP_ADD, o, True)
    def __sub__(self, o): return self._bin(OP_SUB, o)
    def __rsub__(self, o): return self._bin(OP_SUB, o, True)
    def __mul__(self, o): return self._bin(OP_MUL, o)
    def __rmul__(self, o): return self._bin(OP_MUL, o, True)
    def __truediv__(self, o): return self._bin(OP_DIV, o)
    def __rtruediv__(self, o): return self._bin(OP_DIV, o, True)

    def __neg__(self):
        c = self._cv()
        if c is not None:
            return self.tape.const(-c)
        op, a, _, _ = self.tape.ops[self.idx]
        if op == OP_NEG:
            return Var(self.tape, a)
        return Var(self.tape, self.tape.emit(OP_NEG, self.idx))

    def __pos__(self):
        return self

    def __pow__(self, e):
        """Haskell `**` (Floating) for Var/float exponents, `^` for Python ints."""
        t = self.tape
        if isinstance(e, bool):
            raise TypeError("bool exponent")
        if isinstance(e, int):
            return powi(self, e)
        ev = self._lift(e)
        if ev is NotImplemented:
            return NotImplemented
        ce, cs = ev._cv(), self._cv()
        if ce is not None:
            if cs is not None:
                return t.const(_eval_ieee(OP_POWC, [cs], None, 0, 0, ce))
            if ce == math.floor(ce) and abs(ce) <= 64:
                return powi(self, int(ce))     # x ** 2 must stay valid for x < 0 (Examples.hs:154)
            return Var(t, t.emit(OP_POWC, self.idx, 0, ce))
        return Var(t, t.emit(OP_POW, self.idx, ev.idx))

    def __rpow__(self, base):
        b = self._lift(base)
        if b is NotImplemented:
            return NotImplemented
        return b.__pow__(self)

    # -- things a traced function must not do ---------------------------------------
    def _no_ord(self, *_):
        raise TypeError("comparison on a traced value: functions that branch on their "
                        "argument (Ord/RealFrac methods) cannot be recorded")
    __lt__ = __le__ = __gt__ = __ge__ = _no_ord
    # == and != too: identity semantics would silently record one branch of `if x == 0:` (the
    # Haskell shim's Traced instance has no usable Eq either); Vars stay hashable by identity
    __eq__ = __ne__ = _no_ord
    __hash__ = object.__hash__
    __bool__ = _no_ord
    __float__ = _no_ord

    def __abs__(self):
        """Num.abs: recorded (derivative signum x, as `ad` differentiates it; not differentiable at 0)."""
        c = self._cv()
        if c is not None:
            return self.tape.const(abs(c))
        return Var(self.tape, self.tape.emit(OP_ABS, self.idx))

    def __repr__(self):
        return f"Var(v{self.idx})"


def _unary(op: int, pyf: Callable[[float], float]):
    def f(x):
        if isinstance(x, Var):
            c = x._cv()
            if c is not None:
                return x.tape.const(_eval_ieee(op, [c], None, 0, 0, 0.0))
            return Var(x.tape, x.tape.emit(op, x.idx))
        return _eval_ieee(op, [float(x)], None, 0, 0, 0.0)
    f.__name__ = OP_NAMES[op]
    return f


sin = _unary(OP_SIN, math.sin)
cos = _unary(OP_COS, math.cos)
tan = _unary(OP_TAN, math.tan)
asin = _unary(OP_ASIN, math.asin)
acos = _unary(OP_ACOS, math.acos)
atan = _unary(OP_ATAN, math.atan)
sinh = _unary(OP_SINH, math.sinh)
cosh = _unary(OP_COSH, math.cosh)
tanh = _unary(OP_TANH, math.tanh)
asinh = _unary(OP_ASINH, math.asinh)
acosh = _unary(OP_ACOSH, math.acosh)
atanh = _unary(OP_ATANH, math.atanh)
exp = _unary(OP_EXP, math.exp)
log = _unary(OP_LOG, math.log)
sqrt = _unary(OP_SQRT, math.sqrt)
recip = _unary(OP_RECIP, lambda v: 1.0 / v)
signum = _unary(OP_SIGNUM, lambda v: float((v > 0) - (v < 0)))


def powi(x, k: int):
    """Haskell `x ^ k` / `x ^^ k` (integral exponent)."""
    k = int(k)
    if not isinstance(x, Var):
        return _powi(float(x), k)
    c = x._cv()
    if c is not None:
        return x.tape.const(_eval_ieee(OP_POWI, [c], None, 0, k, 0.0))
    if k == 0:
        return x.tape.const(1.0)
    if k == 1:
        return x
    return Var(x.tape, x.tape.emit(OP_POWI, x.idx, k))


def atan2(y, x):
    if isinstance(y, Var) or isinstance(x, Var):
        anchor = y if isinstance(y, Var) else x
        yv, xv = anchor._lift(y), anchor._lift(x)
        cy, cx = yv._cv(), xv._cv()
        if cy is not None and cx is not None:
            return anchor.tape.const(math.atan2(cy, cx))
        return Var(anchor.tape, anchor.tape.emit(OP_ATAN2, yv.idx, xv.idx))
    return math.atan2(float(y), float(x))


def trace(fn: Callable, n_in: int, n_out: int | None = None) -> Tape:
    """Record `fn` applied to n_in fresh inputs.  n_out=None: scalar result."""
    tape = Tape(n_in)
    xs = [tape.input(j) for j in range(n_in)]
    res = fn(xs)
    if n_out is None:
        res = [res]
    else:
        res = list(res)
        if len(res) != n_out:
            raise ValueError(f"coordinate map returned {len(res)} values, expected {n_out}")
    for r in res:
        if not isinstance(r, Var):
            r = tape.const(float(r))
        elif r.tape is not tape:
            raise ValueError("result belongs to a different recording")
        tape.outs.append(r.idx)
    return tape.canonical()
