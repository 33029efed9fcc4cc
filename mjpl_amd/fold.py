"""Straight-line float64 code with the model's constants folded in -- the emitter behind the generated chains.

The interpreting kernels (and the oracle, and MuJoCo) evaluate ``mj_kinematics`` with every table value read at run
time: a hinge about a body's z axis still multiplies by the axis' two zeros, a body placed at the origin of its parent
still rotates a zero vector.  A generated chain knows those values.  ``Fold`` emits the SAME operations in the SAME
order -- one IEEE-754 binary64 rounding per ``+ - *`` (the libraries are compiled with ``-ffp-contract=off``) -- except
those whose result is known without executing them:

* ``x * 1 -> x``, ``x * -1 -> -x``, constants with constants (Python floats are binary64: the product the device would
  round to);
* ``x * 0 -> 0`` and ``x + 0 -> x``.

Negation is free in binary64 (``a - b == a + (-b)``, ``(-a) * b == -(a * b)``, round-to-nearest is symmetric), so signs
travel with the operands instead of costing instructions.  What the folding may change is the SIGN OF AN EXACT ZERO
(``x * 0`` is ``-0`` for negative ``x``; ``-0 + 0 = +0``): every comparison, sum with a non-zero term, product and
square root gives the same value either way, so a folded chain differs from the interpreted one only where an exactly
zero intermediate reaches ``atan2`` or a division as a deciding sign -- not at any configuration the tests or the
planner produce (tests/test_pose_fold.py compares against the oracle's chain for equality, value by value).

A value is ``("c", float)`` (known at generation time) or ``("v", text, negated)`` (a C++ expression naming a register).
"""
from __future__ import annotations

import math

ZERO = ("c", 0.0)
ONE = ("c", 1.0)


def dlit(x) -> str:
    """An exact C++17 hexadecimal literal of a float64."""
    return float(x).hex()


class Fold:
    def __init__(self, indent: str = "    ", prefix: str = "t", fold: bool = True):
        # fold=False: constants are emitted as literals and every operation of the statement is executed -- the
        # unfolded chain tests compare a folded one with (tests/test_pose_fold.py)
        self.fold = fold
        self.lines: list[str] = []
        self.n = 0
        self.ind = indent
        self.prefix = prefix
        self.ops = {"mul": 0, "add": 0, "sqrt": 0, "div": 0, "sel": 0}
        self.memo: dict[str, tuple] = {}  # expression text -> the register that holds it (one definition per expression)

    # ---- values
    def c(self, x) -> tuple:
        x = float(x)
        if not self.fold:
            return ("v", dlit(x) if x >= 0 and not (x == 0 and math.copysign(1.0, x) < 0) else f"({dlit(x)})", False)
        return ("c", 0.0 if x == 0.0 else x)  # (either zero: the folded zero)

    @staticmethod
    def v(name: str) -> tuple:
        return ("v", name, False)

    @staticmethod
    def is_c(a) -> bool:
        return a[0] == "c"

    @staticmethod
    def is0(a) -> bool:
        return a[0] == "c" and a[1] == 0.0

    @staticmethod
    def neg(a) -> tuple:
        if a[0] == "c":
            return ("c", 0.0 if a[1] == 0.0 else -a[1])
        return ("v", a[1], not a[2])

    @staticmethod
    def text(a) -> str:
        """The value as a C++ expression."""
        if a[0] == "c":
            return dlit(a[1]) if a[1] >= 0 or math.isnan(a[1]) else f"({dlit(a[1])})"
        return f"(-{a[1]})" if a[2] else a[1]

    def emit(self, line: str):
        self.lines.append(self.ind + line)

    def tmp(self, expr: str) -> tuple:
        if expr in self.memo:
            return self.memo[expr]
        name = f"{self.prefix}{self.n}"
        self.n += 1
        self.emit(f"const double {name} = {expr};")
        self.memo[expr] = ("v", name, False)
        return self.memo[expr]

    def scope(self):
        """`with f.scope():` -- registers defined inside a C++ block are forgotten when it closes (the memo is restored)."""
        fold = self

        class _Scope:
            def __enter__(self):
                self.saved = dict(fold.memo)

            def __exit__(self, *exc):
                fold.memo = self.saved
        return _Scope()

    def named(self, name: str, a) -> tuple:
        """Bind a value to a declared variable (an output or a loop-carried register)."""
        self.emit(f"{name} = {self.text(a)};")
        return ("v", name, False)

    # ---- arithmetic, one rounding per emitted operation
    def mul(self, a, b) -> tuple:
        if a[0] == "c" and b[0] == "c":
            return self.c(a[1] * b[1])
        if self.is0(a) or self.is0(b):
            return ZERO
        if a[0] == "c":
            a, b = b, a  # (variable first; multiplication commutes exactly)
        if b[0] == "c":
            if b[1] == 1.0:
                return a
            if b[1] == -1.0:
                return self.neg(a)
            self.ops["mul"] += 1
            sign = a[2] != (b[1] < 0)
            t = self.tmp(f"{a[1]} * {dlit(abs(b[1]))}")
            return self.neg(t) if sign else t
        self.ops["mul"] += 1
        x, y = sorted((a[1], b[1]))
        t = self.tmp(f"{x} * {y}")
        return self.neg(t) if a[2] != b[2] else t

    def add(self, a, b) -> tuple:
        if a[0] == "c" and b[0] == "c":
            return self.c(a[1] + b[1])
        if self.is0(a):
            return b
        if self.is0(b):
            return a
        self.ops["add"] += 1
        if a[0] == "c":
            a, b = b, a  # (addition commutes exactly)
        if b[0] == "c":
            # x + c, -x + c = c - x
            if a[2]:
                return self.tmp(f"{dlit(b[1])} - {a[1]}")
            return self.tmp(f"{a[1]} + {dlit(b[1])}" if b[1] >= 0 else f"{a[1]} - {dlit(-b[1])}")
        if a[2] == b[2]:
            x, y = sorted((a[1], b[1]))
            t = self.tmp(f"{x} + {y}")
            return self.neg(t) if a[2] else t
        if not a[2] and b[2]:
            return self.tmp(f"{a[1]} - {b[1]}")
        return self.tmp(f"{b[1]} - {a[1]}")

    def sub(self, a, b) -> tuple:
        return self.add(a, self.neg(b))

    def sum(self, terms) -> tuple:
        """((t0 + t1) + t2) + ... -- the left-to-right order of the statements this mirrors."""
        acc = terms[0]
        for t in terms[1:]:
            acc = self.add(acc, t)
        return acc

    def reg(self, a) -> tuple:
        """The value in a register of its own, sign applied (for values that outlive the block or feed run-time code)."""
        if a[0] == "c" or a[2]:
            return self.tmp(self.text(a))
        return a

    # ---- the routines of mjpl_device.h (Real<double>::exact branches), operation for operation
    def mul_mat_vec3(self, m, v):
        return [self.sum([self.mul(m[3 * r + 0], v[0]), self.mul(m[3 * r + 1], v[1]), self.mul(m[3 * r + 2], v[2])]) for r in range(3)]

    def mul_quat(self, a, b):
        m = self.mul
        return [self.sub(self.sub(self.sub(m(a[0], b[0]), m(a[1], b[1])), m(a[2], b[2])), m(a[3], b[3])),
                self.sub(self.add(self.add(m(a[0], b[1]), m(a[1], b[0])), m(a[2], b[3])), m(a[3], b[2])),
                self.add(self.add(self.sub(m(a[0], b[2]), m(a[1], b[3])), m(a[2], b[0])), m(a[3], b[1])),
                self.add(self.sub(self.add(m(a[0], b[3]), m(a[1], b[2])), m(a[2], b[1])), m(a[3], b[0]))]

    def rot_vec_quat(self, vec, q):
        m = self.mul
        t0 = self.sub(self.add(m(q[0], vec[0]), m(q[2], vec[2])), m(q[3], vec[1]))
        t1 = self.sub(self.add(m(q[0], vec[1]), m(q[3], vec[0])), m(q[1], vec[2]))
        t2 = self.sub(self.add(m(q[0], vec[2]), m(q[1], vec[1])), m(q[2], vec[0]))
        two = self.c(2.0)
        return [self.add(vec[0], m(two, self.sub(m(q[2], t2), m(q[3], t1)))),
                self.add(vec[1], m(two, self.sub(m(q[3], t0), m(q[1], t2)))),
                self.add(vec[2], m(two, self.sub(m(q[1], t1), m(q[2], t0))))]

    def quat2mat(self, q):
        m = self.mul
        q00, q01, q02, q03 = m(q[0], q[0]), m(q[0], q[1]), m(q[0], q[2]), m(q[0], q[3])
        q11, q12, q13 = m(q[1], q[1]), m(q[1], q[2]), m(q[1], q[3])
        q22, q23, q33 = m(q[2], q[2]), m(q[2], q[3]), m(q[3], q[3])
        two = self.c(2.0)
        R = [None] * 9
        R[0] = self.sub(self.sub(self.add(q00, q11), q22), q33)
        R[4] = self.sub(self.add(self.sub(q00, q11), q22), q33)
        R[8] = self.add(self.sub(self.sub(q00, q11), q22), q33)
        R[1] = m(two, self.sub(q12, q03))
        R[2] = m(two, self.add(q13, q02))
        R[3] = m(two, self.add(q12, q03))
        R[5] = m(two, self.sub(q23, q01))
        R[6] = m(two, self.sub(q13, q02))
        R[7] = m(two, self.add(q23, q01))
        return R

    def normalize4(self, v, minval: float = 1e-15):
        """mju_normalize4: unit quaternion if the norm is tiny; rescaled only when the norm is off by more than minval.

        The statement computes norm = sqrt(|v|^2), tiny = norm < minval, scale = |norm - 1| > minval, inv = 1 / norm and
        selects.  A product of unit quaternions is a few ulp off unit length at most, so `scale` is false on almost every
        call -- and then neither the square root nor the division matters.  With a correctly rounded square root (the
        exact path's contract) "not tiny and not scale" is an INTERVAL of |v|^2 (sqrt is monotonic): the generated code
        tests |v|^2 against its two end points (unit_interval) and only executes the statement outside it.  Same
        values, bit for bit; about forty instructions less per body on the common path."""
        sq = self.sum([self.mul(v[k], v[k]) for k in range(4)])
        if all(x[0] == "c" for x in v):
            norm = math.sqrt(sq[1])
            if norm < minval:
                return [ONE, ZERO, ZERO, ZERO]
            if abs(norm - 1.0) > minval:
                inv = 1.0 / norm
                return [self.c(x[1] * inv) for x in v]
            return list(v)
        lo, hi = unit_interval(minval)
        self.ops["sqrt"] += 1
        self.ops["div"] += 1
        k0 = self.n
        self.n += 1
        names = [f"{self.prefix}{k0}n{k}" for k in range(4)]
        live = [k for k in range(4) if not self.is0(v[k]) or k == 0]
        self.emit("double " + ", ".join(f"{names[k]} = {self.text(v[k])}" for k in live) + ";")
        sqt = self.text(sq)
        self.emit(f"if (__builtin_expect(!({sqt} >= {dlit(lo)} && {sqt} <= {dlit(hi)}), 0)) {{  // (off unit length by more than {minval:g}, or not a number)")
        self.emit(f"  const double norm_ = sqrt({sqt});")
        self.emit(f"  const bool tiny_ = norm_ < {dlit(minval)}, scale_ = fabs(norm_ - 1.0) > {dlit(minval)};")
        self.emit("  const double inv_ = 1.0 / norm_;")
        for k in live:
            unit = "1.0" if k == 0 else "0.0"
            if self.is0(v[k]):
                self.emit(f"  {names[k]} = tiny_ ? {unit} : 0.0;")
            else:
                self.ops["sel"] += 1
                self.emit(f"  {names[k]} = tiny_ ? {unit} : (scale_ ? {self.text(v[k])} * inv_ : {self.text(v[k])});")
        self.emit("}")
        return [self.v(names[k]) if k in live else ZERO for k in range(4)]


_UNIT_INTERVALS: dict[float, tuple] = {}


def unit_interval(minval: float = 1e-15) -> tuple:
    """The smallest and largest float64 s with  not (sqrt(s) < minval)  and  not (|sqrt(s) - 1| > minval)  under a
    correctly rounded square root (math.sqrt is): the set is an interval because sqrt is monotonic."""
    if minval in _UNIT_INTERVALS:
        return _UNIT_INTERVALS[minval]

    def inside(s: float) -> bool:
        n = math.sqrt(s)
        return not (n < minval) and not (abs(n - 1.0) > minval)

    assert inside(1.0)
    lo = hi = 1.0
    # walk out from 1 in steps that halve: the boundary lies within a few hundred ulp of 1 for minval ~ 1e-15
    for direction in (-1.0, 1.0):
        x = 1.0
        step = 4.0 * minval
        while step > 0.0:
            y = x + direction * step
            if y != x and inside(y):
                x = y
            else:
                step *= 0.5
                if x + direction * step == x:
                    break
        while inside(math.nextafter(x, direction * math.inf)):  # (to the last float64 inside)
            x = math.nextafter(x, direction * math.inf)
        if direction < 0:
            lo = x
        else:
            hi = x
    assert inside(lo) and inside(hi) and not inside(math.nextafter(lo, -math.inf)) and not inside(math.nextafter(hi, math.inf))
    _UNIT_INTERVALS[minval] = (lo, hi)
    return lo, hi
