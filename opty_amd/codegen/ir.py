"""Scalar expression DAG (the codegen's intermediate representation).

The reference turns SymPy expressions into C text directly
(``opty/utils.py:745-757``: ``cse`` then ``ccode`` per sub-expression) and
differentiates them symbolically with SymPy (``_forward_jacobian``,
``opty/utils.py:82-228``).  This package instead lowers the discretised
equations of motion once into a hash-consed DAG of float64 operations and does
everything else -- common-subexpression sharing, forward-mode
differentiation, node-invariant classification, scheduling, printing HIP --
on that DAG.

A node is ``(op, args)``; its id is its index in the DAG, and operands always
have smaller ids than their users, so increasing id order is a topological
order.  Construction goes through the ``DAG`` methods which fold constants,
apply the usual algebraic identities (``x*1``, ``x+0``, ``x*0``, ``--x`` ...)
and share structurally identical nodes.
"""

import math

# opcodes ------------------------------------------------------------------
CONST = 'const'      # args: (float,)
INPUT = 'in'         # args: (kind, index); see INPUT_KINDS
ADD, SUB, MUL, DIV, NEG = 'add', 'sub', 'mul', 'div', 'neg'
POWI = 'powi'        # args: (a, n) with n a Python int >= 2
POW = 'pow'          # args: (a, b)
MAX, MIN, ATAN2 = 'max', 'min', 'atan2'
SELECT = 'select'    # args: (rel, a, b, x, y): (a rel b) ? x : y
RELATIONS = ('lt', 'le', 'eq', 'ne')
UNARY = ('sqrt', 'sin', 'cos', 'tan', 'exp', 'log', 'abs', 'sign', 'asin',
         'acos', 'atan', 'sinh', 'cosh', 'tanh', 'step', 'erf', 'erfc',
         'floor', 'ceil', 'asinh', 'acosh', 'atanh',
         # the rest of the C99 printer's table (sympy.codegen.cfunctions,
         # gamma / loggamma): printed under their C names
         'log1p', 'expm1', 'log2', 'log10', 'exp2', 'cbrt', 'tgamma',
         'lgamma')
#: ``step(x)`` is 1 for x > 0 else 0 (used for d max / d min).

#: kinds of INPUT nodes.  ``cur``/``adj`` are the per-node values of a
#: trajectory row (state or input trajectory) at the "current" and "adjacent"
#: time node (``opty/direct_collocation.py:2350-2364``); ``par`` is a
#: node-invariant parameter, ``h`` the node time interval, ``free`` one entry
#: of the free vector at a fixed index (instance constraints).
INPUT_KINDS = ('cur', 'adj', 'par', 'h', 'free')


class DAG(object):

    def __init__(self):
        self.op = []
        self.args = []
        self.uni = []          # node-invariant flag, maintained on creation
        self._memo = {}
        self.zero = self.const(0.0)
        self.one = self.const(1.0)

    def __len__(self):
        return len(self.op)

    # -- raw node creation -------------------------------------------------
    def _node(self, op, args):
        key = (op, args)
        i = self._memo.get(key)
        if i is None:
            i = len(self.op)
            self.op.append(op)
            self.args.append(args)
            self._memo[key] = i
            if op == CONST:
                self.uni.append(True)
            elif op == INPUT:
                self.uni.append(args[0] in ('par', 'h', 'free'))
            else:
                self.uni.append(all(self.uni[j] for j in self.operands(i)))
        return i

    def const(self, v):
        v = float(v)
        if v == 0.0:
            v = 0.0            # fold -0.0
        # key on the repr so that nan/inf and 0.0 behave
        return self._node(CONST, (v,))

    def input(self, kind, index):
        assert kind in INPUT_KINDS
        return self._node(INPUT, (kind, int(index)))

    def is_const(self, i):
        return self.op[i] == CONST

    def value(self, i):
        return self.args[i][0]

    # -- arithmetic with simplification -------------------------------------
    def neg(self, a):
        if self.is_const(a):
            return self.const(-self.value(a))
        if self.op[a] == NEG:
            return self.args[a][0]
        if self.op[a] == SUB:
            x, y = self.args[a]
            return self.sub(y, x)
        return self._node(NEG, (a,))

    def add(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.value(a) + self.value(b))
        if a == self.zero:
            return b
        if b == self.zero:
            return a
        if self.op[b] == NEG:
            return self.sub(a, self.args[b][0])
        if self.op[a] == NEG:
            return self.sub(b, self.args[a][0])
        if a > b:
            a, b = b, a
        return self._node(ADD, (a, b))

    def sub(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.value(a) - self.value(b))
        if b == self.zero:
            return a
        if a == self.zero:
            return self.neg(b)
        if a == b:
            return self.zero
        if self.op[b] == NEG:
            return self.add(a, self.args[b][0])
        return self._node(SUB, (a, b))

    def mul(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.value(a)*self.value(b))
        if self.is_const(b):
            a, b = b, a
        if self.is_const(a):
            v = self.value(a)
            if v == 0.0:
                return self.zero
            if v == 1.0:
                return b
            if v == -1.0:
                return self.neg(b)
            if self.op[b] == NEG:
                return self.mul(self.const(-v), self.args[b][0])
            if self.op[b] == MUL and self.is_const(self.args[b][0]):
                c, x = self.args[b]
                return self.mul(self.const(v*self.value(c)), x)
            return self._node(MUL, (a, b))
        if self.op[a] == NEG and self.op[b] == NEG:
            return self.mul(self.args[a][0], self.args[b][0])
        if self.op[a] == NEG:
            return self.neg(self.mul(self.args[a][0], b))
        if self.op[b] == NEG:
            return self.neg(self.mul(a, self.args[b][0]))
        if a > b:
            a, b = b, a
        return self._node(MUL, (a, b))

    def div(self, a, b):
        if self.is_const(b):
            v = self.value(b)
            if self.is_const(a) and v != 0.0:
                return self.const(self.value(a)/v)
            if v == 1.0:
                return a
            if v == -1.0:
                return self.neg(a)
            # x / c is kept as a true division (one correctly rounded op)
        if a == self.zero and b != self.zero:
            return self.zero
        if self.op[a] == NEG and self.op[b] == NEG:
            return self.div(self.args[a][0], self.args[b][0])
        if self.op[a] == NEG:
            return self.neg(self.div(self.args[a][0], b))
        if self.op[b] == NEG:
            return self.neg(self.div(a, self.args[b][0]))
        return self._node(DIV, (a, b))

    def powi(self, a, n):
        n = int(n)
        if n == 0:
            return self.one
        if n == 1:
            return a
        if n < 0:
            return self.div(self.one, self.powi(a, -n))
        if self.is_const(a):
            try:
                return self.const(self.value(a)**n)
            except OverflowError:
                pass
        if self.op[a] == NEG:
            p = self.powi(self.args[a][0], n)
            return p if n % 2 == 0 else self.neg(p)
        return self._node(POWI, (a, n))

    def pow(self, a, b):
        if self.is_const(b):
            v = self.value(b)
            if v == int(v) and abs(v) <= 64:
                return self.powi(a, int(v))
            if v == 0.5:
                return self.unary('sqrt', a)
            if v == -0.5:
                return self.div(self.one, self.unary('sqrt', a))
            if self.is_const(a):
                # a fold that Python refuses (negative base with a
                # fractional exponent, overflow) is left to the device,
                # which returns what C's pow does (NaN / inf)
                try:
                    return self.const(math.pow(self.value(a), v))
                except (ValueError, OverflowError, ZeroDivisionError):
                    pass
        return self._node(POW, (a, b))

    _FOLD = {'sqrt': math.sqrt, 'sin': math.sin, 'cos': math.cos,
             'tan': math.tan, 'exp': math.exp, 'log': math.log,
             'abs': abs, 'asin': math.asin, 'acos': math.acos,
             'atan': math.atan, 'sinh': math.sinh, 'cosh': math.cosh,
             'tanh': math.tanh,
             'sign': lambda v: (v > 0) - (v < 0),
             'step': lambda v: 1.0 if v > 0 else 0.0,
             'erf': math.erf, 'erfc': math.erfc, 'floor': math.floor,
             'ceil': math.ceil, 'asinh': math.asinh, 'acosh': math.acosh,
             'atanh': math.atanh, 'log1p': math.log1p, 'expm1': math.expm1,
             'log2': math.log2, 'log10': math.log10,
             'tgamma': math.gamma, 'lgamma': math.lgamma}
    # (exp2 / cbrt of a constant are left to the device: Python 3.10 has no
    # correctly rounded counterpart of C's)

    def unary(self, name, a):
        assert name in UNARY, name
        if self.is_const(a) and name in self._FOLD:
            try:
                return self.const(self._FOLD[name](self.value(a)))
            except (ValueError, OverflowError):
                pass
        if self.op[a] == NEG:
            x = self.args[a][0]
            if name in ('sin', 'tan', 'asin', 'atan', 'sinh', 'tanh',
                        'sign', 'erf', 'asinh', 'atanh', 'cbrt'):
                return self.neg(self.unary(name, x))
            if name in ('cos', 'cosh', 'abs'):
                return self.unary(name, x)
        return self._node(name, (a,))

    def binary(self, name, a, b):
        assert name in (MAX, MIN, ATAN2)
        if self.is_const(a) and self.is_const(b):
            f = {MAX: max, MIN: min, ATAN2: math.atan2}[name]
            return self.const(f(self.value(a), self.value(b)))
        if name in (MAX, MIN) and a > b:
            a, b = b, a
        return self._node(name, (a, b))

    _REL = {'lt': lambda a, b: a < b, 'le': lambda a, b: a <= b,
            'eq': lambda a, b: a == b, 'ne': lambda a, b: a != b}

    def select(self, rel, a, b, x, y):
        """``(a rel b) ? x : y`` -- the branches of a Piecewise; both sides
        are evaluated (straight-line code), one is kept."""
        assert rel in RELATIONS, rel
        if x == y:
            return x
        if self.is_const(a) and self.is_const(b):
            return x if self._REL[rel](self.value(a), self.value(b)) else y
        return self._node(SELECT, (rel, a, b, x, y))

    def sum(self, terms):
        """Balanced (pairwise) sum: short dependency chains, and the pairwise
        association is also the numerically better one."""
        terms = [t for t in terms if t != self.zero]
        if not terms:
            return self.zero
        while len(terms) > 1:
            nxt = [self.add(terms[k], terms[k + 1])
                   for k in range(0, len(terms) - 1, 2)]
            if len(terms) % 2:
                nxt.append(terms[-1])
            terms = nxt
        return terms[0]

    def prod(self, factors):
        factors = list(factors)
        if not factors:
            return self.one
        while len(factors) > 1:
            nxt = [self.mul(factors[k], factors[k + 1])
                   for k in range(0, len(factors) - 1, 2)]
            if len(factors) % 2:
                nxt.append(factors[-1])
            factors = nxt
        return factors[0]

    # -- analysis ------------------------------------------------------------
    def operands(self, i):
        """Ids of the operand nodes of node ``i``."""
        op = self.op[i]
        if op in (CONST, INPUT):
            return ()
        if op == POWI:
            return (self.args[i][0],)
        if op == SELECT:
            return self.args[i][1:]
        return self.args[i]

    def reachable(self, roots):
        """Sorted list of node ids needed to compute ``roots``."""
        seen = set()
        stack = list(roots)
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i)
            stack.extend(self.operands(i))
        return sorted(seen)

    def is_uniform(self, i):
        """True when node ``i`` does not depend on any per-node input, i.e.
        it is node-invariant (depends on parameters / h only)."""
        return self.uni[i]

    def count_ops(self, roots):
        hist = {}
        for i in self.reachable(roots):
            op = self.op[i]
            if op not in (CONST, INPUT):
                hist[op] = hist.get(op, 0) + 1
        return hist
