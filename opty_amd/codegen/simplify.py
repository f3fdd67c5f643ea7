"""Algebraic clean-up of the expression DAG before it is scheduled: collect the
node-invariant coefficients of like terms.

SymPy hands over the equations of motion of a multibody system *expanded*:
the mass-matrix entry of an n-link pendulum arrives as
``l_i l_j m_j (s_i s_j + c_i c_j) + l_i l_j m_{j+1} (s_i s_j + c_i c_j) + ...``
-- one term per body outboard of the joint, every one the same per-node factor
times a different product of constants -- and forward-mode differentiation
(``lower.forward_jacobian``) reproduces that shape in every partial.  The
reference compiles exactly that (node-invariant
temporaries recomputed per node); evaluated SIMT-across-nodes it costs one
multiply, one add and one coefficient (a scalar load into an SGPR pair) per
term and per node.  Here sums are put into the normal form

    sum_g  C_g * (product of per-node factors of group g),   C_g = sum of the
                                                             group's constants

with ``C_g`` node-invariant, i.e. evaluated once per launch by ``opty_uni``
instead of once per node: for the 24-link stand-in of BASELINE config 5 this
removes most of the per-node arithmetic and of the coefficient traffic.

Only *private* intermediate nodes (used once) are flattened, so no shared
sub-expression is duplicated, and nothing is expanded: a product of two sums
stays a product.  The rewrite re-associates floating-point sums (as the
pairwise summation of ``ir.DAG.sum`` already does); the parity tolerance is
per-entry with a floor derived from each entry's own rounding-error bound
(``tests/dag_interp.py``).
"""

import sys

from . import ir

_SUM = (ir.ADD, ir.SUB, ir.NEG)


def collect_coefficients(dag, roots):
    """Returns new root ids (same order) of equivalent, collected
    expressions; new nodes are added to ``dag``."""
    uses = {}
    for i in dag.reachable(roots):
        for j in dag.operands(i):
            uses[j] = uses.get(j, 0) + 1
    for r in roots:
        uses[r] = uses.get(r, 0) + 1
    op, args, uni = dag.op, dag.args, dag.uni
    memo = {}

    def private(k):
        return uses.get(k, 0) == 1

    def factors_of(k):
        """Product rooted at k (through private MUL / NEG nodes):
        sign, node-invariant factors, per-node factors (old ids)."""
        sign, cu, cv = 1, [], []
        stack = [(k, True)]
        while stack:
            j, top = stack.pop()
            if uni[j]:
                cu.append(j)
            elif op[j] == ir.MUL and (top or private(j)):
                stack.append((args[j][1], False))
                stack.append((args[j][0], False))
            elif op[j] == ir.NEG and (top or private(j)):
                sign = -sign
                stack.append((args[j][0], False))
            else:
                cv.append(j)
        return sign, cu, cv

    def expand(i):
        """[(sign, node-invariant factor ids, per-node factor ids)] whose sum
        is node i."""
        out = []
        stack = [(i, 1, (), True)]
        while stack:
            k, s, coef, top = stack.pop()
            if uni[k]:
                out.append((s, coef + (k,), ()))
                continue
            o = op[k]
            if o in _SUM and (top or private(k)):
                a = args[k]
                if o == ir.ADD:
                    stack.append((a[1], s, coef, False))
                    stack.append((a[0], s, coef, False))
                elif o == ir.SUB:
                    stack.append((a[1], -s, coef, False))
                    stack.append((a[0], s, coef, False))
                else:
                    stack.append((a[0], -s, coef, False))
            elif o == ir.MUL and (top or private(k)):
                sg, cu, cv = factors_of(k)
                if (len(cv) == 1 and op[cv[0]] in _SUM and private(cv[0])):
                    # constant * (private sum): push the constant inside
                    stack.append((cv[0], s*sg, coef + tuple(cu), False))
                else:
                    out.append((s*sg, coef + tuple(cu), tuple(cv)))
            else:
                out.append((s, coef, (k,)))
        return out

    def rewrite(i):
        new = memo.get(i)
        if new is not None:
            return new
        o = op[i]
        if uni[i] or o in (ir.CONST, ir.INPUT):
            new = i
        elif o in _SUM or o == ir.MUL:
            groups = {}
            for s, coef, cv in expand(i):
                key = tuple(sorted(rewrite(v) for v in cv))
                groups.setdefault(key, []).append((s, coef))
            terms = []
            for key, parts in groups.items():
                pos = [dag.prod(c) for s, c in parts if s > 0]
                neg = [dag.prod(c) for s, c in parts if s < 0]
                coef = dag.sub(dag.sum(pos), dag.sum(neg))
                terms.append(dag.mul(coef, dag.prod(key)))
            new = dag.sum(terms)
        else:
            a = args[i]
            if o == ir.DIV:
                new = dag.div(rewrite(a[0]), rewrite(a[1]))
            elif o == ir.POWI:
                new = dag.powi(rewrite(a[0]), a[1])
            elif o == ir.POW:
                new = dag.pow(rewrite(a[0]), rewrite(a[1]))
            elif o in (ir.MAX, ir.MIN, ir.ATAN2):
                new = dag.binary(o, rewrite(a[0]), rewrite(a[1]))
            elif o == ir.SELECT:
                new = dag.select(a[0], *[rewrite(k) for k in a[1:]])
            else:
                new = dag.unary(o, rewrite(a[0]))
        memo[i] = new
        return new

    limit = sys.getrecursionlimit()
    sys.setrecursionlimit(max(limit, 20000))
    try:
        return [rewrite(r) for r in roots]
    finally:
        sys.setrecursionlimit(limit)
