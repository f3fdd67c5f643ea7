"""Test-only NumPy semantics of the instruction tape (``opty_amd.codegen.tape``)
that ``opty_tape_kernel`` executes on the GPU: the device kernel is held to
this instruction by instruction.  Nothing in ``opty_amd`` imports it."""
import numpy as np

from opty_amd.codegen import ir
from opty_amd.codegen.tape import (T_ADD, T_SUB, T_MUL, T_DIV, T_NEG, T_POWI,
                                   T_POW, T_MAX, T_MIN, T_ATAN2, T_SELECT,
                                   T_UNARY0)


def run_on_host(tape, vals):
    """The tape's semantics in NumPy (tests: the device kernel is held to
    this, instruction by instruction).  Fills ``vals`` in place."""
    un = {'abs': np.abs, 'sign': np.sign,
          'step': lambda x: (x > 0).astype(float), 'asin': np.arcsin,
          'acos': np.arccos, 'atan': np.arctan, 'asinh': np.arcsinh,
          'acosh': np.arccosh, 'atanh': np.arctanh}
    rel = [np.less, np.less_equal, np.equal, np.not_equal]
    with np.errstate(all='ignore'):
        for op, dst, a, b, c, d, imm, _ in tape.code.tolist():
            x = vals[a]
            if op == T_ADD:
                v = x + vals[b]
            elif op == T_SUB:
                v = x - vals[b]
            elif op == T_MUL:
                v = x*vals[b]
            elif op == T_DIV:
                v = x/vals[b]
            elif op == T_NEG:
                v = -x
            elif op == T_POWI:
                v = x**imm
            elif op == T_POW:
                v = np.power(x, vals[b])
            elif op == T_MAX:
                v = np.fmax(x, vals[b])
            elif op == T_MIN:
                v = np.fmin(x, vals[b])
            elif op == T_ATAN2:
                v = np.arctan2(x, vals[b])
            elif op == T_SELECT:
                v = np.where(rel[imm](x, vals[b]), vals[c], vals[d])
            else:
                name = ir.UNARY[op - T_UNARY0]
                if name in ('erf', 'erfc', 'tgamma', 'lgamma'):
                    from scipy import special
                    f = {'erf': special.erf, 'erfc': special.erfc,
                         'tgamma': special.gamma,
                         'lgamma': special.gammaln}[name]
                else:
                    f = un.get(name) or getattr(np, name)
                v = f(x)
            vals[dst] = v
    return vals
