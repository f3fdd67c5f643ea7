"""The DAG as a flat instruction tape for ``opty_tape_kernel``.

Why this exists: hipcc 7.2 miscompiles some generated kernels that sit at the
edge of the register file -- deterministically wrong values in
whole strips, with or without vector spills, sometimes identically in two
kernels of the same build.  Comparing compiled builds with each other cannot
settle which one is right.  The tape is evaluated ON THE GPU by one small
hand-written kernel of ``libopty_hip.so`` (``opty_hip_tape_run``: one lane per
node, every value in HBM, a handful of registers -- nothing for a register
allocator to get wrong, and the same device math library as the generated
code), so its results are what the expression DAG *means*; a code object whose
kernels disagree with it on the verification nodes is never handed out
(``ConstraintCollocator._verify_build``).  It referees builds; it never
supplies the values a caller sees.

Encoding: ``TAPE_WORDS`` int32 per instruction -- ``op, dst, a, b, c, d, imm,
0`` -- over a value table ``vals[slot*nodes + node]`` whose slots are DAG node
ids renumbered densely; CONST and INPUT slots are filled by the host before
the run.
"""
import numpy as np

from . import ir

TAPE_WORDS = 8
#: opcodes shared with csrc/referee.cpp (enum TapeOp)
T_ADD, T_SUB, T_MUL, T_DIV, T_NEG, T_POWI, T_POW, T_MAX, T_MIN, T_ATAN2, \
    T_SELECT = range(11)
T_UNARY0 = 16               # + index into ir.UNARY
_BINARY = {ir.ADD: T_ADD, ir.SUB: T_SUB, ir.MUL: T_MUL, ir.DIV: T_DIV,
           ir.POW: T_POW, ir.MAX: T_MAX, ir.MIN: T_MIN, ir.ATAN2: T_ATAN2}
_REL = {name: k for k, name in enumerate(ir.RELATIONS)}


class Tape(object):
    """``code`` (ninstr, TAPE_WORDS) int32; ``slot`` DAG node -> value slot;
    ``consts`` [(slot, value)]; ``inputs`` [(slot, kind, index)];
    ``nslots``."""

    def __init__(self, dag, roots):
        need = dag.reachable(set(roots))
        self.slot = {i: s for s, i in enumerate(need)}
        self.nslots = len(need)
        self.consts, self.inputs, code = [], [], []
        sl = self.slot
        for i in need:
            op, a = dag.op[i], dag.args[i]
            if op == ir.CONST:
                self.consts.append((sl[i], float(a[0])))
            elif op == ir.INPUT:
                self.inputs.append((sl[i], a[0], a[1]))
            elif op in _BINARY:
                code.append((_BINARY[op], sl[i], sl[a[0]], sl[a[1]],
                             0, 0, 0, 0))
            elif op == ir.NEG:
                code.append((T_NEG, sl[i], sl[a[0]], 0, 0, 0, 0, 0))
            elif op == ir.POWI:
                code.append((T_POWI, sl[i], sl[a[0]], 0, 0, 0, int(a[1]), 0))
            elif op == ir.SELECT:
                code.append((T_SELECT, sl[i], sl[a[1]], sl[a[2]], sl[a[3]],
                             sl[a[4]], _REL[a[0]], 0))
            else:
                code.append((T_UNARY0 + ir.UNARY.index(op), sl[i], sl[a[0]],
                             0, 0, 0, 0, 0))
        self.code = np.array(code, dtype=np.int32).reshape(-1, TAPE_WORDS)

    def table(self, nodes, inputs):
        """Value table ``(nslots, nodes)`` with the CONST and INPUT slots
        filled; ``inputs(kind, index)`` -> scalar or ``(nodes,)`` array."""
        vals = np.zeros((self.nslots, int(nodes)))
        for s, v in self.consts:
            vals[s] = v
        for s, kind, idx in self.inputs:
            vals[s] = inputs(kind, idx)
        return vals
