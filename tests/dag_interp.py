"""Test-only NumPy interpreter for the codegen DAG.

Lets the CPU test-suite check the lowering / differentiation
(``opty_amd.codegen``) against the golden vectors without a GPU.  It is NOT a
backend: nothing in ``opty_amd`` imports it.
"""
import numpy as np

from opty_amd.codegen import ir

from opty_amd.codegen.errbound import (_UN, _REL, _DUN,  # noqa: F401
                                       evaluate_with_error_bound)


def evaluate(dag, roots, inputs):
    """``inputs(kind, index)`` -> scalar or (nodes,) array."""
    val = {}
    for i in dag.reachable(roots):
        op, a = dag.op[i], dag.args[i]
        if op == ir.CONST:
            v = a[0]
        elif op == ir.INPUT:
            v = inputs(*a)
        elif op == ir.ADD:
            v = val[a[0]] + val[a[1]]
        elif op == ir.SUB:
            v = val[a[0]] - val[a[1]]
        elif op == ir.MUL:
            v = val[a[0]]*val[a[1]]
        elif op == ir.DIV:
            v = val[a[0]]/val[a[1]]
        elif op == ir.NEG:
            v = -val[a[0]]
        elif op == ir.POWI:
            v = val[a[0]]**a[1]
        elif op == ir.POW:
            v = np.power(val[a[0]], val[a[1]])
        elif op == ir.MAX:
            v = np.maximum(val[a[0]], val[a[1]])
        elif op == ir.MIN:
            v = np.minimum(val[a[0]], val[a[1]])
        elif op == ir.ATAN2:
            v = np.arctan2(val[a[0]], val[a[1]])
        elif op == ir.SELECT:
            v = np.where(_REL[a[0]](val[a[1]], val[a[2]]), val[a[3]],
                         val[a[4]])
        else:
            v = _UN[op](np.asarray(val[a[0]], dtype=float))
        val[i] = v
    return [val[r] for r in roots]


def error_bounds(col, free, nodes=None):
    """Per-entry rounding-error bounds (units of round-off) of
    ``constraints(free)`` and ``jacobian(free)`` in the reference's layouts;
    ``nodes``: only these constraint nodes -> ``con (M, len(nodes))``,
    ``jac (len(nodes), P)`` plus the instance tails."""
    prog = col._build_program()
    inputs, count = _input_getter(col, free, nodes)
    ones = np.ones(count)
    _, ce = evaluate_with_error_bound(prog.dag, prog.con_out, inputs)
    _, je = evaluate_with_error_bound(prog.dag, prog.jac_out, inputs)
    _, ice = evaluate_with_error_bound(prog.dag, prog.inst_con_out, inputs)
    _, ije = evaluate_with_error_bound(prog.dag, prog.inst_jac_out, inputs)
    con = np.stack([np.atleast_1d(c)*ones for c in ce]) if ce \
        else np.zeros((0, count))
    jac = np.stack([np.atleast_1d(v)*ones for v in je], axis=1) if je \
        else np.zeros((count, 0))
    ic, ij = np.array(ice, dtype=float), np.array(ije, dtype=float)
    if nodes is None:
        return np.concatenate((con.ravel(), ic)), \
            np.concatenate((jac.ravel(), ij))
    return con, jac, ic, ij


def _input_getter(col, free, nodes=None):
    prog = col._build_program()
    N, n, q = col.num_collocation_nodes, prog.n, prog.q
    free = np.asarray(free, dtype=float)
    known = np.array([col.known_trajectory_map[f](free)
                      if callable(col.known_trajectory_map[f])
                      else col.known_trajectory_map[f]
                      for f in col.known_input_trajectories], dtype=float)
    tail = free[(n + q)*N:]
    kpar = [float(col.known_parameter_map[p]) for p in col.known_parameters]
    sel = slice(None) if nodes is None else np.asarray(nodes)

    def row(r):
        src, k = prog.rows[r]
        return free[k*N:(k + 1)*N] if src == 'free' else known[k]

    def inputs(kind, idx):
        if kind in ('cur', 'adj'):
            off = prog.cur_offset if kind == 'cur' else prog.adj_offset
            return row(idx)[off:off + N - 1][sel]
        if kind == 'par':
            src, k = prog.pars[idx]
            return kpar[k] if src == 'known' else tail[k]
        if kind == 'h':
            return col.node_time_interval if prog.h[0] == 'fixed' \
                else tail[prog.h[1]]
        if kind == 'free':
            f = col._inst_atoms[idx]
            return free[col.instance_constraints_free_index_map[f]]
        raise AssertionError(kind)

    return inputs, (N - 1 if nodes is None else len(sel))


def evaluate_collocator(col, free):
    """constraints(free), jacobian(free) of an ``opty_amd.ConstraintCollocator``
    through the interpreter (layouts as the reference's)."""
    prog = col._build_program()
    N, n, q = col.num_collocation_nodes, prog.n, prog.q
    free = np.asarray(free, dtype=float)
    known = np.array([col.known_trajectory_map[f](free)
                      if callable(col.known_trajectory_map[f])
                      else col.known_trajectory_map[f]
                      for f in col.known_input_trajectories], dtype=float)
    tail = free[(n + q)*N:]
    kpar = [float(col.known_parameter_map[p]) for p in col.known_parameters]

    def row(r):
        src, k = prog.rows[r]
        return free[k*N:(k + 1)*N] if src == 'free' else known[k]

    def inputs(kind, idx):
        if kind in ('cur', 'adj'):
            off = prog.cur_offset if kind == 'cur' else prog.adj_offset
            return row(idx)[off:off + N - 1]
        if kind == 'par':
            src, k = prog.pars[idx]
            return kpar[k] if src == 'known' else tail[k]
        if kind == 'h':
            return col.node_time_interval if prog.h[0] == 'fixed' \
                else tail[prog.h[1]]
        if kind == 'free':
            f = col._inst_atoms[idx]
            return free[col.instance_constraints_free_index_map[f]]
        raise AssertionError(kind)

    ones = np.ones(N - 1)
    con = evaluate(prog.dag, prog.con_out, inputs)
    con = np.concatenate([np.atleast_1d(c)*ones for c in con]) \
        if con else np.zeros(0)
    jac = evaluate(prog.dag, prog.jac_out, inputs)
    jac = np.stack([np.atleast_1d(v)*ones for v in jac], axis=1).ravel()
    ic = evaluate(prog.dag, prog.inst_con_out, inputs)
    ij = evaluate(prog.dag, prog.inst_jac_out, inputs)
    con = np.concatenate((con, np.array(ic, dtype=float)))
    jac = np.concatenate((jac, np.array(ij, dtype=float)))
    return con, jac
