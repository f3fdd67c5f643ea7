"""Builds the per-problem *collocation program*: one DAG holding the M
discretised defect equations, their M x C analytic partials and the instance
constraints, plus the tables that say where every DAG input lives in device
memory.

This is the HIP backend's counterpart of what the reference assembles inside
``_gen_multi_arg_con_func`` / ``_gen_multi_arg_con_jac_func``
(``opty/direct_collocation.py:2304-2380``, ``:2692-2805``): the argument list
(``args``), the differentiation variables (``wrt``) and the expressions handed
to ``ufuncify_matrix``.
"""

from . import ir
import os

from .lower import Lowerer, forward_jacobian
from .simplify import collect_coefficients


def _collect(dag, con_out, jac):
    """Coefficient collection (``simplify.py``) over the defect equations and
    all their partials at once, so that sharing between them is seen.
    ``OPTY_COLLECT=0`` turns it off (A/B measurements)."""
    if os.environ.get('OPTY_COLLECT', '1') == '0':
        return con_out, jac
    width = len(jac[0]) if jac else 0
    new = collect_coefficients(dag, list(con_out) +
                               [node for row in jac for node in row])
    M = len(con_out)
    return new[:M], [new[M + j*width:M + (j + 1)*width]
                     for j in range(len(jac))]


class CollocationProgram(object):
    """Plain data; consumed by :mod:`opty_amd.codegen.emit_hip`.

    ``layout``: 'coo' or 'csr'; ``row_start[j]`` = first stored entry of
    equation j in ``jac_out`` when the entries are grouped by row (always true
    for 'csr' and for the unpruned 'coo' block).

    Attributes
    ----------
    dag : ir.DAG
    con_out : list of M node ids (defect equations)
    jac_out : list of M*C node ids, row-major ``[j*C + k]``
    inst_con_out / inst_jac_out : node ids of the instance constraints and of
        their partials (in the order of ``jacobian_indices``' tail)
    rows : list, one entry per trajectory row ``('free', k)`` (row ``k`` of
        the ``free`` vector viewed as ``(n+q, N)``) or ``('known', j)``
    cur_offset / adj_offset : time-node offset of the current / adjacent
        value relative to the constraint node (BE: 1/0, midpoint: 0/1)
    pars : list, one entry per parameter ``('known', k)`` or ``('tail', j)``
        (``free[(n+q)*N + j]``)
    h : ``('fixed',)`` or ``('tail', r)``
    """

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def P(self):
        """Values stored per constraint node: M*C for the reference's dense
        block, fewer when structural zeros are pruned."""
        return len(self.jac_out)


def build_program(discrete_eom, state_cur, state_adj, traj_cur, traj_adj,
                  num_known_traj, parameters, num_known_par, h_sym,
                  variable_duration, wrt, method, instance=None,
                  implicit=(), prune_zeros=False, layout='coo'):
    """Lowers the discretised equations and differentiates them.

    Parameters mirror the reference's locals: ``state_cur``/``state_adj`` are
    the ``xi`` and ``xp`` (backward Euler) or ``xn`` (midpoint) symbols,
    ``traj_cur``/``traj_adj`` the ``si``/``sn`` symbols of *all* m input
    trajectories (known first), ``parameters`` known-then-unknown, ``wrt`` the
    column order of ``opty/direct_collocation.py:2719-2737``.

    ``implicit``: ``[(k, state, kd)]`` -- input trajectory ``k`` is a known
    function ``r(x_state(t))`` whose derivative ``dr/dx`` is input trajectory
    ``kd`` (``opty/direct_collocation.py:2080-2093``): its discrete symbol is an
    applied function ``r_i(x_i)``, lowered as a plain input row that carries a
    chain-rule link for the Jacobian.

    ``layout``: ``'coo'`` stores the values in the reference's order (node
    major, ``opty/direct_collocation.py:2644-2675``); ``'csr'`` stores them
    sorted by constraint row, then column -- equation-major rows
    ``j*(N-1) + i``, each row's entries in ascending free index -- the order a
    compressed-sparse-row consumer needs (opt-in: changes the index contract).

    ``instance``: optional ``(expressions, atom_symbols, known_par_syms)`` --
    instance constraints written over one placeholder Symbol per function
    atom; lowered into the same DAG with INPUT kind ``'free'``.
    """
    dag = ir.DAG()
    n = len(state_cur)
    m = len(traj_cur)
    q = m - num_known_traj
    r = len(parameters) - num_known_par
    table = {}
    for k, s in enumerate(state_cur):
        table[s] = dag.input('cur', k)
    for k, s in enumerate(state_adj):
        table[s] = dag.input('adj', k)
    for k, s in enumerate(traj_cur):
        table[s] = dag.input('cur', n + k)
    if method == 'midpoint':
        for k, s in enumerate(traj_adj):
            table[s] = dag.input('adj', n + k)
    for k, s in enumerate(parameters):
        table[s] = dag.input('par', k)
    table[h_sym] = dag.input('h', 0)

    chain = {}
    for k, st, kd in implicit:
        chain[dag.input('cur', n + k)] = [(dag.input('cur', st),
                                           dag.input('cur', n + kd))]
        if method == 'midpoint':
            chain[dag.input('adj', n + k)] = [(dag.input('adj', st),
                                               dag.input('adj', n + kd))]

    low = Lowerer(dag, table)
    con_out = [low.lower(e) for e in discrete_eom]
    wrt_nodes = [table[s] for s in wrt]
    jac = forward_jacobian(dag, con_out, wrt_nodes, chain)
    con_out, jac = _collect(dag, con_out, jac)
    # (j, k) of every stored entry of the block, row-major.  The reference
    # stores all M*C of them, structural zeros included
    # (opty/direct_collocation.py:2589-2593); ``prune_zeros`` (opt-in, changes
    # the index contract) keeps only entries whose partial is not identically
    # zero.
    pattern = [(j, k) for j, row in enumerate(jac) for k, node in
               enumerate(row) if not (prune_zeros and node == dag.zero)]
    if layout == 'csr':
        # the column of wrt entry k of node i is monotone in this key for
        # every i (N >= 2): states interleave adjacent/current, then inputs,
        # then the parameter tail
        key = _column_key(n, q, method)
        pattern.sort(key=lambda jk: (jk[0], key(jk[1])))
    elif layout != 'coo':
        raise ValueError('layout must be "coo" or "csr".')
    jac_out = [jac[j][k] for j, k in pattern]
    row_start = [0]*(len(con_out) + 1)
    for j, _ in pattern:
        row_start[j + 1] += 1
    for j in range(len(con_out)):
        row_start[j + 1] += row_start[j]

    # row r of the slab: states then unknown inputs come from `free`
    # (``free`` viewed as (n+q, N), opty/utils.py:308-318); known inputs from
    # the known-trajectory buffer.  input_trajectories = known + unknown.
    rows = [('free', k) for k in range(n)]
    rows += [('known', j) for j in range(num_known_traj)]
    rows += [('free', n + j) for j in range(q)]
    pars = [('known', k) for k in range(num_known_par)]
    pars += [('tail', j) for j in range(r)]
    h = ('tail', r) if variable_duration else ('fixed',)

    inst_con_out, inst_jac_out, num_atoms = [], [], 0
    if instance is not None:
        exprs, atom_syms, grads = instance
        itable = {s: dag.input('free', a) for a, s in enumerate(atom_syms)}
        for k, s in enumerate(parameters[:num_known_par]):
            itable[s] = dag.input('par', k)
        ilow = Lowerer(dag, itable)
        num_atoms = len(atom_syms)
        for e, atoms in zip(exprs, grads):
            node = ilow.lower(e)
            inst_con_out.append(node)
            if atoms:
                g = forward_jacobian(dag, [node], [itable[s] for s in atoms])
                inst_jac_out += g[0]

    return CollocationProgram(
        dag=dag, con_out=con_out, jac_out=jac_out, n=n, m=m, q=q, r=r,
        s=int(variable_duration), M=len(con_out), C=len(wrt),
        num_known_traj=num_known_traj, num_known_par=num_known_par,
        rows=rows, pars=pars, h=h, method=method,
        cur_offset=1 if method == 'backward euler' else 0,
        adj_offset=0 if method == 'backward euler' else 1,
        inst_con_out=inst_con_out, inst_jac_out=inst_jac_out,
        num_inst_atoms=num_atoms, pattern=pattern,
        pruned=bool(prune_zeros), layout=layout, row_start=row_start)


def matrix_program(dag, outputs, num_vec, num_const, shape):
    """Program of a plain matrix of expressions (the reference's
    ``ufuncify_matrix`` call shape, ``opty/utils.py:639-640``): ``outputs`` are
    the DAG nodes of the ``rows x cols`` entries, row-major, over the inputs
    ``('cur', k)`` -- vector argument ``k`` (one value per evaluation row) --
    and ``('par', k)`` -- constant argument ``k``.  The vector arguments are
    the rows of one packed ``(num_vec, n)`` buffer that the kernels see as
    their ``free`` vector with ``N = n``."""
    rows, cols = shape
    assert len(outputs) == rows*cols
    return CollocationProgram(
        dag=dag, con_out=[], jac_out=list(outputs), n=num_vec, m=0, q=0, r=0,
        s=0, M=rows, C=cols, num_known_traj=0, num_known_par=num_const,
        rows=[('free', k) for k in range(num_vec)],
        pars=[('known', k) for k in range(num_const)], h=('fixed',),
        method='matrix', cur_offset=0, adj_offset=0, inst_con_out=[],
        inst_jac_out=[], num_inst_atoms=0,
        pattern=[(j, k) for j in range(rows) for k in range(cols)],
        pruned=False, layout='coo',
        row_start=[j*cols for j in range(rows + 1)])


def _column_key(n, q, method):
    """Sort key of wrt index k equal to the order of the free-vector column
    it differentiates with respect to (the closed form of
    ``opty/direct_collocation.py:2657-2675`` at a generic node)."""
    big = 1 << 20           # stands for N; node i = 1

    def key(k):
        if method == 'backward euler':
            if k < n:
                return k*big + 2
            if k < 2*n:
                return (k - n)*big + 1
            if k < 2*n + q:
                return (n + k - 2*n)*big + 2
            return (n + q)*big + (k - 2*n - q)
        if k < n:
            return k*big + 1
        if k < 2*n:
            return (k - n)*big + 2
        if k < 2*n + q:
            return (n + k - 2*n)*big + 1
        if k < 2*n + 2*q:
            return (n + k - 2*n - q)*big + 2
        return (n + q)*big + (k - 2*n - 2*q)
    return key


def _static_tester(prog):
    """``is_static(node)``: the node's value is the same at every node and
    every call with the same known parameters and (fixed) interval."""
    dag = prog.dag
    static = {}

    def is_static(root):
        stack = [root]
        while stack:
            i = stack.pop()
            if i in static:
                if not static[i]:
                    static[root] = False
                    return False
                continue
            if dag.op[i] == ir.INPUT:
                kind, idx = dag.args[i]
                ok = ((kind == 'par' and prog.pars[idx][0] == 'known') or
                      (kind == 'h' and prog.h[0] == 'fixed'))
                static[i] = ok
                if not ok:
                    static[root] = False
                    return False
                continue
            if not dag.uni[i]:
                static[i] = False
                static[root] = False
                return False
            stack.extend(dag.operands(i))
        static[root] = True
        return True
    return is_static


def varying_entries(prog):
    """Stored block entries whose value can differ between two evaluations
    with the same known parameters and (fixed) node time interval: those that
    depend on a trajectory value, an unknown parameter or a free interval.
    The rest -- for the 10-link pendulum 660 of 990: the reference's
    structural zeros, +-1, +-1/h, products of masses and lengths
    (``opty/direct_collocation.py:2589-2593`` keeps them all in the value
    vector) -- are the same at every node and every call."""
    is_static = _static_tester(prog)
    return [e for e, node in enumerate(prog.jac_out) if not is_static(node)]


def varying_copies(prog):
    """Splits :func:`varying_entries` into the entries that have to be
    evaluated / moved and the ones that are *the same expression* as an
    earlier one (the same node of the hash-consed DAG: the mass matrix of a
    multibody system is symmetric, and its entries appear in the columns of
    the current and of the adjacent node's speeds).  Returns ``(unique,
    copies)``: ``unique`` ascending entry numbers, ``copies`` a list of
    ``(dst, src)`` with ``src`` in ``unique`` -- for the 10-link pendulum 275
    unique entries and 55 copies of the 330 varying ones."""
    first = {}
    unique, copies = [], []
    for e in varying_entries(prog):
        node = prog.jac_out[e]
        if node in first:
            copies.append((e, first[node]))
        else:
            first[node] = e
            unique.append(e)
    return unique, copies


def scaled_copies(prog):
    """:func:`varying_copies` taken one step further: varying entries that
    are a NODE-INVARIANT multiple of another varying entry -- ``c_1 X`` and
    ``c_2 X`` with the same per-node expression ``X`` and factors that depend
    on known parameters / the fixed interval only (a mass-matrix partial in
    the current node's column and, differently scaled, in another row) -- need
    not both cross PCIe: the host fills ``dst = (c_dst/c_src) src``.

    Returns ``(unique, copies)``: ``unique`` ascending entry numbers that are
    moved; ``copies`` a list of ``(dst, src, num, den)`` sorted by ``dst``,
    ``src`` in ``unique``, ``num`` / ``den`` the factor chains of dst / src --
    lists of ``('neg',)``, ``('mul', node)``, ``('div', node)`` over static
    DAG nodes, so that ``dst = prod(num)/prod(den) * src`` (:func:`chain_value`
    evaluates them for given known values; exact duplicates have two empty
    chains).  For the 10-link pendulum: 275 -> 269 moved entries (the
    numerical rank of the 275 as functions of the node values is 266: what
    ANY linear reconstruction could reach)."""
    dag = prog.dag
    is_static = _static_tester(prog)

    def strip(i):
        chain = []
        while True:
            op = dag.op[i]
            if op == ir.NEG:
                chain.append(('neg',))
                i = dag.args[i][0]
                continue
            if op == ir.MUL:
                a, b = dag.args[i]
                if is_static(a) and not is_static(b):
                    chain.append(('mul', a))
                    i = b
                    continue
                if is_static(b) and not is_static(a):
                    chain.append(('mul', b))
                    i = a
                    continue
            if op == ir.DIV:
                a, b = dag.args[i]
                if is_static(b) and not is_static(a):
                    chain.append(('div', b))
                    i = a
                    continue
            return chain, i

    base = {}                  # core node -> (entry, its chain)
    unique, copies = [], []
    for e in varying_entries(prog):
        chain, core = strip(prog.jac_out[e])
        if core in base:
            src, den = base[core]
            copies.append((e, src, chain, den))
        else:
            base[core] = (e, chain)
            unique.append(e)
    return unique, copies


def chain_value(chain, values):
    """Value of a factor chain of :func:`scaled_copies` given ``values``
    (static DAG node -> float)."""
    v = 1.0
    for step in chain:
        if step[0] == 'neg':
            v = -v
        elif step[0] == 'mul':
            v *= values[step[1]]
        else:
            v /= values[step[1]]
    return v

