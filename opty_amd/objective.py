"""Device-resident objective and objective gradient (SURVEY.md section 8(f),
rank 1): the HIP counterpart of ``create_objective_function``
(``opty/utils.py:329-470``).

The reference lambdifies the symbolic objective and its symbolic gradient and
replaces every ``Integral(g, t)`` by a quadrature over the collocation nodes:

* backward Euler: ``h * sum_i w_i g(node i)`` with ``w = [0, 1, ..., 1]``
  (``opty/utils.py:419-425``);
* midpoint: ``h * sum_i g((node i + node i+1)/2)`` over the N-1 midpoints for
  the objective and the parameter gradient, and -- for the gradient with
  respect to a trajectory value at node i -- ``h * w'_i dg/dz(node i)`` with
  ``w' = [1/2, 1, ..., 1, 1/2]`` evaluated at the NODE values
  (``opty/utils.py:440-464``).  That second rule is reproduced as is.

Here the integrand, its partials and the quadrature run on the GPU: one
elementwise + wave-reduction kernel (``opty_objgrad``: lane == time node,
coalesced loads of ``free`` and coalesced stores of the trajectory part of the
gradient, per-wave partial sums) and one single-wave kernel (``opty_objfin``)
that adds the partial sums in a fixed order (deterministic), applies ``h`` and
the parameter-only terms and writes the value and the parameter part of the
gradient.

Supported objectives are the ones the reference evaluates correctly: linear in
their integrals, ``sum_j a_j(p) * Integral(g_j(x, u, p), t) + b(p)``.
"""

import numpy as np
import sympy as sm
import sympy.physics.mechanics as me

from .utils import sort_sympy
from .codegen import ir
from .codegen.lower import Lowerer, forward_jacobian
from .codegen.emit_hip import _Body, KERNEL_PARAMS
from . import hip_backend as hb

__all__ = ['create_objective_function']


def _split_objective(objective, time_symbol, time_funcs):
    """-> integrand ``G`` (a_j folded in) and parameter-only remainder ``b``."""
    objective = sm.sympify(objective)
    integrals = sorted(objective.atoms(sm.Integral), key=sm.default_sort_key)
    dummies = []
    for integral in integrals:
        if integral.function.has(sm.Integral):
            raise NotImplementedError('Nested integrals are not supported.')
        if integral.limits != ((time_symbol,),):
            raise NotImplementedError('Only indefinite integrals of time are '
                                      'supported.')
        dummies.append(sm.Dummy('I%d' % len(dummies)))
    obj_d = objective.xreplace(dict(zip(integrals, dummies)))
    G = sm.S.Zero
    for j, dj in enumerate(dummies):
        a_j = obj_d.diff(dj)
        if any(a_j.has(dk) for dk in dummies):
            raise NotImplementedError(
                'The objective must be linear in its integrals (the '
                'reference evaluates anything else incorrectly).')
        if any(a_j.has(f) for f in time_funcs):
            raise NotImplementedError(
                'Factors outside an integral may only depend on the unknown '
                'parameters.')
        G += a_j*integrals[j].function
    b = sm.sympify(obj_d.xreplace({dj: sm.S.Zero for dj in dummies}))
    if any(b.has(f) for f in time_funcs):
        raise NotImplementedError('Terms outside an integral may only depend '
                                  'on the unknown parameters.')
    return G, b


def _emit(dag, n_rows, r, N_sym_unused, method, roots):
    """HIP source of ``opty_objgrad`` / ``opty_objfin``."""
    g_quad, dp_quad, dz_node, b_val, db_val = roots
    nq = 1 + r

    def leaf(i):
        if dag.op[i] != ir.INPUT:
            return None
        kind, k = dag.args[i]
        if kind == 'cur':
            return 'zc%d' % k
        if kind == 'adj':
            return 'za%d' % k
        if kind == 'par':
            return 'free_[%dLL*N + %d]' % (n_rows, k)
        raise AssertionError(kind)

    need = set(dag.reachable([g_quad] + dp_quad + dz_node))
    rows_c = sorted({dag.args[i][1] for i in need if dag.op[i] == ir.INPUT
                     and dag.args[i][0] == 'cur'})
    rows_a = sorted({dag.args[i][1] for i in need if dag.op[i] == ir.INPUT
                     and dag.args[i][0] == 'adj'})
    body = _Body(dag, need, leaf)
    lines = ['const int lane = threadIdx.x;',
             'const long long i = (long long)blockIdx.x*64 + lane;',
             'const bool in = i < N;',
             'const long long ic = in ? i : N - 1;',
             'const long long ia = ic + 1 < N ? ic + 1 : N - 1;']
    for k in rows_c:
        lines.append('const double zc%d = free_[%dLL*N + ic];' % (k, k))
    for k in rows_a:
        lines.append('const double za%d = free_[%dLL*N + ia];' % (k, k))
    if method == 'backward euler':
        lines.append('const double wq = (in && i > 0) ? 1.0 : 0.0;')
        lines.append('const double wg = wq;')
    else:
        lines.append('const double wq = (i < N - 1) ? 1.0 : 0.0;')
        lines.append('const double wg = !in ? 0.0 : '
                     '((i == 0 || i == N - 1) ? 0.5 : 1.0);')
    # trajectory part of the gradient: elementwise, coalesced
    for k, node in enumerate(dz_node):
        body.new_scope()
        ref = body.emit(node)
        body.lines.append('if (jac && in) jac[%dLL*N + i] = h*wg*%s;'
                          % (k, ref))
    body.new_scope()
    # quadrature terms: per-wave partial sums, fixed order
    qrefs = [body.emit(g_quad)] + [body.emit(nd) for nd in dp_quad]
    body.end_scope()
    lines += body.lines
    for j, ref in enumerate(qrefs):
        lines.append('double q%d = wq*%s;' % (j, ref))
    lines.append('#pragma unroll')
    lines.append('for (int off = 32; off > 0; off >>= 1) {')
    for j in range(nq):
        lines.append('    q%d += __shfl_down(q%d, off, 64);' % (j, j))
    lines.append('}')
    lines.append('if (lane == 0) {')
    for j in range(nq):
        lines.append('    con[(long long)blockIdx.x*%d + %d] = q%d;'
                     % (nq, j, j))
    lines.append('}')
    src = ['// generated by opty_amd.objective -- do not edit',
           '#include "opty_device.h"', '',
           'extern "C" __global__ void __launch_bounds__(64)',
           'opty_objgrad(%s)' % KERNEL_PARAMS, '{']
    src += ['    ' + ln for ln in lines] + ['}', '']

    # final reduction: one wave, partials summed in a fixed order
    ubody = _Body(dag, set(dag.reachable([b_val] + db_val)), leaf)
    bref = ubody.emit(b_val)
    drefs = [ubody.emit(nd) for nd in db_val]
    ubody.end_scope()
    fin = ['const int lane = threadIdx.x;',
           'const long long nblk = con_stride;']
    for j in range(nq):
        fin.append('double s%d = 0.0;' % j)
    fin.append('for (long long b = lane; b < nblk; b += 64) {')
    for j in range(nq):
        fin.append('    s%d += con[b*%d + %d];' % (j, nq, j))
    fin.append('}')
    fin.append('#pragma unroll')
    fin.append('for (int off = 32; off > 0; off >>= 1) {')
    for j in range(nq):
        fin.append('    s%d += __shfl_down(s%d, off, 64);' % (j, j))
    fin.append('}')
    fin.append('if (lane == 0) {')
    fin += ['    ' + ln for ln in ubody.lines]
    fin.append('    uni_w[0] = h*s0 + %s;' % bref)
    for k in range(r):
        fin.append('    if (jac) jac[%dLL*N + %d] = h*s%d + %s;'
                   % (n_rows, k, 1 + k, drefs[k]))
    fin.append('}')
    src += ['extern "C" __global__ void __launch_bounds__(64)',
            'opty_objfin(%s)' % KERNEL_PARAMS, '{']
    src += ['    ' + ln for ln in fin] + ['}', '']
    return '\n'.join(src)


def build_objective_program(objective, state_symbols,
                            unknown_input_trajectories, unknown_parameters,
                            integration_method='backward euler',
                            time_symbol=None):
    """Lowers the objective into a DAG.  Returns ``(dag, roots, n, q, r)``
    with ``roots = (g_quad, dp_quad, dz_node, b_val, db_val)``: the integrand
    at the quadrature points, its parameter partials there, its trajectory
    partials at the node values, the parameter-only remainder and its
    partials.  INPUT kinds: ``cur``/``adj`` = trajectory row at node i / i+1,
    ``par`` = unknown parameter (name-sorted, ``opty/utils.py:393-394``)."""
    if time_symbol is None:
        time_symbol = me.dynamicsymbols._t
    if integration_method not in ('backward euler', 'midpoint'):
        raise NotImplementedError(
            f"Integration method '{integration_method}' is not implemented.")
    states = list(state_symbols)
    inputs = sort_sympy(unknown_input_trajectories)
    params = sort_sympy(unknown_parameters)
    n, q, r = len(states), len(inputs), len(params)
    funcs = states + inputs
    G, b = _split_objective(objective, time_symbol, funcs)

    dag = ir.DAG()
    zc = [sm.Dummy('zc%d' % k, real=True) for k in range(n + q)]
    za = [sm.Dummy('za%d' % k, real=True) for k in range(n + q)]
    table = {s: dag.input('cur', k) for k, s in enumerate(zc)}
    table.update({s: dag.input('adj', k) for k, s in enumerate(za)})
    table.update({p: dag.input('par', k) for k, p in enumerate(params)})
    low = Lowerer(dag, table)
    G_node = low.lower(G.xreplace(dict(zip(funcs, zc))))
    wrt_nodes = [table[s] for s in zc] + [table[p] for p in params]
    grads = forward_jacobian(dag, [G_node], wrt_nodes)[0]
    dz_node, dp_node = grads[:n + q], grads[n + q:]
    if integration_method == 'backward euler':
        g_quad, dp_quad = G_node, dp_node
    else:
        at_mid = {f: (c + a)/2 for f, c, a in zip(funcs, zc, za)}
        g_quad = low.lower(G.xreplace(at_mid))
        dp_quad = forward_jacobian(dag, [g_quad],
                                   [table[p] for p in params])[0]
    b_val = low.lower(b)
    db_val = forward_jacobian(dag, [b_val], [table[p] for p in params])[0]
    return dag, (g_quad, dp_quad, dz_node, b_val, db_val), n, q, r


def compile_objective(objective, state_symbols, unknown_input_trajectories,
                      unknown_parameters, num_collocation_nodes,
                      integration_method='backward euler', time_symbol=None,
                      tmp_dir=None):
    """Lowers, prints and builds the objective kernels (no device needed):
    ``(code object path, (n, q, r))``."""
    dag, roots, n, q, r = build_objective_program(
        objective, state_symbols, unknown_input_trajectories,
        unknown_parameters, integration_method, time_symbol)
    source = _emit(dag, n + q, r, int(num_collocation_nodes),
                   integration_method, roots)
    return hb.compile_module(source, tmp_dir), (n, q, r)


def create_objective_function(objective, state_symbols,
                              unknown_input_trajectories, unknown_parameters,
                              num_collocation_nodes, node_time_interval,
                              integration_method='backward euler',
                              time_symbol=None, device=0, tmp_dir=None):
    """Returns ``(obj, obj_grad)`` evaluated on the GPU; same arguments and
    return contract as the reference's ``create_objective_function``
    (``opty/utils.py:329-364``): ``obj(free) -> float``,
    ``obj_grad(free) -> ndarray (n*N + q*N + r,)``.

    Both callables also accept a ``torch`` CUDA tensor for ``free``;
    ``obj_grad`` then returns a CUDA tensor (nothing crosses PCIe).
    """
    hsaco, (n, q, r) = compile_objective(
        objective, state_symbols, unknown_input_trajectories,
        unknown_parameters, num_collocation_nodes, integration_method,
        time_symbol, tmp_dir)
    N = int(num_collocation_nodes)
    handle = hb.HipObjective(dict(N=N, n=n, q=q, r=r, device=int(device),
                                  h=float(node_time_interval)), hsaco)
    num_free = (n + q)*N + r

    def _as_input(free):
        if hasattr(free, 'data_ptr'):
            if tuple(free.shape) != (num_free,):
                raise ValueError('free must have shape (%d,)' % num_free)
            # ordered behind whatever produced the tensor on torch's stream
            handle.use_torch_stream()
            return free, hb.DEVICE
        free = np.ascontiguousarray(free, dtype=np.float64)
        if free.shape != (num_free,):
            raise ValueError('free must have shape (%d,)' % num_free)
        return free, hb.HOST

    def obj(free):
        free, mem = _as_input(free)
        return handle.evaluate(free, None, mem)

    def obj_grad(free):
        free, mem = _as_input(free)
        if mem == hb.DEVICE:
            import torch
            grad = torch.empty(num_free, dtype=torch.float64,
                               device=free.device)
        else:
            grad = np.empty(num_free)
        handle.evaluate(free, grad, mem)
        return grad

    obj.handle = obj_grad.handle = handle
    return obj, obj_grad
