"""The reference's plugin call shape on the GPU: ``ufuncify_matrix(args, expr,
const=...) -> f(result, *num_args)`` (``opty/utils.py:639-640``) and the two
multi-argument closures built on it (``opty/direct_collocation.py:2382-2446``,
``:2816-2887``).  Restates ``opty/tests/test_utils.py:244-336`` and
``opty/tests/test_direct_collocation.py:791-966``; the expected values of the
first come from the reference's own ``ufuncify_matrix``
(``tests/golden/ufuncify_matrix.npz``, ``_gen/make_golden.py``)."""
import os

import numpy as np
import pytest
import sympy as sm

import golden_util as gu
from examples import problems
from opty_amd.utils import coo_to_dense

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                      'ufuncify_matrix.npz')


def _case():
    a, b, c, d, I, i = sm.symbols('a, b, if, d_{badsym}, I, i')
    mat = sm.Matrix([[a**2*sm.cos(sm.pi*b)**c,
                      sm.tan(b)/sm.sin(a + b) + c**4],
                     [a**2 + b**2 - sm.sqrt(c),
                      ((a + b + c)*(a + b))/a*sm.sin(b)]])
    z = np.load(GOLDEN)
    n = 10000
    s = [int(k) for k in z['seeds']]
    a_vals = 0.5*(problems.make_free(n, seed=s[0]) + 1.0) + 2.0**-21
    b_vals = 0.5*(problems.make_free(n, seed=s[1]) + 1.0) + 2.0**-21
    c_vals = 0.5*(problems.make_free(n, seed=s[2]) + 1.0) + 10.0
    np.testing.assert_array_equal(a_vals[z['rows']], z['a_rows'])
    return (a, b, c, d, I, i), mat, (a_vals, b_vals, c_vals), z


def _numpy_loop(a, b, c):
    out = np.empty((len(a), 2, 2))
    out[:, 0, 0] = a**2*np.cos(np.pi*b)**c
    out[:, 0, 1] = np.tan(b)/np.sin(a + b) + c**4
    out[:, 1, 0] = a**2 + b**2 - np.sqrt(c)
    out[:, 1, 1] = ((a + b + c)*(a + b))/a*np.sin(b)
    return out


def _assert_same(out, expected, rtol=1e-10):
    """``cos(pi*b)**c`` is NaN wherever the base is negative (the reference's
    test compares those as equal NaNs too): same NaN pattern, finite entries
    within ``rtol`` relative."""
    out, expected = np.asarray(out), np.asarray(expected)
    assert out.shape == expected.shape
    nan = np.isnan(expected)
    assert nan.any() and not nan.all()
    np.testing.assert_array_equal(np.isnan(out), nan)
    gu.assert_close(out[~nan], expected[~nan], rtol, scale=0.0,
                    what='ufuncify_matrix 2x2')


def test_matrix_module_builds_for_gfx950():
    """CPU: the 2 x 2 case lowers, prints and cross-compiles; a const
    argument's sub-expressions go to the node-invariant table."""
    from opty_amd.codegen import ir
    from opty_amd.codegen.lower import Lowerer
    from opty_amd.codegen.program import matrix_program
    from opty_amd.codegen.emit_hip import emit_matrix_module
    from opty_amd import hip_backend as hb
    (a, b, c, *_), mat, _, _ = _case()
    dag = ir.DAG()
    table = {a: dag.input('cur', 0), b: dag.input('cur', 1),
             c: dag.input('par', 0)}
    low = Lowerer(dag, table)
    prog = matrix_program(dag, [low.lower(e) for e in mat], 2, 1, (2, 2))
    src, meta = emit_matrix_module(prog)
    assert 'opty_jac' in src and 'opty_conjac' not in src
    assert meta['num_uniform'] == 3        # if**4, sqrt(if) and if itself
    assert os.path.exists(hb.compile_module(src))


def test_compile_failure_raises_import_error_with_stderr(tmp_path):
    """``opty/utils.py:912-916``: a build failure is an ``ImportError`` that
    carries the compiler's stderr (CPU test: hipcc cross-compiles)."""
    from opty_amd import hip_backend as hb
    bad = ('#include "opty_device.h"\n'
           'extern "C" __global__ void opty_jac(double *x)\n'
           '{ x[0] = this_symbol_does_not_exist; }\n')
    with pytest.raises(ImportError) as err:
        hb.compile_module(bad, str(tmp_path))
    assert err.match('Unable to build the HIP code object')
    assert err.match('this_symbol_does_not_exist')
    assert err.match('STDERR output from compilation')
    # nothing half-built is left behind for the cache to pick up
    assert not [f for f in os.listdir(str(tmp_path)) if f.endswith('.hsaco')]


@pytest.mark.gpu
def test_ufuncify_matrix_reference_case():
    from opty_amd import ufuncify_matrix
    (a, b, c, d, I, i), mat, (a_vals, b_vals, c_vals), z = _case()
    rows, n = z['rows'], len(a_vals)
    c_val = float(z['c_const'][0])
    with np.errstate(invalid='ignore'):
        expected = _numpy_loop(a_vals, b_vals, c_vals)

    f = ufuncify_matrix((a, b, c), mat)
    result = np.empty((n, 4))
    out = f(result, a_vals, b_vals, c_vals)
    assert out.shape == (n, 2, 2) and out.base is result
    # vs the reference's own compiled function (recorded rows), then vs NumPy
    # on every row as the reference's test does
    _assert_same(out[rows], z["result_vec"])
    _assert_same(out, expected)

    f = ufuncify_matrix((a, b, c), mat, const=(c,))
    out = f(np.empty((n, 4)), a_vals, b_vals, c_val)
    _assert_same(out[rows], z["result_const"])
    with np.errstate(invalid='ignore'):
        _assert_same(out, _numpy_loop(a_vals, b_vals, c_val))
        # a new const value on the same function refreshes the invariant
        # table
        out = f(np.empty((n, 4)), a_vals, b_vals, 11.25)
        _assert_same(out, _numpy_loop(a_vals, b_vals, 11.25))

        f = ufuncify_matrix((a, b, c), mat, const=(c,), parallel=True)
        _assert_same(f(np.empty((n, 4)), a_vals, b_vals, c_val),
                     _numpy_loop(a_vals, b_vals, c_val))
    # symbols named I / i / d_{badsym}: names never reach the generated code
    for other in (I, i, d):
        f = ufuncify_matrix((a, b, other), mat.xreplace({c: other}))
        _assert_same(f(np.empty((n, 4)), a_vals, b_vals, c_vals), expected)
    # the 2-tuple cse() returns (opty/utils.py:677-682)
    f = ufuncify_matrix((a, b, c), sm.cse(mat))
    _assert_same(f(np.empty((n, 4)), a_vals, b_vals, c_vals), expected)


@pytest.mark.gpu
def test_ufuncify_matrix_argument_contract():
    """``opty/utils.py:610-617, 778-807``: result C-contiguous float64
    ``(n, rows*cols)``, vector arguments contiguous float64 ``(n,)``."""
    import torch
    from opty_amd import ufuncify_matrix
    x, y, k = sm.symbols('x, y, k')
    mat = sm.Matrix([[x*y + k, sm.sin(x), 3], [y, 0, k**2], [x - y, 1, x]])
    f = ufuncify_matrix((x, k, y), mat, const=(k,))      # const in the middle
    for n in (1, 63, 64, 65, 1000):
        xv = np.linspace(-1.0, 1.0, n)
        yv = np.cos(np.arange(n, dtype=float))
        out = f(np.empty((n, 9)), xv, 0.75, yv)
        want = np.empty((n, 3, 3))
        want[:, 0] = np.stack([xv*yv + 0.75, np.sin(xv), 3 + 0*xv], axis=1)
        want[:, 1] = np.stack([yv, 0*xv, 0.75**2 + 0*xv], axis=1)
        want[:, 2] = np.stack([xv - yv, 1 + 0*xv, xv], axis=1)
        np.testing.assert_allclose(out, want, rtol=1e-14, atol=1e-15)
    n = 100
    xv, yv = np.ones(n), np.ones(n)
    with pytest.raises(ValueError):
        f(np.empty((n, 8)), xv, 1.0, yv)                 # wrong width
    with pytest.raises(ValueError):
        f(np.empty((n, 9), dtype=np.float32), xv, 1.0, yv)
    with pytest.raises(ValueError):
        f(np.empty((9, n)).T, xv, 1.0, yv)               # not C-contiguous
    with pytest.raises(ValueError):
        f(np.empty((n, 9)), xv[:-1], 1.0, yv)            # wrong length
    with pytest.raises(ValueError):
        f(np.empty((n, 9)), np.ones(2*n)[::2], 1.0, yv)  # strided
    with pytest.raises(ValueError):
        f(np.empty((n, 9)), xv.astype(np.float32), 1.0, yv)
    with pytest.raises(TypeError):
        f(np.empty((n, 9)), xv, 1.0)
    with pytest.raises(ValueError):
        ufuncify_matrix((x, y), mat, const=(k,))
    # device-resident call: CUDA tensors in, evaluated in place
    dev = torch.device('cuda:0')
    f.hip.use_torch_stream()
    xt = torch.linspace(-1, 1, 777, dtype=torch.float64, device=dev)
    yt = torch.cos(xt)
    res = torch.empty((777, 9), dtype=torch.float64, device=dev)
    out = f(res, xt, 0.5, yt)
    torch.cuda.synchronize()
    assert out.shape == (777, 3, 3)
    np.testing.assert_allclose(out[:, 0, 0].cpu().numpy(),
                               (xt*yt + 0.5).cpu().numpy(), rtol=1e-13,
                               atol=1e-15)
    f.hip.set_stream(None)


def _msd():
    """Fixture of the reference's ``TestConstraintCollocator.setup_method``
    (``opty/tests/test_direct_collocation.py:658-700``)."""
    import opty_amd
    m, c, k, t = sm.symbols('m, c, k, t')
    x, v, f = [s(t) for s in sm.symbols('x, v, f', cls=sm.Function)]
    eom = sm.Matrix([x.diff() - v, m*v.diff() + c*v + k*x - f])
    state_values = np.array([[1.0, 2.0, 3.0, 4.0], [5.0, 6.0, 7.0, 8.0]])
    specified_values = np.array([2.0, 2.0, 2.0, 2.0])
    constant_values = np.array([1.0, 2.0, 3.0])          # m, c, k
    h = 0.01

    def make(method):
        return opty_amd.ConstraintCollocator(
            equations_of_motion=eom, state_symbols=(x, v),
            num_collocation_nodes=4, node_time_interval=h,
            known_parameter_map={m: 1.0, c: 2.0},
            known_trajectory_map={f: specified_values}, time_symbol=t,
            integration_method=method)
    return make, (m, c, k), state_values, specified_values, constant_values, h


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['backward euler', 'midpoint'])
def test_gen_multi_arg_con_func(method):
    make, syms, sv, uv, cv, h = _msd()
    col = make(method)
    col._gen_multi_arg_con_func()
    consts = np.array([cv[syms.index(p)] for p in col.parameters])
    result = col._multi_arg_con_func(sv, uv, consts, h)
    m, c, k = cv
    kin, dyn = np.zeros(3), np.zeros(3)
    if method == 'backward euler':
        for i in (1, 2, 3):
            xi, vi = sv[:, i]
            xp, vp = sv[:, i - 1]
            dyn[i - 1] = m*(vi - vp)/h + c*vi + k*xi - uv[i]
            kin[i - 1] = (xi - xp)/h - vi
    else:
        for i in (0, 1, 2):
            xi, vi = sv[:, i]
            xn, vn = sv[:, i + 1]
            kin[i] = (xn - xi)/h - (vi + vn)/2
            dyn[i] = (m*(vn - vi)/h + c*(vn + vi)/2 + k*(xn + xi)/2 -
                      (uv[i] + uv[i + 1])/2)
    np.testing.assert_allclose(result, np.hstack((kin, dyn)), rtol=1e-13)
    # 2-D specified values take the other branch (:2415-2426)
    np.testing.assert_allclose(
        col._multi_arg_con_func(sv, uv[None, :], consts, h), result)


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['backward euler', 'midpoint'])
def test_gen_multi_arg_con_jac_func(method):
    make, syms, sv, uv, cv, h = _msd()
    col = make(method)
    col._gen_multi_arg_con_jac_func()
    consts = np.array([cv[syms.index(p)] for p in col.parameters])
    vals = col._multi_arg_con_jac_func(sv, uv, consts, h)
    rows, cols = col.jacobian_indices()
    jac = coo_to_dense(vals, rows, cols)
    x = sv[0]
    m, c, k = cv
    if method == 'backward euler':
        expected = np.array(
            [[-1/h, 1/h, 0, 0, 0, -1, 0, 0, 0],
             [0, -1/h, 1/h, 0, 0, 0, -1, 0, 0],
             [0, 0, -1/h, 1/h, 0, 0, 0, -1, 0],
             [0, k, 0, 0, -m/h, c + m/h, 0, 0, x[1]],
             [0, 0, k, 0, 0, -m/h, c + m/h, 0, x[2]],
             [0, 0, 0, k, 0, 0, -m/h, c + m/h, x[3]]])
    else:
        part1 = np.array([[-1/h, 1/h, 0, 0], [0, -1/h, 1/h, 0],
                          [0, 0, -1/h, 1/h], [k/2, k/2, 0, 0],
                          [0, k/2, k/2, 0], [0, 0, k/2, k/2]])
        part2 = np.array(
            [[-0.5, -0.5, 0, 0, 0], [0, -0.5, -0.5, 0, 0],
             [0, 0, -0.5, -0.5, 0],
             [-m/h + c/2, m/h + c/2, 0, 0, (x[1] + x[0])/2],
             [0, -m/h + c/2, m/h + c/2, 0, (x[2] + x[1])/2],
             [0, 0, -m/h + c/2, m/h + c/2, (x[3] + x[2])/2]])
        expected = np.hstack((part1, part2))
    np.testing.assert_allclose(jac, expected, rtol=1e-13)
    # and it is what the fused path returns for the same numbers
    free = np.hstack((sv.ravel(), [cv[2]]))
    np.testing.assert_allclose(col.generate_jacobian_function()(free), vals,
                               rtol=1e-13)


@pytest.mark.gpu
def test_multi_arg_closures_10link_match_fused_path():
    """The multi-argument closures of the 10-link system (45 vector and 23
    const arguments, 22 x 45 block) against the fused kernels."""
    import opty_amd
    kw = problems.build('config3_10link_small')
    col = opty_amd.ConstraintCollocator(**kw)
    N = col.num_collocation_nodes
    free = problems.make_free(col.num_free, seed=6)
    states = free[:col.num_states*N].reshape(col.num_states, N)
    spec = free[col.num_states*N:].reshape(1, N)
    consts = np.array([kw['known_parameter_map'][p] for p in col.parameters])
    col._gen_multi_arg_con_func()
    col._gen_multi_arg_con_jac_func()
    con = col._multi_arg_con_func(states, spec, consts,
                                  col.node_time_interval)
    jac = col._multi_arg_con_jac_func(states, spec, consts,
                                      col.node_time_interval)
    cb, jb = gu.error_bounds(col, free)
    gu.assert_close(con, col.generate_constraint_function()(free), 1e-12,
                    what='multi-arg con vs fused', bound=cb)
    gu.assert_close(jac, col.generate_jacobian_function()(free), 1e-12,
                    what='multi-arg jac vs fused', bound=jb)


@pytest.mark.gpu
def test_cse_pair_with_shared_conditions():
    """``cse()`` also extracts Boolean sub-expressions that several
    ``Piecewise`` conditions share (``x0 = a < 1/2``); a ``(replacements,
    [matrix])`` pair with such a replacement evaluates like the matrix itself
    (found by tools/matrix_soak.py)."""
    import sympy as sm
    import opty_amd
    a, b = sm.symbols('a, b', real=True)
    half = sm.Rational(1, 2)
    mat = sm.Matrix([[sm.Piecewise((a**2, a < half), (a/4, True)),
                      sm.Piecewise((b, a < half), (b**3, True))],
                     [sm.Piecewise((a + b, sm.And(a < half, b > -half)),
                                   (a - b, True)), a*b]])
    pair = sm.cse(mat)
    from sympy.logic.boolalg import Boolean
    assert any(isinstance(sub, Boolean) for _, sub in pair[0])
    f = opty_amd.ufuncify_matrix((a, b), pair)
    g = opty_amd.ufuncify_matrix((a, b), mat)
    rng = np.random.default_rng(2)
    av, bv = rng.uniform(-1, 1, 130), rng.uniform(-1, 1, 130)
    r1 = f(np.empty((130, 4)), av, bv)
    r2 = g(np.empty((130, 4)), av, bv)
    np.testing.assert_array_equal(r1, r2)
    want = np.where(av < 0.5, av**2, av/4)
    np.testing.assert_allclose(r1[:, 0, 0], want, rtol=1e-14)
