"""GPU tests of the device-resident objective / gradient
(``opty_amd.create_objective_function``, C ABI ``opty_hip_objective_*``)
against the closed-form answers of the reference's
``TestCreateObjectiveFunction`` (``opty/tests/test_utils.py:67-219``)."""
import numpy as np
import pytest

import objective_cases

pytestmark = pytest.mark.gpu


def test_known_answers():
    import opty_amd
    t, cases = objective_cases.cases()
    for case in cases:
        states, inputs, unknowns = case['args']
        N = 20
        obj, obj_grad = opty_amd.create_objective_function(
            case['expr'], states, inputs, unknowns, N, case['h'],
            integration_method=case['method'], time_symbol=t)
        np.testing.assert_allclose(obj(case['free']), case['value'],
                                   rtol=1e-12, err_msg=case['name'])
        np.testing.assert_allclose(obj_grad(case['free']), case['grad'],
                                   rtol=1e-12, atol=1e-15,
                                   err_msg=case['name'])


@pytest.mark.parametrize('method', ['backward euler', 'midpoint'])
def test_large_device_resident(method):
    """N = 100 000, torch tensors in and out, against NumPy closed forms;
    ragged wave (N not a multiple of 64) and determinism."""
    import sympy as sym
    import torch
    import opty_amd
    t = sym.symbols('t')
    x, v, u = [f(t) for f in sym.symbols('x, v, u', cls=sym.Function)]
    p = sym.symbols('p')
    N, h = 100003, 0.01
    expr = sym.Integral(p*u**2 + sym.cos(x)*v**2, t) + 3*p**2
    obj, obj_grad = opty_amd.create_objective_function(
        expr, [x, v], [u], [p], N, h, integration_method=method,
        time_symbol=t)
    rng = np.random.default_rng(3)
    free = rng.standard_normal(3*N + 1)
    xs, vs, us, pv = free[:N], free[N:2*N], free[2*N:3*N], free[-1]
    G = lambda X, V, U: pv*U**2 + np.cos(X)*V**2
    if method == 'backward euler':
        value = h*G(xs[1:], vs[1:], us[1:]).sum() + 3*pv**2
        w = np.hstack((0, np.ones(N - 1)))
        dp = h*(us[1:]**2).sum() + 6*pv
    else:
        mid = lambda a: (a[1:] + a[:-1])/2
        value = h*G(mid(xs), mid(vs), mid(us)).sum() + 3*pv**2
        w = np.hstack((0.5, np.ones(N - 2), 0.5))
        dp = h*(mid(us)**2).sum() + 6*pv
    grad = np.hstack((h*w*(-np.sin(xs)*vs**2), h*w*2*np.cos(xs)*vs,
                      h*w*2*pv*us, dp))
    ft = torch.from_numpy(free).cuda()
    v1 = obj(ft)
    g1 = obj_grad(ft)
    assert g1.is_cuda
    np.testing.assert_allclose(v1, value, rtol=1e-11)
    np.testing.assert_allclose(g1.cpu().numpy(), grad, rtol=1e-11,
                               atol=1e-14)
    assert obj(ft) == v1                       # fixed summation order
    assert torch.equal(obj_grad(ft), g1)
    np.testing.assert_allclose(obj(free), value, rtol=1e-11)   # host path
