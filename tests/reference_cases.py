"""Known-answer fixtures restated from the reference's own unit tests
(``opty/tests/test_direct_collocation.py``): tiny N = 4 collocators with
literal expected index arrays and hand-derived constraint / Jacobian values.
Each ``case_*`` returns a dict: ``kw`` (collocator kwargs), ``free``,
``con`` (expected constraints), ``rows``/``cols`` (expected literal indices)
and ``dense`` (expected dense Jacobian, assembled like ``_coo_matrix``)."""
from collections import OrderedDict

import numpy as np
import sympy as sym


def case_instance_constraints():
    """``TestConstraintCollocatorInstanceConstraints`` (:1403-1710)."""
    I, m, g, d, t = sym.symbols('I, m, g, d, t')
    theta, omega, T = [f(t) for f in sym.symbols('theta, omega, T',
                                                 cls=sym.Function)]
    eom = sym.Matrix([theta.diff() - omega,
                      I*omega.diff() + m*g*d*sym.sin(theta) - T])
    par_map = OrderedDict(zip((I, m, g, d), (1.0, 1.0, 9.81, 1.0)))
    th_f, om_f = sym.symbols('theta, omega', cls=sym.Function)
    inst = (2.0*th_f(0.0), 3.0*th_f(0.03) - sym.pi, 4.0*om_f(0.0),
            5.0*om_f(0.03))
    kw = dict(equations_of_motion=eom, state_symbols=(theta, omega),
              num_collocation_nodes=4, node_time_interval=0.01,
              known_parameter_map=par_map, instance_constraints=inst,
              time_symbol=t)
    th = np.array([1.0, 2.0, 3.0, 4.0])
    om = np.array([5.0, 6.0, 7.0, 8.0])
    T_ = np.array([9.0, 10.0, 11.0, 12.0])
    free = np.hstack((th, om, T_))
    h, Iv, mv, gv, dv = 0.01, 1.0, 1.0, 9.81, 1.0
    kin = (th[1:] - th[:-1])/h - om[1:]
    dyn = Iv*(om[1:] - om[:-1])/h + mv*gv*dv*np.sin(th[1:]) - T_[1:]
    con = np.hstack((kin, dyn, [2.0, 12.0 - np.pi, 20.0, 40.0]))
    rows = np.array([0, 0, 0, 0, 0, 3, 3, 3, 3, 3, 1, 1, 1, 1, 1, 4, 4, 4, 4,
                     4, 2, 2, 2, 2, 2, 5, 5, 5, 5, 5, 6, 7, 8, 9])
    cols = np.array([1, 5, 0, 4, 9, 1, 5, 0, 4, 9, 2, 6, 1, 5, 10, 2, 6, 1,
                     5, 10, 3, 7, 2, 6, 11, 3, 7, 2, 6, 11, 0, 3, 4, 7])
    dense = np.zeros((10, 12))
    for i in range(3):
        dense[i, i + 1], dense[i, i] = 1/h, -1/h
        dense[i, 4 + i + 1] = -1.0
        dense[3 + i, i + 1] = mv*gv*dv*np.cos(th[i + 1])
        dense[3 + i, 4 + i + 1], dense[3 + i, 4 + i] = Iv/h, -Iv/h
        dense[3 + i, 8 + i + 1] = -1.0
    dense[6, 0], dense[7, 3], dense[8, 4], dense[9, 7] = 2.0, 3.0, 4.0, 5.0
    free_index = {'theta(0.0)': 0, 'theta(0.03)': 3, 'omega(0.0)': 4,
                  'omega(0.03)': 7}
    return dict(kw=kw, free=free, con=con, rows=rows, cols=cols, dense=dense,
                free_index=free_index)


def case_variable_duration():
    """``TestConstraintCollocatorVariableDuration`` (:1713-2039)."""
    m, g, d, t, h = sym.symbols('m, g, d, t, h')
    theta, omega, T = [f(t) for f in sym.symbols('theta, omega, T',
                                                 cls=sym.Function)]
    eom = sym.Matrix([theta.diff() - omega,
                      m*d**2*omega.diff() + m*g*d*sym.sin(theta) - T])
    par_map = OrderedDict(zip((m, g, d), (1.0, 9.81, 1.0)))
    th_f, om_f = sym.symbols('theta, omega', cls=sym.Function)
    inst = (th_f(0*h), th_f(3*h) - sym.pi, om_f(0*h), om_f(3*h))
    kw = dict(equations_of_motion=eom, state_symbols=(theta, omega),
              num_collocation_nodes=4, node_time_interval=h,
              known_parameter_map=par_map, instance_constraints=inst,
              time_symbol=t)
    th = np.array([1.0, 2.0, 3.0, 4.0])
    om = np.array([5.0, 6.0, 7.0, 8.0])
    T_ = np.array([9.0, 10.0, 11.0, 12.0])
    hv, mv, gv, dv = 0.01, 1.0, 9.81, 1.0
    free = np.hstack((th, om, T_, hv))
    kin = (th[1:] - th[:-1])/hv - om[1:]
    dyn = mv*dv**2*(om[1:] - om[:-1])/hv + mv*gv*dv*np.sin(th[1:]) - T_[1:]
    con = np.hstack((kin, dyn, [th[0], th[3] - np.pi, om[0], om[3]]))
    rows = np.array([0]*6 + [3]*6 + [1]*6 + [4]*6 + [2]*6 + [5]*6 +
                    [6, 7, 8, 9])
    cols = np.array([1, 5, 0, 4, 9, 12]*2 + [2, 6, 1, 5, 10, 12]*2 +
                    [3, 7, 2, 6, 11, 12]*2 + [0, 3, 4, 7])
    dense = np.zeros((10, 13))
    for i in range(3):
        dense[i, i + 1], dense[i, i] = 1/hv, -1/hv
        dense[i, 4 + i + 1] = -1.0
        dense[i, 12] = -(th[i + 1] - th[i])/hv**2
        dense[3 + i, i + 1] = dv*gv*mv*np.cos(th[i + 1])
        dense[3 + i, 4 + i + 1] = mv*dv**2/hv
        dense[3 + i, 4 + i] = -mv*dv**2/hv
        dense[3 + i, 8 + i + 1] = -1.0
        dense[3 + i, 12] = -dv**2*mv*(om[i + 1] - om[i])/hv**2
    dense[6, 0] = dense[7, 3] = dense[8, 4] = dense[9, 7] = 1.0
    return dict(kw=kw, free=free, con=con, rows=rows, cols=cols, dense=dense)


def case_mass_spring_damper(method):
    """``TestConstraintCollocator`` (:658-1017): known trajectory f, unknown
    parameter k; hand-derived backward Euler / midpoint arithmetic."""
    m, c, k, t = sym.symbols('m, c, k, t')
    x, v, f = [s(t) for s in sym.symbols('x, v, f', cls=sym.Function)]
    eom = sym.Matrix([x.diff() - v, m*v.diff() + c*v + k*x - f])
    fs = np.array([2.0, 2.0, 2.0, 2.0])
    kw = dict(equations_of_motion=eom, state_symbols=(x, v),
              num_collocation_nodes=4, node_time_interval=0.01,
              known_parameter_map=OrderedDict(((m, 1.0), (c, 2.0))),
              known_trajectory_map={f: fs}, time_symbol=t,
              integration_method=method)
    xs = np.array([1.0, 2.0, 3.0, 4.0])
    vs = np.array([5.0, 6.0, 7.0, 8.0])
    kv, h, mv, cv = 3.0, 0.01, 1.0, 2.0
    free = np.hstack((xs, vs, kv))
    dense = np.zeros((6, 9))
    if method == 'backward euler':
        kin = (xs[1:] - xs[:-1])/h - vs[1:]
        dyn = mv*(vs[1:] - vs[:-1])/h + cv*vs[1:] + kv*xs[1:] - fs[1:]
        for i in range(3):
            dense[i, i + 1], dense[i, i] = 1/h, -1/h
            dense[i, 4 + i + 1] = -1.0
            dense[3 + i, i + 1] = kv
            dense[3 + i, 4 + i + 1] = mv/h + cv
            dense[3 + i, 4 + i] = -mv/h
            dense[3 + i, 8] = xs[i + 1]
    else:
        xm, vm = (xs[1:] + xs[:-1])/2, (vs[1:] + vs[:-1])/2
        fm = (fs[1:] + fs[:-1])/2
        kin = (xs[1:] - xs[:-1])/h - vm
        dyn = mv*(vs[1:] - vs[:-1])/h + cv*vm + kv*xm - fm
        for i in range(3):
            dense[i, i], dense[i, i + 1] = -1/h, 1/h
            dense[i, 4 + i] = dense[i, 4 + i + 1] = -0.5
            dense[3 + i, i] = dense[3 + i, i + 1] = kv/2
            dense[3 + i, 4 + i] = -mv/h + cv/2
            dense[3 + i, 4 + i + 1] = mv/h + cv/2
            dense[3 + i, 8] = xm[i]
    con = np.hstack((kin, dyn))
    return dict(kw=kw, free=free, con=con, rows=None, cols=None, dense=dense)


ALL = {
    'instance_constraints': case_instance_constraints,
    'variable_duration': case_variable_duration,
    'msd_backward_euler': lambda: case_mass_spring_damper('backward euler'),
    'msd_midpoint': lambda: case_mass_spring_damper('midpoint'),
}
