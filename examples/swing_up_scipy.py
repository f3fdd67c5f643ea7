#!/usr/bin/env python
"""Minimum-effort pendulum swing-up (the system of the reference's
``examples-gallery/beginner/plot_pendulum_swing_up_fixed_duration.py``) solved
end to end on the GPU callbacks: objective and gradient from
``opty_amd.create_objective_function`` (device kernels), constraints and a
row-sorted Jacobian (``jacobian_layout='csr'``) that becomes a SciPy
``csr_matrix`` without any conversion, driven by SciPy's SLSQP.  IPOPT (the reference's
solver) is not available in this image.
"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__),
                                                '..')))

import numpy as np
import scipy.optimize as so
import scipy.sparse as sp
import sympy as sm
import sympy.physics.mechanics as me

import opty_amd


def main(num_nodes=41, duration=4.0, max_torque=30.0, verbose=True):
    me.dynamicsymbols._t = sm.Symbol('t')
    t = me.dynamicsymbols._t
    I, m, g, d = sm.symbols('I, m, g, d')
    theta, omega, T = me.dynamicsymbols('theta, omega, T')
    eom = sm.Matrix([theta.diff() - omega,
                     I*omega.diff() + m*g*d*sm.sin(theta) - T])
    h = duration/(num_nodes - 1)
    par = {I: 1.0, m: 1.0, g: 9.81, d: 1.0}
    inst = (theta.func(0.0), theta.func(duration) - sm.pi,
            omega.func(0.0), omega.func(duration))
    obj, obj_grad = opty_amd.create_objective_function(
        sm.Integral(T**2, t), (theta, omega), (T,), (), num_nodes, h,
        time_symbol=t)
    prob = opty_amd.Problem(obj, obj_grad, eom, (theta, omega), num_nodes, h,
                            known_parameter_map=par,
                            instance_constraints=inst, time_symbol=t,
                            bounds={T: (-max_torque, max_torque)},
                            jacobian_layout='csr')
    row_ptr, col_idx = prob.collocator.jacobian_csr_structure()
    shape = (prob.num_constraints, prob.num_free)

    def jac(free):
        return sp.csr_matrix((prob.jacobian(free).copy(), col_idx, row_ptr),
                             shape=shape)

    lo = np.where(prob.lower_bound <= -prob.INF, -np.inf, prob.lower_bound)
    hi = np.where(prob.upper_bound >= prob.INF, np.inf, prob.upper_bound)
    x0 = np.hstack((np.linspace(0.0, np.pi, num_nodes),
                    np.full(num_nodes, np.pi/duration),
                    np.zeros(num_nodes)))
    # SLSQP is a dense method; a few hundred variables are fine for it
    con = dict(type='eq', fun=prob.constraints,
               jac=lambda free: jac(free).toarray())
    res = so.minimize(prob.objective, x0, jac=prob.gradient, method='SLSQP',
                      constraints=[con], bounds=so.Bounds(lo, hi),
                      options=dict(maxiter=500, ftol=1e-10,
                                   disp=bool(verbose)))
    free = res.x
    violation = np.abs(prob.constraints(free)).max()
    if verbose:
        print('effort %.4f, max |constraint| %.2e, theta(T) = %.4f, '
              'max |T| = %.3f' % (res.fun, violation, free[num_nodes - 1],
                                  np.abs(free[2*num_nodes:]).max()))
    return free, res, violation


if __name__ == '__main__':
    main()
