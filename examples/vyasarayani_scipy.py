#!/usr/bin/env python
"""Parameter identification of a pendulum (the problem of the reference's
``examples/vyasarayani2011.py``), solved end to end with the MI355X
constraint / Jacobian evaluator.

The reference hands its callbacks to IPOPT through cyipopt; IPOPT is not
available in this image, so this script feeds the very same callbacks
(``Problem.objective/gradient/constraints/jacobian/jacobianstructure``) to
SciPy's ``trust-constr`` instead.  The point is the plumbing: symbolic EoM ->
generated HIP kernels -> C ABI -> NLP solver callbacks.
"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__),
                                                '..')))

import numpy as np
import scipy.optimize as so
import scipy.sparse as sp
import sympy as sym
from scipy.integrate import odeint

import opty_amd


def main(num_nodes=101, duration=5.0, seed=0, verbose=True):
    p, t = sym.symbols('p, t')
    y1, y2 = [f(t) for f in sym.symbols('y1, y2', cls=sym.Function)]
    y = sym.Matrix([y1, y2])
    eom = y.diff(t) - sym.Matrix([y2, -p*sym.sin(y1)])

    interval = duration/(num_nodes - 1)
    time = np.linspace(0.0, duration, num=num_nodes)
    p_true = 10.0
    y_meas = odeint(lambda y, t: [y[1], -p_true*np.sin(y[0])],
                    [np.pi/6.0, 0.0], time)
    rng = np.random.default_rng(seed)
    y1_meas = y_meas[:, 0] + rng.normal(scale=0.01, size=num_nodes)

    def obj(free):
        return interval*np.sum((y1_meas - free[:num_nodes])**2)

    def obj_grad(free):
        grad = np.zeros_like(free)
        grad[:num_nodes] = 2.0*interval*(free[:num_nodes] - y1_meas)
        return grad

    prob = opty_amd.Problem(obj, obj_grad, eom, (y1, y2), num_nodes,
                            interval, time_symbol=t,
                            integration_method='midpoint')
    rows, cols = prob.jacobianstructure()
    shape = (prob.num_constraints, prob.num_free)

    def jac(free):
        return sp.coo_matrix((prob.jacobian(free), (rows, cols)),
                             shape=shape).tocsr()

    con = so.NonlinearConstraint(prob.constraints, 0.0, 0.0, jac=jac)
    x0 = np.hstack((y_meas[:, 0], y_meas[:, 1], 5.0))     # wrong parameter
    res = so.minimize(prob.objective, x0, jac=prob.gradient,
                      constraints=[con], method='trust-constr',
                      options=dict(maxiter=300, gtol=1e-10, xtol=1e-12))
    p_hat = res.x[-1]
    if verbose:
        print('identified p = %.4f (true %.1f), max |constraint| = %.2e, '
              '%d iterations' % (p_hat, p_true,
                                 np.abs(prob.constraints(res.x)).max(),
                                 res.nit))
    return p_hat, res


if __name__ == '__main__':
    main()
