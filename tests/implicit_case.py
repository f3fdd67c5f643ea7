"""The literal known-answer case of the reference's
``test_implicit_known_traj`` (``opty/tests/test_direct_collocation.py:18-212``):
problem definition, free vector and expected constraint / Jacobian values,
restated as data for both the CPU (DAG interpreter) and the GPU (HIP) tests."""
import numpy as np
import sympy as sym
import sympy.physics.mechanics as mech


def build():
    mech.dynamicsymbols._t = sym.Symbol('t')
    m, g, r, h = sym.symbols('m, g, r, h', real=True)
    x, v, f, s = mech.dynamicsymbols('x, v, f, s', real=True)
    t = mech.dynamicsymbols._t
    theta_of_x = sym.Function('theta', real=True)(x)
    omega_of_v = sym.Function('omega', real=True)(v)
    eom = sym.Matrix([x.diff() - v - s + r*omega_of_v,
                      m*v.diff() - f + m*g*sym.sin(theta_of_x)])
    N = 4
    xs = np.linspace(2.0, 5.0, num=N)
    th = np.linspace(0.0, 10.0, num=N)
    calls = dict(
        theta=lambda free: np.interp(free[0:N], xs, th),
        dtheta=lambda free: np.array([3.9, 1.2, -5.6, 12.3]),
        omega=lambda free: np.array([-0.01, -0.98, 3.45, 27.45]),
        domega=lambda free: np.array([0.1, 8.9, -43.4, -2.5]))
    kw = dict(equations_of_motion=eom, state_symbols=(x, v),
              num_collocation_nodes=N, node_time_interval=h,
              known_parameter_map={r: 7.1, m: 3.3, g: 10.2},
              known_trajectory_map={
                  omega_of_v.diff(v): calls['domega'],
                  omega_of_v: calls['omega'],
                  s: np.array([121., 122., 123., 124.]),
                  theta_of_x: calls['theta'],
                  theta_of_x.diff(x): calls['dtheta']},
              time_symbol=t)
    free = np.array([2., 3., 4., 5., 6., 7., 8., 9., 10., 11., 12., 13., 14.])
    thetas, dthetas = calls['theta'](free), calls['dtheta'](free)
    omegas, domegas = calls['omega'](free), calls['domega'](free)
    con = np.array([
        (3. - 2.)/14. - 7. - 122. + 7.1*omegas[1],
        (4. - 3.)/14. - 8. - 123. + 7.1*omegas[2],
        (5. - 4.)/14. - 9. - 124. + 7.1*omegas[3],
        3.3*(7. - 6.)/14. - 11. + 3.3*10.2*np.sin(thetas[1]),
        3.3*(8. - 7.)/14. - 12. + 3.3*10.2*np.sin(thetas[2]),
        3.3*(9. - 8.)/14. - 13. + 3.3*10.2*np.sin(thetas[3])])
    jac = []
    for i in (1, 2, 3):
        jac += [1./14., -1. + 7.1*domegas[i], -1./14., 0., 0.,
                -(free[i] - free[i - 1])/14.**2,
                3.3*10.2*np.cos(thetas[i])*dthetas[i], 3.3/14., 0., -3.3/14.,
                -1., -3.3*(free[4 + i] - free[3 + i])/14.**2]
    symbols = dict(x=x, v=v, f=f, s=s, theta_of_x=theta_of_x,
                   omega_of_v=omega_of_v, t=t, m=m, g=g, r=r, h=h)
    return kw, free, con, np.array(jac), symbols
