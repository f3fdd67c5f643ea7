"""Problem zoo (test, bench and example input -- not part of the product
package): the direct-collocation problems BASELINE.json's ``configs`` name.

Each factory returns a plain ``dict`` of keyword arguments that both this
package's :class:`opty_amd.ConstraintCollocator` and the reference's
``opty.ConstraintCollocator`` (``opty/direct_collocation.py:1406-1411``) accept,
so parity tests, the golden-vector generator and ``bench.py`` all build the very
same symbolic problem.

Sizes (N nodes, n states, M equations, q unknown inputs, r unknown parameters,
s variable duration, o instance constraints) follow SURVEY.md section 8(d).
"""

import numpy as np
import sympy as sm
import sympy.physics.mechanics as me

__all__ = ['vyasarayani', 'pendulum_swing_up', 'n_link_cart_pendulum',
           'mass_spring_damper', 'variable_duration_pendulum',
           'chaplygin_sleigh', 'one_equation', 'implicit_known_trajectory',
           'gait_like_pendulum', 'CONFIGS', 'make_free']


_KANE = {}


def _cart_pendulum(num_links):
    """``(states, eom)`` of ``n_link_pendulum_on_cart`` (Kane's method takes
    15 s for 24 links): derived once per process, SymPy objects are
    immutable."""
    if num_links not in _KANE:
        from sympy.physics.mechanics.models import n_link_pendulum_on_cart
        # The reference's collocator assigns ``me.dynamicsymbols._t``
        # globally (``opty/direct_collocation.py:1492``); restore SymPy's
        # default so the derivation always runs on the same time symbol
        me.dynamicsymbols._t = sm.Symbol('t')
        kane = n_link_pendulum_on_cart(n=num_links, cart_force=True,
                                       joint_torques=False)
        states = kane.q.col_join(kane.u)
        t = me.dynamicsymbols._t
        eom = sm.ImmutableDenseMatrix(
            kane.mass_matrix_full @ states.diff(t) - kane.forcing_full)
        _KANE[num_links] = (sm.ImmutableDenseMatrix(states), eom,
                            tuple(kane.q), tuple(kane.u))
    me.dynamicsymbols._t = sm.Symbol('t')
    return _KANE[num_links]


def vyasarayani(num_nodes=51, duration=50.0, method='backward euler'):
    """Config 1: single-pendulum parameter identification.

    EoM from ``examples/vyasarayani2011.py:44-48`` (``y1' - y2``,
    ``y2' + p sin(y1)``), shrunk to N=51 / backward Euler as BASELINE.json
    config 1 asks.  ``p`` is an unknown parameter.
    """
    p, t = sm.symbols('p, t')
    y1, y2 = [f(t) for f in sm.symbols('y1, y2', cls=sm.Function)]
    y = sm.Matrix([y1, y2])
    eom = y.diff(t) - sm.Matrix([y2, -p*sm.sin(y1)])
    return dict(equations_of_motion=eom, state_symbols=(y1, y2),
                num_collocation_nodes=num_nodes,
                node_time_interval=duration/(num_nodes - 1),
                time_symbol=t, integration_method=method)


def pendulum_swing_up(num_nodes=10000, duration=10.0, method='midpoint'):
    """Config 2: 1-link pendulum swing-up with four instance constraints.

    EoM and instance constraints from
    ``examples-gallery/beginner/plot_pendulum_swing_up_fixed_duration.py:43-84``.
    """
    I, m, g, d, t = sm.symbols('I, m, g, d, t')
    theta, omega, T = sm.symbols('theta, omega, T', cls=sm.Function)
    eom = sm.Matrix([theta(t).diff() - omega(t),
                     I*omega(t).diff() + m*g*d*sm.sin(theta(t)) - T(t)])
    instance_constraints = (theta(0.0), theta(duration) - np.pi,
                            omega(0.0), omega(duration))
    return dict(equations_of_motion=eom,
                state_symbols=(theta(t), omega(t)),
                num_collocation_nodes=num_nodes,
                node_time_interval=duration/(num_nodes - 1),
                known_parameter_map={I: 1.0, m: 1.0, g: 9.81, d: 1.0},
                instance_constraints=instance_constraints,
                time_symbol=t, integration_method=method)


def n_link_cart_pendulum(num_links=10, num_nodes=100000, interval=0.01,
                         method='backward euler', variable_duration=False,
                         unknown_masses=0):
    """Configs 3/4 (num_links=10) and the config-5 stand-in (num_links=24).

    ``sympy.physics.mechanics.models.n_link_pendulum_on_cart`` is the same
    constructor the reference's own test uses with n=3
    (``opty/tests/test_direct_collocation.py:2044-2048``); EoM is
    ``mass_matrix_full @ x' - forcing_full``.  Parameters: g=9.81,
    ``l_k = 1 + 0.01 k``, ``m_k = 1 + 0.01 k``; the
    cart force F(t) is the one unknown input trajectory.

    ``unknown_masses`` leaves the last that many masses free (r > 0) and
    ``variable_duration`` makes ``h`` a free Symbol (s = 1); both are used by
    parity tests to exercise every column class of the Jacobian block.
    """
    states, eom, _, _ = _cart_pendulum(num_links)
    t = me.dynamicsymbols._t
    par_map = {}
    params = sorted((s for s in eom.free_symbols if s != t),
                    key=lambda s: (s.name[0], int(s.name[1:] or 0)))
    masses = [s for s in params if s.name.startswith('m')]
    free_masses = set(masses[len(masses) - unknown_masses:]) \
        if unknown_masses else set()
    for s in params:
        if s in free_masses:
            continue
        if s.name == 'g':
            par_map[s] = 9.81
        else:
            par_map[s] = 1.0 + 0.01*int(s.name[1:])
    h = sm.Symbol('h', real=True) if variable_duration else interval
    return dict(equations_of_motion=sm.Matrix(eom),
                state_symbols=tuple(states),
                num_collocation_nodes=num_nodes,
                node_time_interval=h,
                known_parameter_map=par_map,
                time_symbol=t, integration_method=method)


def mass_spring_damper(num_nodes=4, interval=2.0, method='backward euler'):
    """The tiny fixture system of ``TestConstraintCollocator``
    (``opty/tests/test_direct_collocation.py:658-700``): ``m v' + c v + k x - f``
    with known trajectory ``f`` and unknown parameter ``k``."""
    m, c, k, t = sm.symbols('m, c, k, t')
    x, v, f = [s(t) for s in sm.symbols('x, v, f', cls=sm.Function)]
    eom = sm.Matrix([x.diff() - v, m*v.diff() + c*v + k*x - f])
    return dict(equations_of_motion=eom, state_symbols=(x, v),
                num_collocation_nodes=num_nodes, node_time_interval=interval,
                known_parameter_map={m: 1.0, c: 2.0},
                known_trajectory_map={f: np.linspace(1.0, 4.0, num_nodes)},
                time_symbol=t, integration_method=method)


def variable_duration_pendulum(num_nodes=60, method='midpoint'):
    """Variable-duration (s=1) pendulum with instance constraints given as
    integer multiples of ``h`` (the pattern of
    ``TestConstraintCollocatorVariableDuration``,
    ``opty/tests/test_direct_collocation.py:1713-1760``)."""
    m, g, d, h = sm.symbols('m, g, d, h', real=True)
    t = sm.Symbol('t')
    theta, omega, T = [s(t) for s in sm.symbols('theta, omega, T',
                                                cls=sm.Function)]
    eom = sm.Matrix([theta.diff() - omega,
                     m*d**2*omega.diff() + m*g*d*sm.sin(theta) - T])
    N = num_nodes
    inst = (theta.func(0*h), theta.func((N - 1)*h) - sm.pi,
            omega.func(0*h), omega.func((N - 1)*h))
    return dict(equations_of_motion=eom, state_symbols=(theta, omega),
                num_collocation_nodes=N, node_time_interval=h,
                known_parameter_map={m: 1.0, g: 9.81},
                instance_constraints=inst,
                time_symbol=t, integration_method=method)


def chaplygin_sleigh(num_nodes=100, interval=0.1, method='backward euler'):
    """More equations than states (M = 5, n = 4): the Chaplygin sleigh with
    one algebraic (nonholonomic) equation, three unknown inputs and eight
    instance constraints -- ``test_extra_algebraic``
    (``opty/tests/test_direct_collocation.py:281-345``)."""
    me.dynamicsymbols._t = sm.Symbol('t')
    m = sm.symbols('m', real=True)
    x, y, theta = me.dynamicsymbols('x, y, theta', real=True)
    vx, vy = me.dynamicsymbols('v_x, v_y', real=True)
    Fx, Fy = me.dynamicsymbols('F_x, F_y')
    t = me.dynamicsymbols._t
    eom = sm.Matrix([m*vx.diff() - Fx, x.diff() - vx, m*vy.diff() - Fy,
                     y.diff() - vy, -sm.sin(theta)*vx + sm.cos(theta)*vy])
    dur = interval*(num_nodes - 1)
    inst = (x.func(0.0), y.func(0.0), vx.func(0.0), vy.func(0.0),
            x.func(dur) - 1.0, y.func(dur) - 1.0, vx.func(dur), vy.func(dur))
    return dict(equations_of_motion=eom, state_symbols=(x, y, vx, vy),
                num_collocation_nodes=num_nodes, node_time_interval=interval,
                known_parameter_map={m: 1.0}, instance_constraints=inst,
                time_symbol=t, integration_method=method)


def one_equation(num_nodes=100, method='backward euler'):
    """A single equation of motion (M = n = 1, odd block width P = 3) --
    ``test_one_eom_only``
    (``opty/tests/test_direct_collocation.py:2337-2393``)."""
    me.dynamicsymbols._t = sm.Symbol('t')
    t = me.dynamicsymbols._t
    y, u = me.dynamicsymbols('y u')
    eom = sm.Matrix([-y.diff(t) - y**3 + u])
    t0, tf = 0.0, 10.0
    inst = (y.func(t0) - 1, y.func(tf) - 1.5)
    return dict(equations_of_motion=eom, state_symbols=(y,),
                num_collocation_nodes=num_nodes,
                node_time_interval=(tf - t0)/(num_nodes - 1),
                instance_constraints=inst, time_symbol=t,
                integration_method=method)


def implicit_known_trajectory(num_nodes=40, method='backward euler',
                              variable_duration=True):
    """Known trajectories that are functions of a state, ``theta(x(t))`` and
    ``omega(v(t))``, supplied with their derivatives as callables of ``free``
    -- the system of ``test_implicit_known_traj``
    (``opty/tests/test_direct_collocation.py:18-70``) with smooth callables
    so that any N works."""
    me.dynamicsymbols._t = sm.Symbol('t')
    m, g, r, h = sm.symbols('m, g, r, h', real=True)
    x, v, f, s = me.dynamicsymbols('x, v, f, s', real=True)
    t = me.dynamicsymbols._t
    theta = sm.Function('theta', real=True)(x)
    omega = sm.Function('omega', real=True)(v)
    eom = sm.Matrix([x.diff() - v - s + r*omega,
                     m*v.diff() - f + m*g*sm.sin(theta)])
    N = num_nodes
    traj = {
        omega.diff(v): lambda free: -2.0*free[N:2*N]*np.exp(-free[N:2*N]**2),
        omega: lambda free: np.exp(-free[N:2*N]**2),
        s: np.linspace(0.5, 1.5, N),
        theta: lambda free: 0.3*np.sin(2.0*free[0:N]) + 0.1*free[0:N],
        theta.diff(x): lambda free: 0.6*np.cos(2.0*free[0:N]) + 0.1,
    }
    return dict(equations_of_motion=eom, state_symbols=(x, v),
                num_collocation_nodes=N,
                node_time_interval=h if variable_duration else 0.05,
                known_parameter_map={r: 7.1, m: 3.3, g: 10.2},
                known_trajectory_map=traj, time_symbol=t,
                integration_method=method)


def elementary_functions(num_nodes=45, method='backward euler',
                         variable_duration=False):
    """Not from the reference's examples: a 4-state system that exercises
    every elementary function the C99 printer of ``ufuncify_matrix``
    (``opty/utils.py:748-757``) can meet in practice -- tanh, exp, sqrt, atan,
    log, sinh, cosh, tan, pow with a non-integer exponent, asin, acos, atan2
    and a division by an expression -- with a known trajectory, an unknown
    input, a known non-integer exponent and an unknown parameter."""
    me.dynamicsymbols._t = sm.Symbol('t')
    t = me.dynamicsymbols._t
    a, b, c, hh = sm.symbols('a, b, c, h', real=True)
    x, y, z, w, u, k = me.dynamicsymbols('x, y, z, w, u, k', real=True)
    eom = sm.Matrix([
        x.diff() - (sm.tanh(y) + sm.exp(-x**2)*u + sm.sqrt(1 + z**2)),
        y.diff() - (sm.atan(x*z) + sm.log(1 + w**2)*a - sm.sinh(x/2)),
        z.diff() - (sm.cosh(3*y/10)*sm.tan(w/2) + (sm.Rational(3, 2) + x)**b
                    - k/(1 + y**2)),
        w.diff() - (sm.asin(sm.tanh(z)/2) + sm.acos(x/2)*c
                    + sm.atan2(y, 2 + z)),
    ])
    N = num_nodes
    interval = hh if variable_duration else 0.05
    dur = (N - 1)*(hh if variable_duration else 0.05)
    inst = (x.func(0*hh if variable_duration else 0.0) - 0.25,
            w.func(dur)**2 - 0.5)
    return dict(equations_of_motion=eom, state_symbols=(x, y, z, w),
                num_collocation_nodes=N, node_time_interval=interval,
                known_parameter_map={b: 1.7, c: 0.8},
                known_trajectory_map={k: np.cos(np.linspace(0.0, 3.0, N))},
                instance_constraints=inst, time_symbol=t,
                integration_method=method)


def c99_functions(num_nodes=47, method='backward euler'):
    """Not from the reference's examples: a 3-state system with the rest of
    the C99 printer's function table (``sympy.printing.c.known_functions_C99``
    beyond :func:`elementary_functions`): ``log1p``, ``expm1``, ``log2``,
    ``log10``, ``exp2``, ``Cbrt``, ``hypot``, ``fma`` of
    ``sympy.codegen.cfunctions`` and ``gamma`` / ``loggamma`` of a known
    parameter (their derivative, digamma, has no C counterpart), with an
    unknown input and an unknown parameter.  (``cfunctions.Sqrt`` is lowered
    too, but SymPy 1.14's C printer raises ``KeyError: 'Sqrt'`` for it, so
    the reference cannot build it.)"""
    from sympy.codegen import cfunctions as cf
    me.dynamicsymbols._t = sm.Symbol('t')
    t = me.dynamicsymbols._t
    a, g = sm.symbols('a, g', real=True)
    x, y, z, u = me.dynamicsymbols('x, y, z, u', real=True)
    eom = sm.Matrix([
        x.diff() - (cf.log1p(y**2) + cf.expm1(-x**2/4)*u
                    + cf.hypot(z, 1 + x*y)),
        y.diff() - (cf.log2(2 + z**2)*a - cf.log10(3 + x**2)
                    + cf.Cbrt(1 + y**2)*sm.gamma(g)),
        z.diff() - (cf.exp2(-y**2)*sm.loggamma(g + 1) + cf.fma(x, y, u/2)
                    - sm.sqrt(2 + z**2)),
    ])
    N = num_nodes
    inst = (x.func(0.0) - 0.25, z.func((N - 1)*0.05) - 0.5)
    return dict(equations_of_motion=eom, state_symbols=(x, y, z),
                num_collocation_nodes=N, node_time_interval=0.05,
                known_parameter_map={g: 2.6},
                instance_constraints=inst, time_symbol=t,
                integration_method=method)


def piecewise_functions(num_nodes=53, method='backward euler'):
    """Not from the reference's examples: a 3-state system with the
    branching and special functions SymPy's C99 printer also accepts in
    ``ufuncify_matrix`` (``opty/utils.py:748-757``) -- ``Piecewise`` with
    relational / And / Or conditions (dry friction, a dead band, a one-sided
    spring), ``erf``, ``asinh``, ``Abs``, ``Max`` -- with an
    unknown input and an unknown parameter.  (``sec`` / ``cot`` / ``floor``
    are lowered too, but the reference cannot build them: its symbol renaming
    turns ``(1.0/cos(x))`` into ``(1.0/cos(x))_``, and SymPy's derivative of
    ``floor`` is not printable -- so no fixture can pin those.)"""
    me.dynamicsymbols._t = sm.Symbol('t')
    t = me.dynamicsymbols._t
    mu, kk = sm.symbols('mu, kk', real=True)
    x, v, z, u = me.dynamicsymbols('x, v, z, u', real=True)
    friction = sm.Piecewise((mu*v, v > 0), (2*mu*v, True))
    band = sm.Piecewise((0, sm.And(x > -sm.Rational(1, 4),
                                   x < sm.Rational(1, 4))),
                        (x - sm.Rational(1, 4), x >= sm.Rational(1, 4)),
                        (x + sm.Rational(1, 4), True))
    stop = sm.Piecewise((kk*(z - sm.Rational(1, 2))**2,
                         sm.Or(z > sm.Rational(1, 2), v < -sm.Rational(3, 4))),
                        (0, True))
    eom = sm.Matrix([
        x.diff() - v,
        v.diff() + friction + kk*band + stop - u + sm.erf(x*z),
        z.diff() - (sm.asinh(v) - sm.tanh(3*x)/4 + 1/sm.cos(z/2) -
                    sm.cos(1 + x**2)/sm.sin(1 + x**2) +
                    sm.Max(x, z)*sm.Abs(v)),
    ])
    return dict(equations_of_motion=eom, state_symbols=(x, v, z),
                num_collocation_nodes=num_nodes, node_time_interval=0.04,
                known_parameter_map={mu: 0.3}, time_symbol=t,
                integration_method=method)


def delay_equation(num_nodes=51, method='backward euler'):
    """Six chained delay segments with six algebraic path constraints
    (M = 12 equations for n = 6 states), six unknown inputs and twelve
    instance constraints that couple *two* function atoms each, inputs
    included -- the system of the reference's gallery example
    ``examples-gallery/beginner/plot_betts_10_50.py:52-110``."""
    me.dynamicsymbols._t = sm.Symbol('t')
    t = me.dynamicsymbols._t
    x = me.dynamicsymbols('x1:7')
    u = me.dynamicsymbols('u1:7')
    prev = [sm.Float(0.0), x[0]*0.0] + [x[k]*u[k - 1] for k in range(1, 5)]
    # -x1' + x0*u_{-1}, -x2' + x1*u0 (u_{-1} = u0 = 0), then -x_k' + x_{k-1}*u_{k-2}
    eom = sm.Matrix([-x[k].diff(t) + prev[k] for k in range(6)] +
                    [u[k] + x[k] for k in range(6)])
    t0, tf = 0.0, 1.0
    inst = tuple([x[0].func(t0) - 1.0] +
                 [x[k].func(t0) - x[k - 1].func(tf) for k in range(1, 6)] +
                 [u[k].func(t0) + x[k].func(t0) - 0.5 for k in range(6)])
    return dict(equations_of_motion=eom, state_symbols=tuple(x),
                num_collocation_nodes=num_nodes,
                node_time_interval=(tf - t0)/(num_nodes - 1),
                instance_constraints=inst, time_symbol=t,
                integration_method=method)


def odd_block_chain(num_nodes=75, method='backward euler',
                    unknown_parameter=True):
    """Seven coupled first-order equations with one unknown parameter:
    M = n = 7, C = 15, so the per-node block has an ODD number of entries
    (P = 105 >= 64): node rows alternate between 16-byte aligned and
    misaligned, the case the line-aligned flush's straddling pieces exist
    for.  With ``unknown_parameter=False`` nothing but the states is free
    (q = r = s = 0).  Not from the reference's examples."""
    me.dynamicsymbols._t = sm.Symbol('t')
    t = me.dynamicsymbols._t
    p, c = sm.symbols('p, c', real=True)
    x = me.dynamicsymbols('x1:8', real=True)
    eom = sm.Matrix([x[k].diff() + p*x[k] - c*x[k - 1]*sm.sin(x[k])
                     - sm.cos(x[(k + 2) % 7])*x[(k + 3) % 7]
                     for k in range(7)])
    return dict(equations_of_motion=eom, state_symbols=tuple(x),
                num_collocation_nodes=num_nodes, node_time_interval=0.04,
                known_parameter_map={c: 0.7} if unknown_parameter
                else {c: 0.7, p: 0.3},
                time_symbol=t, integration_method=method)


def gait_like_pendulum(num_links=24, num_nodes=50000, method='backward euler',
                       speed=1.3):
    """Config-5-shaped stand-in: what ``examples-gallery/advanced/
    plot_human_gait.py:100-214`` asks of the path, on a system whose equations
    can be built here (pygait2d is not vendored):

    * ~50 states for ``num_links=24`` (``n_link_pendulum_on_cart``),
      **variable duration** (``h`` free, bounds like ``:131``),
    * an unknown input trajectory (cart force ``F``) and a **known** one
      (a "hand of god" torque on the last link, ``:118-122``),
    * smooth **contact-like ``exp`` terms** in the dynamic rows (gait2d's
      ground contact is a smoothed penalty in the foot's height and speed):
      ``kc*exp(-cc*q0)`` on the cart and ``kd*exp(-cc*q_k)*u_k`` on three links,
    * **instance constraints** of the gait example's kinds (``:163-184``): a
      fixed start, ``q0(T) - speed*q1(T)`` (two atoms, one scaled), periodic
      two-atom pairs that cross states ``q_a(0) - q_b(T)``, speeds
      ``u_k(0) - u_k(T)`` and one on the unknown input, all written as integer
      multiples of ``h``.
    """
    states, eom, q, u = _cart_pendulum(num_links)
    q, u = list(q), list(u)
    t = me.dynamicsymbols._t
    eom = sm.Matrix(eom)
    kc, cc, kd = sm.symbols('kc, cc, kd', real=True)
    Tg = sm.Function('Tg')(t)
    nq = len(q)
    eom[nq] += kc*sm.exp(-cc*q[0])
    for k in sorted({1, nq//2, nq - 1}):
        eom[nq + k] += kd*sm.exp(-cc*q[k])*u[k]
    eom[2*nq - 1] -= Tg
    par_map = {}
    for s in sorted((s for s in eom.free_symbols if s != t),
                    key=lambda s: s.name):
        if s in (kc, cc, kd):
            continue
        par_map[s] = 9.81 if s.name == 'g' else 1.0 + 0.01*int(s.name[1:])
    par_map.update({kc: 0.4, cc: 1.5, kd: 0.25})
    h = sm.Symbol('h', real=True)
    N = num_nodes
    dur = (N - 1)*h
    F = [f for f in eom.atoms(sm.Function) if f.func.__name__ == 'F'][0]
    pairs = [(1, min(2, nq - 1)), (min(2, nq - 1), 1)]
    if nq > 4:
        pairs += [(3, 4), (4, 3)]
    inst = [q[0].func(0*h), q[0].func(dur) - speed*q[1].func(dur)]
    inst += [q[a].func(0*h) - q[b].func(dur) for a, b in pairs]
    inst += [u[k].func(0*h) - u[k].func(dur) for k in range(min(nq, 5))]
    inst += [F.func(0*h) - F.func(dur)]
    return dict(equations_of_motion=eom, state_symbols=tuple(states),
                num_collocation_nodes=N, node_time_interval=h,
                known_parameter_map=par_map,
                known_trajectory_map={
                    Tg: 0.3*np.sin(np.linspace(0.0, 7.0, N))},
                instance_constraints=tuple(inst), time_symbol=t,
                integration_method=method)


def gallery_problem(name='gallery_one_legged_time_trial', num_nodes=None):
    """A problem of the reference's example gallery, from the inputs recorded
    in ``tests/golden/<name>.npz`` (``tests/golden/_gen/gallery_capture.py``),
    optionally on another node count (``tests/gallery_cases.py:rescale``).

    ``gallery_one_legged_time_trial``: ``examples-gallery/advanced/
    plot_one_legged_time_trial.py`` -- a four-bar leg + crank driven by four
    De Groote (2016) musculotendon actuators (``sympy.physics.biomechanics``):
    12 states (4 coordinates, 4 speeds, 4 activations), 4 excitations as
    unknown inputs, variable duration, 13 instance constraints; the one
    musculoskeletal model of the gallery that builds with SymPy alone.
    """
    import os
    import sys
    tests = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'tests')
    if tests not in sys.path:
        sys.path.insert(0, tests)
    import gallery_cases
    _, _, kw = gallery_cases.load(name)
    if num_nodes is not None:
        kw = gallery_cases.rescale(kw, num_nodes)
    return kw


def planar_biped(num_nodes=50000, method='backward euler'):
    """Config-5-shaped stand-in with the STRUCTURE of the gait model the
    reference's ``plot_human_gait.py`` takes from ``pygait2d`` (absent here,
    SURVEY.md 8(c)): a planar seven-segment biped -- trunk, two thighs, two
    shanks, two feet -- with 9 degrees of freedom (hip position, trunk angle,
    six joint angles), i.e. 18 states, driven by six joint torques (unknown
    input trajectories), with a smoothed ground contact under the heel and
    the toe of each foot (normal force from a smooth ``max(0, -y)`` with
    damping, smoothed Coulomb friction), a free node time interval and the
    gait example's periodicity conditions written for half a symmetric
    stride (left and right swap: two-atom instance constraints).  Derived
    with Kane's method; implicit first-order form ``[q' - u; F_r + F_r*]``
    (M = n = 18, q = 6, C = 43, P = 774)."""
    t = sm.Symbol('t')
    me.dynamicsymbols._t = t
    q = list(me.dynamicsymbols('qx qy qa qb qc qd qe qf qg', real=True))
    u = list(me.dynamicsymbols('ux uy ua ub uc ud ue uf ug', real=True))
    T = list(me.dynamicsymbols('Tb Tc Td Te Tf Tg', real=True))
    g, kc, cc, mu, eps = sm.symbols('g kc cc mu eps', real=True)
    seg = {n: sm.symbols('m_%s I_%s l_%s d_%s' % (n, n, n, n), real=True)
           for n in ('tr', 'th', 'sh', 'ft')}
    N = me.ReferenceFrame('N')
    origin = me.Point('O')
    origin.set_vel(N, 0)
    hip = origin.locatenew('hip', q[0]*N.x + q[1]*N.y)
    hip.set_vel(N, u[0]*N.x + u[1]*N.y)
    A = N.orientnew('A', 'Axis', (q[2], N.z))
    A.set_ang_vel(N, u[2]*N.z)
    bodies, loads = [], []

    def body(name, com, frame, m, inertia):
        bodies.append(me.RigidBody(name, com, frame, m,
                                   (me.inertia(frame, 0, 0, inertia), com)))
        loads.append((com, -m*g*N.y))

    m, I, l, d = seg['tr']
    com = hip.locatenew('trc', d*A.y)
    com.v2pt_theory(hip, N, A)
    body('trunk', com, A, m, I)
    for side, k in (('r', 3), ('l', 6)):
        frames, parent = [], A
        for j in range(3):
            F = parent.orientnew('F%s%d' % (side, j), 'Axis',
                                 (q[k + j], parent.z))
            F.set_ang_vel(parent, u[k + j]*parent.z)
            loads.extend([(F, T[k - 3 + j]*N.z), (parent, -T[k - 3 + j]*N.z)])
            frames.append(F)
            parent = F
        B, C, D = frames
        (m1, I1, l1, d1), (m2, I2, l2, d2), (m3, I3, l3, d3) = (
            seg['th'], seg['sh'], seg['ft'])
        thc = hip.locatenew('thc' + side, -d1*B.y)
        knee = hip.locatenew('knee' + side, -l1*B.y)
        for pnt in (thc, knee):
            pnt.v2pt_theory(hip, N, B)
        shc = knee.locatenew('shc' + side, -d2*C.y)
        ankle = knee.locatenew('ankle' + side, -l2*C.y)
        for pnt in (shc, ankle):
            pnt.v2pt_theory(knee, N, C)
        ftc = ankle.locatenew('ftc' + side, d3*D.x)
        heel = ankle.locatenew('heel' + side, -l3/4*D.x - l3/5*D.y)
        toe = ankle.locatenew('toe' + side, 3*l3/4*D.x - l3/5*D.y)
        for pnt in (ftc, heel, toe):
            pnt.v2pt_theory(ankle, N, D)
        body('th' + side, thc, B, m1, I1)
        body('sh' + side, shc, C, m2, I2)
        body('ft' + side, ftc, D, m3, I3)
        for pnt in (heel, toe):
            y = pnt.pos_from(origin).dot(N.y)
            vx, vy = pnt.vel(N).dot(N.x), pnt.vel(N).dot(N.y)
            normal = kc*(sm.sqrt(y**2 + eps**2) - y)/2*(1 - cc*vy)
            loads.append((pnt, normal*N.y -
                          mu*normal*vx/sm.sqrt(vx**2 + eps**2)*N.x))
    kd = [qi.diff(t) - ui for qi, ui in zip(q, u)]
    kane = me.KanesMethod(N, q, u, kd_eqs=kd)
    fr, frstar = kane.kanes_equations(bodies, loads)
    eom = sm.Matrix(kd).col_join(fr + frstar)
    values = {'g': 9.81, 'kc': 2.0e3, 'cc': 0.5, 'mu': 0.8, 'eps': 0.02,
              'm_tr': 50.0, 'I_tr': 3.2, 'l_tr': 0.6, 'd_tr': 0.3,
              'm_th': 7.0, 'I_th': 0.15, 'l_th': 0.44, 'd_th': 0.19,
              'm_sh': 3.4, 'I_sh': 0.05, 'l_sh': 0.43, 'd_sh': 0.19,
              'm_ft': 1.0, 'I_ft': 0.005, 'l_ft': 0.2, 'd_ft': 0.06}
    par_map = {s: values[s.name] for s in sorted(
        (s for s in eom.free_symbols if s != t), key=lambda s: s.name)}
    h = sm.Symbol('h', real=True)
    dur = (num_nodes - 1)*h
    f = lambda x, when: x.func(when)
    swap = [(3, 6), (4, 7), (5, 8), (6, 3), (7, 4), (8, 5)]
    inst = [f(q[0], 0*h), f(q[1], 0*h) - f(q[1], dur),
            f(q[2], 0*h) - f(q[2], dur)]
    inst += [f(q[a], 0*h) - f(q[b], dur) for a, b in swap]
    inst += [f(u[k], 0*h) - f(u[k], dur) for k in range(3)]
    inst += [f(u[a], 0*h) - f(u[b], dur) for a, b in swap[:3]]
    inst += [f(T[0], 0*h) - f(T[3], dur)]
    return dict(equations_of_motion=eom, state_symbols=tuple(q + u),
                num_collocation_nodes=num_nodes, node_time_interval=h,
                known_parameter_map=par_map,
                instance_constraints=tuple(inst), time_symbol=t,
                integration_method=method)


# name -> (factory, kwargs).  "*_small" variants are the sizes the oracle and
# the reference finish in seconds; parity fixtures are generated from them.
CONFIGS = {
    'config1_vyasarayani': (vyasarayani, {}),
    'config2_pendulum': (pendulum_swing_up, {}),
    'config2_pendulum_small': (pendulum_swing_up, {'num_nodes': 101}),
    'config3_10link': (n_link_cart_pendulum, {}),
    'config3_10link_small': (n_link_cart_pendulum, {'num_nodes': 41}),
    'pend3_link_midpoint_small': (n_link_cart_pendulum,
                                  {'num_links': 3, 'num_nodes': 37,
                                   'method': 'midpoint', 'interval': 0.02}),
    'pend2_link_vardur_unkmass_small': (
        n_link_cart_pendulum, {'num_links': 2, 'num_nodes': 33,
                               'variable_duration': True,
                               'unknown_masses': 2}),
    'msd_be_small': (mass_spring_damper, {'num_nodes': 23}),
    'msd_mid_small': (mass_spring_damper, {'num_nodes': 23,
                                           'method': 'midpoint'}),
    'vardur_pendulum_small': (variable_duration_pendulum, {}),
    'chaplygin_be_small': (chaplygin_sleigh, {}),
    'chaplygin_mid_small': (chaplygin_sleigh, {'num_nodes': 77,
                                               'method': 'midpoint'}),
    'one_eom_be_small': (one_equation, {}),
    'one_eom_mid_small': (one_equation, {'num_nodes': 67,
                                         'method': 'midpoint'}),
    'implicit_traj_be_small': (implicit_known_trajectory, {}),
    'implicit_traj_mid_small': (implicit_known_trajectory,
                                {'num_nodes': 33, 'method': 'midpoint',
                                 'variable_duration': False}),
    'states_only_mid_small': (odd_block_chain,
                              {'num_nodes': 66, 'method': 'midpoint',
                               'unknown_parameter': False}),
    'odd_block_be_small': (odd_block_chain, {}),
    'odd_block_mid_small': (odd_block_chain, {'num_nodes': 130,
                                              'method': 'midpoint'}),
    'delay_be_small': (delay_equation, {}),
    'delay_mid_small': (delay_equation, {'num_nodes': 66,
                                         'method': 'midpoint'}),
    'c99_be_small': (c99_functions, {}),
    'c99_mid_small': (c99_functions, {'method': 'midpoint'}),
    'elementary_be_small': (elementary_functions, {}),
    'elementary_mid_small': (elementary_functions,
                             {'num_nodes': 70, 'method': 'midpoint',
                              'variable_duration': True}),
    'piecewise_be_small': (piecewise_functions, {}),
    'piecewise_mid_small': (piecewise_functions, {'num_nodes': 68,
                                                  'method': 'midpoint'}),
    'config5_standin_24link': (n_link_cart_pendulum,
                               {'num_links': 24, 'num_nodes': 50000,
                                'variable_duration': True}),
    'config5_standin_24link_small': (n_link_cart_pendulum,
                                     {'num_links': 24, 'num_nodes': 6,
                                      'variable_duration': True}),
    # config-5-shaped: variable duration, known + unknown input, exp contact
    # terms, the gait example's instance constraints
    'gaitlike_3link_be_small': (gait_like_pendulum,
                                {'num_links': 3, 'num_nodes': 41}),
    'gaitlike_3link_mid_small': (gait_like_pendulum,
                                 {'num_links': 3, 'num_nodes': 38,
                                  'method': 'midpoint'}),
    'config5_gaitlike_24link': (gait_like_pendulum, {}),
    'config5_gaitlike_24link_small': (gait_like_pendulum, {'num_nodes': 6}),
    # a real muscle-driven DAE with a reference golden (config-5 class)
    # seven-segment planar biped: the structure of the gait model of config 5
    'biped_small': (planar_biped, {'num_nodes': 9}),
    'biped_mid_small': (planar_biped, {'num_nodes': 9,
                                       'method': 'midpoint'}),
    'config5_biped': (planar_biped, {}),
    'one_legged_small': (gallery_problem, {'num_nodes': 43}),
    'config5_one_legged': (gallery_problem, {'num_nodes': 50000}),
}


def build(name):
    """Instantiates the keyword dict for a named config."""
    factory, kwargs = CONFIGS[name]
    return factory(**kwargs)


def make_free(num_free, seed=0, variable_duration=False, interval=0.01):
    """Deterministic synthetic ``free`` vector: dyadic rationals in [-1, 1).

    An integer hash (no libm, no RNG library state) so that every machine and
    both the reference and this package see bit-identical inputs (SURVEY.md
    section 8(c)/(d)).  Values are k/2**20 with k from a 64-bit mix of the
    index and ``seed``; if the problem has a variable duration the last entry
    (h) is set to ``interval``.
    """
    idx = np.arange(num_free, dtype=np.uint64)
    with np.errstate(over='ignore'):
        x = idx + np.uint64((0x9E3779B97F4A7C15*(seed + 1)) % (1 << 64))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    k = (x >> np.uint64(43)).astype(np.int64) - (1 << 20)   # 21 bits signed
    free = k.astype(np.float64)/float(1 << 20)
    if variable_duration:
        free[-1] = interval
    return free
