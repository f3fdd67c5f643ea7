"""CPU tests of the product's host logic and codegen (no GPU, no oracle
dependency except as the checker): the DAG lowering + forward-mode Jacobian
evaluated by a test-only NumPy interpreter must reproduce the reference's
golden vectors; the emitted HIP must build for gfx950; the C-ABI library must
load and export every symbol of ``include/opty_hip.h``."""
import ctypes
import os
import re
import shutil

import numpy as np
import pytest
import sympy as sm

import golden_util as gu
import dag_interp
from opty_amd import ConstraintCollocator, Problem, parse_free
from examples import problems
from opty_amd import hip_backend as hb
from opty_amd.codegen import ir
from opty_amd.codegen.emit_hip import EmitOptions, emit_module
from opty_amd.codegen.lower import Lowerer, forward_jacobian

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.mark.parametrize('name', gu.FULL)
def test_program_matches_reference(name):
    meta, z = gu.load(name)
    col = ConstraintCollocator(**problems.build(name))
    assert col.num_free == meta['num_free']
    assert col.num_constraints == meta['num_constraints']
    assert col.num_block_columns == meta['C']
    for attr, key in (('state_symbols', 'states'),
                      ('known_parameters', 'known_parameters'),
                      ('unknown_parameters', 'unknown_parameters'),
                      ('known_input_trajectories', 'known_trajectories'),
                      ('unknown_input_trajectories',
                       'unknown_trajectories')):
        assert [str(s) for s in getattr(col, attr)] == meta[key], attr
    con, jac = dag_interp.evaluate_collocator(col, z['free'])
    cb, jb = gu.error_bounds(col, z['free'])
    N1, M, C = meta['N'] - 1, meta['M'], meta['C']
    ccap, jcap = gu.row_caps(z['jac'][:N1*M*C].reshape(N1, M, C))
    ccap = np.concatenate((ccap.ravel(),
                           np.full(len(z['con']) - N1*M, np.inf)))
    jcap = np.concatenate((jcap.ravel(),
                           np.full(len(z['jac']) - N1*M*C, np.inf)))
    gu.assert_close(con, z['con'], 1e-10, what=name + ' con (interp)',
                    bound=cb, cap=ccap)
    gu.assert_close(jac, z['jac'], 1e-10, what=name + ' jac (interp)',
                    bound=jb, cap=jcap)
    r, c = col._instance_constraints_jacobian_indices()
    if meta['nnz_inst']:
        np.testing.assert_array_equal(r, z['rows'][-meta['nnz_inst']:])
        np.testing.assert_array_equal(c, z['cols'][-meta['nnz_inst']:])


def test_dag_simplification_and_sharing():
    d = ir.DAG()
    x, y = d.input('cur', 0), d.input('cur', 1)
    assert d.add(x, d.zero) == x and d.mul(x, d.one) == x
    assert d.mul(x, d.zero) == d.zero and d.sub(x, x) == d.zero
    assert d.neg(d.neg(x)) == x
    assert d.add(x, y) == d.add(y, x) and d.mul(x, y) == d.mul(y, x)
    assert d.add(x, d.neg(y)) == d.sub(x, y)
    assert d.value(d.mul(d.const(2.0), d.const(3.0))) == 6.0
    assert d.powi(x, 1) == x and d.powi(x, 0) == d.one
    assert d.pow(x, d.const(2.0)) == d.powi(x, 2)
    assert d.unary('cos', d.neg(x)) == d.unary('cos', x)
    p = d.input('par', 0)
    assert d.uni[d.mul(p, p)] and not d.uni[d.mul(p, x)]


def test_forward_jacobian_against_sympy():
    """Every derivative rule, against SymPy's own differentiation."""
    a, b, c = sm.symbols('a b c', real=True)
    exprs = [sm.sin(a)*sm.cos(b) + sm.tan(c), sm.exp(a*b)/(1 + c**2),
             sm.sqrt(a**2 + b**2 + 1)*sm.log(c**2 + 2), a**3*b**-2 + c**b,
             sm.atan2(a, b) + sm.asin(c/3) + sm.acos(c/4) + sm.atan(a*b),
             sm.sinh(a) + sm.cosh(b)*sm.tanh(c) + sm.Abs(a - b),
             sm.Max(a, b*c) + sm.Min(a, b),
             sm.erf(a*b) + sm.erfc(c/2) + sm.asinh(a) + sm.acosh(1 + b)
             + sm.atanh(c/3)]
    # the rest of the C99 printer's table (sympy.codegen.cfunctions)
    from sympy.codegen import cfunctions as cf
    exprs += [cf.log1p(a*b) + cf.expm1(-c) + cf.log2(a + b) + cf.log10(c + 2),
              cf.exp2(a - b)*cf.Cbrt(1 + c**2) + cf.hypot(a, b*c)
              + cf.fma(a, b, c) + cf.Sqrt(a + c), sm.sinc(a*b)*c]
    d = ir.DAG()
    table = {s: d.input('cur', k) for k, s in enumerate((a, b, c))}
    low = Lowerer(d, table)
    outs = [low.lower(e) for e in exprs]
    jac = forward_jacobian(d, outs, [table[s] for s in (a, b, c)])
    rng = np.random.default_rng(1)
    vals = rng.uniform(0.3, 1.7, size=(3, 50))
    num = dag_interp.evaluate(d, [n for row in jac for n in row],
                              lambda kind, k: vals[k])
    sym = sm.Matrix(exprs).jacobian([a, b, c])
    import oracle_bounds
    f = sm.lambdify((a, b, c), sym, oracle_bounds._MODULES)
    for i in range(50):
        ref = np.array(f(*vals[:, i]), dtype=float).ravel()
        got = np.array([np.broadcast_to(v, (50,))[i] for v in num])
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)


def test_parse_free_and_layout():
    free = np.arange(2*5 + 2*5 + 3 + 1, dtype=float)
    st, sp, cs, h = parse_free(free, 2, 2, 5, variable_duration=True)
    assert st.shape == (2, 5) and sp.shape == (2, 5)
    np.testing.assert_array_equal(cs, [20, 21, 22])
    assert h == 23
    st, sp, cs = parse_free(free[:13], 2, 0, 5)
    assert sp is None and len(cs) == 3
    st, sp, cs = parse_free(free[:15], 2, 1, 5)
    assert sp.shape == (5,)


def test_api_errors_match_reference_behaviour():
    kw = problems.mass_spring_damper(num_nodes=5)
    with pytest.raises(ValueError, match='backend'):
        ConstraintCollocator(backend='cython', **kw)
    with pytest.raises(ValueError, match='integration method'):
        ConstraintCollocator(**dict(kw, integration_method='rk4'))
    with pytest.raises(ValueError, match='unique'):
        ConstraintCollocator(**dict(
            kw, state_symbols=kw['state_symbols'] + kw['state_symbols'][:1]))
    f = list(kw['known_trajectory_map'])[0]
    with pytest.raises(ValueError, match='not length'):
        ConstraintCollocator(**dict(kw, known_trajectory_map={f: np.ones(3)}))
    # too few states for the derivatives in the equations
    with pytest.raises(ValueError, match='Too few state'):
        ConstraintCollocator(**dict(kw,
                                    state_symbols=kw['state_symbols'][:1]))
    col = ConstraintCollocator(**kw)
    assert col.num_states == 2 and col.num_eom == 2
    assert [s.name for s in col.current_discrete_state_symbols] == ['xi',
                                                                    'vi']
    assert [s.name for s in col.previous_discrete_state_symbols] == ['xp',
                                                                     'vp']
    assert str(col.time_interval_symbol) == 'h_opty'


def test_known_and_unknown_order():
    """Ordering rules of ``test_known_and_unknown_order``
    (``opty/tests/test_direct_collocation.py:2042-2088``)."""
    from sympy.physics.mechanics.models import n_link_pendulum_on_cart
    import sympy.physics.mechanics as me
    from opty_amd.utils import sort_sympy
    me.dynamicsymbols._t = sm.Symbol('t')
    kane = n_link_pendulum_on_cart(n=3, cart_force=True, joint_torques=True)
    states = kane.q.col_join(kane.u)
    eom = kane.mass_matrix_full @ states.diff() - kane.forcing_full
    g, l0, l1, l2, m0, m1, m2, m3, t = sort_sympy(eom.free_symbols)
    par_map = {l1: 1.5, l0: 1.0, m3: 2.5, g: 9.81, m1: 1.5}
    funcs = sort_sympy(f for f in me.find_dynamicsymbols(eom)
                       if not isinstance(f, sm.Derivative))
    F, T1, T2, T3 = funcs[:4]
    N = 51
    col = ConstraintCollocator(eom, states, N, 0.1,
                               known_parameter_map=par_map,
                               known_trajectory_map={T1: np.zeros(N),
                                                     F: np.ones(N)},
                               time_symbol=t)
    assert col.input_trajectories == (T1, F, T2, T3)
    assert col.known_parameters == (l1, l0, m3, g, m1)
    assert col.unknown_parameters == (l2, m0, m2)
    assert col.unknown_input_trajectories == (T2, T3)
    assert col.num_free == (8 + 2)*N + 3


def test_problem_facade_bounds_without_ipopt():
    """``Problem`` bound arrays (``opty/direct_collocation.py:370-440``) need
    the device only for the callbacks; check validation that happens first."""
    kw = problems.mass_spring_damper(num_nodes=5)
    with pytest.raises(ValueError, match='No time derivatives'):
        Problem(lambda f: 0.0, lambda f: f, sm.Matrix([sm.Symbol('a')]),
                kw['state_symbols'], 5, 1.0)


def test_emit_builds_for_gfx950(tmp_path, monkeypatch):
    """The printed HIP for the 10-link pendulum cross-compiles for gfx950 with
    zero scratch (no register spills to memory) in the Jacobian kernels."""
    if shutil.which('hipcc') is None and not os.path.exists(
            '/opt/rocm/bin/hipcc'):
        pytest.skip('hipcc not available')
    col = ConstraintCollocator(**problems.build('config3_10link_small'))
    source, meta = col.generate_source()
    assert meta['P'] == 990 and meta['kernels']['jac']['groups'] >= 1
    for kern in ('opty_con', 'opty_jac', 'opty_conjac', 'opty_uni'):
        assert 'void __launch_bounds__(64)\n%s(' % kern in source
    hsaco = hb.compile_module(source, cache_dir=str(tmp_path))
    assert os.path.getsize(hsaco) > 0
    # N does not enter the generated code: with the printer's own rules one
    # code object serves every large N ...
    monkeypatch.setenv('OPTY_LAUNCH_PLANS', 'off')
    col2 = ConstraintCollocator(**problems.build('config3_10link'))
    assert col2.generate_source()[1]['sha'] == meta['sha']
    # ... and a measured launch plan (opty_amd/launch_plans.json) only moves
    # the strip counts: the fused kernel of BASELINE config 3 keeps the seed
    monkeypatch.delenv('OPTY_LAUNCH_PLANS')
    col3 = ConstraintCollocator(**problems.build('config3_10link'))
    meta3 = col3.generate_source()[1]
    assert meta3['kernels']['conjac']['sha'] == meta['kernels']['conjac']['sha']
    assert meta3['P'] == meta['P']


def test_group_ranges_cover_block():
    col = ConstraintCollocator(**problems.build('pend3_link_midpoint_small'))
    prog = col._build_program()
    for opts in (EmitOptions(), EmitOptions(chunk=8, groups=3),
                 EmitOptions(interleave=0), EmitOptions(chunk=2, groups=100)):
        _, meta = emit_module(prog, opts)
        # strips of all waves, in block order, tile the block
        g = sorted(rg for grp in meta['groups'] for rg in grp)
        assert g[0][0] == 0 and g[-1][1] == prog.P
        assert all(a[1] == b[0] for a, b in zip(g, g[1:]))
        assert all(a[0] % 2 == 0 for a in g)


def test_c_abi_library_exports_every_declared_symbol():
    """``libopty_hip.so`` loads (no GPU needed) and exports exactly what
    ``include/opty_hip.h`` declares; ``create`` fails loudly without a
    device -- there is no CPU fallback."""
    header = open(os.path.join(REPO, 'include', 'opty_hip.h')).read()
    declared = set(re.findall(r'\b(opty_hip_[a-z_]+)\s*\(', header))
    declared -= {'opty_hip_problem', 'opty_hip_desc'}
    assert declared == set(hb._SIGNATURES), declared ^ set(hb._SIGNATURES)
    hb.build_runtime_library()
    lib = hb.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    # the build verification's device side is a library of its own
    # (include/opty_hip_referee.h -> libopty_hip_referee.so): the runtime
    # library does not carry it
    rheader = open(os.path.join(REPO, 'include',
                                'opty_hip_referee.h')).read()
    rdeclared = set(re.findall(r'\b(opty_hip_[a-z_]+)\s*\(', rheader))
    assert rdeclared == set(hb._REFEREE_SIGNATURES), \
        rdeclared ^ set(hb._REFEREE_SIGNATURES)
    ref = hb.load_referee()
    for name in rdeclared:
        assert hasattr(ref, name), name
        if name != 'opty_hip_referee_last_error':
            assert not hasattr(lib, name), name
    import torch
    if not torch.cuda.is_available():
        assert lib.opty_hip_device_count() == 0
        desc = hb._Desc(N=10, n=1, M=1, C=2, P=2, jac_wgs_per_block=1,
                        fused_wgs_per_block=1, con_wgs_per_block=1,
                        jac_waves_per_wg=1, fused_waves_per_wg=1,
                        con_waves_per_wg=1)
        handle = ctypes.c_void_p()
        rc = lib.opty_hip_create(ctypes.byref(desc), b'/nonexistent.hsaco',
                                 ctypes.byref(handle))
        assert rc != 0
        assert b'no HIP device' in lib.opty_hip_last_error()
        col = ConstraintCollocator(**problems.build('msd_be_small'))
        with pytest.raises(hb.HipBackendError, match='no CPU fallback'):
            col.generate_constraint_function()


def test_implicit_known_trajectories_known_answer():
    """``test_implicit_known_traj`` of the reference, on the DAG."""
    import implicit_case
    kw, free, con_exp, jac_exp, sy = implicit_case.build()
    col = ConstraintCollocator(**kw)
    assert col._deriv_in_knw_traj
    th, om = sy['theta_of_x'], sy['omega_of_v']
    assert col.known_input_trajectories == (
        om.diff(sy['v']), om, sy['s'], th, th.diff(sy['x']))
    assert col.unknown_input_trajectories == (sy['f'],)
    names = [str(a) for a in col.current_known_discrete_specified_symbols]
    assert names == ['domegai_dvi', 'omegai(vi)', 'si', 'thetai(xi)',
                     'dthetai_dxi']
    repl = col._create_function_replacements()
    assert sorted(str(v) for v in repl.values()) == sorted(
        ['dthetai_dxi', 'thetaixi', 'dthetan_dxn', 'thetanxn', 'domegai_dvi',
         'omegaivi', 'domegan_dvn', 'omeganvn'])
    con, jac = dag_interp.evaluate_collocator(col, free)
    np.testing.assert_allclose(con, con_exp)
    np.testing.assert_allclose(jac, jac_exp)
    # theta(x, v) and theta(x) + theta(v) are rejected like in the reference
    x, v = sy['x'], sy['v']
    bad = sm.Function('theta', real=True)(x, v)
    eom = sm.Matrix([x.diff() - v, sy['m']*v.diff() + sm.sin(bad)])
    with pytest.raises(ValueError, match='more than one'):
        ConstraintCollocator(eom, (x, v), 4, 0.1,
                             known_trajectory_map={bad: np.zeros(4)},
                             time_symbol=sy['t'])
    th_v = sm.Function('theta', real=True)(v)
    eom = sm.Matrix([x.diff() - v + th_v, sy['m']*v.diff() + sm.sin(th)])
    with pytest.raises(ValueError, match='Repeated'):
        ConstraintCollocator(eom, (x, v), 4, 0.1, time_symbol=sy['t'],
                             known_trajectory_map={th_v: np.zeros(4),
                                                   th: np.zeros(4)})


def _interp_objective(case, t):
    """Objective value and gradient from the objective DAG, with the
    quadrature rules applied on the host (test-only)."""
    from opty_amd.objective import build_objective_program
    states, inputs, unknowns = case['args']
    dag, roots, n, q, r = build_objective_program(
        case['expr'], states, inputs, unknowns, case['method'], t)
    g_quad, dp_quad, dz_node, b_val, db_val = roots
    free, h = case['free'], case['h']
    N = (len(free) - r)//(n + q)
    rows = free[:(n + q)*N].reshape(n + q, N)
    par = free[(n + q)*N:]

    def inputs_at(cur_idx, adj_idx):
        def f(kind, k):
            if kind == 'cur':
                return rows[k, cur_idx]
            if kind == 'adj':
                return rows[k, adj_idx]
            return par[k]
        return f

    nodes = np.arange(N)
    ones = np.ones(N)
    dz = dag_interp.evaluate(dag, dz_node, inputs_at(nodes, nodes))
    if case['method'] == 'backward euler':
        wq = np.hstack((0, np.ones(N - 1)))
        wg = wq
        quad = dag_interp.evaluate(dag, [g_quad] + dp_quad,
                                   inputs_at(nodes, nodes))
        sums = [float(np.sum(np.atleast_1d(v)*ones*wq)) for v in quad]
    else:
        wg = np.hstack((0.5, np.ones(N - 2), 0.5))
        quad = dag_interp.evaluate(dag, [g_quad] + dp_quad,
                                   inputs_at(nodes[:-1], nodes[1:]))
        sums = [float(np.sum(np.atleast_1d(v)*np.ones(N - 1))) for v in quad]
    bb = dag_interp.evaluate(dag, [b_val] + db_val, inputs_at(0, 0))
    value = h*sums[0] + float(bb[0])
    grad = np.hstack([h*wg*np.atleast_1d(v)*ones for v in dz] +
                     [np.array([h*sums[1 + k] + float(bb[1 + k])
                                for k in range(r)])])
    return value, grad


def test_objective_program_known_answers():
    """``TestCreateObjectiveFunction`` of the reference
    (``opty/tests/test_utils.py:67-219``) on the objective DAG."""
    import objective_cases
    t, cases = objective_cases.cases()
    for case in cases:
        value, grad = _interp_objective(case, t)
        np.testing.assert_allclose(value, case['value'], rtol=1e-12,
                                   err_msg=case['name'])
        np.testing.assert_allclose(grad, case['grad'], rtol=1e-12,
                                   atol=1e-15, err_msg=case['name'])


def test_objective_rejections():
    from opty_amd.objective import build_objective_program
    t = sm.symbols('t')
    x = sm.Function('x')(t)
    with pytest.raises(NotImplementedError):
        build_objective_program(sm.Integral(x**2, t), [x], [], [],
                                'not_existing_method', t)
    with pytest.raises(NotImplementedError):
        build_objective_program(sm.Integral(x**2, (t, 0, 1)), [x], [], [],
                                'backward euler', t)
    with pytest.raises(NotImplementedError):
        build_objective_program(sm.Integral(x**2, t)**2, [x], [], [],
                                'backward euler', t)
    with pytest.raises(NotImplementedError):
        build_objective_program(x**2, [x], [], [], 'backward euler', t)


@pytest.mark.parametrize('name', ['config3_10link_small',
                                  'pend3_link_midpoint_small',
                                  'chaplygin_mid_small'])
def test_pruned_block_pattern(name):
    """``prune_zeros=True`` keeps exactly the entries whose partial is not
    identically zero, and the kept values are the reference's values."""
    meta, z = gu.load(name)
    dense = ConstraintCollocator(**problems.build(name))
    pruned = ConstraintCollocator(prune_zeros=True, **problems.build(name))
    pd, pp = dense._build_program(), pruned._build_program()
    assert pd.P == meta['M']*meta['C'] and pp.P < pd.P
    kept = [(j, k) for (j, k), node in zip(pd.pattern, pd.jac_out)
            if node != pd.dag.zero]
    assert pp.pattern == kept
    _, jac = dag_interp.evaluate_collocator(pruned, z['free'])
    N, P = meta['N'], meta['M']*meta['C']
    full = z['jac'][:P*(N - 1)].reshape(N - 1, P)
    sel = [j*meta['C'] + k for j, k in kept]
    gu.assert_close(jac[:pp.P*(N - 1)].reshape(N - 1, pp.P), full[:, sel],
                    1e-10, what='pruned values')
    # everything that was dropped is exactly zero in the reference's output
    dropped = sorted(set(range(P)) - set(sel))
    assert not full[:, dropped].any()


@pytest.mark.parametrize('name', ['instance_constraints',
                                  'variable_duration', 'msd_backward_euler',
                                  'msd_midpoint'])
def test_reference_unit_test_fixtures_on_dag(name):
    """The reference's own N = 4 fixtures on the product's host logic + DAG
    (instance-constraint free-index map and literal tail indices included)."""
    import reference_cases
    case = reference_cases.ALL[name]()
    col = ConstraintCollocator(**case['kw'])
    con, jac = dag_interp.evaluate_collocator(col, case['free'])
    np.testing.assert_allclose(con, case['con'], rtol=1e-12)
    if 'free_index' in case:
        got = {str(k): v for k, v in
               col.instance_constraints_free_index_map.items()}
        assert got == case['free_index']
    if case['rows'] is not None:
        r, c = col._instance_constraints_jacobian_indices()
        np.testing.assert_array_equal(r, case['rows'][-len(r):])
        np.testing.assert_array_equal(c, case['cols'][-len(c):])


@pytest.mark.parametrize('name,prune', [
    ('config3_10link_small', False), ('config3_10link_small', True),
    ('pend3_link_midpoint_small', False),
    ('pend2_link_vardur_unkmass_small', True),
    ('elementary_mid_small', False), ('chaplygin_mid_small', True),
    ('one_eom_be_small', False)])
def test_csr_layout_program_order(name, prune):
    """``jacobian_layout='csr'``: the stored entries of a block are grouped
    by equation and ascend in the reference's column index
    (``jacobian_indices``, golden) at every node; the values are the
    reference's values, re-ordered."""
    meta, z = gu.load(name)
    col = ConstraintCollocator(jacobian_layout='csr', prune_zeros=prune,
                               **problems.build(name))
    prog = col._build_program()
    N, C, P = meta['N'], meta['C'], meta['M']*meta['C']
    assert prog.layout == 'csr' and prog.P == len(prog.pattern)
    assert [j for j, _ in prog.pattern] == sorted(j for j, _ in prog.pattern)
    for j in range(prog.M):
        a, b = prog.row_start[j], prog.row_start[j + 1]
        assert all(jj == j for jj, _ in prog.pattern[a:b])
    gcols = z['cols'][:P*(N - 1)].reshape(N - 1, P)
    sel = [j*C + k for j, k in prog.pattern]
    for i in (0, N - 2):
        c = gcols[i, sel]
        for j in range(prog.M):
            row = c[prog.row_start[j]:prog.row_start[j + 1]]
            assert np.all(np.diff(row) > 0), (i, j, row)
    _, jac = dag_interp.evaluate_collocator(col, z['free'])
    full = z['jac'][:P*(N - 1)].reshape(N - 1, P)
    gu.assert_close(jac[:prog.P*(N - 1)].reshape(N - 1, prog.P),
                    full[:, sel], 1e-10, what='csr values')
    if prune:
        assert not full[:, sorted(set(range(P)) - set(sel))].any()


@pytest.mark.parametrize('L', [1, 2, 3, 5, 7, 10, 12, 16, 24, 40, 45, 55, 63, 64])
def test_flat_flush_model(L):
    """Python model of ``opty_flush_flat`` (opty_amd/csrc/opty_device.h):
    every element of the span is written exactly once, from the right
    (entry, node) of the tile, with 16-byte aligned pairs, for all sixteen
    positions of the span start within a 128-byte line and ragged node
    counts.  It sweeps the rows of the row-sorted layout and -- ``L = P`` --
    the whole 64-node tile of a small (P < 64) node-major block
    (``emit_hip._strip_flat``)."""
    steps = (64*L + 15 + 127)//128
    for phase in range(16):
        for nvalid in (64, 1, 37):
            total = nvalid*L
            written = {}
            for it in range(steps):
                for lane in range(64):
                    g = it*64 + lane
                    f = 2*g - phase
                    f1 = f + 1
                    if not (f >= 0 and f1 < total):
                        continue            # out-of-range buffer offset
                    assert 16*g < (total + phase)*8   # inside the buffer
                    for e in (f, f1):
                        assert e not in written
                        written[e] = (e % L, e//L)       # tile[k][node]
            if phase & 1:                                # lane 0
                assert 0 not in written
                written[0] = (0, 0)
            if (total + phase) & 1:                      # lane 1
                assert total - 1 not in written
                written[total - 1] = ((total - 1) % L, (total - 1)//L)
            assert sorted(written) == list(range(total))
            for g, (k, nd) in written.items():
                assert g == nd*L + k and nd < nvalid


def test_objective_modules_build_for_gfx950(tmp_path):
    """The objective / gradient kernels of every reference case cross-compile
    (they include the same device header as the collocation kernels)."""
    import objective_cases
    from opty_amd import hip_backend as hb
    from opty_amd.objective import build_objective_program, _emit
    t, cases = objective_cases.cases()
    seen = set()
    for case in cases:
        states, inputs, unknowns = case['args']
        dag, roots, n, q, r = build_objective_program(
            case['expr'], states, inputs, unknowns, case['method'], t)
        source = _emit(dag, n + q, r, 11, case['method'], roots)
        if source not in seen:
            seen.add(source)
            assert os.path.getsize(hb.compile_module(source, str(tmp_path)))


def test_strip_count_rules():
    """The launch geometry the printer picks (emit_hip.py: ``group_ranges``,
    ``_fit_one_round``; measurements in profiles/r02_strip_sweeps.txt):
    opty_jac about one wave per 100 entries, the fused kernel about
    0.286 sqrt(P) strips, and a node shard that would need 1.x rounds gets the
    coarser cut that fits the 1024 resident waves."""
    from opty_amd.codegen.emit_hip import _fit_one_round, RESIDENT_WAVES
    col = ConstraintCollocator(**problems.build('config3_10link_small'))
    prog = col._build_program()

    def waves(node_blocks):
        k = emit_module(prog, EmitOptions(),
                        node_blocks=node_blocks)[1]['kernels']
        return (k['jac']['groups'], k['conjac']['groups'],
                k['conjac']['waves_per_wg'])

    assert waves(None) == (10, 10, 1)           # 10 strips; 9 strips + 1 con
    assert waves(1563) == (10, 10, 1)           # N = 100 000
    assert waves(391) == (10, 10, 1)            # 4 shards: several rounds
    # 8 shards (12 500 nodes): two waves per SIMD, 6 strips + 2 constraint
    # waves (the 22 rows cut for half the register file: one wave needs 258
    # VGPRs, two spills under the 256 cap) = 2 workgroups of 4 waves per
    # block, all 1568 waves resident
    jac, fused, wpw = waves(196)
    assert (jac, fused, wpw) == (6, 8, 4)
    assert 196*fused <= 2*RESIDENT_WAVES
    src = emit_module(prog, EmitOptions(), node_blocks=196)[0]
    assert 'amdgpu_waves_per_eu(2, 2)' in src and 'chunk=16' in src
    assert waves(1) == (10, 10, 1)              # tiny launches fit anyway
    # explicit strip counts are never overridden
    k = emit_module(prog, EmitOptions(groups=6), node_blocks=196)[1]
    assert k['kernels']['jac']['groups'] == 6
    assert k['kernels']['jac']['waves_per_wg'] == 1
    # hand-set printer options switch the small-launch geometry off; the
    # coarser one-round cut at one wave per SIMD is what is left
    k = emit_module(prog, EmitOptions(waves=1), node_blocks=196)[1]
    assert (k['kernels']['jac']['groups'],
            k['kernels']['conjac']['groups']) == (4, 5)
    # the rule itself
    assert _fit_one_round(9, 1, 196, live_groups=5) == 4
    assert _fit_one_round(9, 1, 1563, live_groups=5) == 9
    assert _fit_one_round(9, 1, 100) == 9
    assert _fit_one_round(32, 5, 100, live_groups=16) == 32   # nothing fits


def test_merge_fixed_free_static_helper():
    """``ConstraintCollocator._merge_fixed_free`` with the reference's static
    signature: the cases of ``opty/tests/test_direct_collocation.py:1337-1400``
    plus a callable known value (``opty/direct_collocation.py:2916-2917``)."""
    import sympy as sm
    merge = ConstraintCollocator._merge_fixed_free
    free = np.ones(10)
    m, c, k = sm.symbols('m, c, k')
    np.testing.assert_allclose(
        merge((m, c, k), {m: 1.0, c: 2.0}, np.array([3.0]), 'par', free),
        [1.0, 2.0, 3.0])
    a, b, c, d = sm.symbols('a, b, c, d')
    np.testing.assert_allclose(
        merge((a, b, c, d), {a: 1.0, b: 2.0}, np.array([3.0, 4.0]), 'par',
              free), [1.0, 2.0, 3.0, 4.0])
    t = sm.symbols('t')
    f, k = [g(t) for g in sm.symbols('f, k', cls=sm.Function)]
    np.testing.assert_allclose(
        merge((f, k), {f: np.array([1.0, 2.0])}, np.array([3.0, 4.0]),
              'traj', free), [[1.0, 2.0], [3.0, 4.0]])
    a, b, c, d = [g(t) for g in sm.symbols('a, b, c, d', cls=sm.Function)]
    np.testing.assert_allclose(
        merge([a, b, c, d], {a: np.array([1.0, 2.0]),
                             b: np.array([3.0, 4.0])},
              np.array([[5.0, 6.0], [7.0, 8.0]]), 'traj', free),
        [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]])
    np.testing.assert_allclose(
        merge([a, b], {a: lambda fr: 2.0*fr[:2]}, np.array([5.0, 6.0]),
              'traj', free), [[2.0, 2.0], [5.0, 6.0]])


@pytest.mark.parametrize('name', ['elementary_be_small', 'chaplygin_be_small',
                                  'one_eom_be_small', 'msd_mid_small'])
def test_small_blocks_are_flushed_as_one_span(name):
    """Node-major blocks with P < 64: one wave stages the whole P x 64 tile
    and sweeps the contiguous 64*P-double span with ``opty_flush_flat<P>``
    (whole 128-byte lines) instead of K-entry pieces per node."""
    col = ConstraintCollocator(**problems.build(name))
    prog = col._build_program()
    assert prog.P < 64
    source, meta = col.generate_source()
    assert 'opty_flush16<' not in source and 'opty_flush8<' not in source
    assert source.count('opty_flush_flat<%d>(ring, jrow, nvalid, lane)'
                        % prog.P) == 2           # opty_jac and opty_conjac
    # every entry is staged exactly once per kernel, in its own tile row
    body = source[source.index('\nopty_jac('):]
    body = body[:body.index('\n}\n')]
    rows = [int(m) for m in re.findall(r'ring\[(\d+) \+ lane\] = ', body)]
    assert rows == [e*65 for e in range(prog.P)]
    assert meta['kernels']['jac']['lds_bytes'] >= 8*65*prog.P


@pytest.mark.parametrize('name', ['config3_10link_small',
                                  'pend2_link_vardur_unkmass_small',
                                  'gaitlike_3link_be_small',
                                  'chaplygin_be_small', 'msd_be_small'])
def test_varying_entries_against_the_reference_values(name):
    """``program.varying_entries``: every block entry NOT in the list has one
    value at all nodes of the reference's golden Jacobian (and, for problems
    without unknown parameters / free interval, that value does not depend on
    ``free``: the oracle is not needed, the static entries' DAG nodes have no
    trajectory or free-tail input)."""
    from opty_amd.codegen.program import varying_entries
    meta, z = gu.load(name)
    col = ConstraintCollocator(**problems.build(name))
    prog = col._build_program()
    var = varying_entries(prog)
    assert var == sorted(set(var)) and all(0 <= e < prog.P for e in var)
    P, ncn = prog.P, meta['N'] - 1
    blk = z['jac'][:P*ncn].reshape(ncn, P)
    static = sorted(set(range(P)) - set(var))
    assert (blk[:, static] == blk[0, static]).all()
    # the list is not vacuous: most listed entries do change between nodes
    moving = (blk[:, var] != blk[0, var]).any(axis=0)
    if not (meta['r'] or meta['s']):
        assert moving.sum() >= 0.6*len(var)
    else:
        # entries that read the free tail count as varying
        dag = prog.dag
        tail = {i for i in range(len(dag)) if dag.op[i] == ir.INPUT and (
            (dag.args[i][0] == 'par' and
             prog.pars[dag.args[i][1]][0] == 'tail') or
            (dag.args[i][0] == 'h' and prog.h[0] != 'fixed'))}
        for e in static:
            assert not tail & set(dag.reachable([prog.jac_out[e]]))
    # entries that repeat another one's expression (filled on the host from
    # it, opty_hip_set_entry_copies) hold the same values in the reference's
    # vector -- which evaluates each of them separately
    from opty_amd.codegen.program import varying_copies
    unique, copies = varying_copies(prog)
    assert sorted(unique + [d for d, _ in copies]) == var
    assert all(s in unique and s < d for d, s in copies)
    for d, s in copies:
        np.testing.assert_allclose(blk[:, d], blk[:, s], rtol=1e-13, atol=0)
    if name == 'config3_10link_small':
        assert (len(unique), len(copies)) == (275, 55)


@pytest.mark.parametrize('name,launch_nodes,layout', [
    ('config3_10link_small', None, 'coo'),
    ('config3_10link_small', 12500, 'coo'),
    ('config5_standin_24link_small', None, 'coo'),
    ('config5_standin_24link_small', None, 'csr'),
    ('config5_standin_24link_small', 6250, 'coo'),
    ('config5_gaitlike_24link_small', 6250, 'coo'),
    ('elementary_mid_small', None, 'coo')])
def test_built_kernels_do_not_spill_vector_registers(name, launch_nodes,
                                                     layout):
    """The code object a collocator uses has no kernel that spills VECTOR
    registers to scratch memory (``ConstraintCollocator._build_code_object``
    re-cuts until that holds): builds of the 24-link systems that did spill
    returned wrong, run-to-run different values on MI355X
    (``hip_backend.vgpr_spills``)."""
    col = ConstraintCollocator(launch_nodes=launch_nodes,
                               jacobian_layout=layout,
                               **problems.build(name))
    hsaco, meta = col._build_code_object()
    assert hb.vgpr_spills(hsaco) == {}
    res = hb.kernel_resources(hsaco)
    for kern in ('opty_con', 'opty_jac', 'opty_conjac'):
        assert res[kern]['.private_segment_fixed_size'] == 0
        assert res[kern]['.vgpr_count'] <= 512


def test_small_problems_are_single_launch_modules():
    """Problems small enough for the runtime's latency path cost what their
    launches cost: the instance tails ride in the main kernels (one more
    workgroup, ``inst_folded``), and a node-invariant table that opty_uni
    would refill before every evaluation (unknown parameters, variable
    duration) is not used at all -- those values are computed in every lane.
    Large launches keep the table; their instance tails are folded as
    well."""
    # BASELINE config 2: 4 instance constraints, a static table
    col = ConstraintCollocator(**problems.build('config2_pendulum'))
    src, meta = col.generate_source()
    assert meta['inst_folded'] and meta['num_uniform'] > 0
    assert not meta['uniform_dynamic']
    assert col._descriptor(meta)['inst_folded'] == 1
    for kern in ('opty_con', 'opty_jac', 'opty_conjac'):
        body = src[src.index(kern + '('):]
        body = body[:body.index('\nextern "C"')]
        assert 'if (blockIdx.x >= ((node_end - node_begin + 63)/64 + 7)' \
            in body and 'con[2LL*con_stride + 3]' in body
    # the launch opty_hip_eval_instance makes for node shards is still there
    assert 'opty_inst(' in src
    # unknown parameters / variable duration, small: no table
    for name in ('msd_be_small', 'vardur_pendulum_small'):
        col = ConstraintCollocator(**problems.build(name))
        src, meta = col.generate_source()
        assert meta['num_uniform'] == 0 and not meta['uniform_dynamic']
        assert 'uni_c[' not in src
        off = ConstraintCollocator(
            emit_options=EmitOptions(inline_uniform=0, inline_dynamic=0,
                                     fold_instance=0),
            **problems.build(name))
        src0, meta0 = off.generate_source()
        assert meta0['num_uniform'] > 0 and meta0['uniform_dynamic']
        assert not meta0['inst_folded']
        # with a table: only what does NOT depend on `free` is in it -- the
        # few operations behind 1/h or an unknown parameter are evaluated in
        # the lanes, and opty_uni does not run before every evaluation
        part = ConstraintCollocator(
            emit_options=EmitOptions(inline_uniform=0, fold_instance=0),
            **problems.build(name))
        src1, meta1 = part.generate_source()
        assert not meta1['uniform_dynamic']
        assert meta1['num_uniform'] < meta0['num_uniform']
    # the tails are folded at every size (a launch of opty_inst is 5-10 % of
    # a 50-70 us evaluation too); the table stays a table
    big = ConstraintCollocator(**dict(
        problems.CONFIGS['config2_pendulum'][0](num_nodes=100001)))
    src, meta = big.generate_source()
    assert meta['inst_folded'] and meta['num_uniform'] > 0
    unfolded = ConstraintCollocator(
        emit_options=EmitOptions(fold_instance=0), **dict(
            problems.CONFIGS['config2_pendulum'][0](num_nodes=100001)))
    src, meta = unfolded.generate_source()
    assert not meta['inst_folded'] and 'blockIdx.x >= ((node_end' not in src
    # a table too large to recompute per lane stays a table
    col = ConstraintCollocator(**problems.build(
        'config5_gaitlike_24link_small'))
    assert col.generate_source()[1]['num_uniform'] > 1000


def test_gamma_of_a_trajectory_has_no_derivative_rule():
    """``gamma`` / ``loggamma`` are lowered (``tgamma`` / ``lgamma``), but
    their derivative is the digamma function, which C -- and therefore the
    reference's generated code -- does not have: usable on known parameters
    (``c99_functions``), rejected where the Jacobian would need it."""
    import sympy.physics.mechanics as me
    from opty_amd.codegen.lower import LoweringError
    t = sm.Symbol('t')
    me.dynamicsymbols._t = t
    x = me.dynamicsymbols('x', real=True)
    eom = sm.Matrix([x.diff() + sm.gamma(2 + x**2)])
    col = ConstraintCollocator(eom, (x,), 11, 0.1, time_symbol=t)
    with pytest.raises(LoweringError, match='derivative'):
        col.generate_source()


def test_unavoidable_spills_fall_back_to_the_safe_scheduler(monkeypatch,
                                                            caplog):
    """When no cut of a module is free of vector-register spills, the
    least-spilling one is built without the pre-RA scheduler stage that the
    wrong values of round 3 followed (``hip_backend.SAFE_SCHEDULER_FLAGS``),
    with a warning that names ``cross_check()``."""
    import logging
    col = ConstraintCollocator(**problems.build('msd_be_small'))
    plain, _ = col._build_code_object()
    calls = []
    real = hb.compile_module

    def spy(source, *args, **kw):
        calls.append(tuple(kw.get('extra_flags', ())))
        return real(source, *args, **kw)
    monkeypatch.setattr(hb, 'compile_module', spy)
    monkeypatch.setattr(hb, 'vgpr_spills',
                        lambda hsaco, kernels=None: {'opty_conjac': 3,
                                                     'opty_jac': 1})
    with caplog.at_level(logging.WARNING, logger='opty_amd'):
        hsaco, meta = col._build_code_object()
    assert calls[-1] == hb.SAFE_SCHEDULER_FLAGS
    assert all(c == () for c in calls[:-1])
    assert hsaco != plain and os.path.exists(hsaco)
    assert any('cross_check' in r.getMessage() for r in caplog.records)
    assert not meta.get('banned_kernels')


@pytest.mark.parametrize('spilling,bit', [
    ('opty_jac', hb.ROUTE_NO_JAC_KERNEL),
    ('opty_conjac', hb.ROUTE_NO_FUSED_KERNEL)])
def test_one_spilling_jacobian_kernel_is_banned_not_shipped(
        monkeypatch, caplog, spilling, bit):
    """When ONE of the two Jacobian kernels spills vector registers whatever
    the cut and the other is clean, the spilling one is never launched: the
    build is used as it is, marked (``meta['banned_kernels']``), and the
    descriptor tells the library to serve that entry point with the clean
    kernel (``OPTY_HIP_ROUTE_NO_*``) -- no last-resort scheduler flags, no
    spilling kernel in service (the biped's 6 250-node shard, r05)."""
    import logging
    col = ConstraintCollocator(**problems.build('msd_be_small'))
    plain, _ = col._build_code_object()
    calls = []
    real = hb.compile_module

    def spy(source, *args, **kw):
        calls.append(tuple(kw.get('extra_flags', ())))
        return real(source, *args, **kw)
    monkeypatch.setattr(hb, 'compile_module', spy)
    monkeypatch.setattr(hb, 'vgpr_spills',
                        lambda hsaco, kernels=None: {spilling: 2})
    with caplog.at_level(logging.WARNING, logger='opty_amd'):
        hsaco, meta = col._build_code_object()
    assert all(c == () for c in calls)          # no last-resort flags
    assert meta['banned_kernels'] == [spilling]
    assert meta['vector_spills'] == {spilling: 2}
    assert not [r for r in caplog.records if r.levelno >= logging.WARNING]
    bits = col._routing_bits(meta)
    assert bits & bit and bits & hb.ROUTE_CALIBRATE
    assert col._descriptor(meta)['routing'] == bits


def test_real_biped_shard_bans_its_spilling_kernel():
    """The build the bench line of r05 carried with a warning: the
    seven-segment biped's 6 250-node launches, whose ``opty_jac`` spills two
    vector registers at every cut.  It is banned; what stays in service is
    spill-free."""
    col = ConstraintCollocator(launch_nodes=6250,
                               **problems.build('biped_small'))
    hsaco, meta = col.prebuild()
    banned = meta.get('banned_kernels', [])
    in_service = {k: v for k, v in hb.vgpr_spills(hsaco).items()
                  if k not in banned}
    assert in_service == {}, in_service
    if banned:
        assert banned == ['opty_jac']
        assert col._routing_bits(meta) & hb.ROUTE_NO_JAC_KERNEL


def test_hand_set_workgroup_width_is_narrowed_to_the_lds():
    """64-entry chunks in a 4-wave workgroup next to the slab of a 22-state
    system need more than a CU's 160 KB of LDS: the printer narrows the
    workgroup instead of leaving the failure to hipcc (found by
    tools/geometry_soak.py)."""
    col = ConstraintCollocator(
        emit_options=EmitOptions(chunk=64, waves=4, groups=8),
        **problems.build('config3_10link_small'))
    src, meta = col.generate_source()
    for kern in meta['kernels'].values():
        assert kern['lds_bytes'] <= 160*1024
    assert meta['kernels']['jac']['waves_per_wg'] < 4


@pytest.mark.parametrize('name,hot', [
    ('config5_standin_24link_small', True), ('one_legged_small', True),
    ('config3_10link_small', True),      # opty_con: 255 VGPRs, 2 SGPR spills
    ('pend3_link_midpoint_small', False), ('msd_be_small', False)])
def test_high_pressure_builds_are_marked_for_verification(name, hot):
    """Kernels at the edge of the register file (>= 480 VGPRs or spilled
    SGPRs: the 24-link stand-ins, the musculoskeletal model) are the ones
    ``ConstraintCollocator._verify_build`` holds to the instruction tape
    before a handle is handed out."""
    col = ConstraintCollocator(**problems.build(name))
    hsaco, meta = col.prebuild()
    marked = hb.high_pressure_kernels(hsaco)
    assert bool(marked) == hot, marked


def test_pinned_plans_are_built_exactly_as_recorded(tmp_path, monkeypatch):
    """A ``"pinned"`` entry of the plan file -- the build that replaced one
    the verification refused -- is reproduced as recorded (options and hipcc
    switches), without the spill loop; caller-fixed options ignore it."""
    from opty_amd import launch_plan
    from opty_amd.codegen.emit_hip import EmitOptions
    monkeypatch.setenv('OPTY_LAUNCH_PLANS', str(tmp_path/'plans.json'))
    kw = problems.build('pend3_link_midpoint_small')
    col = ConstraintCollocator(tmp_dir=str(tmp_path), **kw)
    assert col._pinned_build() is None
    default, _ = col._build_code_object()
    opts = EmitOptions(groups=3, fused_groups=4, chunk=16)
    launch_plan.record(
        launch_plan.key_of(col._build_program(), col._launch_blocks()),
        dict(options=launch_plan.options_kwargs(opts),
             pinned=dict(opt_level='-O1', label='test')))
    assert launch_plan.options_kwargs(opts) == dict(
        groups=3, fused_groups=4, chunk=16)
    col2 = ConstraintCollocator(tmp_dir=str(tmp_path), **kw)
    pinned_opts, how = col2._pinned_build()
    assert pinned_opts.key() == opts.key() and how['opt_level'] == '-O1'
    calls = []
    real = hb.compile_module

    def spy(source, *args, **kwargs):
        calls.append(kwargs)
        return real(source, *args, **kwargs)
    monkeypatch.setattr(hb, 'compile_module', spy)
    hsaco, meta = col2._build_code_object()
    assert hsaco != default and len(calls) == 1
    assert calls[0]['opt_level'] == '-O1'
    assert meta['geometry']['jac'] == 3 and meta['geometry']['fused'] == 4
    assert meta['chunk'] == 16
    col3 = ConstraintCollocator(tmp_dir=str(tmp_path),
                                emit_options=EmitOptions(), **kw)
    assert col3._pinned_build() is None


def test_automatic_work_aware_cut():
    """Blocks whose even cut would make the launch wait for recomputed
    arithmetic -- the seven-segment biped: 24 even strips evaluate 4.7 times
    the block's operations -- are cut where the work is, into the fewest
    strips the register file allows; both Jacobian kernels use that cut.
    Store-bound blocks keep their even strips; an explicit strip count is an
    even cut as before."""
    from opty_amd.codegen import emit_hip as eh
    col = ConstraintCollocator(**problems.build('biped_small'))
    prog = col._build_program()
    src, meta = emit_module(prog, EmitOptions(), node_blocks=782)
    geo = meta['geometry']
    assert geo['cut'] == 'work' and geo['jac'] == geo['fused'] <= 6
    assert meta['groups'] == meta['fused_groups']
    strips = [tuple(rg) for grp in meta['groups'] for rg in grp]
    assert strips[0][0] == 0 and strips[-1][1] == prog.P
    assert all(a[1] == b[0] and a[1] % 16 == 0
               for a, b in zip(strips, strips[1:]))
    w = eh._ModuleWriter(prog, EmitOptions())
    leaf = lambda i: w._is_vec_input(i) or w._uniform_leaf(i)
    work = sum(w._weighted_cost(*rg) for rg in strips)
    for e0, e1 in strips:
        chunks = [[prog.jac_out[v % prog.P] for v in range(a, b)]
                  for a, b in w._chunks(e0, w._virtual_end(e1))]
        assert eh._max_live(prog.dag, chunks, leaf) <= eh.WORK_CUT_MAX_LIVE
    _, even = emit_module(prog, EmitOptions(cut='even'), node_blocks=782)
    assert even['geometry']['cut'] == 'even'
    work_even = sum(w._weighted_cost(*rg) for grp in even['groups']
                    for rg in grp)
    assert work <= 0.8*work_even
    # no strip count admits the bounds: None, and the explicit option falls
    # back on even strips
    assert w._work_cut(2, 16, prog.P//16) is None
    _, two = emit_module(prog, EmitOptions(cut='work', groups=2,
                                           fused_groups=2), node_blocks=782)
    assert [rg for grp in two['groups'] for rg in grp] == \
        [[0, 384], [384, prog.P]]
    # store-bound: even strips
    for name in ('config3_10link_small', 'pend3_link_midpoint_small',
                 'one_legged_small'):
        col = ConstraintCollocator(**problems.build(name))
        _, m = emit_module(col._build_program(), EmitOptions(),
                           node_blocks=782)
        assert m['geometry']['cut'] == 'even', name


def test_spill_loop_falls_back_on_recomputation_per_chunk():
    """The muscle-driven leg under the midpoint rule: every cut of its block
    spills 24+ vector registers (with or without the constraint rows in the
    Jacobian waves); the build loop's last phase -- 16-entry chunks whose
    temporaries are dropped at every chunk boundary (``EmitOptions.forget``)
    -- finds a build without vector spills, and that is the one used."""
    import gallery_cases as gc
    _, _, kw = gc.load('gallery_one_legged_time_trial__flipped')
    col = ConstraintCollocator(**kw)
    hsaco, meta = col._build_code_object()
    assert hb.vgpr_spills(hsaco) == {}
    assert meta['geometry']['chunk'] == 16
    assert 'forget=1' in col._built_source.splitlines()[1]
    # the script's own rule (backward Euler) needs no fallback
    _, _, kw = gc.load('gallery_one_legged_time_trial')
    col = ConstraintCollocator(**kw)
    hsaco, meta = col._build_code_object()
    assert hb.vgpr_spills(hsaco) == {}
    assert 'forget=1' not in col._built_source.splitlines()[1]


def test_kernel_metadata_guard_fails_closed(tmp_path):
    """A code object whose metadata cannot be read, or that holds none of the
    expected kernels, is an error -- not "no spills"."""
    bogus = tmp_path/'x.hsaco'
    bogus.write_bytes(b'not a code object')
    with pytest.raises(hb.HipBackendError):
        hb.vgpr_spills(str(bogus))
    col = ConstraintCollocator(**problems.build('msd_be_small'))
    hsaco, _ = col._build_code_object()
    with pytest.raises(hb.HipBackendError, match='none of the kernels'):
        hb.vgpr_spills(hsaco, kernels=('no_such_kernel',))
    assert hb.vgpr_spills(hsaco) == {}
    assert os.path.exists(hsaco + '.resources.json')


def test_scatter_pool_honours_the_process_affinity_mask():
    """The host threads that scatter the varying Jacobian entries stay inside
    the affinity mask of the thread that created the pool and never outnumber
    its CPUs (a ``taskset`` / cpuset of the host application);
    ``OPTY_HIP_HOST_AFFINITY=wide`` lifts the cap."""
    import subprocess
    import sys
    code = ('from opty_amd import hip_backend as hb; hb.set_host_threads(5); '
            'print(hb.host_threads())')
    cpu = sorted(os.sched_getaffinity(0))[0]

    def run(env_extra):
        env = dict(os.environ, **env_extra)
        env.pop('OPTY_HIP_HOST_THREADS', None)
        out = subprocess.run(['taskset', '-c', str(cpu), sys.executable, '-c',
                              code], capture_output=True, text=True,
                             cwd=REPO, env=env, check=True)
        return int(out.stdout.strip().splitlines()[-1])
    if shutil.which('taskset') is None:
        pytest.skip('no taskset')
    assert run({}) == 1
    assert run({'OPTY_HIP_HOST_AFFINITY': 'wide'}) == 5


def test_public_attributes_exist_and_are_read_only():
    """The reference's public read-only attributes
    (``opty/direct_collocation.py:1556-1892``; its
    ``test_attributes_read_only`` lists them) exist under the same names and
    cannot be assigned; ``integration_method`` alone has a setter."""
    col = ConstraintCollocator(**problems.build('gaitlike_3link_be_small'))
    names = '''current_discrete_state_symbols
        current_discrete_specified_symbols
        current_known_discrete_specified_symbols
        current_unknown_discrete_specified_symbols discrete_eom eom
        input_trajectories instance_constraints known_input_trajectories
        known_parameters known_parameter_map known_trajectory_map
        next_known_discrete_specified_symbols next_discrete_state_symbols
        next_discrete_specified_symbols
        next_unknown_discrete_specified_symbols node_time_interval
        num_collocation_nodes num_constraints num_free
        num_input_trajectories num_instance_constraints
        num_known_input_trajectories num_known_parameters num_parameters
        num_states num_eom num_unknown_input_trajectories
        num_unknown_parameters parameters parallel
        previous_discrete_state_symbols show_compile_output
        state_derivative_symbols state_symbols time_interval_symbol
        time_symbol tmp_dir unknown_input_trajectories
        unknown_parameters'''.split()
    for name in names:
        getattr(col, name)
        with pytest.raises(AttributeError):
            setattr(col, name, 5)
    col.integration_method = 'midpoint'
    assert col.integration_method == 'midpoint'
    assert col.num_block_columns == 2*col.num_states + \
        2*col.num_unknown_input_trajectories + \
        col.num_unknown_parameters + 1


def test_printer_expansions_against_sympy():
    """Functions outside the C math table are lowered through the expansion
    the reference's C99 printer prints for them (``lower._printer_expansion``:
    printer hooks of ``sympy.physics.biomechanics``, ``implemented_function``,
    the printer's rewrite table, ``UnevaluatedExpr``): values and forward-mode
    derivatives against SymPy's own evaluation and differentiation."""
    import sympy.physics.biomechanics as bm
    from sympy.utilities.lambdify import implemented_function
    a, b, c = sm.symbols('a b c', real=True)
    xx = sm.Symbol('xx')
    # (the printer inlines SymPy Lambdas only, not Python callables)
    sq = implemented_function('sq', sm.Lambda(xx, xx**2 + sm.sin(xx)))
    exprs = [
        bm.FiberForceLengthActiveDeGroote2016.with_defaults(a)*b,
        bm.FiberForceLengthPassiveDeGroote2016.with_defaults(a + c/4),
        bm.FiberForceVelocityDeGroote2016.with_defaults(b - 1),
        bm.TendonForceLengthDeGroote2016.with_defaults(1 + c/20),
        bm.FiberForceVelocityInverseDeGroote2016.with_defaults(a),
        bm.TendonForceLengthInverseDeGroote2016.with_defaults(b),
        sm.acot(a*b) + sm.asec(2 + c) + sm.acsc(2 + a),
        sq(a*c) + sm.UnevaluatedExpr(a + b)*c,
        sm.sec(a) + sm.csc(b) + sm.cot(c) + sm.coth(a) + sm.sech(b) +
        sm.csch(c),
    ]
    d = ir.DAG()
    table = {s: d.input('cur', k) for k, s in enumerate((a, b, c))}
    low = Lowerer(d, table)
    outs = [low.lower(e) for e in exprs]
    jac = forward_jacobian(d, outs, [table[s] for s in (a, b, c)])
    rng = np.random.default_rng(3)
    vals = rng.uniform(0.6, 1.4, size=(3, 40))
    got = dag_interp.evaluate(d, outs, lambda kind, k: vals[k])
    gjac = dag_interp.evaluate(d, [n for row in jac for n in row],
                               lambda kind, k: vals[k])
    # the same functions written out for SymPy (it cannot differentiate an
    # implemented function or look through UnevaluatedExpr)
    plain = list(exprs)
    plain[7] = (a*c)**2 + sm.sin(a*c) + (a + b)*c
    plain = [e.doit() for e in plain]
    f = sm.lambdify((a, b, c), plain, 'numpy')
    df = sm.lambdify((a, b, c), [sm.diff(e, s) for e in plain
                                 for s in (a, b, c)], 'numpy')
    ref = [np.broadcast_to(np.asarray(v, dtype=float), (40,))
           for v in f(*vals)]
    dref = [np.broadcast_to(np.asarray(v, dtype=float), (40,))
            for v in df(*vals)]
    for k in range(len(exprs)):
        np.testing.assert_allclose(np.broadcast_to(got[k], (40,)), ref[k],
                                   rtol=1e-12, atol=1e-13, err_msg=str(k))
    for k in range(3*len(exprs)):
        np.testing.assert_allclose(np.broadcast_to(gjac[k], (40,)), dref[k],
                                   rtol=1e-11, atol=1e-12, err_msg=str(k))
    # an undefined function has no expansion, here as in the reference
    from opty_amd.codegen.lower import LoweringError
    with pytest.raises(LoweringError):
        low.lower(sm.Function('mystery')(a))


def test_automatic_parameter_specialisation_follows_the_scalar_spills():
    """``specialize_parameters=None`` (the default): a problem whose generic
    fused kernel spills hundreds of scalar registers into vector lanes -- the
    muscle-driven leg -- gets its known parameters printed as literals
    without being asked; a small problem, a caller who said ``False``, a
    ``deterministic`` build and a caller who fixed the printer options keep
    the generic module."""
    kw = problems.build('one_legged_small')
    col = ConstraintCollocator(**kw)
    hsaco, meta = col.prebuild()
    assert col._auto_specialized is True and meta['auto_specialized']
    assert col._specialize and col._literal_values
    assert 'uni_c[' not in col._built_source.split('opty_conjac')[1][:200000]
    spilled = hb.cached_kernel_resources(hsaco)['opty_conjac'][
        '.sgpr_spill_count']
    assert spilled < ConstraintCollocator._AUTO_SPECIALIZE_SGPR_SPILLS
    never = ConstraintCollocator(specialize_parameters=False, **kw)
    h2, m2 = never.prebuild()
    assert not never._specialize and not m2.get('auto_specialized')
    assert hb.cached_kernel_resources(h2)['opty_conjac'][
        '.sgpr_spill_count'] >= ConstraintCollocator._AUTO_SPECIALIZE_SGPR_SPILLS
    assert h2 != hsaco
    det = ConstraintCollocator(deterministic=True, **kw)
    det.prebuild()
    assert not det._specialize
    fixed = ConstraintCollocator(emit_options=never._built_options, **kw)
    fixed.prebuild()
    assert not fixed._specialize
    small = ConstraintCollocator(**problems.build('msd_be_small'))
    small.prebuild()
    assert not small._specialize and small._auto_specialized is False
    with pytest.raises(ValueError, match='specialize_parameters'):
        ConstraintCollocator(specialize_parameters='yes', **kw)


def test_integration_stub_descriptor_matches_the_abi():
    """INTEGRATION.md shows the ctypes binding a maintainer of the reference
    would add: its descriptor must be the header's, field for field (it went
    stale between ABI versions 4 and 7 once)."""
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    blk = text[text.index('class OptyHipDesc'):
               text.index('class OptyHipBinding')]
    names = re.findall(r"'([a-zA-Z_]+)'", blk)
    assert names == [n for n, _ in hb._Desc._fields_]
    assert 'ABI version %d' % hb.ABI_VERSION in blk
    header = open(os.path.join(REPO, 'include', 'opty_hip.h')).read()
    struct = header[header.index('typedef struct opty_hip_desc {'):
                    header.index('} opty_hip_desc;')]
    fields = re.findall(r'^\s+(?:int64_t|int32_t|float)\s+([a-z_A-Z]+)',
                        struct, re.M)
    assert fields == names
