"""The oracle (``oracle/collocation_oracle.py``) pinned to the reference:

* every golden vector generated from the real reference
  (``tests/golden/_gen/make_golden.py``), values and int64 indices;
* the literal index arrays the reference's own tests hold.
"""
import numpy as np
import pytest
import sympy as sm

import golden_util as gu
from examples import problems
from oracle.collocation_oracle import OracleCollocator, dense_from_coo


_ORACLES = {}


def _oracle(name):
    """One oracle per fixture and session: its symbolic Jacobian (SymPy
    ``jacobian`` + ``cse``) is a minute and a half for the midpoint biped,
    and two tests want it."""
    if name not in _ORACLES:
        _ORACLES[name] = OracleCollocator(name=name.replace('_small', ''),
                                          **problems.build(name))
    return _ORACLES[name]


@pytest.mark.parametrize('name', gu.FULL_FAST)
def test_oracle_matches_reference_full(name):
    meta, z = gu.load(name)
    orc = _oracle(name)
    assert orc.num_free == meta['num_free']
    assert orc.num_constraints == meta['num_constraints']
    assert [str(p) for p in orc.unknown_parameters] == \
        meta['unknown_parameters']
    assert [str(p) for p in orc.known_parameters] == meta['known_parameters']
    assert [str(p) for p in orc.unknown_trajectories] == \
        meta['unknown_trajectories']
    free = z['free']
    np.testing.assert_array_equal(
        free, problems.make_free(orc.num_free, seed=meta['seed'],
                                 variable_duration=bool(meta['s'])))
    rows, cols = orc.jacobian_indices()
    assert rows.dtype == np.int64 and cols.dtype == np.int64
    np.testing.assert_array_equal(rows, z['rows'])
    np.testing.assert_array_equal(cols, z['cols'])
    r2, c2 = orc.jacobian_indices_loop()
    np.testing.assert_array_equal(r2, z['rows'])
    np.testing.assert_array_equal(c2, z['cols'])
    # 1e-12 relative per entry, floors from the entries' own error bounds
    import opty_amd
    cb, jb = gu.error_bounds(
        opty_amd.ConstraintCollocator(**problems.build(name)), free)
    # (floors capped at 1e-12 of the largest entry of the entry's own block
    # row of the REFERENCE's Jacobian: the product's error analysis cannot
    # widen the oracle's tolerance beyond that)
    ccap, jcap = gu.caps_for(z['jac'], len(z['con']), meta['N'] - 1,
                             meta['M'], meta['C'])
    gu.assert_close(orc.generate_constraint_function()(free), z['con'],
                    1e-12, what=name + ' con (oracle)', bound=cb, cap=ccap)
    gu.assert_close(orc.generate_jacobian_function()(free), z['jac'], 1e-12,
                    what=name + ' jac (oracle)', bound=jb, cap=jcap)


def test_oracle_matches_reference_config2_full_size():
    """BASELINE config 2 at its real size (N = 10 000, midpoint, 4 instance
    constraints): sampled nodes + checksums from the reference."""
    name = 'config2_pendulum'
    meta, z = gu.load(name)
    orc = _oracle(name)
    N, M, C = meta['N'], meta['M'], meta['C']
    P = M*C
    free = problems.make_free(orc.num_free, seed=meta['seed'])
    con = orc.generate_constraint_function()(free)
    jac = orc.generate_jacobian_function()(free)
    rows, cols = orc.jacobian_indices()
    nodes = z['nodes']
    blk = jac[:P*(N - 1)].reshape(N - 1, P)
    gu.assert_close(blk[nodes], z['jac_nodes'], 1e-12, what='jac nodes')
    gu.assert_close(con[:M*(N - 1)].reshape(M, N - 1)[:, nodes],
                    z['con_nodes'], 1e-12, what='con nodes')
    gu.assert_close(blk.sum(axis=0), z['jac_entry_sums'], 1e-10,
                    what='jac sums')
    gu.assert_close(con[M*(N - 1):], z['con_tail'], 1e-12, what='con tail')
    gu.assert_close(jac[P*(N - 1):], z['jac_tail'], 1e-12, what='jac tail')
    np.testing.assert_array_equal(rows[P*(N - 1):], z['rows_tail'])
    np.testing.assert_array_equal(cols[P*(N - 1):], z['cols_tail'])
    np.testing.assert_array_equal(
        rows[:P*(N - 1)].reshape(N - 1, P)[nodes], z['rows_nodes'])
    np.testing.assert_array_equal(
        cols[:P*(N - 1)].reshape(N - 1, P)[nodes], z['cols_nodes'])


def test_literal_indices_unknown_trajectories():
    """Known-answer arrays restated from the reference's
    ``TestConstraintCollocatorUnknownTrajectories``
    (``opty/tests/test_direct_collocation.py:1163-1177``): mass-spring-damper,
    N = 4, backward Euler, one unknown trajectory f, one unknown parameter k."""
    m, c, k, t = sm.symbols('m, c, k, t')
    x, v, f = [s(t) for s in sm.symbols('x, v, f', cls=sm.Function)]
    eom = sm.Matrix([x.diff() - v, m*v.diff() + c*v + k*x - f])
    orc = OracleCollocator(eom, (x, v), 4, 2.0,
                           known_parameter_map={m: 1.0, c: 2.0},
                           time_symbol=t, name='msd_unknown_traj')
    rows, cols = orc.jacobian_indices()
    np.testing.assert_array_equal(rows, np.array(
        [0, 0, 0, 0, 0, 0, 3, 3, 3, 3, 3, 3,
         1, 1, 1, 1, 1, 1, 4, 4, 4, 4, 4, 4,
         2, 2, 2, 2, 2, 2, 5, 5, 5, 5, 5, 5]))
    np.testing.assert_array_equal(cols, np.array(
        [1, 5, 0, 4, 9, 12, 1, 5, 0, 4, 9, 12,
         2, 6, 1, 5, 10, 12, 2, 6, 1, 5, 10, 12,
         3, 7, 2, 6, 11, 12, 3, 7, 2, 6, 11, 12]))


def test_hand_derived_values_mass_spring_damper():
    """Hand arithmetic in the style of ``TestConstraintCollocator``
    (``opty/tests/test_direct_collocation.py:791-966``): backward Euler
    defects and their dense-block partials for N = 4."""
    kw = problems.mass_spring_damper(num_nodes=4, interval=2.0)
    orc = OracleCollocator(name='msd', **kw)
    h, mm, cc = 2.0, 1.0, 2.0
    xs = np.array([1.0, 2.0, 3.0, 4.0])
    vs = np.array([5.0, 6.0, 7.0, 8.0])
    kk = 10.0
    fs = np.linspace(1.0, 4.0, 4)
    free = np.hstack((xs, vs, kk))
    con = orc.generate_constraint_function()(free)
    exp1 = (xs[1:] - xs[:-1])/h - vs[1:]
    exp2 = mm*(vs[1:] - vs[:-1])/h + cc*vs[1:] + kk*xs[1:] - fs[1:]
    np.testing.assert_allclose(con, np.hstack((exp1, exp2)), rtol=1e-13)
    jac = orc.generate_jacobian_function()(free)
    rows, cols = orc.jacobian_indices()
    dense = dense_from_coo(jac, rows, cols)
    expected = np.zeros((6, 9))
    for i in range(3):
        expected[i, i + 1] = 1/h         # d eom1 / d x_i
        expected[i, i] = -1/h            # d eom1 / d x_p
        expected[i, 4 + i + 1] = -1.0    # d eom1 / d v_i
        expected[3 + i, i + 1] = kk      # d eom2 / d x_i
        expected[3 + i, 4 + i + 1] = mm/h + cc
        expected[3 + i, 4 + i] = -mm/h
        expected[3 + i, 8] = xs[i + 1]   # d eom2 / d k
    np.testing.assert_allclose(dense, expected, rtol=1e-13, atol=1e-15)


def test_error_behaviour():
    kw = problems.mass_spring_damper(num_nodes=5)
    with pytest.raises(ValueError):
        OracleCollocator(**dict(kw, integration_method='rk4'))
    bad = dict(kw)
    bad['known_trajectory_map'] = {list(kw['known_trajectory_map'])[0]:
                                   np.zeros(3)}
    with pytest.raises(ValueError):
        OracleCollocator(**bad)


@pytest.mark.parametrize('name', ['instance_constraints',
                                  'variable_duration', 'msd_backward_euler',
                                  'msd_midpoint'])
def test_reference_unit_test_fixtures(name):
    """The reference's own N = 4 fixtures (literal index arrays, hand-derived
    values) against the oracle."""
    import reference_cases
    case = reference_cases.ALL[name]()
    orc = OracleCollocator(name='ref_' + name, **case['kw'])
    con = orc.generate_constraint_function()(case['free'])
    jac = orc.generate_jacobian_function()(case['free'])
    rows, cols = orc.jacobian_indices()
    np.testing.assert_allclose(con, case['con'], rtol=1e-12)
    if case['rows'] is not None:
        np.testing.assert_array_equal(rows, case['rows'])
        np.testing.assert_array_equal(cols, case['cols'])
    np.testing.assert_allclose(dense_from_coo(jac, rows, cols),
                               case['dense'], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('name', gu.FULL_FAST)
def test_tolerance_floors_are_justified_by_the_oracle(name):
    """The per-entry floors of the parity tolerance come from a running
    rounding-error bound of the PRODUCT's expression DAG
    (``golden_util.error_bounds``); a numerically poor rewrite there could
    widen its own tolerance.  Hold them to an independent measure: the term
    magnitudes of the ORACLE's SymPy expressions (``tests/oracle_bounds.py``:
    every sum adds absolute values) -- no entry's bound may exceed 32 of them
    (measured over the fixtures: 1.0 ... 14.6) -- and the reference's values
    themselves must not exceed their own term magnitudes."""
    import opty_amd
    import oracle_bounds
    meta, z = gu.load(name)
    kw = problems.build(name)
    orc = _oracle(name)
    orc.generate_jacobian_function()(z['free'])
    cmag, jmag = oracle_bounds.magnitudes(orc, z['free'])
    col = opty_amd.ConstraintCollocator(**kw)
    cb, jb = gu.error_bounds(col, z['free'])
    N1, M, C = meta['N'] - 1, meta['M'], meta['C']
    for bound, mag, ref, what in (
            (cb[:M*N1], cmag, z['con'][:M*N1], 'con'),
            (jb[:M*C*N1], jmag, z['jac'][:M*C*N1], 'jac')):
        assert np.isfinite(mag).all(), what
        assert (np.abs(ref) <= mag*(1 + 1e-12) + 1e-300).all(), what
        assert (bound <= 32.0*mag + 1e-300).all(), (
            what, float(np.max(bound/np.maximum(mag, 1e-300))))
