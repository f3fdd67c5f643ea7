"""Parity on the problems users actually run: every script of the reference's
example gallery that builds with SymPy alone (28 of 31; the other three need
``pygait2d`` / ``pydy`` / ``yeadon``) -- each under the discretisation rule the
script chooses and once more under the other one (``__flipped``: 24 of the 28
use backward Euler) --, against goldens recorded from the REAL reference for
the very arguments each script hands to ``Problem``
(``tests/golden/_gen/gallery_capture.py``; inputs rebuilt from data by
:mod:`gallery_cases`).

* CPU (``-m "not gpu"``): the product's lowering + forward-mode Jacobian
  through the test-only DAG interpreter, and the oracle;
* GPU (``-m gpu``): ``constraints`` / ``jacobian`` / ``jacobian_indices``
  through the C ABI, separate and fused launches.

Bar: int64 indices bit-exact (instance tails as sorted triplets), values
within 1e-10 relative with the per-entry floors of :mod:`golden_util`.
"""
import numpy as np
import pytest

import dag_interp
import gallery_cases as gc
import golden_util as gu

RTOL = 1e-10

#: the oracle is SymPy ``jacobian`` + ``cse`` + gcc per problem; the one
#: fixture that takes it minutes was checked in the build container (the
#: other 25 full fixtures take 25 s together)
ORACLE_HEAVY = {'gallery_ball_rolling_on_spinning_disc',
                'gallery_ball_rolling_on_spinning_disc__flipped'}
ORACLE_CASES = [k for k in gc.FULL if k not in ORACLE_HEAVY]


def _check_attributes(col, meta):
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


def _caps(meta, z):
    N1, M, C = meta['N'] - 1, meta['M'], meta['C']
    ccap, jcap = gu.row_caps(z['jac'][:N1*M*C].reshape(N1, M, C))
    ccap = np.concatenate((ccap.ravel(),
                           np.full(len(z['con']) - N1*M, np.inf)))
    jcap = np.concatenate((jcap.ravel(),
                           np.full(len(z['jac']) - N1*M*C, np.inf)))
    return ccap, jcap


def _check_full(meta, z, col, con, jac, label):
    cb, jb = gu.error_bounds(col, z['free'])
    ccap, jcap = _caps(meta, z)
    gu.assert_close(con, z['con'], RTOL, what=label + ' con', bound=cb,
                    cap=ccap)
    gu.assert_close(jac, z['jac'], RTOL, what=label + ' jac', bound=jb,
                    cap=jcap)


def _check_sampled(meta, z, col, con, jac, label):
    N, M, C = meta['N'], meta['M'], meta['C']
    P = M*C
    nodes = z['nodes']
    blk = jac[:P*(N - 1)].reshape(N - 1, P)
    cb = con[:M*(N - 1)].reshape(M, N - 1)
    cbn, jbn, icb, ijb = gu.error_bounds(col, z['free'], nodes)
    ccap, jcap = gu.row_caps(z['jac_nodes'].reshape(len(nodes), M, C))
    gu.assert_close(blk[nodes], z['jac_nodes'], RTOL,
                    what=label + ' jac nodes', bound=jbn,
                    cap=jcap.reshape(len(nodes), P))
    gu.assert_close(cb[:, nodes], z['con_nodes'], RTOL,
                    what=label + ' con nodes', bound=cbn, cap=ccap)
    scale = float(z['jac_abs_sum'][0])
    gu.assert_close(blk.sum(axis=0), z['jac_entry_sums'], 1e-9,
                    scale=scale/P, what=label + ' jac entry sums')
    gu.assert_close(cb.sum(axis=1), z['con_eq_sums'], 1e-9,
                    scale=float(z['con_abs_sum'][0])/M,
                    what=label + ' con sums')
    gu.assert_close(np.abs(blk).sum(), scale, 1e-9,
                    what=label + ' jac abs sum')
    gu.assert_close(con[M*(N - 1):], z['con_tail'], RTOL,
                    what=label + ' con tail', bound=icb)
    gu.assert_close(jac[P*(N - 1):], z['jac_tail'], RTOL,
                    what=label + ' jac tail', bound=ijb)


def _check_values(meta, z, col, con, jac, label):
    assert len(jac) == meta['nnz']
    (_check_full if meta['kind'] == 'full' else _check_sampled)(
        meta, z, col, con, jac, label)


def _check_indices(meta, z, rows, cols):
    assert rows.dtype == np.int64 and cols.dtype == np.int64
    assert len(rows) == len(cols) == meta['nnz']
    if meta['kind'] == 'full':
        np.testing.assert_array_equal(rows, z['rows'])
        np.testing.assert_array_equal(cols, z['cols'])
        return
    N, P = meta['N'], meta['M']*meta['C']
    nodes = z['nodes']
    np.testing.assert_array_equal(
        rows[:P*(N - 1)].reshape(N - 1, P)[nodes], z['rows_nodes'])
    np.testing.assert_array_equal(
        cols[:P*(N - 1)].reshape(N - 1, P)[nodes], z['cols_nodes'])
    np.testing.assert_array_equal(rows[P*(N - 1):], z['rows_tail'])
    np.testing.assert_array_equal(cols[P*(N - 1):], z['cols_tail'])


@pytest.mark.parametrize('name', gc.NAMES)
def test_gallery_program_matches_reference(name):
    """Lowering, discretisation, symbol ordering, forward-mode Jacobian and
    the instance-constraint index map, without a GPU."""
    from opty_amd import ConstraintCollocator
    meta, z, kw = gc.load(name)
    col = ConstraintCollocator(**kw)
    _check_attributes(col, meta)
    con, jac = dag_interp.evaluate_collocator(col, z['free'])
    _check_values(meta, z, col, con, jac, name + ' (interp)')
    r, c = col._instance_constraints_jacobian_indices()
    tail_r = z['rows'] if meta['kind'] == 'full' else z['rows_tail']
    tail_c = z['cols'] if meta['kind'] == 'full' else z['cols_tail']
    if meta['nnz_inst']:
        np.testing.assert_array_equal(r, tail_r[-meta['nnz_inst']:])
        np.testing.assert_array_equal(c, tail_c[-meta['nnz_inst']:])


@pytest.mark.parametrize('name', ORACLE_CASES)
def test_oracle_matches_gallery_reference(name):
    """The oracle (test infrastructure) against the same reference record."""
    from oracle.collocation_oracle import OracleCollocator
    meta, z, kw = gc.load(name)
    orc = OracleCollocator(name=name, **kw)
    con = orc.generate_constraint_function()(z['free'])
    jac = orc.generate_jacobian_function()(z['free'])
    rows, cols = orc.jacobian_indices()
    base = meta['nnz'] - meta['nnz_inst']
    order = base + np.lexsort((cols[base:], rows[base:]))
    rows[base:], cols[base:], jac[base:] = rows[order], cols[order], \
        jac[order]
    _check_indices(meta, z, rows, cols)
    np.testing.assert_allclose(con, z['con'], rtol=1e-12,
                               atol=1e-12*np.abs(z['con']).max())
    np.testing.assert_allclose(jac, z['jac'], rtol=1e-12,
                               atol=1e-12*np.abs(z['jac']).max())


#: the gallery problems whose parity leans on the tolerance floor (entries
#: that cancel: VERDICT r05 "what's weak" 1) plus a sample of the others
SECOND_OPINION = [k for k in (
    'gallery_light_diffraction', 'gallery_one_legged_time_trial',
    'gallery_light_diffraction__flipped',
    'gallery_one_legged_time_trial__flipped', 'gallery_countersteer',
    'gallery_pendulum_swing_up_fixed_duration', 'gallery_car_around_pylons')
    if k in ORACLE_CASES]


@pytest.mark.parametrize('name', SECOND_OPINION)
def test_gallery_floors_have_an_oracle_second_opinion(name):
    """The per-entry floors of the gallery comparisons come from a running
    error analysis of the PRODUCT's DAG (``golden_util.error_bounds``).  A
    numerically poor rewrite there could widen its own tolerance -- so the
    floors are held to an independent measure on the problems where they
    matter: the term magnitudes of the ORACLE's SymPy expressions
    (``tests/oracle_bounds.py``).  No entry's bound may exceed 32 of them,
    and the reference's values must sit inside their own term magnitudes."""
    import opty_amd
    import oracle_bounds
    from oracle.collocation_oracle import OracleCollocator
    meta, z, kw = gc.load(name)
    if meta['kind'] != 'full':
        pytest.skip('sampled record')
    orc = OracleCollocator(name=name, **kw)
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
        ratio = float(np.max(bound/np.maximum(mag, 1e-300)))
        gu.STATS.setdefault(name + ' ' + what + ' (second opinion)', dict(
            worst_rel=0.0, worst_bound_units=0.0, entries=int(bound.size),
            entries_passed_by_floor=0, worst_rel_passed_by_floor=0.0,
            floor_capped=True))['bound_over_oracle_magnitude'] = ratio
        assert (bound <= 32.0*mag + 1e-300).all(), (what, ratio)


@pytest.mark.gpu
@pytest.mark.parametrize('name', gc.NAMES)
def test_gallery_hip(name):
    """The HIP path through the C ABI: separate launches (IPOPT's call
    pattern), then the fused launch."""
    import opty_amd
    from opty_amd import hip_backend as hb
    meta, z, kw = gc.load(name)
    col = opty_amd.ConstraintCollocator(**kw)
    _check_attributes(col, meta)
    con = col.generate_constraint_function()(z['free'])
    jac = col.generate_jacobian_function()(z['free'])
    rows, cols = col.jacobian_indices()
    _check_indices(meta, z, rows, cols)
    _check_values(meta, z, col, con, jac, name)
    con2 = np.empty_like(con)
    jac2 = hb.pinned_empty(len(jac))
    col.hip.eval_con_jac(z['free'], con2, jac2, hb.HOST)
    _check_values(meta, z, col, con2, jac2, name + ' fused')


#: constraint nodes of the scaled runs: 23 full waves and a ragged one
SCALED_NODES = 1501


@pytest.mark.gpu
@pytest.mark.parametrize('name', [k for k in gc.NAMES
                                  if k not in ORACLE_HEAVY])
def test_gallery_scaled_against_oracle(name):
    """The same problems on ``SCALED_NODES`` collocation nodes (duration
    kept, instance times moved with the grid, known trajectories
    interpolated: ``gallery_cases.rescale``) -- several node blocks, a ragged
    last wave, strips and windows the recorded sizes do not reach -- against
    the oracle (which the recorded sizes pin to the reference)."""
    import opty_amd
    from oracle.collocation_oracle import OracleCollocator
    from examples import problems
    meta, z, kw = gc.load(name)
    kw = gc.rescale(kw, SCALED_NODES)
    col = opty_amd.ConstraintCollocator(**kw)
    orc = OracleCollocator(name=name, **kw)
    vd = col._variable_duration
    free = problems.make_free(col.num_free, seed=11, variable_duration=vd)
    if vd:
        free[-1] = z['free'][-1]
    c_ref = orc.generate_constraint_function()(free)
    j_ref = np.asarray(orc.generate_jacobian_function()(free))
    assert np.isfinite(c_ref).all() and np.isfinite(j_ref).all()
    r_ref, k_ref = orc.jacobian_indices()
    base = col.num_eom*col.num_block_columns*(SCALED_NODES - 1)
    order = base + np.lexsort((k_ref[base:], r_ref[base:]))
    r_ref[base:], k_ref[base:], j_ref[base:] = (r_ref[order], k_ref[order],
                                                j_ref[order])
    con = col.generate_constraint_function()(free)
    jac = col.generate_jacobian_function()(free)
    rows, cols = col.jacobian_indices()
    np.testing.assert_array_equal(rows, r_ref)
    np.testing.assert_array_equal(cols, k_ref)
    cb, jb = gu.error_bounds(col, free)
    N1, M, C = SCALED_NODES - 1, col.num_eom, col.num_block_columns
    ccap, jcap = gu.row_caps(j_ref[:N1*M*C].reshape(N1, M, C))
    ccap = np.concatenate((ccap.ravel(), np.full(len(con) - N1*M, np.inf)))
    jcap = np.concatenate((jcap.ravel(),
                           np.full(len(jac) - N1*M*C, np.inf)))
    gu.assert_close(con, c_ref, RTOL, what=name + ' scaled con', bound=cb,
                    cap=ccap)
    gu.assert_close(jac, j_ref, RTOL, what=name + ' scaled jac', bound=jb,
                    cap=jcap)


@pytest.mark.gpu
@pytest.mark.parametrize('name', gc.NAMES)
def test_gallery_problem_facade(name):
    """``opty_amd.Problem`` built as the script builds the reference's:
    the variable and constraint bound arrays IPOPT is handed
    (``opty/direct_collocation.py:370-440``) are the reference's, bit for bit
    (scalar, per-node and ``eom_bounds`` bounds: 23 of the scripts set
    some), and the callbacks return what the collocator returns."""
    import opty_amd
    meta, z, kw = gc.load(name)
    bounds, eom_bounds, want = gc.facade(name)
    prob = opty_amd.Problem(lambda free: 0.0, lambda free: free,
                            bounds=bounds, eom_bounds=eom_bounds, **kw)
    assert prob.num_free == meta['num_free']
    assert prob.num_constraints == meta['num_constraints']
    np.testing.assert_array_equal(prob.lower_bound, want['lower_bound'])
    np.testing.assert_array_equal(prob.upper_bound, want['upper_bound'])
    np.testing.assert_array_equal(prob._low_con_bounds, want['low_con'])
    np.testing.assert_array_equal(prob._upp_con_bounds, want['upp_con'])
    rows, cols = prob.jacobianstructure()
    _check_indices(meta, z, rows, cols)
    _check_values(meta, z, prob.collocator, prob.constraints(z['free']),
                  prob.jacobian(z['free']), name + ' facade')


OBJECTIVES = [k for k in gc.NAMES if gc.MANIFEST[k].get('has_objective')]


@pytest.mark.gpu
@pytest.mark.parametrize('name', OBJECTIVES)
def test_gallery_objective(name):
    """The objectives the scripts build with ``create_objective_function``
    (``opty/utils.py:329-470``), on the device: value and gradient against
    what the reference's lambdified functions returned."""
    import opty_amd
    args, free, value, grad = gc.objective(name)
    obj, obj_grad = opty_amd.create_objective_function(**args)
    got = obj(free)
    assert abs(got - value) <= 1e-12*max(1.0, abs(value)), (got, value)
    g = obj_grad(free)
    assert g.shape == grad.shape
    np.testing.assert_allclose(g, grad, rtol=1e-12,
                               atol=1e-13*max(1.0, np.abs(grad).max()))
