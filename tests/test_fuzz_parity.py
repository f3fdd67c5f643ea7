"""Fuzzed parity: seeded random problems (``tests/random_problems.py``: n <= 6,
known / unknown parameters and trajectories, both discretisations, variable
duration, M = n or n + 1, 1-3 instance constraints, every lowered function)
against the oracle -- the product's expression DAG through the test
interpreter on CPU, the HIP kernels through the C ABI on the GPU.

The tolerance does not come from the product: 1e-10 relative per entry, an
entry that cancels being held to 1e-10 of the largest entry of its own
equation row at its own node of the ORACLE's Jacobian."""
import numpy as np
import pytest

import random_problems as rp
from examples import problems

RTOL = 1e-10
_WORST = {}


def _reference(seed):
    from oracle.collocation_oracle import OracleCollocator
    kw = rp.generate(seed)
    orc = OracleCollocator(name='fuzz%d' % seed, **kw)
    free = problems.make_free(orc.num_free, seed=seed,
                              variable_duration=orc.variable_duration,
                              interval=0.02)
    con = orc.generate_constraint_function()(free)
    jac = np.asarray(orc.generate_jacobian_function()(free))
    rows, cols = orc.jacobian_indices()
    return kw, orc, free, con, jac, rows, cols


def _check(tag, seed, orc, con, jac, c_ref, j_ref):
    N1, M, C = orc.N - 1, orc.M, orc.C
    assert np.isfinite(j_ref).all() and np.isfinite(c_ref).all()
    rowmax = np.abs(j_ref[:N1*M*C].reshape(N1, M, C)).max(axis=2)
    ref_j = np.concatenate([np.repeat(rowmax, C, axis=1).ravel(),
                            np.abs(j_ref[N1*M*C:])])
    ref_c = np.concatenate([rowmax.T.ravel(),
                            np.maximum(np.abs(c_ref[N1*M:]), 1.0)])
    worst = 0.0
    for got, want, ref, what in ((jac, j_ref, ref_j, 'jac'),
                                 (con, c_ref, ref_c, 'con')):
        assert got.shape == want.shape, (seed, what)
        err = np.abs(got - want)/np.maximum(np.abs(want),
                                            np.maximum(ref, 1e-300))
        k = int(np.nanargmax(err)) if err.size else 0
        assert np.isfinite(got).all() and (err <= RTOL).all(), (
            tag, seed, what, k, got[k], want[k])
        worst = max(worst, float(err.max()) if err.size else 0.0)
        import golden_util as gu
        gu.note_floor_only('fuzz %s %s' % (tag, what), np.abs(got - want),
                           want, RTOL*np.maximum(np.abs(want), ref), RTOL)
    _WORST.setdefault(tag, []).append(worst)


@pytest.mark.parametrize('seed', rp.SEEDS)
def test_expression_dag_against_the_oracle(seed):
    import opty_amd
    import dag_interp
    kw, orc, free, c_ref, j_ref, rows, cols = _reference(seed)
    col = opty_amd.ConstraintCollocator(**kw)
    assert col.num_free == orc.num_free
    assert col.num_constraints == orc.num_constraints
    con, jac = dag_interp.evaluate_collocator(col, free)
    _check('dag', seed, orc, con, jac, c_ref, j_ref)
    r, c = col._instance_constraints_jacobian_indices()
    np.testing.assert_array_equal(r, rows[len(rows) - len(r):])
    np.testing.assert_array_equal(c, cols[len(cols) - len(c):])


def test_the_generator_covers_what_it_claims():
    import sympy as sm
    seen, methods, variable, algebraic, two_atom = set(), set(), set(), 0, 0
    for seed in rp.SEEDS:
        kw = rp.generate(seed)
        eom = kw['equations_of_motion']
        for cls in (sm.Piecewise, sm.Max, sm.Min, sm.Abs, sm.atan2, sm.erf,
                    sm.tan, sm.exp, sm.log, sm.asin, sm.acos, sm.asinh,
                    sm.sinh, sm.cosh, sm.tanh, sm.atan, sm.sin, sm.cos):
            if eom.has(cls):
                seen.add(cls.__name__)
        if any(p.exp.is_Rational and not p.exp.is_Integer
               for p in eom.atoms(sm.Pow)):
            seen.add('pow')
        methods.add(kw['integration_method'])
        variable.add(isinstance(kw['node_time_interval'], sm.Symbol))
        algebraic += eom.shape[0] > len(kw['state_symbols'])
        two_atom += any(len(c.atoms(sm.Function)) > 1
                        for c in kw['instance_constraints'])
    assert len(seen) == 19, sorted(seen)
    assert methods == {'backward euler', 'midpoint'}
    assert variable == {True, False} and algebraic >= 3 and two_atom >= 5
    assert len(rp.SEEDS) >= 30


@pytest.mark.gpu
@pytest.mark.parametrize('seed', rp.SEEDS)
def test_hip_kernels_against_the_oracle(seed):
    import opty_amd
    kw, orc, free, c_ref, j_ref, rows, cols = _reference(seed)
    col = opty_amd.ConstraintCollocator(**kw)
    con = col.generate_constraint_function()(free)
    jac = np.array(col.generate_jacobian_function()(free))
    _check('hip', seed, orc, con, jac, c_ref, j_ref)
    r, c = col.jacobian_indices()
    assert r.dtype == np.int64
    np.testing.assert_array_equal(r, rows)
    np.testing.assert_array_equal(c, cols)
    # the fused launch of the same module
    from opty_amd import hip_backend as hb
    con2, jac2 = np.empty_like(con), np.empty_like(jac)
    col.hip.eval_con_jac(free, con2, jac2, hb.HOST)
    _check('hip fused', seed, orc, con2, jac2, c_ref, j_ref)


def test_zz_report():
    """Stats line (like the parity summary of the golden tests)."""
    for tag, worst in sorted(_WORST.items()):
        print('fuzz parity [%s]: %d problems, worst error %.2e of the '
              'tolerance reference' % (tag, len(worst), max(worst)))
