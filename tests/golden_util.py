"""Helpers shared by the parity tests: load the committed golden vectors
(produced from the real reference by ``tests/golden/_gen/make_golden.py``)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

with open(os.path.join(GOLDEN, 'MANIFEST.json')) as _f:
    MANIFEST = json.load(_f)

FULL = sorted(k for k, v in MANIFEST.items() if v['kind'] == 'full')
#: fixtures whose ORACLE takes minutes to build (SymPy differentiating a
#: 24-link pendulum); the oracle was checked against them once in the build
#: container (5e-16), the product is checked against them in every run.
HEAVY = {'config5_standin_24link_small', 'config5_gaitlike_24link_small'}
FULL_FAST = [k for k in FULL if k not in HEAVY]
SAMPLED = sorted(k for k, v in MANIFEST.items() if v['kind'] == 'sampled')


def load(name):
    return MANIFEST[name], np.load(os.path.join(GOLDEN, name + '.npz'))


#: floor of the parity tolerance, in units of the per-entry rounding-error
#: bound (``dag_interp.evaluate_with_error_bound``).  The reference's own
#: values sit within 1 unit of ours on every fixture but one: it prints
#: ``theta(T) - pi`` with 15 digits (7 units).
BOUND_UNITS = 32.0
UNIT_ROUNDOFF = 2.0**-53

#: worst errors seen by ``assert_close`` in this session, per label
#: (``tests/conftest.py`` prints them at the end of the run)
STATS = {}


def error_bounds(col, free, nodes=None):
    """Per-entry rounding-error bounds of ``col``'s outputs at ``free`` (test
    infrastructure: the product's expression DAG run through the NumPy
    interpreter with running error analysis)."""
    import dag_interp
    return dag_interp.error_bounds(col, free, nodes)


def caps_for(jac_ref, num_con, N1, M, C):
    """``(con_cap, jac_cap)`` for whole output vectors: :func:`row_caps` of
    the reference Jacobian's collocation part, no cap (inf) on the instance
    tails (their bound is their own size)."""
    jac_ref = np.asarray(jac_ref, dtype=float)
    ccap, jcap = row_caps(jac_ref[:N1*M*C].reshape(N1, M, C))
    return (np.concatenate((ccap.ravel(), np.full(num_con - N1*M, np.inf))),
            np.concatenate((jcap.ravel(),
                            np.full(len(jac_ref) - N1*M*C, np.inf))))


def note_floor_only(what, err, desired, tol, rtol):
    """STATS accounting for checks that do not go through
    :func:`assert_close` (the fuzz tests: floor = rtol x row maximum of the
    ORACLE's Jacobian, nothing from the product)."""
    with np.errstate(all='ignore'):
        rel = np.where(desired != 0, err/np.abs(desired), 0.0)
    st = STATS.setdefault(what, dict(worst_rel=0.0, worst_bound_units=0.0,
                                     entries=0, entries_passed_by_floor=0,
                                     worst_rel_passed_by_floor=0.0,
                                     floor_capped=True))
    if not desired.size:
        return
    st['worst_rel'] = max(st['worst_rel'], float(np.nanmax(rel)))
    st['entries'] += int(desired.size)
    by_floor = (err > rtol*np.abs(desired)) & (err <= tol)
    if by_floor.any():
        st['entries_passed_by_floor'] += int(by_floor.sum())
        st['worst_rel_passed_by_floor'] = max(
            st['worst_rel_passed_by_floor'],
            float(np.nanmax(np.where(by_floor, rel, 0.0))))


def row_caps(jac_blocks):
    """Per-entry cap on the tolerance floor from the reference values
    themselves: ``max |entry|`` over the entry's equation row of its node's
    block -- the magnitude of the 1/h terms of that equation at that node.
    ``jac_blocks``: ``(nodes, M, C)``.  Returns ``(con_cap (M, nodes),
    jac_cap (nodes, M, C))``."""
    rowmax = np.abs(jac_blocks).max(axis=2)                  # (nodes, M)
    return rowmax.T.copy(), np.repeat(rowmax[:, :, None],
                                      jac_blocks.shape[2], axis=2)


def assert_close(actual, desired, rtol=1e-10, scale=None, what='',
                 bound=None, cap=None):
    """Per-entry parity check, ``|a - d| <= max(rtol*|d|, floor)``.

    The bar is 1e-10 *relative* (BASELINE.json north_star).  An entry that is
    a sum of cancelling terms cannot be held to 1e-10 of its *result*
    (SURVEY.md section 7), so every entry gets a floor:

    * ``bound`` (preferred; array like ``desired``): the entry's own
      rounding-error bound in units of round-off -- floor =
      ``BOUND_UNITS * 2**-53 * bound``, i.e. a constant, a single product or
      a small entry of a block with large neighbours is held to 1e-10
      relative, and only genuine cancellation widens the tolerance, by what
      the entry's own terms justify;
    * ``scale`` (checksums over all nodes): floor = ``rtol*scale``;
    * neither: ``rtol * max|desired|`` (array-wide, coarse).

    ``cap`` (array like ``desired``, see :func:`row_caps`): the floor is never
    allowed above ``rtol*cap`` -- whatever the error analysis says, an entry
    is held at least to 1e-10 of the largest entry of its own equation row at
    its own node.
    """
    actual = np.asarray(actual, dtype=float)
    desired = np.asarray(desired, dtype=float)
    assert actual.shape == desired.shape, (what, actual.shape, desired.shape)
    if bound is not None:
        bound = np.asarray(bound, dtype=float)
        assert bound.shape == desired.shape, (what, bound.shape)
        floor = BOUND_UNITS*UNIT_ROUNDOFF*bound
    else:
        if scale is None:
            scale = float(np.max(np.abs(desired))) if desired.size else 1.0
        floor = rtol*scale
    if cap is not None:
        cap = np.asarray(cap, dtype=float)
        assert cap.shape == desired.shape, (what, cap.shape)
        floor = np.minimum(floor, rtol*cap)
    tol = np.maximum(rtol*np.abs(desired), floor)
    err = np.abs(actual - desired)
    if desired.size:
        with np.errstate(all='ignore'):
            rel = np.where(desired != 0, err/np.abs(desired), 0.0)
            units = (np.where(bound > 0, err/(UNIT_ROUNDOFF*bound), 0.0)
                     if bound is not None else np.zeros(1))
        st = STATS.setdefault(what or '?', dict(worst_rel=0.0,
                                                worst_bound_units=0.0,
                                                entries=0,
                                                entries_passed_by_floor=0,
                                                worst_rel_passed_by_floor=0.0,
                                                floor_capped=cap is not None))
        st['worst_rel'] = max(st['worst_rel'], float(np.nanmax(rel)))
        st['worst_bound_units'] = max(st['worst_bound_units'],
                                      float(np.nanmax(units)))
        st['entries'] += int(desired.size)
        # entries that do NOT meet rtol of their own value and pass only
        # because of the floor (cancellation inside the entry): how many,
        # and how far off relative to themselves
        by_floor = (err > rtol*np.abs(desired)) & (err <= tol)
        if by_floor.any():
            st['entries_passed_by_floor'] += int(by_floor.sum())
            st['worst_rel_passed_by_floor'] = max(
                st['worst_rel_passed_by_floor'],
                float(np.nanmax(np.where(by_floor, rel, 0.0))))
        st['floor_capped'] = bool(st['floor_capped'] and cap is not None)
    bad = ~(err <= tol)          # NaNs are bad
    if bad.any():
        with np.errstate(all='ignore'):
            ratio = np.where(bad, np.where(tol > 0, err/tol, np.inf), -1.0)
        k = int(np.argmax(np.where(np.isnan(ratio), np.inf, ratio)))
        raise AssertionError(
            '%s: %d/%d entries off; worst at %d: %r vs %r (err %.3g, tol '
            '%.3g)' % (what, bad.sum(), bad.size, k, actual.flat[k],
                       desired.flat[k], err.flat[k], tol.flat[k]))
