"""GPU tests of the row-sorted Jacobian layout (``jacobian_layout='csr'``,
SURVEY.md 8(f) rank 3): the same triplets as the reference's COO order --
checked against the golden vectors of the real reference -- stored sorted by
constraint row, then column, with a ``row_ptr`` / ``col_idx`` structure that
``scipy.sparse.csr_matrix`` accepts as is."""
import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as gu
from examples import problems

pytestmark = pytest.mark.gpu


def _pair(name, prune=False, **over):
    import opty_amd
    factory, fkw = problems.CONFIGS[name]
    mk = lambda **kw: opty_amd.ConstraintCollocator(
        prune_zeros=prune, **kw, **factory(**dict(fkw, **over)))
    return mk(), mk(jacobian_layout='csr')


def _check_structure(csr, rs, cs, nnz):
    row_ptr, col_idx = csr.jacobian_csr_structure()
    assert row_ptr.dtype == np.int64 and col_idx.dtype == np.int64
    assert row_ptr[0] == 0 and row_ptr[-1] == nnz
    assert len(row_ptr) == csr.num_constraints + 1
    np.testing.assert_array_equal(col_idx, cs)
    np.testing.assert_array_equal(
        np.repeat(np.arange(csr.num_constraints), np.diff(row_ptr)), rs)
    # sorted: rows never decrease, columns ascend within a row
    assert np.all(np.diff(rs) >= 0)
    same = np.diff(rs) == 0
    assert np.all(np.diff(cs)[same] >= 0)
    return row_ptr, col_idx


@pytest.mark.parametrize('name,prune', [(n, False) for n in gu.FULL] + [
    ('config3_10link_small', True), ('chaplygin_mid_small', True),
    ('pend2_link_vardur_unkmass_small', True)])
def test_csr_equals_reference_triplets(name, prune):
    meta, z = gu.load(name)
    _, csr = _pair(name, prune)
    jac = csr.generate_jacobian_function()(z['free']).copy()
    rs, cs = csr.jacobian_indices()
    row_ptr, col_idx = _check_structure(csr, rs, cs, len(jac))
    shape = (meta['num_constraints'], meta['num_free'])
    ref = sp.coo_matrix((z['jac'], (z['rows'], z['cols'])), shape=shape)
    got = sp.csr_matrix((jac, col_idx, row_ptr), shape=shape)
    if not prune:
        # the very same triplets, re-ordered (stable within equal (row, col))
        o_ref = np.lexsort((z['cols'], z['rows']))
        np.testing.assert_array_equal(z['rows'][o_ref], rs)
        np.testing.assert_array_equal(z['cols'][o_ref], cs)
        gu.assert_close(jac, z['jac'][o_ref], 1e-10, what=name + ' csr')
    diff = abs(got - ref.tocsr())
    scale = abs(z['jac']).max()
    assert (diff.max() if diff.nnz else 0.0) <= 1e-10*scale


@pytest.mark.parametrize('N', [2, 65, 1000, 4099])
def test_csr_matches_coo_layout_ragged(N):
    """10-link pendulum at node counts that are not multiples of the 64-node
    block: CSR values are the COO values permuted; fused launch too."""
    import torch
    from opty_amd import hip_backend as hb
    coo, csr = _pair('config3_10link', num_nodes=N)
    free = problems.make_free(coo.num_free, seed=N)
    jc = coo.generate_jacobian_function()(free).copy()
    rc, cc = coo.jacobian_indices()
    js = csr.generate_jacobian_function()(free).copy()
    rs, cs = csr.jacobian_indices()
    _check_structure(csr, rs, cs, len(js))
    o = np.lexsort((cc, rc))
    np.testing.assert_array_equal(rc[o], rs)
    np.testing.assert_array_equal(cc[o], cs)
    tol = 1e-13*np.abs(jc).max()
    assert np.abs(js - jc[o]).max() <= tol
    # fused launch on device pointers
    dev = torch.device('cuda', 0)
    hip = csr.hip
    f = torch.from_numpy(free).to(dev)
    con = torch.empty(csr.num_constraints, dtype=torch.float64, device=dev)
    jac = torch.full((hip.nnz,), float('nan'), dtype=torch.float64,
                     device=dev)
    hip.use_torch_stream()
    hip.eval_con_jac(f, con, jac, hb.DEVICE)
    torch.cuda.synchronize()
    assert np.abs(jac.cpu().numpy() - js).max() <= tol
    assert np.abs(con.cpu().numpy() -
                  coo.generate_constraint_function()(free)).max() <= tol


def test_csr_wide_rows():
    """Rows wider than the staging tile (24-link pendulum, 102 columns) take
    the chunked path."""
    coo, csr = _pair('config5_standin_24link_small')
    free = problems.make_free(coo.num_free, seed=5, variable_duration=True)
    jc = coo.generate_jacobian_function()(free).copy()
    rc, cc = coo.jacobian_indices()
    js = csr.generate_jacobian_function()(free).copy()
    rs, cs = csr.jacobian_indices()
    o = np.lexsort((cc, rc))
    np.testing.assert_array_equal(rc[o], rs)
    np.testing.assert_array_equal(cc[o], cs)
    assert np.abs(js - jc[o]).max() <= 1e-13*np.abs(jc).max()


def test_csr_structure_needs_csr_layout():
    coo, _ = _pair('config1_vyasarayani')
    with pytest.raises(ValueError):
        coo.jacobian_csr_structure()


def test_problem_facade_with_csr_layout():
    """``Problem(..., jacobian_layout='csr', prune_zeros=True)``: the IPOPT
    callbacks ``jacobianstructure`` / ``jacobian`` describe the same matrix as
    the default layout."""
    import opty_amd
    kw = problems.pendulum_swing_up(num_nodes=130)
    obj, grad = (lambda f: 0.0), (lambda f: f)
    ref = opty_amd.Problem(obj, grad, **kw)
    alt = opty_amd.Problem(obj, grad, jacobian_layout='csr',
                           prune_zeros=True,
                           **problems.pendulum_swing_up(num_nodes=130))
    free = problems.make_free(ref.num_free, seed=9)
    shape = (ref.num_constraints, ref.num_free)
    A = sp.coo_matrix((ref.jacobian(free), ref.jacobianstructure()),
                      shape=shape).tocsr()
    B = sp.coo_matrix((alt.jacobian(free), alt.jacobianstructure()),
                      shape=shape).tocsr()
    assert len(alt.jacobian(free)) < len(ref.jacobian(free))
    diff = abs(A - B)
    assert (diff.max() if diff.nnz else 0.0) <= 1e-12*abs(A).max()
    np.testing.assert_array_equal(alt.constraints(free),
                                  ref.constraints(free))


def test_end_to_end_swing_up_with_csr_jacobian():
    """``examples/swing_up_scipy.py``: device objective + gradient, device
    constraints and a CSR Jacobian drive SciPy's SLSQP to a feasible
    minimum-effort swing-up."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(__file__), '..', 'examples',
                        'swing_up_scipy.py')
    spec = importlib.util.spec_from_file_location('swing_up_scipy', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    free, res, violation = mod.main(verbose=False)
    N = 41
    assert res.success and violation < 1e-8
    assert abs(free[N - 1] - np.pi) < 1e-8 and abs(free[0]) < 1e-8
    assert abs(res.fun - 59.65) < 0.5
