"""``jacobian_layout='varying_first'`` (opt-in, SURVEY.md 8(f) rank 3 family):
the reference's triplets in another ORDER -- ``[entries that can change, all
nodes | entries that repeat one of those | node-invariant entries | instance
partials]`` -- so that the host-visible Jacobian is one PCIe stream into the
head of the persistent array (``opty_hip_set_segments``).

Bar: the triplet SET equals the reference's -- int64 (row, col) pairs
bit-exact after sorting, values within 1e-10 -- on reference goldens; the
default layout is untouched.
"""
import numpy as np
import pytest

import gallery_cases as gc
import golden_util as gu
from examples import problems

RTOL = 1e-10
CASES = ['config3_10link_small', 'config2_pendulum_small',
         'pend3_link_midpoint_small', 'pend2_link_vardur_unkmass_small',
         'gaitlike_3link_be_small', 'implicit_traj_mid_small',
         'chaplygin_mid_small', 'one_legged_small', 'msd_be_small']
GALLERY = ['gallery_wheel_on_bumpy_road', 'gallery_friction_slack',
           'gallery_drone', 'gallery_betts_10_50']


def _load(name):
    if name.startswith('gallery_'):
        meta, z, kw = gc.load(name)
        return meta, z, kw
    meta, z = gu.load(name)
    return meta, z, problems.build(name)


@pytest.mark.parametrize('name', CASES + GALLERY)
def test_segments_partition_the_block(name):
    """CPU: ``jacobian_segments`` is a permutation of the block; in the
    REFERENCE's own values every entry of segment 2 has one value at all
    nodes and every entry of segment 1 equals its source to rounding."""
    from opty_amd import ConstraintCollocator
    meta, z, kw = _load(name)
    col = ConstraintCollocator(jacobian_layout='varying_first', **kw)
    order, seg_len, source = col.jacobian_segments()
    P = meta['M']*meta['C']
    assert sorted(order) == list(range(P)) and seg_len.sum() == P
    L0, L1, L2 = (int(v) for v in seg_len)
    assert len(source) == L1 and (L1 == 0 or source.max() < L0)
    blk = z['jac'][:P*(meta['N'] - 1)].reshape(meta['N'] - 1, P)
    inv = blk[:, order[L0 + L1:]]
    assert np.all(inv == inv[0]), 'a node-invariant entry varies'
    rep, src = blk[:, order[L0:L0 + L1]], blk[:, order[:L0][source]]
    np.testing.assert_allclose(rep, src, rtol=1e-9,
                               atol=1e-12*max(1.0, np.abs(blk).max()))
    with pytest.raises(ValueError, match='prune_zeros'):
        ConstraintCollocator(jacobian_layout='varying_first',
                             prune_zeros=True, **kw)


def _sorted_triplets(rows, cols, vals):
    order = np.lexsort((cols, rows))
    return rows[order], cols[order], vals[order], order


@pytest.mark.gpu
@pytest.mark.parametrize('windows', [None, '1', '3'])
@pytest.mark.parametrize('name', CASES + GALLERY)
def test_triplet_set_equals_the_reference(name, windows, monkeypatch):
    """``windows``: node windows of the host pipeline (upload / evaluate /
    pack of one window while the previous one crosses PCIe; default: by
    size), forced here so that small problems exercise ragged windows,
    instance tails and a free ``h`` behind them."""
    if windows:
        monkeypatch.setenv('OPTY_HIP_HOST_WINDOWS', windows)
    import opty_amd
    from opty_amd import hip_backend as hb
    meta, z, kw = _load(name)
    col = opty_amd.ConstraintCollocator(jacobian_layout='varying_first',
                                        **kw)
    ref = opty_amd.ConstraintCollocator(**kw)
    jac = col.generate_jacobian_function()
    rows, cols = col.jacobian_indices()
    assert rows.dtype == np.int64 and cols.dtype == np.int64
    gr, gc_, gv, _ = _sorted_triplets(z['rows'], z['cols'], z['jac'])
    _, jb = gu.error_bounds(ref, z['free'])
    _, _, jb, _ = _sorted_triplets(z['rows'], z['cols'], jb)
    # the (row, col) pairs are unique, so sorting defines the pairing
    assert len(set(zip(gr.tolist(), gc_.tolist()))) == len(gr)
    order, seg_len, source = col.jacobian_segments()
    N1, P = meta['N'] - 1, meta['M']*meta['C']
    for call in range(3):           # first call: everything; then seg 0 + 1
        vals = jac(z['free']).copy()
        r, c, v, perm = _sorted_triplets(rows, cols, vals)
        np.testing.assert_array_equal(r, gr)
        np.testing.assert_array_equal(c, gc_)
        gu.assert_close(v, gv, RTOL, what=name + ' varying_first jac',
                        bound=jb)
        if call == 0:
            other = z['free'] + 0.03125      # another point in between
            other[-1] = z['free'][-1]
            jac(other)
    # layout: segment g of node i at S_g*(N-1) + i*L_g
    L0, L1, L2 = (int(x) for x in seg_len)
    ref_vals = ref.generate_jacobian_function()(z['free'])
    blk = ref_vals[:P*N1].reshape(N1, P)
    seg0 = vals[:L0*N1].reshape(N1, L0)
    seg1 = vals[L0*N1:(L0 + L1)*N1].reshape(N1, L1)
    seg2 = vals[(L0 + L1)*N1:P*N1].reshape(N1, L2)
    # (to rounding: which wave evaluates the entries next to a strip boundary
    # depends on the alignment of the destination)
    close = dict(rtol=1e-12, atol=1e-13*max(1.0, np.abs(blk).max()))
    np.testing.assert_allclose(seg0, blk[:, order[:L0]], **close)
    np.testing.assert_array_equal(seg1, seg0[:, source])   # bit-equal copies
    np.testing.assert_allclose(seg2, blk[:, order[L0 + L1:]], **close)
    np.testing.assert_allclose(vals[P*N1:], ref_vals[P*N1:], **close)
    # device-pointer evaluation returns the same layout
    import torch
    dfree = torch.from_numpy(z['free']).cuda()
    dcon = torch.empty(col.num_constraints, dtype=torch.float64,
                       device='cuda')
    djac = torch.empty(col.hip.nnz, dtype=torch.float64, device='cuda')
    col.hip.eval_con_jac(dfree, dcon, djac, hb.DEVICE)
    col.hip.synchronize()
    dv = djac.cpu().numpy()
    np.testing.assert_allclose(dv[:L0*N1], vals[:L0*N1], **close)
    np.testing.assert_allclose(dv, vals, rtol=1e-9, atol=1e-11*max(
        1.0, np.abs(vals).max()))
    # (the fused kernel's constraint rows against the separate kernel's)
    np.testing.assert_allclose(
        dcon.cpu().numpy(), ref.generate_constraint_function()(z['free']),
        rtol=1e-10, atol=1e-10)


@pytest.mark.gpu
def test_large_problem_streams_into_the_head_of_the_vector():
    """Config 3's block at N = 20 001 (chunked DMA + host copies): equal to
    the default layout's persistent path entry by entry, after a change of a
    known parameter too, with one and the default number of host threads."""
    import opty_amd
    from opty_amd import hip_backend as hb
    factory, fkw = problems.CONFIGS['config3_10link']
    kw = factory(**dict(fkw, num_nodes=20001))
    col = opty_amd.ConstraintCollocator(jacobian_layout='varying_first',
                                        **kw)
    ref = opty_amd.ConstraintCollocator(**kw)
    order, seg_len, source = col.jacobian_segments()
    assert tuple(seg_len) == (275, 55, 660)
    N1, P = 20000, 990
    jac, rjac = (col.generate_jacobian_function(),
                 ref.generate_jacobian_function())
    rows, cols = col.jacobian_indices()
    rr, rc = ref.jacobian_indices()

    def check(free):
        v = jac(free)
        blk = rjac(free)[:P*N1].reshape(N1, P)
        got = np.empty_like(blk)
        at = 0
        for g, L in enumerate(int(x) for x in seg_len):
            seg = v[at*N1:(at + L)*N1].reshape(N1, L)
            got[:, order[at:at + L]] = seg
            at += L
        np.testing.assert_allclose(got, blk, rtol=1e-12, atol=1e-10)
    # indices: the same permutation
    at = 0
    for L in (int(x) for x in seg_len):
        sel = order[at:at + L]
        np.testing.assert_array_equal(
            rows[at*N1:(at + L)*N1].reshape(N1, L),
            rr[:P*N1].reshape(N1, P)[:, sel])
        np.testing.assert_array_equal(
            cols[at*N1:(at + L)*N1].reshape(N1, L),
            rc[:P*N1].reshape(N1, P)[:, sel])
        at += L
    frees = [problems.make_free(col.num_free, seed=s) for s in (1, 2, 3)]
    try:
        check(frees[0])
        check(frees[1])
        hb.set_host_threads(1)
        check(frees[2])
        key = list(kw['known_parameter_map'])[-1]
        for c in (col, ref):
            c.known_parameter_map[key] = 1.375
        check(frees[0])
        check(frees[1])
    finally:
        hb.set_host_threads(0)
