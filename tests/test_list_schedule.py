"""Persistent kernels with a list schedule (dispatch order ``'list'``,
``opty_hip_desc.jac_persist / fused_persist``): the schedule the library
builds (host arithmetic, CPU), what the printer prints, and -- on the GPU --
that such kernels return what the one-workgroup-per-item kernels return."""
import collections
import ctypes

import numpy as np
import pytest

import golden_util as gu
from examples import problems


def _loads(sched, cost):
    return [sum(cost[c] for c, _ in w) for w in sched]


@pytest.mark.parametrize('persist,nblk,cost', [
    (1024, 782, [20.5, 11.6, 4.7]),                 # the muscle-driven leg
    (1024, 782, [15.7, 16.1, 15.0, 13.5, 8.4, 7.2]),
    (1024, 1563, [1.0]*10),
    (1024, 98, [20.0, 12.0, 5.0]),                  # fewer items than SIMDs
    (1024, 1, [3.0, 1.0]), (1024, 7, [1.0]), (1024, 9, [2.0, 1.0, 1.0]),
    (64, 100, [5.0, 1.0]), (8, 3, [0.0, 0.0])])
def test_list_schedule_holds_every_item_once(persist, nblk, cost):
    from opty_amd import hip_backend as hb
    sched = hb.list_schedule(persist, nblk, cost)
    nslot = (nblk + 7)//8
    assert len(sched) == min(persist, nslot*8*len(cost))
    seen = collections.Counter(item for w in sched for item in w)
    assert set(seen.values()) == {1}
    assert set(seen) == {(g, b) for g in range(len(cost))
                         for b in range(nblk)}
    for w, items in enumerate(sched):
        # a node block's strips stay on one XCD (workgroup w runs on w % 8)
        assert all(b % 8 == w % 8 for _, b in items)
    if max(cost) > 0:
        # longest-processing-time-first: no workgroup is further above the
        # mean of its XCD than one item
        loads = np.array(_loads(sched, cost))
        for x in range(8):
            mine = loads[x::8]
            assert mine.max() <= mine.mean() + max(cost) + 1e-9


def test_list_schedule_spreads_the_short_strips():
    """Every workgroup starts somewhere else in its list: the store-heavy
    short strips do not all run at the end of the launch."""
    from opty_amd import hip_backend as hb
    sched = hb.list_schedule(1024, 782, [20.5, 11.6, 4.7])
    first = collections.Counter(w[0][0] for w in sched if len(w) > 1)
    assert len(first) == 3 and min(first.values()) > 100
    # the biped's shape: six classes
    sched = hb.list_schedule(1024, 782, [15.7, 16.1, 15.0, 13.5, 8.4, 7.2])
    first = collections.Counter(w[0][0] for w in sched)
    assert len(first) == 6


def test_list_schedule_rejects_bad_requests():
    from opty_amd import hip_backend as hb
    lib = hb.load_library()
    count = ctypes.c_int64()
    cost = (ctypes.c_float*2)(1.0, 1.0)
    for persist, nblk, classes in ((0, 4, 2), (12, 4, 2), (1024, -1, 2),
                                   (1024, 4, 0), (1024, 4, 33)):
        assert lib.opty_hip_list_schedule(persist, nblk, classes, cost, None,
                                          0, ctypes.byref(count)) != 0
        assert b'list-schedule' in lib.opty_hip_last_error()
    table = (ctypes.c_int32*4)()
    assert lib.opty_hip_list_schedule(1024, 4, 2, cost, table, 4,
                                      ctypes.byref(count)) != 0


def _options(**kw):
    from opty_amd.codegen.emit_hip import EmitOptions
    return EmitOptions(groups=4, fused_groups=4, **kw)


def test_list_order_prints_persistent_kernels():
    import opty_amd
    from opty_amd import hip_backend as hb
    kw = problems.n_link_cart_pendulum(num_links=3, num_nodes=300)
    plain = opty_amd.ConstraintCollocator(emit_options=_options(), **kw)
    fused = opty_amd.ConstraintCollocator(
        emit_options=_options(fused_order='list'), **kw)
    both = opty_amd.ConstraintCollocator(
        emit_options=_options(order='list'), **kw)
    src0, meta0 = plain.generate_source()
    src1, meta1 = fused.generate_source()
    src2, meta2 = both.generate_source()
    assert 'sched' not in src0 and 'opty_opaque' not in src0
    assert meta0['kernels']['conjac']['persist'] == 0
    # only the fused kernel changes
    assert meta1['kernels']['jac']['sha'] == meta0['kernels']['jac']['sha']
    assert meta1['kernels']['con']['sha'] == meta0['kernels']['con']['sha']
    k = meta1['kernels']['conjac']
    assert k['persist'] == 1024 and k['waves_per_wg'] == 1
    assert len(k['class_cost']) == k['wgs_per_block'] and \
        min(k['class_cost']) > 0
    assert src1.count('const int *__restrict__ sched') == 1
    assert src2.count('const int *__restrict__ sched') == 2
    # the constraint kernel has nothing to schedule
    assert meta2['kernels']['con']['persist'] == 0
    d = fused._descriptor(meta1)
    assert d['fused_persist'] == 1024 and d['jac_persist'] == 0
    assert len(d['fused_class_cost']) == k['wgs_per_block']
    desc = hb._Desc(**d)
    assert list(desc.fused_class_cost)[:k['wgs_per_block']] == \
        [pytest.approx(c) for c in k['class_cost']]
    # measured durations given with the options replace the estimate
    given = opty_amd.ConstraintCollocator(emit_options=_options(
        fused_order='list', fused_class_cost='4;3;2;1;1'), **kw)
    if k['wgs_per_block'] == 5:
        assert given.generate_source()[1]['kernels']['conjac'][
            'class_cost'] == [4.0, 3.0, 2.0, 1.0, 1.0]
    # builds (without MachineLICM: hb.LOOP_FLAGS) and does not spill
    hsaco, _ = both._build_code_object()
    assert hb.vgpr_spills(hsaco) == {}


def test_create_rejects_a_bad_persistent_geometry(tmp_path):
    from opty_amd import hip_backend as hb
    lib = hb.load_library()
    bogus = tmp_path/'x.hsaco'
    bogus.write_bytes(b'')
    base = dict(N=10, n=1, M=1, C=2, P=2, jac_wgs_per_block=1,
                jac_waves_per_wg=1, fused_wgs_per_block=1,
                con_wgs_per_block=1, fused_waves_per_wg=1,
                con_waves_per_wg=1)
    for bad in (dict(jac_persist=12), dict(fused_persist=-8),
                dict(jac_persist=1024, jac_waves_per_wg=2),
                dict(fused_persist=1024, fused_wgs_per_block=40)):
        desc = hb._Desc(**dict(base, **bad))
        out = ctypes.c_void_p()
        rc = lib.opty_hip_create(ctypes.byref(desc), str(bogus).encode(),
                                 ctypes.byref(out))
        assert rc != 0
        msg = lib.opty_hip_last_error()
        assert b'persistent' in msg or b'list schedule' in msg, msg


CASES = [
    # (problem, nodes, problem overrides): blocks x strips above and below
    # the 1024 persistent workgroups, ragged last block, one block
    ('config3_10link', 20001, {}),
    ('config3_10link', 130, {}),
    ('config3_10link', 40, {}),
    ('pend2_link_vardur_unkmass_small', 30001, {}),   # table from `free`
    ('config2_pendulum', 70001, {}),                  # folded instance tails
    ('chaplygin_be_small', 20000, {}),
]


@pytest.mark.gpu
@pytest.mark.parametrize('name,N,over', CASES)
def test_list_order_matches_the_default_dispatch(name, N, over):
    """The same DAG through persistent kernels: whole problem (separate and
    fused), node shards of changing sizes (a schedule per launch size, more
    sizes than the handle keeps), against the default geometry of the same
    options."""
    import torch
    import opty_amd
    from opty_amd import hip_backend as hb
    factory, fkw = problems.CONFIGS[name]
    kw = factory(**dict(fkw, num_nodes=N, **over))
    ref = opty_amd.ConstraintCollocator(emit_options=_options(), **kw)
    col = opty_amd.ConstraintCollocator(
        emit_options=_options(order='list', fused_order='list'), **kw)
    meta = col.generate_source()[1]
    if meta['kernels']['jac']['groups'] > 0 and ref.hip.desc['P'] >= 32:
        assert col.hip.desc['jac_persist'] == 1024
        assert col.hip.desc['fused_persist'] == 1024
    for seed in (1, 2):
        free = problems.make_free(col.num_free, seed=seed,
                                  variable_duration=col._variable_duration)
        cb, jb = gu.error_bounds(ref, free)
        c0 = ref.generate_constraint_function()(free)
        j0 = np.array(ref.generate_jacobian_function()(free))
        gu.assert_close(col.generate_constraint_function()(free), c0, 1e-12,
                        what='list con', bound=cb)
        gu.assert_close(col.generate_jacobian_function()(free), j0, 1e-12,
                        what='list jac', bound=jb)
        c2, j2 = np.empty_like(c0), np.empty_like(j0)
        col.hip.eval_con_jac(free, c2, j2, hb.HOST)
        gu.assert_close(c2, c0, 1e-12, what='list fused con', bound=cb)
        gu.assert_close(j2, j0, 1e-12, what='list fused jac', bound=jb)
    # node shards
    dev = torch.device('cuda:0')
    hip = col.hip
    hip.use_torch_stream()
    ncn, P, M = N - 1, hip.desc['P'], hip.desc['M']
    f = torch.from_numpy(free).to(dev)
    rng = np.random.default_rng(3)
    cuts = sorted(set(int(v) for v in rng.integers(0, ncn, 24)) | {0, ncn})
    blk0 = j0[:ncn*P].reshape(ncn, P)
    con0 = c0[:M*ncn].reshape(M, ncn)
    jbn = np.broadcast_to(jb, j0.shape)[:ncn*P].reshape(ncn, P) \
        if np.ndim(jb) else jb
    for a, b in zip(cuts[:-1], cuts[1:]):
        for what in (hb.EVAL_FUSED_KERNEL, hb.EVAL_PAIR):
            con = torch.full((M, b - a), np.nan, dtype=torch.float64,
                             device=dev)
            jac = torch.full(((b - a)*P,), np.nan, dtype=torch.float64,
                             device=dev)
            hip.eval_shard(what, f, con, b - a, jac, a, b)
            torch.cuda.synchronize()
            gu.assert_close(jac.cpu().numpy().reshape(b - a, P), blk0[a:b],
                            1e-12, what='shard jac [%d, %d)' % (a, b),
                            bound=jbn[a:b] if np.ndim(jbn) else jbn)
            np.testing.assert_allclose(
                con.cpu().numpy(), con0[:, a:b], rtol=1e-11,
                atol=1e-11*max(1.0, np.abs(con0).max()))
