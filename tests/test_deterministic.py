"""``ConstraintCollocator(deterministic=True)``: values that do not depend on
the launch a node is evaluated in -- node window, node shard (its own code
object: the kernels' geometry follows the launch size), strip count, fused or
separate kernels -- bit for bit, as the reference's are (one scalar function
per node whatever the OpenMP team, ``opty/utils.py:483-494, 524-526``).  The
default build only promises that to rounding."""
import numpy as np
import pytest

import opty_amd
from examples import problems

NAMES = ['config3_10link_small', 'one_legged_small',
         'gaitlike_3link_be_small', 'chaplygin_mid_small']


def test_deterministic_builds_are_their_own_modules():
    """CPU: the option reaches the printer (one ``sincos`` form) and the
    compiler (no FMA contraction); the default build is untouched."""
    from opty_amd import hip_backend as hb
    kw = problems.build('config3_10link_small')
    det = opty_amd.ConstraintCollocator(deterministic=True, **kw)
    std = opty_amd.ConstraintCollocator(**kw)
    assert det._printer_options().deterministic == 1
    assert std._printer_options().deterministic == 0
    src, _ = det.generate_source()
    assert '= sin(' not in src and '= cos(' not in src and 'sincos(' in src
    assert hb.DETERMINISTIC_FLAGS == ('-ffp-contract=off',)
    h_det, _ = det.prebuild()
    h_std, _ = std.prebuild()
    assert h_det != h_std
    assert not hb.vgpr_spills(h_det)


def prebuild():
    """Code objects of the GPU test below (``__graft_entry__.build``)."""
    from opty_amd.sharded import partition_nodes
    nodes = 2051
    for name in NAMES:
        _scaled(name, nodes).prebuild()
        for world in (2, 3, 8):
            ranges = partition_nodes(nodes - 1, world)
            _scaled(name, nodes,
                    launch_nodes=max(b - a for a, b in ranges)).prebuild()


def _scaled(name, nodes, **extra):
    factory, fkw = problems.CONFIGS[name]
    return opty_amd.ConstraintCollocator(
        deterministic=True, **extra, **factory(**dict(fkw, num_nodes=nodes)))


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_any_launch_returns_the_same_bits(name):
    import torch
    from opty_amd import hip_backend as hb
    from opty_amd.sharded import partition_nodes
    dev = torch.device('cuda:0')
    nodes = 2051
    col = _scaled(name, nodes)
    hip = col.hip
    free_h = problems.make_free(col.num_free, seed=11,
                                variable_duration=col._variable_duration)
    col._sync_known(hip, free_h)
    free = torch.from_numpy(free_h).to(dev)
    ncn, M, P = nodes - 1, col.num_eom, hip.desc['P']
    f64 = dict(dtype=torch.float64, device=dev)
    con = torch.empty(col.num_constraints, **f64)
    jac = torch.empty(hip.nnz, **f64)
    hip.eval_con_jac(free, con, jac, hb.DEVICE)
    torch.cuda.synchronize()
    ref_c = con[:M*ncn].view(M, ncn).clone()
    ref_j = jac[:P*ncn].view(ncn, P).clone()
    # fused == separate
    hip.eval_con(free, con, hb.DEVICE)
    hip.eval_jac(free, jac, hb.DEVICE)
    torch.cuda.synchronize()
    assert torch.equal(con[:M*ncn].view(M, ncn), ref_c)
    assert torch.equal(jac[:P*ncn].view(ncn, P), ref_j)
    # a deterministic build is still the reference's function: the default
    # build of the same problem agrees to rounding
    std = opty_amd.ConstraintCollocator(**dict(
        problems.CONFIGS[name][0](**dict(problems.CONFIGS[name][1],
                                         num_nodes=nodes))))
    std._sync_known(std.hip, free_h)
    c2, j2 = torch.empty_like(con), torch.empty_like(jac)
    std.hip.eval_con_jac(free, c2, j2, hb.DEVICE)
    torch.cuda.synchronize()
    # (1e-10 of the largest value: the muscle model's 1/h^2-sized entries
    # round at 1e-5 absolute, and contraction moves an entry by a few such)
    for got, want in ((con, c2), (jac, j2)):
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 1e-10*scale
    std.hip.close()
    # node windows at every alignment, through every selector
    rng = np.random.default_rng(5)
    for k in range(24):
        a = int(rng.integers(0, ncn - 1))
        b = int(min(ncn, a + rng.choice([1, 2, 63, 64, 65, 200, 777])))
        what = [hb.EVAL_FUSED, hb.EVAL_PAIR, hb.EVAL_CON, hb.EVAL_JAC][k % 4]
        cb = torch.full((M, b - a), float('nan'), **f64)
        jb = torch.full(((b - a)*P,), float('nan'), **f64)
        hip.eval_shard(what, free, cb if what != hb.EVAL_JAC else None,
                       b - a, jb if what != hb.EVAL_CON else None, a, b)
        torch.cuda.synchronize()
        if what != hb.EVAL_JAC:
            assert torch.equal(cb, ref_c[:, a:b]), (a, b, what)
        if what != hb.EVAL_CON:
            assert torch.equal(jb.view(b - a, P), ref_j[a:b]), (a, b, what)
    # node shards of 2 / 3 / 8 ranks, each with the code object of ITS launch
    # size, written in place: the gathered vectors are the 1-rank launch's
    for world in (2, 3, 8):
        ranges = partition_nodes(ncn, world)
        gc = torch.full((M, ncn), float('nan'), **f64)
        gj = torch.full((ncn*P,), float('nan'), **f64)
        shard = _scaled(name, nodes,
                        launch_nodes=max(b - a for a, b in ranges))
        shard._sync_known(shard.hip, free_h)
        for a, b in ranges:
            shard.hip.eval_shard(hb.EVAL_FUSED, free, gc[:, a:b], ncn,
                                 gj[a*P:b*P], a, b)
        torch.cuda.synchronize()
        assert torch.equal(gc, ref_c), world
        assert torch.equal(gj.view(ncn, P), ref_j), world
        shard.hip.close()
    hip.close()
