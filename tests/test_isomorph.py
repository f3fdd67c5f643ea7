"""Isomorphic sub-models of the expression DAG (``codegen/isomorph.py``) and
the cooperative geometry built on them (``EmitOptions(publish=1)``: one
workgroup per node block, instances evaluated once -- one per wave -- and
published through LDS before the strips)."""
import copy

import numpy as np
import pytest

from examples import problems
from opty_amd import ConstraintCollocator
from opty_amd import hip_backend as hb
from opty_amd.codegen import emit_hip, isomorph
from opty_amd.codegen.emit_hip import EmitOptions

LEG_CUT = '0:112;112:144;144:176;176:348'


def _groups(name):
    col = ConstraintCollocator(**problems.build(name))
    prog = col._build_program()
    d = prog.dag
    roots = list(prog.con_out) + list(prog.jac_out)
    return d, roots, isomorph.instance_groups(
        d, roots, lambda i: emit_hip._node_weight(d, i))


def test_muscles_of_the_leg_are_found_and_the_bound_is_amdahl():
    """The gallery's muscle-driven leg: three musculotendon actuators of the
    same shape (one of the four is wired differently: the four-bar knee) and
    four activation dynamics, each with a handful of interface values -- and
    a lane-vectorised wave could shorten a lane's work by a third at best,
    while the launch's SIMD-time more than doubles."""
    d, roots, groups = _groups('one_legged_small')
    shapes = sorted((g['k'], g['weight']) for g in groups)
    assert (3, 496) in shapes and (4, 115) in shapes, shapes
    muscles = [g for g in groups if g['k'] == 3][0]
    assert muscles['shared'] == 0 and muscles['interface'] <= 16
    is_leaf = isomorph.default_leaf(d)
    cones = [isomorph.cone(d, r, is_leaf) for r in muscles['roots']]
    # (nearly) disjoint: a few cheap common sub-expressions at most
    for a in range(3):
        for b in range(a + 1, 3):
            both = sum(emit_hip._node_weight(d, i)
                       for i in cones[a] & cones[b])
            assert both <= 0.1*muscles['weight'], (a, b, both)
    total = sum(emit_hip._node_weight(d, i) for i in d.reachable(roots)
                if not is_leaf(i))
    b = isomorph.lane_vectorisation_bounds(total, groups)
    assert 1.2 < b['latency_gain'] < 1.5 and b['simd_time'] > 2.0, b


def test_mirror_legs_of_the_biped_have_a_wide_interface():
    """The biped's two legs ARE isomorphic (828 operations each), but forward
    mode hands ~150 values of each to the rest of the block: nothing to gain
    from evaluating them side by side."""
    d, roots, groups = _groups('biped_small')
    legs = max(groups, key=lambda g: g['weight'])
    assert legs['k'] == 2 and legs['weight'] > 700
    assert legs['interface'] > 100
    is_leaf = isomorph.default_leaf(d)
    total = sum(emit_hip._node_weight(d, i) for i in d.reachable(roots)
                if not is_leaf(i))
    assert isomorph.lane_vectorisation_bounds(
        total, groups)['latency_gain'] < 1.1


def test_shape_hash_ignores_leaves_and_operand_order():
    from opty_amd.codegen import ir
    d = ir.DAG()
    x, y, z, w = (d.input('cur', k) for k in range(4))
    a = d.add(d.mul(x, y), d.unary('sin', x))
    b = d.add(d.unary('sin', z), d.mul(w, z))      # same shape, other leaves
    c = d.add(d.mul(x, y), d.unary('cos', x))      # other operation
    h = isomorph.shape_hashes(d, sorted(d.reachable([a, b, c])))
    assert h[a] == h[b] != h[c]


def _publishing_options(col):
    o = copy.copy(col._printer_options())
    o.chunk, o.groups, o.fused_groups, o.pad = 16, 4, 4, 0
    o.strips = o.fused_strips = LEG_CUT
    o.waves, o.publish = 4, 1
    return o


def test_publication_stage_is_printed_and_builds_without_spills():
    kw = problems.build('one_legged_small')
    base = ConstraintCollocator(launch_nodes=6250, **kw)
    col = ConstraintCollocator(launch_nodes=6250,
                               emit_options=_publishing_options(base), **kw)
    src, meta = col.generate_source()
    for kern in ('jac', 'conjac'):
        k = meta['kernels'][kern]
        assert k['waves_per_wg'] == 4 and k['wgs_per_block'] == 1
        assert k['published_rows'] == 82
        assert 400 < k['publish_stage_weight'] < 700
        assert k['lds_bytes'] <= 160*1024
    assert meta['kernels']['con']['published_rows'] == 0
    assert 'double *const pub = lds +' in src and '__syncthreads();' in src
    hsaco = col._compile(src)
    assert hb.vgpr_spills(hsaco) == {}
    # without instances to publish the option changes nothing
    kw3 = problems.build('config3_10link_small')
    a, _ = ConstraintCollocator(**kw3).generate_source()
    o = copy.copy(ConstraintCollocator(**kw3)._printer_options())
    o.publish = 1
    col3 = ConstraintCollocator(emit_options=o, **kw3)
    _, m3 = col3.generate_source()
    assert m3['kernels']['conjac']['published_rows'] in (0, 45)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['one_legged_small'])
def test_publishing_kernels_match_the_reference(name):
    """The cooperative geometry against the reference's golden record and
    against the instruction tape (every build is verified before use)."""
    import golden_util as gu
    meta, z = gu.load(name)
    kw = problems.build(name)
    base = ConstraintCollocator(**kw)
    col = ConstraintCollocator(emit_options=_publishing_options(base), **kw)
    hip = col.hip
    assert col._build_verdict['ok'] and col._build_verdict['worst'] < 1e-11
    assert col._kernel_meta['kernels']['conjac']['published_rows'] > 0
    con = col.generate_constraint_function()(z['free'])
    jac = col.generate_jacobian_function()(z['free'])
    cb, jb = gu.error_bounds(col, z['free'])
    ccap, jcap = gu.caps_for(z['jac'], len(z['con']), meta['N'] - 1,
                             meta['M'], meta['C'])
    gu.assert_close(con, z['con'], 1e-10, what=name + ' publish con',
                    bound=cb, cap=ccap)
    gu.assert_close(jac, z['jac'], 1e-10, what=name + ' publish jac',
                    bound=jb, cap=jcap)
    con2, jac2 = np.empty_like(con), np.empty_like(np.asarray(jac))
    hip.eval_con_jac(z['free'], con2, jac2, hb.HOST)
    gu.assert_close(jac2, z['jac'], 1e-10, what=name + ' publish fused jac',
                    bound=jb, cap=jcap)
    gu.assert_close(con2, z['con'], 1e-10, what=name + ' publish fused con',
                    bound=cb, cap=ccap)
