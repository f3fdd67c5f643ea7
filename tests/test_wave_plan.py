"""CPU tests of the round-5 printer features: planned waves with LDS parking
(``emit_hip._WavePlan``), the dispatch orders of a launch's workgroups, explicit
strip cuts and the host evaluation of node-invariant values
(``codegen/evaluate.py``)."""
import re

import numpy as np
import pytest

import opty_amd
from opty_amd.codegen import ir
from opty_amd.codegen.emit_hip import (EmitOptions, RESIDENT_WAVES,
                                       _ModuleWriter, _WavePlan, emit_module)
from examples import problems


def _replay(plan, dag, is_leaf):
    """Executes a wave plan symbolically: every operand of every operation
    must be in a register (computed or reloaded and not evicted since) or in
    the LDS slot it was parked in (and that slot must still hold IT); every
    ring write finds its value; parked values are stored while they are
    still in a register."""
    inreg, slot_holds, slot_of = set(), {}, {}
    computed = set()
    ring_writes = 0
    for t, ev in enumerate(plan.events):
        if ev[0] == 'op':
            i = ev[1]
            for j in dag.operands(i):
                if is_leaf(j):
                    continue
                assert j in computed, (t, i, j)
                if j not in inreg:
                    assert j in slot_of and slot_holds[slot_of[j]] == j, \
                        'value %d is neither in a register nor parked' % j
                    inreg.add(j)                    # the reload
            for v in ev[1:]:
                if v is not None:
                    assert v not in computed, 'computed twice: %d' % v
                    computed.add(v)
                    inreg.add(v)
        elif ev[0] == 'con':
            r = plan.targets[ev[1]][2]
            if not is_leaf(r):
                assert r in inreg or slot_holds.get(slot_of.get(r)) == r
        elif ev[0] == 'chunk':
            for k in plan.chunks[ev[1]]:
                r = plan.targets[k][2]
                ring_writes += 1
                if not is_leaf(r):
                    assert r in computed
                    assert r in inreg or slot_holds.get(slot_of.get(r)) == r
        for v, sl in plan.parks.get(t, ()):
            assert v in inreg, 'parked after it left the registers'
            slot_holds[sl] = v
            slot_of[v] = sl
            assert plan.slot_of[v] == sl
        for v in plan.drops.get(t, ()):
            assert v in slot_of and slot_holds[slot_of[v]] == v, \
                'dropped without a copy in LDS'
            inreg.discard(v)
    assert ring_writes == sum(len(c) for c in plan.chunks)
    assert max(list(slot_holds) + [-1]) < plan.slots


def _leg_plan(budget):
    col = opty_amd.ConstraintCollocator(**problems.build('one_legged_small'))
    prog = col._build_program()
    d = prog.dag
    w = _ModuleWriter(prog, EmitOptions(chunk=16))

    def is_leaf(i):
        return d.op[i] == ir.CONST or w._is_vec_input(i) or \
            w._uniform_leaf(i)
    e0, e1, K = 96, 160, 16
    targets = [('c', j, prog.con_out[j]) for j in range(prog.M)]
    chunks = []
    for c0 in range(e0, e1 + 15, K):
        ks = []
        for v in range(c0, min(c0 + K, e1 + 15)):
            ks.append(len(targets))
            targets.append(('e', v, prog.jac_out[v % prog.P]))
        chunks.append(ks)
    return _WavePlan(d, targets, chunks, is_leaf, budget), d, is_leaf


@pytest.mark.parametrize('budget', [235, 200, 150, 64])
def test_wave_plan_is_executable(budget):
    """The merged heavy strip of the muscle-driven leg (both dynamic rows in
    one wave, constraint rows attached), planned for several register
    budgets: the plan can be executed as it is printed, every value is
    computed once, the slots it needs shrink with the budget's slack."""
    plan, d, is_leaf = _leg_plan(budget)
    _replay(plan, d, is_leaf)
    assert plan.order in ('memory', 'cheapest')
    assert plan.peak <= 300
    assert plan.slots >= max(0, plan.peak - budget - 2)
    ops = [e[1] for e in plan.events if e[0] == 'op']
    assert len(ops) == len(set(ops))


def test_planned_module_parks_only_what_it_needs():
    """Printed module: the merged fused strip is planned (park / unpark
    through ``lane_z``), the kernels' LDS grows by exactly the parking rows,
    and a cut that does not need parking prints none."""
    col = opty_amd.ConstraintCollocator(**problems.build('one_legged_small'))
    prog = col._build_program()
    opts = EmitOptions(chunk=16, groups=5, park=48, park_live=235,
                       fused_strips='0:96;96:160;160:348')
    src, meta = emit_module(prog, opts, node_blocks=782)
    (plan,) = meta['plans']
    assert plan['strips'] == [(96, 160)] and 0 < plan['slots'] <= 31
    k = meta['kernels']
    assert k['conjac']['park_rows'] == plan['slots'] and \
        k['jac']['park_rows'] == 0
    assert k['conjac']['lds_bytes'] - k['jac']['lds_bytes'] == \
        plan['slots']*64*8
    assert k['conjac']['lds_bytes'] <= 40*1024          # four waves per CU
    body = src[src.index('opty_conjac('):]
    parks = re.findall(r'opty_park\(park, (\d+), lane, (\w+)\);', body)
    loads = re.findall(r'opty_unpark\(park, (\d+), lane_z\)', body)
    assert parks and loads and 'lane_z = lane + (int)(N >> 62)' in body
    assert {int(s) for s, _ in parks} == set(range(plan['slots']))
    assert {int(s) for s in loads} <= {int(s) for s, _ in parks}
    plain, m2 = emit_module(prog, EmitOptions(chunk=16, groups=5, park=48,
                                              park_live=235),
                            node_blocks=782)
    # (a wave may still be PLANNED -- evaluated cheapest-next -- without
    # needing a single row)
    assert 'opty_park(' not in plain
    assert all(q['slots'] == 0 for q in m2.get('plans', []))


def _order_model(order, nblk, sets, W=1):
    """Python twin of the prologue's workgroup -> (block, set) map."""
    nblk8 = (nblk + 7)//8*8
    seen = []
    for b in range(nblk8*sets):
        xcd, slot = b & 7, b >> 3
        if order == 'tail' and sets > 1:
            tail = max(1, RESIDENT_WAVES//(8*max(1, sets - 1)))
            nslot = (nblk + 7) >> 3
            ntail = min(nslot, tail)
            nhead = nslot - ntail
            stail = slot - nhead*sets
            blk = (slot//sets if stail < 0 else nhead + stail % ntail)*8 + xcd
            grp = (slot % sets if stail < 0 else stail//ntail)
        elif order in ('class', 'tail'):
            nslot = (nblk + 7) >> 3
            blk = (slot % nslot)*8 + xcd
            grp = slot//nslot
        else:
            blk = (slot//sets)*8 + xcd
            grp = slot % sets
        if blk < nblk:
            seen.append((blk, grp))
    return seen


@pytest.mark.parametrize('order', ['block', 'class', 'tail'])
@pytest.mark.parametrize('nblk,sets', [(1, 3), (7, 5), (8, 2), (98, 5),
                                       (782, 3), (782, 6), (1563, 10),
                                       (40, 1)])
def test_dispatch_orders_cover_every_workgroup_once(order, nblk, sets):
    """Every (block, set) pair is handed out exactly once by each order, a
    block's workgroups stay on one XCD (workgroup id mod 8), and the printed
    prologue is the formula modelled here."""
    seen = _order_model(order, nblk, sets)
    assert len(seen) == nblk*sets == len(set(seen))
    assert {g for _, g in seen} == set(range(sets))
    if order == 'class':
        # strip class by strip class
        assert [g for _, g in seen] == sorted(g for _, g in seen)


def test_printed_prologue_matches_the_order_model():
    col = opty_amd.ConstraintCollocator(**problems.build('biped_small'))
    prog = col._build_program()
    for order, needle in (('class', 'blk = (slot % nslot)*8 + xcd'),
                          ('tail', 'nhead + stail % ntail'),
                          ('block', 'blk = (slot/5)*8 + xcd')):
        src, meta = emit_module(prog, EmitOptions(cut='work', groups=5,
                                                  fused_groups=5, order=order),
                                node_blocks=782)
        jac = src[src.index('opty_jac('):src.index('opty_conjac(')]
        assert needle in jac, order
    # per-kernel orders: the fused kernel may differ from opty_jac
    src, _ = emit_module(prog, EmitOptions(cut='work', groups=5,
                                           fused_groups=5, order='block',
                                           fused_order='tail'),
                         node_blocks=782)
    assert 'stail' not in src[src.index('opty_jac('):
                              src.index('opty_conjac(')]
    assert 'stail' in src[src.index('opty_conjac('):]


def test_explicit_strips_are_checked():
    col = opty_amd.ConstraintCollocator(**problems.build('one_legged_small'))
    prog = col._build_program()
    _, meta = emit_module(prog, EmitOptions(
        chunk=16, strips='96:160;160:348+0:96'), node_blocks=782)
    assert meta['groups'] == [[[96, 160]], [[160, 348], [0, 96]]]
    for bad in ('0:100;100:348', '0:96;112:348', '0:96;96:200'):
        with pytest.raises((AssertionError, ValueError)):
            emit_module(prog, EmitOptions(chunk=16, strips=bad),
                        node_blocks=782)


@pytest.mark.parametrize('name', ['config3_10link_small', 'one_legged_small',
                                  'c99_be_small', 'elementary_be_small'])
def test_host_evaluation_of_node_invariant_values(name):
    """``specialize_parameters=True``: the literals are the node-invariant
    nodes' values, evaluated on the host one rounded operation per DAG node
    -- equal to the test-only DAG interpreter (NumPy semantics) to rounding;
    nodes that depend on `free` (unknown parameters, a free interval) are left
    to the kernels."""
    import dag_interp
    col = opty_amd.ConstraintCollocator(specialize_parameters=True,
                                        **problems.build(name))
    lits = col._literals()
    prog = col._build_program()
    d = prog.dag
    assert lits, name
    par, h = col._known_scalars()
    for i, v in lits.items():
        assert d.uni[i] and d.op[i] != ir.CONST
    # the interpreter's values of the same nodes
    free = problems.make_free(col.num_free, seed=1,
                              variable_duration=col._variable_duration)
    inputs, _ = dag_interp._input_getter(col, free)
    roots = list(lits)
    out = dag_interp.evaluate(d, roots, inputs)
    vals = dict(zip(roots, out))
    for i, v in lits.items():
        want = float(np.asarray(vals[i]).ravel()[0])
        assert np.isclose(v, want, rtol=1e-13, atol=0.0) or \
            (np.isnan(v) and np.isnan(want)), (name, i, v, want)
    src, meta = col.generate_source()
    assert meta['literals'] == len(lits)
    dyn = col._variable_duration or col.num_unknown_parameters
    assert meta['num_uniform'] == 0 or dyn
