#!/usr/bin/env python
"""CPU tool: the isomorphic sub-models of the arithmetic-bound problems
(``opty_amd.codegen.isomorph``) and what evaluating them across the lanes of
a wave could buy -- the analysis half of ``profiles/r06_isomorphic.txt``
(VERDICT r05 item 1).  Per problem: the groups found, the heaviest wave of the
launch plan in use, the Amdahl bound on per-lane work (latency of an
under-filled launch: the 1/8 shards) and the SIMD-time factor (full-size
launches).

    isomorphic_report.py [problem ...]
"""
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import opty_amd                                               # noqa: E402
from examples import problems                                 # noqa: E402
from opty_amd.codegen import emit_hip, ir, isomorph           # noqa: E402

DEFAULT = ('one_legged_small', 'biped_small', 'gaitlike_3link_be_small',
           'config3_10link_small', 'config5_gaitlike_24link_small')


def describe(dag, root):
    """A few words on what an instance computes: its transcendental
    operations and the trajectory rows it reads."""
    is_leaf = isomorph.default_leaf(dag)
    c = isomorph.cone(dag, root, is_leaf)
    ops, rows = {}, set()
    for v in c:
        if dag.op[v] in ir.UNARY or dag.op[v] in (ir.POW, ir.DIV):
            ops[dag.op[v]] = ops.get(dag.op[v], 0) + 1
        for j in dag.operands(v):
            if dag.op[j] == ir.INPUT and dag.args[j][0] in ('cur', 'adj'):
                rows.add(dag.args[j][1])
    return ', '.join('%d %s' % (n, k) for k, n in sorted(ops.items())) + \
        '; rows %s' % sorted(rows)


def report(name):
    kw = problems.build(name)
    col = opty_amd.ConstraintCollocator(**kw)
    prog = col._build_program()
    d = prog.dag
    is_leaf = isomorph.default_leaf(d)

    def w(i):
        return emit_hip._node_weight(d, i)
    roots = list(prog.con_out) + list(prog.jac_out)
    need = [i for i in d.reachable(roots) if not is_leaf(i)]
    total = sum(w(i) for i in need)
    depth = {}
    for i in need:
        depth[i] = w(i) + max([depth.get(j, 0) for j in d.operands(i)] + [0])
    print('%s: %d states, block of %d entries (%d not structurally zero), '
          '%d weighted operations per node, critical path %d'
          % (name, prog.n, prog.P, sum(1 for e in prog.jac_out
                                       if e != d.zero), total,
             max(depth.values()) if depth else 0))
    groups = isomorph.instance_groups(d, roots, w, is_leaf)
    if not groups:
        print('  no isomorphic instances of >= 60 weighted operations')
    for g in groups:
        print('  group: k = %d instances x %d operations (%d shared), %d '
              'interface values, %d distinct leaves each; lane work saved '
              '%d  [%s]' % (g['k'], g['weight'], g['shared'], g['interface'],
                            g['leaves'], g['saved'],
                            describe(d, g['roots'][0])))
    b = isomorph.lane_vectorisation_bounds(total, groups)
    print('  whole block in the lanes of one node group (k = %d): per-lane '
          'work %d -> %d (latency x %.2f at best), SIMD-time x %.2f'
          % (b['k'], total, b['lane_work'], b['latency_gain'],
             b['simd_time']))
    # the heaviest wave the plan in use evaluates (what a shard's latency is)
    src, meta = col.generate_source()
    plans = getattr(meta, 'get', lambda *_: None)('plans')
    heavy = None
    for kern in ('conjac', 'jac'):
        cost = meta['kernels'][kern].get('class_cost')
        if cost:
            heavy = (kern, max(cost), sum(cost))
            break
    if heavy:
        print('  plan in use: heaviest wave of opty_%s ~%.0f of %.0f '
              'weighted operations per block (all waves)'
              % (heavy[0], heavy[1], heavy[2]))
    print()


def main():
    for name in sys.argv[1:] or DEFAULT:
        report(name)


if __name__ == '__main__':
    main()
