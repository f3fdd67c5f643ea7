#!/usr/bin/env python
"""GPU tool (r06): A/B of the cooperative geometry for launches that
under-fill the chip -- ``EmitOptions(publish=1)``: one workgroup per node
block, the block's isomorphic sub-models evaluated once, one instance per
wave, and published through LDS before the strips -- on the 1/8 node shard of
the muscle-driven leg (what each GPU of an 8-GPU node launches), against the
launch plan in use and against the same strips without the publication stage.
Every variant is held to the instruction tape before it is timed.

    ab_publish.py [workload [world]]       env: OPTY_AB_ROUNDS (7), OPTY_AB_ITERS (200)
    ab_publish.py --prebuild               (no GPU: compiles into the cache)
"""
import copy
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from opty_amd.sharded import partition_nodes                  # noqa: E402
from examples import problems                                 # noqa: E402

CUT = '0:112;112:144;144:176;176:348'       # rows 4 and 5 in waves of their own


def variants(base):
    yield 'plan (auto)', None
    for label, kw in (
            ('4 strips', {}),
            ('4 strips + share_rcp', dict(share_rcp=1)),
            ('4 strips + publish', dict(publish=1)),
            ('4 strips + publish + share_rcp', dict(publish=1, share_rcp=1))):
        o = copy.copy(base)
        o.chunk, o.groups, o.fused_groups, o.pad = 16, 4, 4, 0
        o.strips = o.fused_strips = CUT
        o.waves = 4
        # (hardware dispatch, no LDS parking: what the publication stage is
        # printed for; the full-size plan of the leg has a list schedule)
        o.order = o.fused_order = None
        o.park, o.share_rcp = 0, 0
        for k, v in kw.items():
            setattr(o, k, v)
        yield label, o


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    prebuild = '--prebuild' in sys.argv
    workload = args[0] if args else 'config5_one_legged'
    world = int(args[1]) if len(args) > 1 else 8
    kw = problems.build(workload)
    ncn = kw['num_collocation_nodes'] - 1
    a, b = partition_nodes(ncn, world)[world//2 - 1 if world > 1 else 0]
    base = opty_amd.ConstraintCollocator(launch_nodes=b - a, **kw)
    prog = base._build_program()
    cols = []
    for label, opts in variants(base._printer_options()):
        col = opty_amd.ConstraintCollocator(
            launch_nodes=b - a, emit_options=opts, **kw)
        col._program = prog
        if prebuild:
            hsaco, meta = col._build_code_object()
            print('%-34s %s spills %s' % (label, os.path.basename(hsaco),
                                         hb.vgpr_spills(hsaco)), flush=True)
            continue
        try:
            col.hip
        except hb.BuildRejected as err:
            print('%-34s REFUSED: %s' % (label, err.verdict['errors']))
            continue
        cols.append((label, col))
    if prebuild:
        return
    free = hb.DeviceVector(problems.make_free(
        base.num_free, variable_duration=base._variable_duration))
    con = hb.DeviceVector(np.zeros(prog.M*(b - a)))
    jac = hb.DeviceVector(np.zeros(prog.P*(b - a)))
    rounds = int(os.environ.get('OPTY_AB_ROUNDS', 7))
    iters = int(os.environ.get('OPTY_AB_ITERS', 200))
    times = {label: {'fused': [], 'jac': []} for label, _ in cols}
    for r in range(rounds + 1):
        for label, col in cols:
            for what, tag in ((hb.EVAL_FUSED_KERNEL, 'fused'),
                              (hb.EVAL_JAC, 'jac')):
                ms = col.hip.time_eval_shard(what, free, con, b - a, jac, a,
                                             b, iters)
                if r:
                    times[label][tag].append(ms)
    print('# %s, nodes [%d, %d) of %d (1/%d shard), %d rounds x %d launches, '
          'median ms' % (workload, a, b, ncn, world, rounds, iters))
    ref = None
    for label, col in cols:
        f = float(np.median(times[label]['fused']))
        j = float(np.median(times[label]['jac']))
        ref = ref or f
        k = col._kernel_meta['kernels']['conjac']
        print('%-34s opty_conjac %.4f ms (x%.2f)  EVAL_JAC %.4f ms   %d waves '
              'per workgroup, %d published rows, LDS %d KB, verified %.1e'
              % (label, f, ref/f, j, k['waves_per_wg'],
                 k.get('published_rows', 0), k['lds_bytes']//1024,
                 col._build_verdict['worst']), flush=True)


if __name__ == '__main__':
    main()
