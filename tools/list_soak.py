#!/usr/bin/env python
"""Developer tool: every problem of the zoo (``__graft_entry__.
prebuilt_collocators``: configs, layouts, shard launch sizes, seeded random
problems, the gallery) once more with PERSISTENT kernels (dispatch order
'list' forced on the plan's options) against its default build: separate and
fused evaluation, tape-refereed builds.

    python tools/list_soak.py --prebuild [i n]   # CPU: compile (share i of n)
    python tools/list_soak.py [substring ...]    # GPU box: compare
env: SOAK_OPTIONS = printer options to force (default order=list,fused_order=list)
     LIST_SOAK_SKIP = comma-separated substrings of names to leave out
     (default: the 24-link stand-ins, whose modules take minutes to compile)
"""
import copy
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np                                            # noqa: E402
import __graft_entry__ as ge                                  # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402

SKIP = os.environ.get('LIST_SOAK_SKIP', '24link,config5_biped,'
                      'config5_one_legged').split(',')


def siblings(shard=None, only=()):
    for name, kw, col, pkw in ge.prebuilt_collocators(shard=shard,
                                                      with_problem=True):
        label = '%s %s' % (name, kw or '')
        if any(s and s in label for s in SKIP):
            continue
        if only and not any(a in label for a in only):
            continue
        opts = copy.copy(col._printer_options())
        # SOAK_OPTIONS="fast_trig=2": other printer options than the
        # persistent kernels' (e.g. the uniform-sincos replacement build)
        forced = os.environ.get('SOAK_OPTIONS', 'order=list,fused_order=list')
        for item in forced.split(','):
            k, v = item.split('=')
            setattr(opts, k, int(v) if v.lstrip('-').isdigit() else v)
        yield label, col, opty_amd.ConstraintCollocator(
            emit_options=opts, **kw, **pkw)


def main():
    args = sys.argv[1:]
    if args and args[0] == '--prebuild':
        shard = (int(args[1]), int(args[2])) if len(args) > 2 else None
        for label, col, sib in siblings(shard):
            hsaco, meta = sib._build_code_object()
            k = meta['kernels']
            print(label, 'persist', k['jac'].get('persist'),
                  k['conjac'].get('persist'), hb.vgpr_spills(hsaco),
                  flush=True)
        return
    worst, count, t00 = 0.0, 0, time.time()
    escaped, refused = [], []
    for label, col, sib in siblings(only=args):
        t0 = time.time()
        free = problems.make_free(col.num_free, seed=11,
                                  variable_duration=col._variable_duration)
        try:
            sib.hip                    # build + tape referee
        except hb.BuildRejected as err:
            print('%-60s REFUSED %s' % (label, err.verdict['errors']),
                  flush=True)
            refused.append(label)
            continue
        c0 = col.generate_constraint_function()(free)
        j0 = np.array(col.generate_jacobian_function()(free))
        c1 = sib.generate_constraint_function()(free)
        j1 = np.array(sib.generate_jacobian_function()(free))
        c2, j2 = np.empty_like(c0), np.empty_like(j0)
        sib.hip.eval_con_jac(free, c2, j2, hb.HOST)
        scale_c = max(1.0, np.abs(c0).max())
        scale_j = max(1.0, np.abs(j0).max())
        err = max(np.abs(c1 - c0).max()/scale_c, np.abs(c2 - c0).max()/scale_c,
                  np.abs(j1 - j0).max()/scale_j, np.abs(j2 - j0).max()/scale_j)
        d = sib.hip.desc
        worst, count = max(worst, err), count + 1
        print('%-60s persist %4d %4d  max error %.2e of the scale  [%.0f s]'
              % (label[:60], d['jac_persist'], d['fused_persist'], err,
                 time.time() - t0), flush=True)
        if err >= 1e-11:
            escaped.append(label)   # wrong values from a build the referee
            #                         accepted
        col.hip.close()
        sib.hip.close()
    print('%d problems compared, worst %.2e, %.0f s; refused by the referee: '
          '%s; WRONG AND ACCEPTED: %s' % (count, worst, time.time() - t00,
                                          refused, escaped))


if __name__ == '__main__':
    main()
