#!/usr/bin/env python
"""Developer tool (GPU box): 1/8 shard of a 24-link stand-in under the launch
plan vs the printer's own geometry vs the whole-problem launch, entry by
entry."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import torch
import opty_amd
from opty_amd import hip_backend as hb
from opty_amd.sharded import ShardedCollocator
from examples import problems

name = sys.argv[1] if len(sys.argv) > 1 else 'config5_standin_24link'
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kw = problems.build(name)
res = {}
for tag, env in (('plan', None), ('seed', 'off')):
    if env:
        os.environ['OPTY_LAUNCH_PLANS'] = env
    else:
        os.environ.pop('OPTY_LAUNCH_PLANS', None)
    sh = ShardedCollocator(rank=rank, world_size=8, **kw)
    meta = sh.collocator.generate_source()[1]
    print(tag, meta['geometry'], {k: v['groups'] for k, v in meta['kernels'].items()})
    free = problems.make_free(sh.collocator.num_free, seed=0, variable_duration=True)
    dfree = torch.from_numpy(free).cuda()
    outs = []
    for rep in range(3):
        con, jac = sh.evaluate(dfree)
        torch.cuda.synchronize()
        outs.append((con.cpu().numpy().copy(), jac.cpu().numpy().copy()))
    print(tag, 'repeatable:', all(np.array_equal(outs[0][1], o[1]) for o in outs[1:]),
          all(np.array_equal(outs[0][0], o[0]) for o in outs[1:]))
    # separate kernels of the same module
    con2 = torch.empty_like(sh.con_local); jac2 = torch.empty_like(sh.jac_local)
    sh.collocator.hip.eval_shard(hb.EVAL_PAIR, dfree, con2, con2.stride(0), jac2, sh.a, sh.b)
    torch.cuda.synchronize()
    res[tag] = (outs[0][1], jac2.cpu().numpy(), sh.a, sh.b, meta)
P = 5100
def diff(a, b, label, groups):
    rel = np.abs(a - b)/np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300)
    bad = np.nonzero(rel > 1e-9)[0]
    print(label, 'differ at', len(bad), 'of', len(a))
    ents = {}
    for k in bad[:100000]:
        node, e = divmod(int(k), P)
        ents.setdefault(e, 0); ents[e] += 1
    strips = [(g[0][0], g[0][1]) for g in groups]
    for e, cnt in sorted(ents.items())[:60]:
        s = [i for i, (e0, e1) in enumerate(strips) if e0 <= e < e1]
        print('   entry %d (eq %d col %d) strip %s: %d nodes' % (e, e//102, e % 102, s, cnt))
fp, jp, a, b, mp = res['plan']
fs, js, _, _, ms = res['seed']
diff(fp, jp, 'plan fused vs plan jac', mp['fused_groups'])
diff(fs, js, 'seed fused vs seed jac', ms['fused_groups'])
diff(fp, fs, 'plan fused vs seed fused', mp['fused_groups'])
diff(jp, js, 'plan jac vs seed jac', mp['groups'])
