"""Developer tool (GPU): where a build's Jacobian differs from the reference
golden -- per kernel (separate / fused), per strip, per (equation, column).
Usage: dbg_golden.py <golden name> [-O1] [key=value printer options ...]"""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
os.environ['OPTY_CROSS_CHECK'] = 'off'
os.environ['OPTY_LAUNCH_PLANS'] = 'off'
import numpy as np
import opty_amd
from opty_amd import hip_backend as hb
from opty_amd.codegen.emit_hip import EmitOptions
from examples import problems
import golden_util as gu

name = sys.argv[1]
okw, opt = {}, None
for a in sys.argv[2:]:
    if a.startswith('-O'):
        opt = a
    else:
        k, v = a.split('=')
        okw[k] = v if k in ('cut', 'ablate', 'con_split', 'small_flush') \
            else int(v)
meta, z = gu.load(name)
kw = problems.build(name)
col = opty_amd.ConstraintCollocator(
    emit_options=EmitOptions(**okw) if okw else None, **kw)
if opt:
    os.environ['OPTY_HIPCC_OPT'] = opt
hsaco, m = col._build_code_object(opt_level=opt)
print('geometry', m['geometry'], 'attached', m['con_attached'])
print('resources', hb.kernel_resources(hsaco) if hasattr(
    hb, 'kernel_resources') else '')
print('spills', hb.vgpr_spills(hsaco))
import torch
if not torch.cuda.is_available():
    sys.exit('built; no GPU here')
h = col.hip
N1, M, C = meta['N'] - 1, meta['M'], meta['C']
P = M*C
free = z['free']
con = np.full(col.num_constraints, np.nan)
jac = np.full(h.nnz, np.nan)
jac2 = np.full(h.nnz, np.nan)
con2 = np.full(col.num_constraints, np.nan)
h.eval_con(free, con, hb.HOST)
h.eval_jac(free, jac, hb.HOST)
h.eval_con_jac(free, con2, jac2, hb.HOST)
want = z['jac']
scale = np.abs(want[:N1*P]).reshape(N1, M, C).max(axis=2, keepdims=True)
for label, got, groups in (('opty_jac', jac, m['groups']),
                           ('opty_conjac', jac2, m['fused_groups'])):
    err = np.abs(got[:N1*P] - want[:N1*P]).reshape(N1, M, C)
    bad = ~(err <= 1e-9*np.maximum(scale, 1e-300))
    print(label, 'bad entries', int(bad.sum()), 'of', bad.size,
          'nan', int(np.isnan(got).sum()))
    if not bad.any():
        continue
    flat = bad.reshape(N1, P)
    per_entry = flat.sum(axis=0)
    ents = np.nonzero(per_entry)[0]
    print('  block entries wrong (entry: nodes wrong):',
          {int(e): int(per_entry[e]) for e in ents[:60]})
    print('  nodes with wrong entries:', np.nonzero(flat.sum(axis=1))[0])
    for g, grp in enumerate(groups):
        for e0, e1 in grp:
            n = int(per_entry[e0:e1].sum())
            if n:
                print('  strip %d [%d, %d): %d wrong' % (g, e0, e1, n))
    k = int(np.argmax(np.where(flat, err.reshape(N1, P), -1)))
    nd, e = divmod(k, P)
    print('  worst node %d entry %d (eq %d col %d): got %r want %r' % (
        nd, e, e//C, e % C, got[k], want[k]))
print('con max rel err sep/fused',
      float(np.nanmax(np.abs(con - z['con']))/np.abs(z['con']).max()),
      float(np.nanmax(np.abs(con2 - z['con']))/np.abs(z['con']).max()))
