import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, opty_amd
from opty_amd import hip_backend as hb
from examples import problems
name = sys.argv[1]
pkw = problems.build(name)
col = opty_amd.ConstraintCollocator(jacobian_layout='csr', **pkw)
opts = copy.copy(col._printer_options())
opts.order = opts.fused_order = 'list'
free = problems.make_free(col.num_free, seed=11, variable_duration=col._variable_duration)
j0 = np.array(col.generate_jacobian_function()(free))
c0 = col.generate_constraint_function()(free)
sib = opty_amd.ConstraintCollocator(jacobian_layout='csr', emit_options=opts, verify_builds='off', **pkw)
c2, j2 = np.empty_like(c0), np.empty_like(j0)
sib.hip.eval_con_jac(free, c2, j2, hb.HOST)
meta = sib.generate_source()[1]
print('fused groups', meta['fused_groups'] if 'fused_groups' in meta else None)
print('con wrong', int((np.abs(c2-c0) > 1e-9*np.abs(c0).max()).sum()))
N = col.num_collocation_nodes; P = sib.hip.desc['P']; ncn = N - 1
prog = sib._build_program(); rs = list(prog.row_start)
bad = np.flatnonzero(np.abs(j2 - j0) > 1e-9*np.abs(j0).max())
ent = {}
for b in bad:
    if b >= P*ncn: ent.setdefault('tail', []).append(b); continue
    j = np.searchsorted(np.array(rs)*ncn, b, side='right') - 1
    L = rs[j+1] - rs[j]; off = b - rs[j]*ncn; i = off//L; e = rs[j] + off % L
    ent.setdefault(int(e), []).append(int(i))
print('N', N, 'P', P, 'rows', rs)
for e in sorted(k for k in ent if k != 'tail'):
    k0 = [b for b in bad if b < P*ncn][0]
print({e: ent[e] for e in sorted(k for k in ent if k != 'tail')})
print('tail', ent.get('tail'))
k = bad[:8]; print('got', j2[k], 'want', j0[k])
