"""Developer tool (GPU box): how often does a build return wrong values?
dbg_list_csr.py <problem> [layout] -- the plan's options with dispatch order
'list' against the default build, 30 evaluations each of the separate and the
fused Jacobian, register files poisoned before every other one."""
import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, opty_amd
from opty_amd import hip_backend as hb
from examples import problems
name = sys.argv[1]
layout = sys.argv[2] if len(sys.argv) > 2 else 'csr'
pkw = problems.build(name)
col = opty_amd.ConstraintCollocator(jacobian_layout=layout, **pkw)
opts = copy.copy(col._printer_options())
opts.order = opts.fused_order = 'list'
free = problems.make_free(col.num_free, seed=11,
                          variable_duration=col._variable_duration)
j0 = np.array(col.generate_jacobian_function()(free))
c0 = col.generate_constraint_function()(free)
for level in ('-O2', '-O1'):
    import os
    os.environ['OPTY_HIPCC_OPT'] = level
    sib = opty_amd.ConstraintCollocator(jacobian_layout=layout,
                                        emit_options=opts,
                                        verify_builds='off', **pkw)
    hsaco, meta = sib._build_code_object()
    res = hb.cached_kernel_resources(hsaco)
    print(level, {k: (res[k]['.vgpr_count'], res[k]['.sgpr_spill_count'],
                      res[k]['.vgpr_spill_count'])
                  for k in ('opty_jac', 'opty_conjac')})
    jf = sib.generate_jacobian_function()
    bad = {'jac': 0, 'fused': 0}
    counts = []
    for k in range(30):
        if k % 2:
            hb.poison_registers()
        j1 = np.array(jf(free))
        c2, j2 = np.empty_like(c0), np.empty_like(j0)
        sib.hip.eval_con_jac(free, c2, j2, hb.HOST)
        w1 = int((np.abs(j1 - j0) > 1e-9*np.abs(j0).max()).sum())
        w2 = int((np.abs(j2 - j0) > 1e-9*np.abs(j0).max()).sum())
        bad['jac'] += w1 > 0
        bad['fused'] += w2 > 0
        counts.append((w1, w2))
    print(level, 'wrong evaluations of 30:', bad, 'wrong values per call',
          counts[:12])
    sib.hip.close()
