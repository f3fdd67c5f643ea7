#!/usr/bin/env python
"""Developer tool (GPU box): where the time of one small host-path call goes
(BASELINE config 2: N = 10 000, 30 000 free values, 20 002 constraints,
119 992 Jacobian values)."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import torch
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
col = opty_amd.ConstraintCollocator(**problems.build('config2_pendulum'))
hip = col.hip
cf, jf = col.generate_constraint_function(), col.generate_jacobian_function()
free = problems.make_free(col.num_free)
if 'trace' in sys.argv[1:]:
    # child of the run below: OPTY_HIP_TRACE=1 prints the phases of each call
    for fn in (cf, jf):
        for _ in range(6):
            fn(free)
    sys.exit(0)
print('launch geometry', hip.desc['num_uniform'], 'uniform values (dynamic: %s),'
      % bool(hip.desc['uniform_dynamic']), hip.desc['num_inst'],
      'instance constraints')
con = np.empty(col.num_constraints)
jac = hb.pinned_empty(hip.nnz)
dev = torch.device('cuda:0')
fd = torch.from_numpy(free).to(dev)
cd = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
jd = torch.empty(hip.nnz, dtype=torch.float64, device=dev)

def med(fn, n=200):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e6*float(np.median(ts))

print('constraints(free) callback            %6.1f us' % med(lambda: cf(free)))
print('  opty_hip_eval_con, host buffers     %6.1f us' % med(lambda: hip.eval_con(free, con, hb.HOST)))
print('  _sync_known + np.empty              %6.1f us' % med(lambda: (col._sync_known(hip, free), np.empty(col.num_constraints))))
print('  device pointers + synchronize       %6.1f us' % med(lambda: (hip.eval_con(fd, cd, hb.DEVICE), hip.synchronize())))
print('jacobian(free) callback               %6.1f us' % med(lambda: jf(free)))
print('  opty_hip_eval_jac, host buffers     %6.1f us' % med(lambda: hip.eval_jac(free, jac, hb.HOST)))
print('  device pointers + synchronize       %6.1f us' % med(lambda: (hip.eval_jac(fd, jd, hb.DEVICE), hip.synchronize())))
print('torch H2D 240 KB (pageable) + sync    %6.1f us' % med(lambda: (fd.copy_(torch.from_numpy(free)), torch.cuda.synchronize())))

import subprocess
sys.stdout.flush()
subprocess.run([sys.executable, __file__, 'trace'],
               env=dict(os.environ, OPTY_HIP_TRACE='1'))
