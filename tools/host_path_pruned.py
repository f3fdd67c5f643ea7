#!/usr/bin/env python
"""Developer tool (GPU box): jacobian(free) of config 3 with prune_zeros=True
through the NumPy callback (OPTY_HIP_PACK_RATIO: share of the stored block up
to which only the varying entries are moved)."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import opty_amd
from examples import problems
col = opty_amd.ConstraintCollocator(prune_zeros=True,
                                    **problems.build('config3_10link'))
jac = col.generate_jacobian_function()
frees = [problems.make_free(col.num_free, seed=s) for s in range(3)]
jac(frees[0]); jac(frees[1])
ts = []
for k in range(10):
    t0 = time.perf_counter(); jac(frees[k % 3]); ts.append(time.perf_counter() - t0)
print('ratio %s: pruned jacobian(free) min %.2f ms median %.2f ms (%d values)'
      % (os.environ.get('OPTY_HIP_PACK_RATIO', 'default'), 1e3*min(ts),
         1e3*sorted(ts)[len(ts)//2], col.hip.nnz))
