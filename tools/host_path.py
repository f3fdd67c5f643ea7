#!/usr/bin/env python
"""Developer tool (GPU box): wall time of the host (cyipopt-style) path,
NumPy in / NumPy out through PCIe, for config 3."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import opty_amd
from examples import problems
col = opty_amd.ConstraintCollocator(**problems.build('config3_10link'))
con = col.generate_constraint_function()
jac = col.generate_jacobian_function()
frees = [problems.make_free(col.num_free, seed=s) for s in range(3)]
con(frees[0]); jac(frees[0])
for name, f in (('constraints(free)', con), ('jacobian(free)', jac)):
    ts = []
    for k in range(10):
        t0 = time.perf_counter(); f(frees[k % 3]); ts.append(time.perf_counter() - t0)
    print('%-18s min %.2f ms  median %.2f ms' % (name, 1e3*min(ts), 1e3*sorted(ts)[len(ts)//2]))
nb = 8*col.hip.nnz
print('jacobian D2H-inclusive rate: %.1f GB/s' % (nb/min(ts)/1e9))
