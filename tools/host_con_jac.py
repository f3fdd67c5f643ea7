#!/usr/bin/env python
"""Developer tool (GPU box): wall time of ``constraints(free)`` and
``jacobian(free)`` (default layout) of BASELINE config 3 through the NumPy
callbacks, pageable and page-locked ``free``.  env: OPTY_HIP_HOST_WINDOWS,
OPTY_HIP_TRACE."""
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402

kw = problems.build('config3_10link')
col = opty_amd.ConstraintCollocator(**kw)
con, jac = col.generate_constraint_function(), col.generate_jacobian_function()
frees = [problems.make_free(col.num_free, seed=s) for s in range(3)]
pinned = []
for f in frees:
    x = hb.pinned_empty(len(f))
    x[:] = f
    pinned.append(x)
for label, vecs in (('pageable', frees), ('pinned', pinned)):
    for name, fn in (('con', con), ('jac', jac)):
        fn(vecs[0])
        fn(vecs[1])
        ts = []
        for k in range(11):
            t0 = time.perf_counter()
            fn(vecs[k % 3])
            ts.append(time.perf_counter() - t0)
        print('windows', os.environ.get('OPTY_HIP_HOST_WINDOWS', 'auto'),
              label, name, 'median ms %.3f min %.3f max %.3f' % (
                  1e3*sorted(ts)[5], 1e3*min(ts), 1e3*max(ts)), flush=True)
