#!/usr/bin/env python
"""Developer tool (GPU box): wall time of ``jacobian(free)`` of BASELINE config
3 in the ``varying_first`` layout, with pageable and page-locked ``free``
vectors.  env: OPTY_HIP_HOST_WINDOWS, OPTY_HIP_TRACE."""
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402

kw = problems.build('config3_10link')
col = opty_amd.ConstraintCollocator(jacobian_layout='varying_first', **kw)
jac = col.generate_jacobian_function()
frees = [problems.make_free(col.num_free, seed=s) for s in range(3)]
pinned = []
for f in frees:
    x = hb.pinned_empty(len(f))
    x[:] = f
    pinned.append(x)
for label, vecs in (('pageable', frees), ('pinned', pinned)):
    jac(vecs[0])
    jac(vecs[1])
    ts = []
    for k in range(9):
        t0 = time.perf_counter()
        jac(vecs[k % 3])
        ts.append(time.perf_counter() - t0)
    print('windows', os.environ.get('OPTY_HIP_HOST_WINDOWS', 'auto'), label,
          'median ms %.3f min %.3f' % (1e3*sorted(ts)[4], 1e3*min(ts)),
          flush=True)
