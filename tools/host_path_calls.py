#!/usr/bin/env python
"""Developer tool (GPU box): per-call times of the default-layout host
Jacobian of config 3 (what bench.py's ``host_path_ms.jac`` is the median of),
next to the scatter-free layout's, interleaved, with the scatter pool's own
trace (lag of the scatter behind the last DMA chunk) under OPTY_HIP_TRACE=1.

    [OPTY_HIP_TRACE=1] python tools/host_path_calls.py [calls]
"""
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    kw = problems.build('config3_10link')
    # HOST_CALLS_PRUNE=1: the pruned block (structural zeros dropped) instead
    a = opty_amd.ConstraintCollocator(
        prune_zeros=os.environ.get('HOST_CALLS_PRUNE') == '1', **kw)
    b = opty_amd.ConstraintCollocator(jacobian_layout='varying_first', **kw)
    ja, jb = a.generate_jacobian_function(), b.generate_jacobian_function()
    frees = [problems.make_free(a.num_free, seed=s) for s in range(3)]
    ta, tb = [], []
    for k in range(calls):
        for fn, ts in ((ja, ta), (jb, tb)):
            t0 = time.perf_counter()
            fn(frees[k % 3])
            ts.append(1e3*(time.perf_counter() - t0))
    print('load average', os.getloadavg(), 'cpus', os.cpu_count(),
          'threads', hb.host_threads(), 'placement', hb.host_placement())
    print('default       ', ' '.join('%.1f' % t for t in ta))
    print('varying_first ', ' '.join('%.1f' % t for t in tb))
    for name, ts in (('default', ta), ('varying_first', tb)):
        tail = np.array(ts[14:])
        print('%-14s after 14 calls: median %.2f  min %.2f  p90 %.2f ms'
              % (name, np.median(tail), tail.min(), np.percentile(tail, 90)))


if __name__ == '__main__':
    main()
