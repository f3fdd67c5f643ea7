#!/usr/bin/env python
"""Same-box wall time of the REAL reference (``/root/reference/opty``, compiled
Cython + OpenMP path, stub ``cyipopt``) next to the oracle's C/OpenMP port on
BASELINE config 3 (10-link pendulum, N = 100 000, backward Euler): the check
that lets ``bench.py`` report the port as ``cpu_baseline`` (``kind: "port"``) on
the GPU box, where the reference cannot travel.  Run from the repo root in the
build container::

    OMP_PROC_BIND=close OMP_PLACES=cores python tests/golden/_gen/time_reference.py

Prints min / median of the constraint + Jacobian pair for both, and the
largest per-entry difference of their values.
"""
import os
import sys
import time

os.environ.setdefault('OMP_PROC_BIND', 'close')
os.environ.setdefault('OMP_PLACES', 'cores')

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..', '..'))
sys.path.insert(0, os.path.join(HERE, 'stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
from opty.direct_collocation import ConstraintCollocator     # noqa: E402
from examples import problems                                 # noqa: E402
from oracle.collocation_oracle import OracleCollocator        # noqa: E402


def pair_times(con, jac, frees, reps=20):
    for k in range(3):
        con(frees[k]), jac(frees[k])
    ts = []
    for k in range(reps):
        f = frees[k % len(frees)]
        t0 = time.perf_counter()
        con(f)
        jac(f)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e3*ts[0], 1e3*ts[len(ts)//2]


def main():
    kw = problems.build('config3_10link')
    ref = ConstraintCollocator(parallel=True, **kw)
    rcon, rjac = (ref.generate_constraint_function(),
                  ref.generate_jacobian_function())
    orc = OracleCollocator(name='config3_10link', parallel=True, **kw)
    ocon, ojac = (orc.generate_constraint_function(),
                  orc.generate_jacobian_function())
    frees = [problems.make_free(ref.num_free, seed=s) for s in range(3)]
    print('cpus: %d, OMP_PROC_BIND=%s' % (os.cpu_count(),
                                          os.environ['OMP_PROC_BIND']))
    for label, c, j in (('reference (cython+openmp)', rcon, rjac),
                        ('oracle port (gcc -O2 -fopenmp)', ocon, ojac)):
        lo, med = pair_times(c, j, frees)
        print('%-32s pair min %.1f ms  median %.1f ms' % (label, lo, med))
    dj = np.abs(np.asarray(rjac(frees[0])) - np.asarray(ojac(frees[0])))
    dc = np.abs(rcon(frees[0]) - ocon(frees[0]))
    print('largest |reference - port|: jac %.3g, con %.3g' % (dj.max(),
                                                             dc.max()))


if __name__ == '__main__':
    main()
