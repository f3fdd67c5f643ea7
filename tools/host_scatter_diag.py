#!/usr/bin/env python
"""Developer tool (GPU box): why is the default-layout host Jacobian slow on
some boxes?  Prints the NUMA topology, where the page-locked vector lives
(get_mempolicy at several offsets), the scatter pool's own trace over the first
calls (OPTY_HIP_TRACE: worker -> cpu map, lag behind the DMA, the placement
it settles on), then the median jacobian(free) time under explicit settings
(OPTY_HIP_HOST_NUMA = each node / off, 8 / 16 / 32 threads), each in a process
of its own."""
import glob
import os
import subprocess
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)


def child():
    import numpy as np
    import opty_amd
    from opty_amd import hip_backend as hb
    from examples import problems
    col = opty_amd.ConstraintCollocator(**problems.build('config3_10link'))
    jac = col.generate_jacobian_function()
    frees = [problems.make_free(col.num_free, seed=s) for s in range(3)]
    ts = []
    for k in range(int(os.environ.get('DIAG_CALLS', 14))):
        t0 = time.perf_counter()
        out = jac(frees[k % 3])
        ts.append(1e3*(time.perf_counter() - t0))
    n = len(out)
    where = [hb.host_numa_node(out[o:]) for o in
             (0, n//4, n//2, 3*n//4, n - 512)]
    tail = sorted(ts[len(ts)//2:])
    print('RESULT calls %s  median of the last half %.2f ms  vector on nodes '
          '%s  threads %d' % (' '.join('%.1f' % t for t in ts),
                              tail[len(tail)//2], where, hb.host_threads()),
          flush=True)


def main():
    if os.environ.get('DIAG_CHILD'):
        return child()
    for path in sorted(glob.glob('/sys/devices/system/node/node*/cpulist')):
        print(path.split('/')[-2], open(path).read().strip())
    print('affinity of this process: %d cpus' % len(os.sched_getaffinity(0)))
    try:
        print(subprocess.run(['rocm-smi', '--showtoponuma'],
                             capture_output=True, text=True).stdout[-600:])
    except OSError:
        pass
    runs = [('default + trace', {'OPTY_HIP_TRACE': '1'})]
    nodes = len(glob.glob('/sys/devices/system/node/node*/cpulist'))
    for node in range(nodes):
        runs.append(('fixed node %d' % node,
                     {'OPTY_HIP_HOST_NUMA': str(node),
                      'OPTY_HIP_HOST_PLACEMENT': 'fixed', 'DIAG_CALLS': '8'}))
    runs.append(('unplaced', {'OPTY_HIP_HOST_NUMA': 'off',
                              'OPTY_HIP_HOST_PLACEMENT': 'fixed',
                              'DIAG_CALLS': '8'}))
    for thr in (8, 32):
        runs.append(('default, %d threads' % thr,
                     {'OPTY_HIP_HOST_THREADS': str(thr), 'DIAG_CALLS': '12'}))
    for label, env in runs:
        proc = subprocess.run([sys.executable, os.path.abspath(__file__)],
                              env=dict(os.environ, DIAG_CHILD='1', **env),
                              capture_output=True, text=True)
        lines = [ln for ln in (proc.stdout + proc.stderr).splitlines()
                 if ln.startswith(('RESULT', 'opty_hip:'))]
        print('== %s' % label)
        for ln in lines[-40:] if 'trace' in label else lines[-1:]:
            print('   ' + ln[:400])


if __name__ == '__main__':
    main()
