#!/usr/bin/env python
"""Reproducer kept for the hipcc 7.2 ``-O3`` miscompile that made
``hip_backend.compile_module`` default to ``-O2`` (round 1): one generated
kernel -- the wave that evaluates equation 47 of the 24-link pendulum's
row-sorted (CSR) Jacobian, exactly as round 1's printer emitted it, all other
waves emptied -- plus ``opty_uni``; ``tools/o3_repro/*.hip.xz``.

Builds it at the given optimisation levels, evaluates the N = 6 problem of
``config5_standin_24link_small`` and compares the stored entries of equation
47 between the levels (and with the reference's golden values).  In round 1
two entries came out as ~1e16 at every node under ``-O3``.

    python tools/o3_repro.py [-O2 -O3 ...]          (GPU box)
"""
import json
import lzma
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
from opty_amd import hip_backend as hb              # noqa: E402
from examples import problems              # noqa: E402

HERE = os.path.join(REPO, 'tools', 'o3_repro')
NAME = 'o3_repro_24link_csr_row47'


def main():
    args = sys.argv[1:]
    full = None
    if args and args[0] == '--source':      # the whole module it came from
        full = args[1]
        args = args[2:]
    levels = args or ['-O2', '-O3']
    if full:
        with open(full) as f:
            source = f.read()
    else:
        with lzma.open(os.path.join(HERE, NAME + '.hip.xz'), 'rt') as f:
            source = f.read()
    with open(os.path.join(HERE, NAME + '.json')) as f:
        info = json.load(f)
    desc, row = info['desc'], info['row']
    kw = problems.build('config5_standin_24link_small')
    import opty_amd
    col = opty_amd.ConstraintCollocator(jacobian_layout='csr', **kw)
    prog = col._build_program()              # for the (unchanged) pattern
    free = problems.make_free(col.num_free, seed=0, variable_duration=True)
    z = np.load(os.path.join(REPO, 'tests', 'golden',
                             'config5_standin_24link_small.npz'))
    assert np.array_equal(free, z['free'])
    ncn = desc['N'] - 1
    S, E = info['row_start']
    L = E - S
    sel = [j*desc['C'] + k for j, k in prog.pattern][S:E]
    want = z['jac'].reshape(ncn, desc['M']*desc['C'])[:, sel]
    out = {}
    for lvl in levels:
        hsaco = hb.compile_module(source, opt_level=lvl)
        hip = hb.HipProblem(desc, hsaco)
        hip.set_known_parameters([float(kw['known_parameter_map'][p])
                                  for p in col.known_parameters])
        hip.set_block_pattern(prog.pattern)
        jac = np.full(hip.nnz, np.nan)
        hip.eval_jac(free, jac, hb.HOST)
        out[lvl] = jac[S*ncn:E*ncn].reshape(ncn, L).copy()
        err = np.abs(out[lvl] - want)
        print('%s: equation %d, %d entries x %d nodes: max |value| %.3g, '
              'max |value - reference| %.3g' % (
                  lvl, row, L, ncn, np.abs(out[lvl]).max(), err.max()))
        hip.close()
    bad = False
    for lvl in levels[1:]:
        d = np.abs(out[lvl] - out[levels[0]]).max()
        print('%s vs %s: max |diff| %.3g' % (levels[0], lvl, d))
        bad |= not d <= 1e-9*np.abs(want).max()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
