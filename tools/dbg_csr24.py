#!/usr/bin/env python
"""Developer tool (GPU box): compare printer-option variants of the 24-link
row-sorted module entry by entry (full vector) and against the golden window."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import opty_amd
import golden_util as gu
from examples import problems
from opty_amd.codegen.emit_hip import EmitOptions
from tune_jac import parse

name = 'config5_standin_24link_small'
layout = sys.argv[1] if len(sys.argv) > 1 else 'csr'
specs = sys.argv[2:] or ['default', 'fast_trig=0']
N = 50000
meta, z = gu.load(name)
factory, fkw = problems.CONFIGS[name]
n_small, M, C = meta['N'], meta['M'], meta['C']
P = M*C
nrows = meta['n'] + meta['q']
small = z['free']
free = problems.make_free((nrows)*N + 1, seed=0)
for r in range(nrows):
    free[r*N:r*N + n_small] = small[r*n_small:(r + 1)*n_small]
free[nrows*N:] = small[nrows*n_small:]
out = {}
for spec in specs:
    opts = EmitOptions() if spec == 'default' else parse(spec)
    col = opty_amd.ConstraintCollocator(
        jacobian_layout=layout, emit_options=opts,
        **factory(**dict(fkw, num_nodes=N)))
    jac = np.array(col.generate_jacobian_function()(free))
    jac2 = np.array(col.generate_jacobian_function()(free))
    print(spec, 'repeatable:', np.array_equal(jac, jac2), 'nan', np.isnan(jac).sum())
    out[spec] = jac
    prog = col._build_program()
a, b = out[specs[0]], out[specs[1]]
rel = np.abs(a - b)/np.maximum(np.abs(a), 1e-300)
bad = np.nonzero(rel > 1e-9)[0]
print('variants differ (>1e-9 rel) at', len(bad), 'of', len(a))
if layout == 'csr':
    rs = np.array(prog.row_start)
    for k in bad[:40]:
        j = np.searchsorted(rs*(N - 1), k, side='right') - 1
        L = rs[j + 1] - rs[j]
        node, pos = divmod(k - rs[j]*(N - 1), L)
        print('  idx %d: eq %d node %d pos %d: %r vs %r' % (k, j, node, pos, a[k], b[k]))
else:
    for k in bad[:40]:
        node, e = divmod(k, P)
        print('  idx %d: node %d entry %d (eq %d col %d): %r vs %r' % (k, node, e, e//C, e % C, a[k], b[k]))
