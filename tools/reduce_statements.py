#!/usr/bin/env python
"""Developer tool (GPU box): statement-level reduction of a reduced wrong
module (``tools/o3_repro/reduced_biped_csr_persistent_O2.hip.xz``: a fused
kernel whose wave of strip 0 never stores its row).  A compile-and-run step
takes about a second on this problem (8 nodes), so plain delta debugging is
affordable: definitions ``const double x = <expr>;`` (and ``sincos`` pairs)
are turned into constants chunk by chunk -- which keeps every candidate
compilable -- dead code is eliminated, and a candidate is kept when the row
is still not stored.

    python tools/reduce_statements.py [seconds [chunks to start with]]
Writes gpurun_out/reduced3_biped_csr_persistent_O2.hip / .txt after every
improvement."""
import json
import lzma
import os
import re
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'tools'))

import numpy as np                                            # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402
import reduce_miscompile as rm                                # noqa: E402

TAG = 'biped_csr_persistent_O2'
KERNEL = 'opty_conjac'


def main():
    limit = float(sys.argv[1]) if len(sys.argv) > 1 else 1200.0
    t00 = time.time()
    src = os.path.join(REPO, 'tools', 'o3_repro', 'reduced_%s.hip.xz' % TAG)
    with lzma.open(src, 'rt') as f:
        lines = f.read().splitlines()
    info = json.load(open(os.path.join(REPO, 'tools', 'o3_repro',
                                       TAG + '.json')))
    col = opty_amd.ConstraintCollocator(**info['collocator_kwargs'],
                                        **problems.build(info['problem']))
    meta = info['meta']
    N, free = col._verification_inputs()
    ncn = N - 1
    rs = list(col._build_program().row_start)
    tmp = os.path.join(REPO, 'gpurun_out', 'reduce_cache')
    os.makedirs(tmp, exist_ok=True)
    row = 11
    S, L = rs[row], rs[row + 1] - rs[row]
    steps = [0]
    desc = dict(col._descriptor(meta), N=N, num_inst=0, nnz_inst=0,
                num_inst_atoms=0, inst_folded=0)
    par = np.array([float(col.known_parameter_map[p])
                    for p in col.known_parameters])

    def unstored(text):
        steps[0] += 1
        hsaco = hb.compile_module(text, tmp, opt_level=info['opt_level'],
                                  extra_flags=tuple(info['extra_flags']))
        h = hb.HipProblem(desc, hsaco)
        try:
            if not col._variable_duration:
                h.set_interval(col.node_time_interval)
            if len(par):
                h.set_known_parameters(par)
            h.set_block_pattern(col._program.pattern)
            d = hb.DeviceVector(free)
            dj = hb.DeviceVector(np.full(h.nnz, np.nan))
            dc = hb.DeviceVector(np.full(col.num_eom*ncn, np.nan))
            hb.poison_registers(0)
            h.eval_con_jac(d, dc, dj, hb.DEVICE)
            h.synchronize()
            jac = dj.numpy()
            for v in (d, dj, dc):
                v.close()
        finally:
            h.close()
        os.remove(hsaco)
        return int(np.isnan(jac[S*ncn:(S + L)*ncn]).sum())

    ks = rm.kernels(lines)
    a, b = ks[KERNEL]
    head, body, rest = lines[:a], lines[a:b], lines[b:]

    def assemble(body):
        return '\n'.join(head + body + rest) + '\n'

    n0 = unstored(assemble(body))
    print('start: %d lines, %d values never stored' % (len(lines), n0),
          flush=True)
    assert n0 > 0, 'this box builds it right'

    def candidates(body):
        return [k for k, ln in enumerate(body)
                if (rm.DEF.match(ln) and not re.match(
                    r'^\s*const double \w+ = -?[0-9.]+;\s*$', ln))
                or rm.PAIR.match(ln)]

    def neutralise(body, which):
        out = []
        for k, ln in enumerate(body):
            if k in which:
                m = rm.DEF.match(ln)
                if m:
                    ln = re.sub(r'= .*;', '= 0.5;', ln, count=1)
                else:
                    m = rm.PAIR.match(ln)
                    pad = ln[:len(ln) - len(ln.lstrip())]
                    ln = '%sdouble %s = 0.5, %s = 0.25;' % (
                        pad, m.group(1), m.group(2))
            out.append(ln)
        return rm.dce(out)

    out = os.path.join(REPO, 'gpurun_out', 'reduced3_%s.hip' % TAG)

    def save(body, n):
        text = assemble(body)
        with open(out, 'w') as f:
            f.write(text)
        with open(out[:-4] + '.txt', 'w') as f:
            f.write('%d lines after %d compile-and-run steps (%.0f s); %d '
                    'values of row %d never stored\n'
                    % (len(text.splitlines()), steps[0], time.time() - t00,
                       n, row))

    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    while time.time() - t00 < limit:
        cand = candidates(body)
        if not cand:
            break
        chunks = min(chunks, len(cand))
        size = -(-len(cand)//chunks)
        progress = False
        k = 0
        while k < len(cand) and time.time() - t00 < limit:
            which = set(cand[k:k + size])
            trial = neutralise(body, which)
            try:
                n = unstored(assemble(trial))
            except Exception as exc:            # noqa
                n = 0
                print('   (%s)' % str(exc)[:80], flush=True)
            if n > 0:
                body = trial
                progress = True
                save(body, n)
                print('   %d of %d definitions constant: %d lines, %d '
                      'values never stored [%d steps, %.0f s]'
                      % (len(which), len(cand), len(head + body + rest), n,
                         steps[0], time.time() - t00), flush=True)
                cand = candidates(body)
                # (same position: the list moved up)
            else:
                k += size
        if not progress:
            if size == 1:
                break
            chunks *= 2
    print('done: %d lines, %d steps, %.0f s' % (len(head + body + rest),
                                               steps[0], time.time() - t00))


if __name__ == '__main__':
    main()
