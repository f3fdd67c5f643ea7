#!/usr/bin/env python
"""Developer tool (GPU box): per-wave timeline of one launch of a generated
kernel (``EmitOptions(trace=1)``: every wave records wall-clock start / end,
shader cycles, its strip and the SIMD it ran on behind the Jacobian values).

    python tools/wave_timeline.py <workload> <fused|jac> "<emit spec>" ...

Prints, per strip class: waves, duration (us: median / p90 / max), first and
last start, last end; and how many waves of each class were running at a few
points in time -- what a launch is waiting for at its end.
"""
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402
from opty_amd.codegen.emit_hip import TRACE_OFFSET            # noqa: E402
from tune_jac import parse                                    # noqa: E402

TICK_US = 0.01          # wall_clock64: 100 MHz


def main():
    workload, what = sys.argv[1], sys.argv[2]
    os.environ.setdefault('OPTY_CROSS_CHECK', 'off')
    factory, fkw = problems.CONFIGS[workload]
    if os.environ.get('OPTY_TUNE_NODES'):
        fkw = dict(fkw, num_nodes=int(os.environ['OPTY_TUNE_NODES']))
    kw = factory(**fkw)
    dev = torch.device('cuda:0') if torch.cuda.is_available() else None
    for spec in sys.argv[3:]:
        if spec == 'auto':
            # the geometry a collocator picks by itself (launch plan or the
            # printer's rules), traced
            import copy
            probe = opty_amd.ConstraintCollocator(**kw)
            opts = copy.copy(probe._printer_options())
            opts.trace = 1
        else:
            opts = parse(spec + ',trace=1')
        col = opty_amd.ConstraintCollocator(emit_options=opts, **kw)
        if dev is None:
            hsaco, meta = col._build_code_object()
            print(spec, 'built', hb.vgpr_spills(hsaco), flush=True)
            continue
        hip = col.hip
        hip.use_torch_stream()
        key = 'fused' if what == 'fused' else 'jac'
        d = hip.desc
        wgs = d['%s_wgs_per_block' % key]
        W = d['%s_waves_per_wg' % key]
        ncn = col.num_collocation_nodes - 1
        nblk8 = ((ncn + 63)//64 + 7)//8*8
        nwaves = (nblk8*wgs + 1)*W
        free = torch.from_numpy(problems.make_free(
            col.num_free, variable_duration=col._variable_duration)).to(dev)
        con = torch.empty(col.num_constraints, dtype=torch.float64,
                          device=dev)
        jac = torch.zeros(hip.nnz + TRACE_OFFSET + 4*nwaves + 64,
                          dtype=torch.float64, device=dev)
        sel = hb.EVAL_FUSED if what == 'fused' else hb.EVAL_JAC
        for _ in range(3):
            ms = hip.time_eval(sel, free, con, jac, 50)
        jac[hip.nnz:].zero_()
        torch.cuda.synchronize()
        hip.time_eval(sel, free, con, jac, 1)
        torch.cuda.synchronize()
        P = d['P']
        rec = jac[ncn*P + TRACE_OFFSET:ncn*P + TRACE_OFFSET + 4*nwaves] \
            .view(torch.int64).cpu().numpy().reshape(-1, 4)
        rec = rec[rec[:, 1] != 0]
        t0 = rec[:, 0].min()
        start = (rec[:, 0] - t0)*TICK_US
        end = (rec[:, 1] - t0)*TICK_US
        grp = rec[:, 2] >> 40
        cyc = rec[:, 3] >> 24
        hw = rec[:, 3] & 0xffffff
        print('%s  %s: %.4f ms by events; %d waves traced, span %.1f us, '
              '%d distinct SIMDs' % (spec, what, ms, len(rec), end.max(),
                                     len(np.unique(hw & 0xf7f3f))), flush=True)
        for g in np.unique(grp):
            m = grp == g
            dur = end[m] - start[m]
            ghz = cyc[m]/np.maximum(dur, 0.01)/1e3
            print('  strip %2d: %4d waves  dur med %5.1f p90 %5.1f max %5.1f us'
                  '  start first %5.1f med %5.1f last %5.1f  end last %5.1f'
                  '  clock %.2f GHz' % (g, m.sum(), np.median(dur),
                                        np.percentile(dur, 90), dur.max(),
                                        start[m].min(), np.median(start[m]),
                                        start[m].max(), end[m].max(),
                                        np.median(ghz)))
        # what a SIMD does between two waves: the gap from one wave's end to
        # the start of the next wave on the same SIMD
        simd = hw & 0xf7f3f
        gaps, busy, last = [], [], []
        for k in np.unique(simd):
            m = np.flatnonzero(simd == k)
            m = m[np.argsort(start[m])]
            gaps.extend(start[m][1:] - end[m][:-1])
            busy.append((end[m] - start[m]).sum())
            last.append(end[m].max())
        gaps = np.array(gaps)
        print('  per SIMD: busy med %.1f us of %.1f (%.0f %%), waves %.1f, gap '
              'between waves med %.2f p90 %.2f max %.2f us (sum %.1f us per '
              'SIMD), idle after the last wave med %.1f us'
              % (np.median(busy), end.max(), 100*np.mean(busy)/end.max(),
                 len(rec)/len(busy), np.median(gaps), np.percentile(gaps, 90),
                 gaps.max(), gaps.sum()/len(busy),
                 np.median(end.max() - np.array(last))), flush=True)
        ts = np.linspace(0, end.max(), 9)[1:-1]
        for t in ts:
            run = (start <= t) & (end > t)
            print('  t=%5.1f us: running %4d  by strip %s' % (
                t, run.sum(), dict(zip(*np.unique(grp[run],
                                                  return_counts=True)))))


if __name__ == '__main__':
    main()
