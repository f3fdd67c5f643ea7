#!/usr/bin/env python
"""Developer tool (GPU box): the NumPy callbacks (host buffers: latency path
for small problems, staged copies, the persistent varying-entries path for
large ones) against a device-pointer evaluation of the same handle, for the
zoo's systems at node counts on both sides of every size threshold."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import torch
import opty_amd
from opty_amd import hip_backend as hb
from opty_amd.codegen.program import varying_copies
from examples import problems

dev = torch.device('cuda:0')
bad = total = 0
for name in ('config3_10link_small', 'gaitlike_3link_be_small',
             'chaplygin_mid_small', 'config2_pendulum_small',
             'pend2_link_vardur_unkmass_small', 'odd_block_mid_small',
             'c99_be_small', 'piecewise_be_small'):
    for nodes in (37, 700, 9001, 60001, 300001):
        factory, fkw = problems.CONFIGS[name]
        col = opty_amd.ConstraintCollocator(**factory(**dict(
            fkw, num_nodes=nodes)))
        hip = col.hip
        if hip.nnz > 120e6:
            hip.close()
            continue
        cf, jf = col.generate_constraint_function(), \
            col.generate_jacobian_function()
        P, ncn = hip.desc['P'], nodes - 1
        unique, copies = varying_copies(col._build_program())
        if 2*len(unique) > 0.8*2*P or hip.nnz < col._PERSISTENT_MIN_NNZ:
            copies = []
        if 10*len(unique) > 8*P:
            copies = []
        for seed in (1, 2, 3):
            free = problems.make_free(col.num_free, seed=seed,
                                      variable_duration=col._variable_duration)
            c, j = cf(free), np.array(jf(free))
            d_free = torch.from_numpy(free).to(dev)
            dc = torch.empty(col.num_constraints, dtype=torch.float64,
                             device=dev)
            dj = torch.empty(hip.nnz, dtype=torch.float64, device=dev)
            hip.eval_con(d_free, dc, hb.DEVICE)
            hip.eval_jac(d_free, dj, hb.DEVICE)
            hip.synchronize()
            want_c, want_j = dc.cpu().numpy(), dj.cpu().numpy()
            blk = want_j[:P*ncn].reshape(ncn, P)
            for d, s in copies:
                blk[:, d] = blk[:, s]
            ok = np.array_equal(c, want_c) and np.array_equal(j, want_j)
            total += 1
            if not ok:
                bad += 1
                dj_ = np.abs(j - want_j)
                print('MISMATCH', name, nodes, seed, 'con',
                      np.abs(c - want_c).max(), 'jac', dj_.max(),
                      'at entries', sorted(set(np.nonzero(dj_)[0][:50] % P)),
                      flush=True)
        # the same triplets in the varying-first order (opt-in layout): as a
        # permutation of the default layout's vector
        vf = opty_amd.ConstraintCollocator(
            jacobian_layout='varying_first',
            **factory(**dict(fkw, num_nodes=nodes)))
        order, seg_len, source = vf.jacobian_segments()
        vjf = vf.generate_jacobian_function()
        for seed in (1, 2):
            free = problems.make_free(col.num_free, seed=seed,
                                      variable_duration=col._variable_duration)
            got = np.array(vjf(free))
            want = np.array(jf(free))
            blk = want[:P*ncn].reshape(ncn, P)
            re = np.empty_like(blk)
            at = 0
            for L in (int(x) for x in seg_len):
                re[:, order[at:at + L]] = got[at*ncn:(at + L)*ncn].reshape(
                    ncn, L)
                at += L
            total += 1
            scale = max(1.0, float(np.abs(want).max()))
            if not (np.abs(re - blk).max() <= 1e-11*scale and
                    np.array_equal(got[P*ncn:], want[P*ncn:])):
                bad += 1
                print('MISMATCH varying_first', name, nodes, seed,
                      np.abs(re - blk).max(), flush=True)
        vf.hip.close()
        hip.close()
        del cf, jf, col, vjf, vf
print('host path soak: %d evaluations, %d mismatches' % (total, bad))
sys.exit(1 if bad else 0)
