#!/usr/bin/env python
"""Developer tool (GPU box): random node windows through opty_hip_eval_shard
(fused, separate, constraints only, Jacobian only; contiguous shard buffers
and strided in-place destinations) against the whole-problem evaluation of
the same handle -- the values of a node must not depend on the launch it is
evaluated in.  OPTY_SOAK_DETERMINISTIC=1: the collocators are built with
``deterministic=True`` and EVERY window -- fused or separate kernels, any
offset -- must equal ONE whole-problem evaluation bit for bit (and the fused
launch the separate ones).  Without a GPU the modules are only prebuilt."""
import os, sys, random
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import torch
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(3)
DET = os.environ.get('OPTY_SOAK_DETERMINISTIC') == '1'
dev = torch.device('cuda:0') if torch.cuda.is_available() else None
bad = total = rounding = 0
for name, nodes in (('config3_10link_small', 5003),
                    ('gaitlike_3link_be_small', 3001),
                    ('chaplygin_mid_small', 2500),
                    ('config2_pendulum_small', 4097),
                    # arithmetic-bound blocks: constraint rows ride in the
                    # Jacobian waves of the fused kernel (r04)
                    ('one_legged_small', 2051),
                    ('elementary_be_small', 3333)):
    factory, fkw = problems.CONFIGS[name]
    col = opty_amd.ConstraintCollocator(deterministic=DET, **factory(
        **dict(fkw, num_nodes=nodes)))
    if dev is None:
        print(name, os.path.basename(col.prebuild()[0]), flush=True)
        continue
    hip = col.hip
    hip.use_torch_stream()
    free_h = problems.make_free(col.num_free, seed=5,
                                variable_duration=col._variable_duration)
    col._sync_known(hip, free_h)
    free = torch.from_numpy(free_h).to(dev)
    ncn = nodes - 1
    M, P = col.num_eom, hip.desc['P']
    con = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
    jac = torch.empty(hip.nnz, dtype=torch.float64, device=dev)
    hip.eval_con_jac(free, con, jac, hb.DEVICE)
    torch.cuda.synchronize()
    # like with like: the fused kernel's windows against the fused kernel's
    # whole evaluation, the separate kernels' against theirs (blocks whose
    # constraint rows ride in the Jacobian waves of the fused kernel evaluate
    # them with other code than opty_con: a cancelling entry may differ by
    # more than a rounding between the two)
    fused_ref = (con[:M*ncn].view(M, ncn).clone(),
                 jac[:P*ncn].view(ncn, P).clone())
    hip.eval_con(free, con, hb.DEVICE)
    hip.eval_jac(free, jac, hb.DEVICE)
    torch.cuda.synchronize()
    separate_ref = (con[:M*ncn].view(M, ncn), jac[:P*ncn].view(ncn, P))
    if DET:
        # one set of values, whatever kernel produced them
        if not (torch.equal(fused_ref[0], separate_ref[0]) and
                torch.equal(fused_ref[1], separate_ref[1])):
            bad += 1
            print('MISMATCH', name, 'fused != separate', flush=True)
        fused_ref = separate_ref
    for k in range(count):
        a = rng.randrange(0, ncn)
        b = min(ncn, a + rng.choice([1, 2, 63, 64, 65, 127, 200,
                                     rng.randrange(1, ncn)]))
        what = rng.choice([hb.EVAL_FUSED, hb.EVAL_PAIR, hb.EVAL_CON,
                           hb.EVAL_JAC])
        in_place = rng.random() < 0.4
        if in_place:
            cbuf = torch.full((M, ncn), float('nan'), dtype=torch.float64,
                              device=dev)
            cs = ncn
            cview = cbuf[:, a:b]
            cptr = cview
        else:
            cbuf = torch.full((M, b - a), float('nan'), dtype=torch.float64,
                              device=dev)
            cs = b - a
            cview = cbuf
            cptr = cbuf
        jbuf = torch.full(((b - a)*P,), float('nan'), dtype=torch.float64,
                          device=dev)
        con2d, jac2d = fused_ref if what == hb.EVAL_FUSED else separate_ref
        hip.eval_shard(what, free,
                       cptr if what != hb.EVAL_JAC else None, cs,
                       jbuf if what != hb.EVAL_CON else None, a, b)
        torch.cuda.synchronize()
        ok = True
        if what != hb.EVAL_JAC:
            ok &= bool(torch.equal(cview, con2d[:, a:b]))
            if in_place:        # nothing outside the window was written
                mask = torch.ones(ncn, dtype=torch.bool, device=dev)
                mask[a:b] = False
                ok &= bool(torch.isnan(cbuf[:, mask]).all())
        if what != hb.EVAL_CON:
            ok &= bool(torch.equal(jbuf.view(b - a, P), jac2d[a:b]))
        total += 1
        if not ok:
            # strips own whole 128-byte lines of the destination, so which
            # wave evaluates the entries next to a strip boundary depends on
            # the destination's alignment: the same expression from another
            # wave's code may round differently -- never more than that
            worst, nans = 0.0, 0
            if what != hb.EVAL_JAC:
                d = (cview - con2d[:, a:b]).abs()
                nans += int(torch.isnan(cview).sum())
                worst = max(worst, float(torch.nan_to_num(d).max() /
                                         con2d[:, a:b].abs().max()))
            if what != hb.EVAL_CON:
                d = (jbuf.view(b - a, P) - jac2d[a:b]).abs()
                nans += int(torch.isnan(jbuf).sum())
                worst = max(worst, float(torch.nan_to_num(d).max() /
                                         jac2d[a:b].abs().max()))
            rounding += 1
            if nans or worst > 1e-13 or DET:
                bad += 1
                print('MISMATCH', name, a, b, what, in_place, 'NaN', nans,
                      'worst', worst, flush=True)
    hip.close()
if dev is None:
    sys.exit(0)
print('window soak%s: %d windows, %d bit-identical, %d equal to rounding '
      '(<= 1e-13 of the largest value, no unwritten value), %d mismatches'
      % (' (deterministic builds)' if DET else '', total, total - rounding,
         max(0, rounding - bad), bad))
sys.exit(1 if bad else 0)
