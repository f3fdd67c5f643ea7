#!/usr/bin/env python
"""Developer tool (GPU box): dense vs pruned Jacobian of config 3, device
kernel time and host-path (PCIe) time."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np, torch
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
for prune in (False, True):
    col = opty_amd.ConstraintCollocator(prune_zeros=prune, **problems.build('config3_10link'))
    hip = col.hip
    dev = torch.device('cuda:0')
    free = problems.make_free(col.num_free)
    ft = torch.from_numpy(free).to(dev)
    con = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
    jac = torch.empty(hip.nnz, dtype=torch.float64, device=dev)
    hip.time_eval(hb.EVAL_JAC, ft, con, jac, 3)
    ms = min(hip.time_eval(hb.EVAL_JAC, ft, con, jac, 30) for _ in range(3))
    fms = min(hip.time_eval(hb.EVAL_FUSED, ft, con, jac, 30) for _ in range(3))
    f = col.generate_jacobian_function(); f(free)
    ts = []
    for k in range(6):
        t0 = time.perf_counter(); f(free); ts.append(time.perf_counter() - t0)
    print('prune_zeros=%-5s nnz %9d  opty_jac %.4f ms (%.0f GB/s written)  opty_conjac %.4f ms  host jacobian(free) %.2f ms'
          % (prune, hip.nnz, ms, 8*hip.nnz/ms/1e6, fms, 1e3*min(ts)))
