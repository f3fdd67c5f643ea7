#!/usr/bin/env python
"""Developer tool (GPU box): host-side cost of one opty_hip_eval_con_jac call
through ctypes (enqueue rate), measured on a tiny problem whose kernels take
less time than the call."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import torch
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
col = opty_amd.ConstraintCollocator(**problems.build('config3_10link_small'))
hip = col.hip
dev = torch.device('cuda:0')
hip.use_torch_stream()
free = torch.from_numpy(problems.make_free(col.num_free)).to(dev)
con = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
jac = torch.empty(hip.nnz, dtype=torch.float64, device=dev)
for fn, label in ((lambda: hip.eval_con_jac(free, con, jac, hb.DEVICE), 'eval_con_jac(tensor args)'),
                  (lambda f=free.data_ptr(), c=con.data_ptr(), j=jac.data_ptr(): hip.eval_con_jac(f, c, j, hb.DEVICE), 'eval_con_jac(int args)')):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-28s enqueue %.2f us/call, incl. drain %.2f us/call' % (label, (t1-t0)/5000*1e6, (t2-t0)/5000*1e6))
