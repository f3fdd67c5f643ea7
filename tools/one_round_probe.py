#!/usr/bin/env python
"""Developer tool (GPU box): fused launch time of the 10-link kernels with 4
strips (5 waves per 64-node block) as a function of the number of node blocks,
around the 1024-resident-wave boundary."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import torch
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
from opty_amd.codegen.emit_hip import EmitOptions
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4
factory, fkw = problems.CONFIGS['config3_10link']
col = opty_amd.ConstraintCollocator(emit_options=EmitOptions(groups=groups), **factory(**dict(fkw, num_nodes=20001)))
hip = col.hip
dev = torch.device('cuda:0')
hip.use_torch_stream()
prog = col._build_program()
M, P, ncn = prog.M, prog.P, 20000
free = torch.from_numpy(problems.make_free(col.num_free)).to(dev)
con = torch.empty(M*ncn, dtype=torch.float64, device=dev)
jac = torch.empty(P*ncn, dtype=torch.float64, device=dev)
t0 = time.time()
while time.time() - t0 < 0.3:
    hip.time_eval_shard(hb.EVAL_FUSED, free, con, ncn, jac, 0, 12500, 50)
for rnd in range(2):
    for blocks in (100, 150, 180, 190, 196, 200, 204, 205, 206, 210, 220, 250, 300):
        n = blocks*64
        ts = [hip.time_eval_shard(hb.EVAL_FUSED, free, con, ncn, jac, 0, n, 200) for _ in range(5)]
        print('groups %d  %4d blocks  %5d waves  fused med %.4f ms  (%.2f us per 100 blocks)' % (
            groups, blocks, blocks*(groups + 1), float(np.median(ts)), 1e3*float(np.median(ts))/blocks*100), flush=True)
    print()
