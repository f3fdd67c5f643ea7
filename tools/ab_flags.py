#!/usr/bin/env python
"""Developer tool (GPU box): A/B timing of hipcc switch sets (OPTY_HIPCC_FLAGS)
for one workload's default build, interleaved rounds.

    python tools/ab_flags.py config5_one_legged "" "-mllvm -amdgpu-sched-strategy=max-ilp" ...
"""
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402

workload = sys.argv[1]
kw = problems.build(workload)
cols = []
for flags in sys.argv[2:]:
    os.environ['OPTY_HIPCC_FLAGS'] = flags
    col = opty_amd.ConstraintCollocator(**kw)
    col.hip.use_torch_stream()
    cols.append((flags or '(default)', col))
os.environ.pop('OPTY_HIPCC_FLAGS', None)
dev = torch.device('cuda:0')
col = cols[0][1]
free = torch.from_numpy(problems.make_free(
    col.num_free, variable_duration=col._variable_duration)).to(dev)
con = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
jac = torch.empty(col.hip.nnz, dtype=torch.float64, device=dev)
for _ in range(20):
    col.hip.time_eval(hb.EVAL_FUSED, free, con, jac, 20)
res = {f: {'fused': [], 'jac': [], 'con': []} for f, _ in cols}
for _ in range(int(os.environ.get('OPTY_AB_ROUNDS', 5))):
    for f, c in cols:
        for what, label in ((hb.EVAL_FUSED, 'fused'), (hb.EVAL_JAC, 'jac'),
                            (hb.EVAL_CON, 'con')):
            res[f][label].append(c.hip.time_eval(what, free, con, jac, 100))
for f, _ in cols:
    print('%-52s ' % f + '  '.join('%s %.4f' % (k, float(np.median(v)))
                                   for k, v in res[f].items()), flush=True)
