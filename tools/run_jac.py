#!/usr/bin/env python
"""Developer tool: launch the jac kernel of config3 a few times (for rocprofv3)."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import torch
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
from tune_jac import parse
from opty_amd.codegen.emit_hip import EmitOptions
spec = sys.argv[1] if len(sys.argv) > 1 else 'default'
what = {'jac': hb.EVAL_JAC, 'con': hb.EVAL_CON, 'fused': hb.EVAL_FUSED}[sys.argv[2] if len(sys.argv) > 2 else 'jac']
# 'default': what a collocator builds by itself (launch plan, spill-free cut)
opts = None if spec == 'default' else parse(spec)
workload = os.environ.get('OPTY_WORKLOAD', 'config3_10link')
col = opty_amd.ConstraintCollocator(
    emit_options=opts,
    specialize_parameters=os.environ.get('OPTY_SPECIALIZE') == '1',
    **problems.build(workload))
hip = col.hip
dev = torch.device('cuda:0')
hip.use_torch_stream()
free = torch.from_numpy(problems.make_free(col.num_free, variable_duration=col._variable_duration)).to(dev)
con = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
jac = torch.empty(hip.nnz, dtype=torch.float64, device=dev)
print(hip.time_eval(what, free, con, jac, 10))
