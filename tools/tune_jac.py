#!/usr/bin/env python
"""Developer tool (GPU box): time the generated kernels of one workload for a
sweep of printer options.  Usage:

    python tools/tune_jac.py [workload] "chunk=32,groups=4" "chunk=16" ...
"""
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb              # noqa: E402
from examples import problems
from opty_amd.codegen.emit_hip import EmitOptions             # noqa: E402


#: OPTY_TUNE_LAYOUT=csr / OPTY_TUNE_PRUNE=1 select the opt-in output layouts
LAYOUT = os.environ.get('OPTY_TUNE_LAYOUT', 'coo')
PRUNE = os.environ.get('OPTY_TUNE_PRUNE') == '1'


def parse(spec):
    kw = {}
    for item in filter(None, spec.split(',')):
        k, v = item.split('=')
        kw[k] = None if v == 'None' else (
            v if k in ('ablate', 'con_split', 'small_flush', 'cut', 'order',
                     'strips', 'fused_strips', 'fused_order', 'class_cost',
                     'fused_class_cost')
            else int(v))
    return EmitOptions(**kw)


def main():
    args = sys.argv[1:]
    workload = 'config3_10link'
    if args and '=' not in args[0] and args[0] != 'default':
        workload = args.pop(0)
    specs = args or ['default']
    dev = torch.device('cuda:0')
    factory, fkw = problems.CONFIGS[workload]
    if os.environ.get('OPTY_TUNE_NODES'):
        fkw = dict(fkw, num_nodes=int(os.environ['OPTY_TUNE_NODES']))
    kw = factory(**fkw)
    iters = int(os.environ.get('OPTY_TUNE_ITERS', 30))
    for spec in specs:
        opts = EmitOptions() if spec == 'default' else parse(spec)
        t0 = time.time()
        col = opty_amd.ConstraintCollocator(
            emit_options=opts, prune_zeros=PRUNE, jacobian_layout=LAYOUT,
            **kw)
        hip = col.hip
        build_s = time.time() - t0
        hip.use_torch_stream()
        free = torch.from_numpy(problems.make_free(
            col.num_free, variable_duration=col._variable_duration)).to(dev)
        con = torch.empty(col.num_constraints, dtype=torch.float64,
                          device=dev)
        jac = torch.empty(hip.nnz, dtype=torch.float64, device=dev)
        # clock ramp: an idle GPU starts at its lowest clock
        t_ramp = time.time()
        while (time.time() - t_ramp)*1e3 < float(os.environ.get(
                'OPTY_TUNE_PREWARM_MS', 150)):
            hip.time_eval(hb.EVAL_FUSED, free, con, jac, 20)
        res = {}
        for what, label in ((hb.EVAL_JAC, 'jac'), (hb.EVAL_CON, 'con'),
                            (hb.EVAL_FUSED, 'fused')):
            hip.time_eval(what, free, con, jac, 3)
            res[label] = min(hip.time_eval(what, free, con, jac, iters)
                             for _ in range(3))
        prog = col._build_program()
        nb = 8.0*prog.P*(col.num_collocation_nodes - 1) + 8.0*col.num_free
        print('%-44s G=%-2d jac %.4f ms (%.0f GB/s, %.1f%% of 8TB/s) '
              'con %.4f fused %.4f  [build %.0fs]'
              % (spec, hip.desc['jac_wgs_per_block']*hip.desc['jac_waves_per_wg'], res['jac'],
                 nb/res['jac']/1e6, nb/res['jac']/1e6/80.0, res['con'],
                 res['fused'], build_s), flush=True)
        hip.close()


if __name__ == '__main__':
    main()
