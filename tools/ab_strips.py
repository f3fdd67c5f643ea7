#!/usr/bin/env python
"""Developer tool (GPU box): A/B/A/B timing of printer-option variants of one
workload in ONE process (shared buffers, interleaved rounds), so that clock
and thermal drift hit all variants alike.  Usage:

    python tools/ab_strips.py [workload] "groups=6" "groups=8" ...
env: OPTY_TUNE_NODES, OPTY_AB_ROUNDS (7), OPTY_AB_ITERS (100)
"""
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb              # noqa: E402
from examples import problems
from opty_amd.codegen.emit_hip import EmitOptions             # noqa: E402
from tune_jac import parse                                    # noqa: E402


def main():
    args = sys.argv[1:]
    workload = 'config3_10link'
    if args and '=' not in args[0] and args[0] not in ('default', 'auto'):
        workload = args.pop(0)
    dev = torch.device('cuda:0') if torch.cuda.is_available() else None
    factory, fkw = problems.CONFIGS[workload]
    if os.environ.get('OPTY_TUNE_NODES'):
        fkw = dict(fkw, num_nodes=int(os.environ['OPTY_TUNE_NODES']))
    if os.environ.get('OPTY_TUNE_LINKS'):
        fkw = dict(fkw, num_links=int(os.environ['OPTY_TUNE_LINKS']))
    kw = factory(**fkw)
    rounds = int(os.environ.get('OPTY_AB_ROUNDS', 7))
    iters = int(os.environ.get('OPTY_AB_ITERS', 100))
    cols = []
    for spec in args:
        # "auto+specialize" / "<options>,specialize=1": parameter-specialised
        # kernels (ConstraintCollocator(specialize_parameters=True))
        special = 'specialize' in spec
        determ = '+deterministic' in spec
        uniform = '+uniform_trig' in spec
        rcp = '+share_rcp' in spec
        bare = spec.replace('+specialize', '').replace(',specialize=1', '') \
            .replace('specialize=1', '').replace('+deterministic', '') \
            .replace('+uniform_trig', '').replace('+share_rcp', '')
        opts = None if bare in ('auto', '') else (
            EmitOptions() if bare == 'default' else parse(bare))
        if uniform:
            # the plan's (or the given) options with sincos behind a
            # wave-uniform test (EmitOptions.fast_trig = 2)
            import copy
            if opts is None:
                opts = opty_amd.ConstraintCollocator(
                    specialize_parameters=special, **kw)._printer_options()
            opts = copy.copy(opts)
            opts.fast_trig = 2
        if rcp:
            # ... with shared reciprocals (EmitOptions.share_rcp)
            import copy
            if opts is None:
                opts = opty_amd.ConstraintCollocator(
                    specialize_parameters=special, **kw)._printer_options()
            opts = copy.copy(opts)
            opts.share_rcp = 1
        col = opty_amd.ConstraintCollocator(
            emit_options=opts, specialize_parameters=special,
            deterministic=determ, **kw)
        if not torch.cuda.is_available():
            hsaco, meta = col._build_code_object()      # prebuild only
            print(spec, hb.vgpr_spills(hsaco), flush=True)
            continue
        try:
            col.hip.use_torch_stream()
        except hb.BuildRejected as err:
            print('%-28s REFUSED by the verification: %s'
                  % (spec, err.verdict['errors']), flush=True)
            continue
        cols.append((spec, col))
    if not cols:
        return
    col = cols[0][1]
    free = torch.from_numpy(problems.make_free(
        col.num_free, variable_duration=col._variable_duration)).to(dev)
    con = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
    jac = torch.empty(col.hip.nnz, dtype=torch.float64, device=dev)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        cols[0][1].hip.time_eval(hb.EVAL_FUSED, free, con, jac, 20)
    res = {spec: {'jac': [], 'con': [], 'fused': []} for spec, _ in cols}
    for _ in range(rounds):
        for spec, c in cols:
            for what, label in ((hb.EVAL_FUSED, 'fused'), (hb.EVAL_JAC, 'jac'),
                                (hb.EVAL_CON, 'con')):
                res[spec][label].append(
                    c.hip.time_eval(what, free, con, jac, iters))
    for spec, c in cols:
        d = c.hip.desc
        print('%-28s G=%-2d ' % (spec, d['jac_wgs_per_block'] *
                                 d['jac_waves_per_wg']) +
              '  '.join('%s med %.4f min %.4f' % (
                  k, float(np.median(v)), min(v))
                  for k, v in res[spec].items()), flush=True)


if __name__ == '__main__':
    main()
