#!/usr/bin/env python
"""Developer tool (CPU container): cross-compile the code objects of one
workload for a list of printer options, so that a later GPU run of
tools/tune_jac.py finds them in the cache.  Usage:

    python tools/precompile.py [workload] "groups=4" "chunk=16,groups=5" ...
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb              # noqa: E402
from examples import problems
from opty_amd.codegen.emit_hip import EmitOptions             # noqa: E402
from tune_jac import parse                                    # noqa: E402


def main():
    args = sys.argv[1:]
    workload = 'config3_10link'
    if args and '=' not in args[0] and args[0] != 'default':
        workload = args.pop(0)
    kw = problems.build(workload.replace('config3_10link', 'config3_10link_small')
                        if workload == 'config3_10link' else workload)
    if os.environ.get('OPTY_TUNE_LINKS'):
        factory, fkw = problems.CONFIGS['config3_10link_small']
        kw = factory(**dict(fkw, num_links=int(os.environ['OPTY_TUNE_LINKS'])))
    layout = os.environ.get('OPTY_TUNE_LAYOUT', 'coo')
    prune = os.environ.get('OPTY_TUNE_PRUNE') == '1'
    with ThreadPoolExecutor(8) as pool:
        jobs = []
        for spec in args or ['default']:
            opts = EmitOptions() if spec == 'default' else parse(spec)
            col = opty_amd.ConstraintCollocator(
                emit_options=opts, prune_zeros=prune, jacobian_layout=layout,
                **kw)
            source, meta = col.generate_source()
            jobs.append((spec, meta, pool.submit(hb.compile_module, source)))
        for spec, meta, job in jobs:
            hsaco = job.result()
            res = hb.cached_kernel_resources(hsaco)
            print(spec, os.path.basename(hsaco),
                  {k: (v['groups'], v['waves_per_wg'], v['lds_bytes'])
                   for k, v in meta['kernels'].items()
                   if 'waves_per_wg' in v},
                  'vgpr/spill/sgpr-spill', {
                      k: (r['.vgpr_count'], r['.vgpr_spill_count'],
                          r['.sgpr_spill_count']) for k, r in res.items()
                      if k in ('opty_con', 'opty_jac', 'opty_conjac')},
                  flush=True)


if __name__ == '__main__':
    main()
