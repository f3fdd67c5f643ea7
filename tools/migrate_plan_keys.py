#!/usr/bin/env python
"""One-off developer tool (CPU): re-keys ``opty_amd/launch_plans.json`` from
the r01-r04 key (sha of the module the printer emitted with default options)
to the printer-independent structural key (``launch_plan.problem_sha``).

Must run with the printer the plans were recorded with (the old key is
recomputed by emitting every known problem); entries whose old key matches no
known problem are stale leftovers of earlier printers and are dropped.
"""
import json
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import opty_amd                                               # noqa: E402
from opty_amd import launch_plan as lp                        # noqa: E402
from examples import problems                                 # noqa: E402


def main():
    path = lp.DEFAULT_FILE
    with open(path) as f:
        plans = json.load(f)
    wanted = {k.split(':')[0] for k in plans}
    names = sorted({v.get('problem', '').split(',')[0] for v in plans.values()})
    # a plan's "problem" names the workload it was tuned on; the full-size
    # configs share their module (and program) with the small twin
    twins = {'config3_10link': 'config3_10link_small',
             'config2_pendulum': 'config2_pendulum_small',
             'config5_standin_24link': 'config5_standin_24link_small',
             'config5_gaitlike_24link': 'config5_gaitlike_24link_small',
             'config5_one_legged': 'one_legged_small',
             'config5_biped': 'biped_small'}
    mapping = {}
    for name in names:
        small = twins.get(name, name)
        for kw in ({}, dict(prune_zeros=True), dict(jacobian_layout='csr')):
            col = opty_amd.ConstraintCollocator(**kw, **problems.build(small))
            prog = col._build_program()
            old = lp.emitted_sha(prog)
            if old in wanted:
                mapping[old] = lp.problem_sha(prog)
                print('%-34s %s %s -> %s' % (name, kw, old, mapping[old]),
                      flush=True)
    out = {}
    for key, entry in plans.items():
        sha, bucket, arch = key.split(':')
        if sha not in mapping:
            print('dropped (stale printer):', key, entry.get('problem'))
            continue
        entry = dict(entry, problem_sha=mapping[sha])
        out['%s:%s:%s' % (mapping[sha], bucket, arch)] = entry
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('%d of %d entries kept' % (len(out), len(plans)))


if __name__ == '__main__':
    main()
