#!/usr/bin/env python
"""Developer tool (GPU box): evaluate one workload with code objects built at
different hipcc optimisation levels and report the largest disagreement.

The generated kernels are thousands of lines of straight-line float64 code at
the register allocator's limit; hipcc 7.2 -O3 was caught miscompiling one of
them (DESIGN.md 4.6 / hip_backend.compile_module).  -O1 and -O2 going through
different pipelines and agreeing to rounding is cheap evidence that a new
system's kernels are sound when no reference fixture exists for it.

    python tools/check_opt_levels.py [workload] [layout] [-O1 -O2 -O3 ...]
"""
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
import opty_amd                                               # noqa: E402
from examples import problems                                 # noqa: E402


def main():
    args = sys.argv[1:]
    workload = args.pop(0) if args and not args[0].startswith('-') \
        else 'config3_10link_small'
    layout = args.pop(0) if args and not args[0].startswith('-') else 'coo'
    levels = args or ['-O1', '-O2']
    results = {}
    for lvl in levels:
        os.environ['OPTY_HIPCC_OPT'] = lvl
        col = opty_amd.ConstraintCollocator(jacobian_layout=layout,
                                            **problems.build(workload))
        free = problems.make_free(col.num_free, seed=3,
                                  variable_duration=col._variable_duration)
        results[lvl] = (col.generate_constraint_function()(free).copy(),
                        col.generate_jacobian_function()(free).copy())
        col.hip.close()
    base = levels[0]
    bad = False
    for lvl in levels[1:]:
        for what, a, b in zip(('con', 'jac'), results[base], results[lvl]):
            scale = max(np.abs(a).max(), 1e-300)
            err = np.abs(a - b).max()/scale
            print('%s vs %s  %s: max |diff| / max |value| = %.3g'
                  % (base, lvl, what, err))
            bad |= not err < 1e-12
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
