#!/usr/bin/env python
"""Developer tool: the fuzzed parity test (tests/test_fuzz_parity.py) over a
range of seeds beyond the committed ones.

    python tools/fuzz_soak.py dag 36 200      # DAG interpreter vs oracle (CPU)
    python tools/fuzz_soak.py build 36 200    # prebuild code objects (CPU)
    python tools/fuzz_soak.py hip 36 200      # HIP kernels vs oracle (GPU box)
"""
import os
import sys
import traceback

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np                                            # noqa: E402


def main():
    mode, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import opty_amd
    import random_problems as rp
    import test_fuzz_parity as tf
    bad = []
    if mode == 'build':
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(8) as pool:
            jobs = []
            for seed in range(lo, hi):
                try:
                    col = opty_amd.ConstraintCollocator(**rp.generate(seed))
                    jobs.append((seed, pool.submit(col._build_code_object)))
                    tf._reference(seed)         # the oracle's library
                except Exception:
                    bad.append(seed)
                    traceback.print_exc()
            for seed, job in jobs:
                job.result()
        print('built', hi - lo - len(bad), 'failed', bad)
        return
    for seed in range(lo, hi):
        try:
            if mode == 'dag':
                tf.test_expression_dag_against_the_oracle(seed)
            else:
                tf.test_hip_kernels_against_the_oracle(seed)
        except Exception as err:
            bad.append(seed)
            print('seed', seed, 'FAILED:', repr(err)[:300], flush=True)
    tf.test_zz_report()
    print('%s: seeds %d..%d, %d failed: %s' % (mode, lo, hi - 1, len(bad),
                                               bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
