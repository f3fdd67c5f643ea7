#!/usr/bin/env python
"""Developer tool: the fuzzed parity test (tests/test_fuzz_parity.py) over a
range of seeds beyond the committed ones.

    python tools/fuzz_soak.py dag 36 200      # DAG interpreter vs oracle (CPU)
    python tools/fuzz_soak.py build 36 200    # prebuild code objects (CPU)
    python tools/fuzz_soak.py hip 36 200      # HIP kernels vs oracle (GPU box)
    python tools/fuzz_soak.py layouts 0 100   # CSR / pruned layouts (GPU box)
"""
import os
import sys
import traceback

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np                                            # noqa: E402


def layouts(seed, tf):
    """Row-sorted, pruned and row-sorted pruned layouts of one random problem
    (GPU) as sparse matrices against the oracle's triplets."""
    import scipy.sparse as sp
    import opty_amd
    kw, orc, free, c_ref, j_ref, rows, cols = tf._reference(seed)
    shape = (orc.num_constraints, orc.num_free)
    ref = sp.coo_matrix((j_ref, (rows, cols)), shape=shape).tocsr()
    scale = max(float(np.abs(j_ref).max()), 1.0)
    for layout, prune in (('csr', False), ('coo', True), ('csr', True)):
        col = opty_amd.ConstraintCollocator(jacobian_layout=layout,
                                            prune_zeros=prune, **kw)
        jac = np.array(col.generate_jacobian_function()(free))
        r, c = col.jacobian_indices()
        assert r.dtype == np.int64 and len(r) == len(jac)
        if layout == 'csr':
            row_ptr, col_idx = col.jacobian_csr_structure()
            got = sp.csr_matrix((jac, col_idx, row_ptr), shape=shape)
            assert np.all(np.diff(r) >= 0)
            np.testing.assert_array_equal(c, col_idx)
        else:
            got = sp.coo_matrix((jac, (r, c)), shape=shape).tocsr()
        if not prune:
            o = np.lexsort((cols, rows))
            np.testing.assert_array_equal(rows[o], r)
            np.testing.assert_array_equal(cols[o], c)
        diff = abs(got - ref)
        assert (diff.max() if diff.nnz else 0.0) <= 1e-10*scale, \
            (seed, layout, prune)
        con = col.generate_constraint_function()(free)
        assert np.abs(con - c_ref).max() <= 1e-10*max(
            1.0, float(np.abs(c_ref).max()))
        col.hip.close()


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
            elif mode == 'layouts':
                layouts(seed, tf)
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
