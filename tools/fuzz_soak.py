#!/usr/bin/env python
"""Developer tool: the fuzzed parity test (tests/test_fuzz_parity.py) over a
range of seeds beyond the committed ones.

    python tools/fuzz_soak.py dag 36 200      # DAG interpreter vs oracle (CPU)
    python tools/fuzz_soak.py build 36 200    # prebuild code objects (CPU)
    python tools/fuzz_soak.py hip 36 200      # HIP kernels vs oracle (GPU box)
    python tools/fuzz_soak.py layouts 0 100   # CSR / pruned layouts (GPU box)
    python tools/fuzz_soak.py sharded 0 100   # 2 and 3 node shards (GPU box)
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
    # the varying-first order (opt-in): the same triplet SET, twice (the
    # second call moves only the varying segment)
    col = opty_amd.ConstraintCollocator(jacobian_layout='varying_first',
                                        **kw)
    r, c = col.jacobian_indices()
    o_ref, o_got = np.lexsort((cols, rows)), np.lexsort((c, r))
    np.testing.assert_array_equal(rows[o_ref], r[o_got])
    np.testing.assert_array_equal(cols[o_ref], c[o_got])
    jf = col.generate_jacobian_function()
    for _ in range(2):
        jac = np.array(jf(free))
        assert np.abs(jac[o_got] - j_ref[o_ref]).max() <= 1e-10*scale, \
            (seed, 'varying_first')
    col.hip.close()


def sharded(seed, tf):
    """One random problem node-sharded over 2 and 3 ranks (all in this
    process, on one GPU): the shards' values and index slices, the instance
    tails and the tail indices, assembled, against the oracle."""
    import torch
    from opty_amd.sharded import ShardedCollocator
    kw, orc, free, c_ref, j_ref, rows, cols = tf._reference(seed)
    N1, M = orc.N - 1, orc.M
    for world in (2, 3):
        if N1 < world:
            continue
        con = np.full(len(c_ref), np.nan)
        jac = np.full(len(j_ref), np.nan)
        r_all = np.full(len(rows), -1, dtype=np.int64)
        c_all = np.full(len(cols), -1, dtype=np.int64)
        con2d = con[:M*N1].reshape(M, N1)
        for rank in range(world):
            sh = ShardedCollocator(rank=rank, world_size=world, **kw)
            f = torch.from_numpy(free).cuda()
            c, j = sh.evaluate(f)
            torch.cuda.synchronize()
            con2d[:, sh.a:sh.b] = c.cpu().numpy()
            jac[sh.a*sh.P:sh.b*sh.P] = j.cpu().numpy()
            r, cc = sh.jacobian_indices_local()
            r_all[sh.a*sh.P:sh.b*sh.P] = r
            c_all[sh.a*sh.P:sh.b*sh.P] = cc
            if rank == world - 1 and sh.o:
                ct, jt = sh.evaluate_instance()
                torch.cuda.synchronize()
                con[M*N1:] = ct.cpu().numpy()
                jac[N1*sh.P:] = jt.cpu().numpy()
                ri, ci = sh.instance_indices()
                r_all[N1*sh.P:] = ri
                c_all[N1*sh.P:] = ci
            sh.collocator.hip.close()
        tf._check('sharded', seed, orc, con, jac, c_ref, j_ref)
        np.testing.assert_array_equal(r_all, rows)
        np.testing.assert_array_equal(c_all, cols)


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
            elif mode == 'sharded':
                sharded(seed, tf)
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
