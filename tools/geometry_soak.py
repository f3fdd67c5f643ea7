#!/usr/bin/env python
"""Developer tool: random printer geometries (strip counts, chunk widths,
workgroup widths, flush kinds, constraint-wave cuts, dispatch order) of a few
systems against the default build of the same system -- every geometry must
produce the same values (to rounding: the expressions are the same, only
their grouping into waves changes).

    python tools/geometry_soak.py build 40     # CPU container: compile
    python tools/geometry_soak.py run 40       # GPU box
"""
import os
import sys
import random

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from opty_amd.codegen.emit_hip import EmitOptions             # noqa: E402
from examples import problems                                 # noqa: E402

SYSTEMS = [('config3_10link_small', 1003), ('chaplygin_be_small', 777),
           ('gaitlike_3link_mid_small', 515), ('odd_block_mid_small', 1029),
           ('elementary_be_small', 333), ('one_legged_small', 457),
           ('piecewise_be_small', 301)]


def variants(count):
    rng = random.Random(5)
    out = []
    for k in range(count):
        name, nodes = SYSTEMS[k % len(SYSTEMS)]
        kw = dict(chunk=rng.choice([16, 32, 32, 64]),
                  waves=rng.choice([None, None, 1, 2, 4]),
                  dear_first=rng.choice([0, 1]),
                  small_flush=rng.choice(['flat', 'chunk']),
                  con_split=rng.choice(['work', 'count']),
                  fold_instance=rng.choice([None, 0, 1]),
                  inline_uniform=rng.choice([None, 0, 1]),
                  con_attach=rng.choice([None, None, 0, 1]),
                  cut=rng.choice(['even', 'even', 'work']))
        if rng.random() < 0.7:
            kw['groups'] = rng.randint(1, 12)
            if rng.random() < 0.6:
                kw['fused_groups'] = rng.randint(1, 12)
        if rng.random() < 0.3:
            kw['con_rows_per_wave'] = rng.randint(1, 8)
        out.append((name, nodes, kw))
    return out


#: OPTY_SOAK_LAYOUT=csr / csr+prune / prune: the opt-in output layouts
LAYOUT = os.environ.get('OPTY_SOAK_LAYOUT', 'coo')


def collocator(name, nodes, kw):
    factory, fkw = problems.CONFIGS[name]
    pkw = factory(**dict(fkw, num_nodes=nodes))
    opts = EmitOptions(**kw) if kw is not None else None
    return opty_amd.ConstraintCollocator(
        emit_options=opts,
        jacobian_layout='csr' if LAYOUT.startswith('csr') else 'coo',
        prune_zeros=LAYOUT.endswith('prune'), **pkw)


def main():
    mode, count = sys.argv[1], int(sys.argv[2])
    todo = variants(count)
    if mode == 'build':
        from concurrent.futures import ThreadPoolExecutor
        cols = [collocator(n, N, None) for n, N in SYSTEMS]
        skipped = 0
        with ThreadPoolExecutor(8) as pool:
            jobs = [pool.submit(c._build_code_object) for c in cols]
            for name, nodes, kw in todo:
                try:
                    c = collocator(name, nodes, kw)
                    c.generate_source()
                except AssertionError:
                    skipped += 1        # a combination the printer rejects
                    continue
                jobs.append(pool.submit(c._build_code_object))
            for j in jobs:
                j.result()
        print('built', len(jobs), 'skipped', skipped)
        return
    ref = {}
    bad = 0
    done = 0
    spilled = refused = 0
    for name, nodes, kw in todo:
        if name not in ref:
            col = collocator(name, nodes, None)
            free = problems.make_free(col.num_free, seed=9,
                                      variable_duration=col._variable_duration)
            con = np.empty(col.num_constraints)
            jac = np.empty(col.hip.nnz)
            col.hip.eval_con_jac(free, con, jac, hb.HOST)
            c2 = col.generate_constraint_function()(free)
            j2 = np.array(col.generate_jacobian_function()(free))
            assert np.array_equal(c2, con) or np.allclose(c2, con, rtol=1e-12)
            ref[name] = (free, con, jac, j2)
        free, con, jac, j2 = ref[name]
        try:
            col = collocator(name, nodes, kw)
            col.generate_source()
        except AssertionError:
            continue
        # hand-set geometries are built as asked, spills and all; builds
        # that spill vector registers are the known-bad class
        # and are not part of this soak, and a build the verification refuses
        # is counted, not compared
        hsaco, _ = col._build_code_object()
        if hb.vgpr_spills(hsaco):
            spilled += 1
            continue
        try:
            col.hip
        except hb.HipBackendError as err:
            refused += 1
            print('REFUSED', name, kw, str(err)[:200], flush=True)
            continue
        for what in ('fused', 'separate'):
            c = np.empty_like(con)
            j = np.empty_like(jac)
            if what == 'fused':
                col.hip.eval_con_jac(free, c, j, hb.HOST)
            else:
                col.hip.eval_con(free, c, hb.HOST)
                col.hip.eval_jac(free, j, hb.HOST)
            scale_c = np.abs(con).max()
            scale_j = np.abs(jac).max()
            ec = np.abs(c - con).max()/scale_c
            ej = np.abs(j - jac).max()/scale_j
            if not (ec <= 1e-12 and ej <= 1e-12):
                bad += 1
                print('MISMATCH', name, kw, what, ec, ej, flush=True)
        col.hip.close()
        done += 1
    print('geometry soak (%s layout): %d variants of %d systems, %d '
          'mismatches; %d more spill vector registers (not run), %d refused '
          'by the build verification' % (LAYOUT, done, len(ref), bad, spilled,
                                         refused))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
