#!/usr/bin/env python
"""Measures the launch plans of the benchmark configurations and of the parity
zoo (``opty_amd/launch_plan.py``).

    python tools/tune_plans.py --prebuild      # CPU container: hipcc all candidates
    python tools/tune_plans.py --tune [names]  # GPU box: time them, write the plans

``--tune`` writes ``opty_amd/launch_plans.json`` of the working copy and a copy
under ``gpurun_out/`` (what comes back from a GPU box)."""
import inspect
import os
import shutil
import sys
import time
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)

import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb, launch_plan          # noqa: E402
from opty_amd.codegen.emit_hip import EmitOptions, emit_module  # noqa: E402
from opty_amd.sharded import partition_nodes                  # noqa: E402
from examples import problems                                 # noqa: E402

#: (label, config to lower, collocation nodes, world sizes whose shard launch
#: is tuned as well)
ZOO_NODES = 400000
TARGETS = [
    ('config3_10link', 'config3_10link_small', 100000, (1, 2, 4, 8)),
    ('config5_standin_24link', 'config5_standin_24link_small', 50000, (1, 8)),
    ('config5_gaitlike_24link', 'config5_gaitlike_24link_small', 50000,
     (1, 8)),
    ('config2_pendulum', 'config2_pendulum_small', 10000, (1,)),
    ('config5_one_legged', 'one_legged_small', 50000, (1, 8)),
    ('config5_biped', 'biped_small', 50000, (1, 8)),
] + [(name, name, ZOO_NODES, (1,)) for name in (
    'pend3_link_midpoint_small', 'delay_be_small', 'odd_block_mid_small',
    'chaplygin_be_small', 'elementary_be_small',
    'pend2_link_vardur_unkmass_small', 'piecewise_be_small',
    'gaitlike_3link_be_small')]


def launches(names=None):
    for label, small, nodes, worlds in TARGETS:
        if names and label not in names:
            continue
        for w in worlds:
            yield label, small, nodes, w, max(
                b - a for a, b in partition_nodes(nodes - 1, w))


def collocator(small, nodes, launch_nodes, lower_small):
    factory, fkw = problems.CONFIGS[small]
    n = fkw.get('num_nodes', inspect.signature(
        factory).parameters['num_nodes'].default) if lower_small else nodes
    return opty_amd.ConstraintCollocator(
        launch_nodes=launch_nodes, **factory(**dict(fkw, num_nodes=n)))


def prebuild(names):
    jobs, seen = [], set()
    with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as pool:
        for label, small, nodes, w, ln in launches(names):
            col = collocator(small, nodes, ln, True)
            prog = col._build_program()
            cands, _ = launch_plan.candidates(prog, (ln + 63)//64)
            for tag, kw in cands:
                source, meta = emit_module(prog, EmitOptions(**kw),
                                           node_blocks=(ln + 63)//64)
                if meta['sha'] in seen:
                    continue
                seen.add(meta['sha'])
                def both(source=source, kw=kw, prog=prog, ln=ln, col=col):
                    # mirror launch_plan.tune: a candidate whose fused kernel
                    # spills vector registers is re-cut by count
                    h = col._compile(source)
                    if 'opty_conjac' in hb.vgpr_spills(h) and \
                            'con_split' not in kw:
                        src2, _ = emit_module(
                            prog, EmitOptions(**dict(kw, con_split='count')),
                            node_blocks=(ln + 63)//64)
                        h = col._compile(src2)
                    return h
                jobs.append((label, w, tag, pool.submit(both)))
        for label, w, tag, job in jobs:
            print(label, 'world', w, tag, os.path.basename(job.result()),
                  flush=True)


def tune(names):
    out = os.path.join(REPO, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    for label, small, nodes, w, ln in launches(names):
        t0 = time.time()
        col = collocator(small, nodes, ln, False)
        entry = col.tune_launch(
            log=lambda msg: print('   ', msg, flush=True))
        entry['problem'] = '%s, %d nodes per launch' % (label, ln)
        prog = col._build_program()
        launch_plan.record(launch_plan.key_of(prog, (ln + 63)//64), entry)
        print('%s world %d (%d nodes): seed %s -> %s  [%.0f s]' % (
            label, w, ln, entry['seed'], entry['options'], time.time() - t0),
            flush=True)
        shutil.copy(launch_plan.DEFAULT_FILE,
                    os.path.join(out, 'launch_plans.json'))


if __name__ == '__main__':
    args = sys.argv[1:]
    mode = args.pop(0) if args else '--prebuild'
    {'--prebuild': prebuild, '--tune': tune}[mode](args or None)
