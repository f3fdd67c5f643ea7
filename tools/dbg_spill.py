#!/usr/bin/env python
"""Developer tool: the 24-link stand-in's fused kernel cut into 19 strips
(spills vector registers; returned wrong, run-to-run different values on
MI355X) built with the hipcc flags in OPTY_HIPCC_FLAGS, against the spill-free
build a collocator uses.

    python tools/dbg_spill.py --prebuild     # CPU container: compile all variants
    OPTY_HIPCC_FLAGS="..." python tools/dbg_spill.py        # GPU box
"""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import opty_amd
from opty_amd import hip_backend as hb
from opty_amd.codegen.emit_hip import EmitOptions
from examples import problems

VARIANTS = ['', '-mllvm -amdgpu-spill-sgpr-to-vgpr=0',
            '-mllvm -amdgpu-spill-vgpr-to-agpr=0', '-O1']
kw = problems.build('config5_standin_24link')
BAD = dict(groups=31, fused_groups=19)


def collocator(opts):
    return opty_amd.ConstraintCollocator(launch_nodes=6250, emit_options=opts,
                                         **kw)


if '--prebuild' in sys.argv:
    small = problems.build('config5_standin_24link_small')
    for flags in VARIANTS:
        os.environ['OPTY_HIPCC_FLAGS'] = flags
        col = opty_amd.ConstraintCollocator(
            launch_nodes=6250, emit_options=EmitOptions(**BAD), **small)
        src, meta = col.generate_source()
        h = hb.compile_module(src)
        r = hb.kernel_resources(h)['opty_conjac']
        print(repr(flags), os.path.basename(h), 'vgpr', r['.vgpr_count'],
              'spills v', r['.vgpr_spill_count'], 's',
              r['.sgpr_spill_count'], 'scratch',
              r['.private_segment_fixed_size'], flush=True)
    sys.exit(0)

import torch
flags = os.environ.get('OPTY_HIPCC_FLAGS', '')
bad = collocator(EmitOptions(**BAD))
os.environ['OPTY_HIPCC_FLAGS'] = ''
good = collocator(None)
os.environ['OPTY_HIPCC_FLAGS'] = flags
dev = torch.device('cuda:0')
free = torch.from_numpy(problems.make_free(bad.num_free, seed=0,
                                           variable_duration=True)).to(dev)
a, b = 18750, 25000
P, M = 5100, 50
outs = {}
for tag, col in (('bad', bad), ('good', good)):
    hip = col.hip
    hip.use_torch_stream()
    con = torch.empty((M, b - a), dtype=torch.float64, device=dev)
    jac = torch.empty((b - a)*P, dtype=torch.float64, device=dev)
    runs = []
    for rep in range(4):
        jac.fill_(float('nan'))
        hip.eval_shard(hb.EVAL_FUSED, free, con, b - a, jac, a, b)
        torch.cuda.synchronize()
        runs.append(jac.cpu().numpy().copy())
    outs[tag] = runs
    print(tag, repr(flags) if tag == 'bad' else '', 'repeatable:',
          all(np.array_equal(runs[0], r) for r in runs[1:]), flush=True)
ref = outs['good'][0]
for k, r in enumerate(outs['bad']):
    rel = np.abs(r - ref)/np.maximum(np.abs(ref), 1e-300)
    n = int((rel > 1e-9).sum())
    ents = sorted({int(i) % P for i in np.nonzero(rel > 1e-9)[0][:200000]})
    print('run %d: %d entries differ from the spill-free build; block entries %s'
          % (k, n, ents[:20]), flush=True)
