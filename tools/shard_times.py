#!/usr/bin/env python
"""Developer tool (GPU box): the fused launch of one node shard of BASELINE
config 4 (and of the config-5 stand-ins) for world sizes 1, 2, 4, 8, timed on
ONE GPU with hipEvents -- the compute-only strong-scaling projection of
DESIGN.md section 7."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import torch
from opty_amd import hip_backend as hb
from opty_amd.sharded import ShardedCollocator
from examples import problems

names = sys.argv[1:] or ['config3_10link']
for name in names:
    kw = problems.build(name)
    base = None
    program = None
    for world in [int(w) for w in os.environ.get('SHARD_WORLDS', '1,2,4,8').split(',')]:
        sh = ShardedCollocator(rank=world//2, world_size=world, **kw)
        col = sh.collocator
        if program is not None:
            col._program = program          # same equations: lower once
        program = col._build_program()
        hip = col.hip
        hip.use_torch_stream()
        free = torch.from_numpy(problems.make_free(
            col.num_free, variable_duration=col._variable_duration)).cuda()
        cs = sh.con_local.stride(0)
        for _ in range(3):
            hip.time_eval_shard(hb.EVAL_FUSED, free, sh.con_local, cs,
                                sh.jac_local, sh.a, sh.b, 200)
        ts = [hip.time_eval_shard(hb.EVAL_FUSED, free, sh.con_local, cs,
                                  sh.jac_local, sh.a, sh.b, 200)
              for _ in range(7)]
        ms = float(np.median(ts))
        base = base or ms
        g = col._kernel_meta['geometry']
        print('%-26s world %d  %6d nodes  %2d strips + %d constraint waves%s'
              '  fused %.4f ms (min %.4f)  speed-up %.2f' % (
                  name, world, sh.b - sh.a, g['fused'], g['con_waves'],
                  ', two waves per SIMD' if g['occupancy'] == 2 else '', ms,
                  min(ts), base/ms), flush=True)
        hip.close()
        del sh, free
        torch.cuda.empty_cache()
