#!/usr/bin/env python
"""Developer tool (GPU box): host RSS and free device memory around a few
hundred callbacks (config 3, host path) and collocator create / close cycles."""
import os, sys, gc
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import numpy as np
import psutil
import torch
import opty_amd
from examples import problems

proc = psutil.Process()


def snap(tag):
    free, total = torch.cuda.mem_get_info()
    print('%-34s host RSS %7.1f MB   device used %8.1f MB' % (
        tag, proc.memory_info().rss/2**20, (total - free)/2**20), flush=True)


snap('start')
col = opty_amd.ConstraintCollocator(**problems.build('config3_10link'))
con, jac = col.generate_constraint_function(), col.generate_jacobian_function()
frees = [problems.make_free(col.num_free, seed=s) for s in range(3)]
con(frees[0]); jac(frees[0]); jac(frees[1])
snap('config 3 built, first calls')
for rep in range(3):
    for k in range(100):
        con(frees[k % 3]); jac(frees[k % 3])
    snap('after %d pairs' % (100*(rep + 1)))
col.hip.close()
del col, con, jac
gc.collect()
snap('closed')
for rep in range(3):
    for k in range(20):
        c = opty_amd.ConstraintCollocator(**problems.build('config2_pendulum'))
        f = problems.make_free(c.num_free)
        c.generate_constraint_function()(f)
        c.generate_jacobian_function()(f)
        c.hip.close()
    gc.collect()
    snap('after %d create/close cycles' % (20*(rep + 1)))
