#!/usr/bin/env python
"""Developer tool (GPU box): where does page-locked memory land?"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import numpy as np
from opty_amd import hip_backend as hb
hb.load_library()
import torch
torch.zeros(1).cuda()
a = hb.pinned_empty(50_000_000)
a[:] = 0
b = np.zeros(50_000_000)
import ctypes; print('cpu now', ctypes.CDLL(None).sched_getcpu(), 'pinned node', hb.host_numa_node(a), 'pageable node', hb.host_numa_node(b),
      )
