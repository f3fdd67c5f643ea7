#!/usr/bin/env python
"""Developer tool (GPU box): device-to-host rate into a SharedHostVector
(file-backed /dev/shm mapping, hipHostRegister) vs a torch pinned tensor."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import torch
from opty_amd.sharded import SharedHostVector
n = 98999010
dev = torch.device('cuda:0')
src = torch.randn(n, dtype=torch.float64, device=dev)
vec = SharedHostVector('opty_rate_%d' % os.getpid(), n, 0)
dst = vec.torch_view()
pin = torch.empty(n, dtype=torch.float64).pin_memory()
print('registered mapping is_pinned:', dst.is_pinned())
for label, d in (('shared /dev/shm + hipHostRegister', dst), ('torch pinned', pin)):
    for _ in range(2):
        d.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0)/5
    print('%-36s %.2f ms  %.1f GB/s' % (label, dt*1e3, n*8/dt/1e9))
assert torch.equal(dst, src.cpu())
vec.close()
