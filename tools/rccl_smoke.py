#!/usr/bin/env python
"""Developer tool (GPU box): the torch.distributed calls bench.py makes, on
RCCL with however many ranks the launcher gives (1 on a 1-GPU box)."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
local = int(os.environ.get('LOCAL_RANK', 0))
dist.init_process_group('nccl', rank=rank, world_size=world,
                        device_id=torch.device('cuda', local))
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
t = torch.ones(4, dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); dist.broadcast(t, 0)
from opty_amd.sharded import SharedHostVector
v = SharedHostVector('opty_rccl_smoke_%d' % os.getppid(), 1000, rank)
v.torch_view().copy_(torch.arange(1000, dtype=torch.float64, device=dev), non_blocking=True)
torch.cuda.synchronize(); assert v.array[999] == 999.0
ops = []
if world > 1:
    peer = (rank + 1) % world
    a = torch.full((10,), float(rank), dtype=torch.float64, device=dev); b = torch.empty_like(a)
    for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, peer), dist.P2POp(dist.irecv, b, (rank - 1) % world)]):
        r.wait()
    torch.cuda.synchronize(); assert b[0].item() == float((rank - 1) % world)
print('rank %d/%d ok: backend %s' % (rank, world, dist.get_backend()))
v.close(); dist.destroy_process_group()
