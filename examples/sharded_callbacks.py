#!/usr/bin/env python
"""BASELINE config 4 as a solver callback: ONE 10-link pendulum problem,
node-sharded over the GPUs of a node, serving ``constraints(free)`` /
``jacobian(free)`` to the process that runs the NLP solver.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \\
        --master-addr 127.0.0.1 examples/sharded_callbacks.py [num_nodes] [gait]

(``gait``: the gait-like problem instead -- variable duration, a known and an
unknown input trajectory, instance constraints with periodic two-atom pairs;
its instance tails are evaluated by the rank that assembles each vector.)

Rank 0 plays the solver: it evaluates the callbacks of
``opty_amd.ShardedProblem`` at a few points (IPOPT would, through
``prob.solve(x0)``, once ``cyipopt`` is installed) and reports the
host-visible rate; the other ranks ``serve()``.  Every rank loads ``free``
and returns its shard over its own PCIe link through page-locked host vectors
shared by all processes.
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__),
                                                '..')))

import numpy as np
import torch
import torch.distributed as dist

import opty_amd
from examples import problems


def main():
    num_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    shared_gpu = int(os.environ.get('WORLD_SIZE', 1)) > \
        torch.cuda.device_count()
    if shared_gpu:            # development: several ranks on one GPU
        local %= torch.cuda.device_count()
        dist.init_process_group('gloo')
    else:
        dist.init_process_group('nccl',
                                device_id=torch.device('cuda', local))
    torch.cuda.set_device(local)
    if 'gait' in sys.argv[2:]:
        kw = problems.gait_like_pendulum(num_links=10, num_nodes=num_nodes)
    else:
        kw = problems.n_link_cart_pendulum(num_links=10,
                                           num_nodes=num_nodes)
    # any objective: the callbacks under test are the constraints
    prob = opty_amd.ShardedProblem(lambda free: float(free @ free),
                                   lambda free: 2.0*free, device=local, **kw)
    if rank != 0:
        prob.serve()
        dist.destroy_process_group()
        return
    rows, cols = prob.jacobianstructure()
    vd = prob.collocator._variable_duration
    frees = [problems.make_free(prob.num_free, seed=s, variable_duration=vd)
             for s in range(3)]
    prob.constraints(frees[0]), prob.jacobian(frees[0])
    t0 = time.perf_counter()
    reps = 10
    for k in range(reps):
        g = prob.constraints(frees[k % 3])
        vals = prob.jacobian(frees[k % 3])
    el = (time.perf_counter() - t0)/reps
    print('%d ranks, N = %d: constraints + jacobian in %.2f ms (%.0f '
          'evals/s), %d constraints, %d Jacobian values (%.0f MB), '
          'max |g| %.3g' % (dist.get_world_size(), num_nodes, 1e3*el, 1/el,
                           len(g), len(vals), 8e-6*len(vals),
                           np.abs(g).max()))
    assert len(vals) == len(rows) == len(cols)
    prob.shutdown()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
