"""The node-sharding logic (partition, one-node halo slabs, all-gather
re-assembly) on 2 CPU processes with the ``gloo`` backend.  The local evaluator
is the oracle (test infrastructure) -- on GPUs it is the HIP collocator; the
sharding code is identical."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from opty_amd import problems
from opty_amd.sharded import ShardedCollocator, partition_nodes


class _OracleLocal(object):
    def __init__(self, kw):
        from oracle.collocation_oracle import OracleCollocator
        self.o = OracleCollocator(name='shard', **kw)
        self.num_free = self.o.num_free
        self.num_states = self.o.n
        self.num_unknown_input_trajectories = self.o.q

    def generate_constraint_function(self):
        return self.o.generate_constraint_function()

    def generate_jacobian_function(self):
        return self.o.generate_jacobian_function()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, N, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        factory, fkw = problems.CONFIGS[name]
        kw = factory(**dict(fkw, num_nodes=N))
        sh = ShardedCollocator(local_factory=_OracleLocal, **kw)
        free = problems.make_free(
            sh.local.o.num_free - (sh.b - sh.a + 1)*(sh.local.o.n +
                                                     sh.local.o.q) +
            N*(sh.local.o.n + sh.local.o.q), seed=7,
            variable_duration=sh.local.o.variable_duration)
        con = sh.constraints(free)
        jac = sh.jacobian(free)
        if rank == 0:
            np.savez(out, con=con, jac=jac, free=free)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name,N', [('msd_be_small', 24),
                                    ('pend3_link_midpoint_small', 31),
                                    ('pend2_link_vardur_unkmass_small', 26)])
def test_two_rank_shards_reassemble_to_full(tmp_path, name, N):
    from oracle.collocation_oracle import OracleCollocator
    out = str(tmp_path/'gathered.npz')
    mp.spawn(_worker, args=(2, _free_port(), name, N, out), nprocs=2,
             join=True)
    z = np.load(out)
    factory, fkw = problems.CONFIGS[name]
    full = OracleCollocator(name='shard', **factory(**dict(fkw,
                                                           num_nodes=N)))
    np.testing.assert_allclose(
        z['con'], full.generate_constraint_function()(z['free']),
        rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(
        z['jac'], full.generate_jacobian_function()(z['free']),
        rtol=1e-13, atol=1e-13)


def test_partition():
    assert partition_nodes(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert partition_nodes(99999, 8)[-1][1] == 99999
    sizes = [b - a for a, b in partition_nodes(99999, 8)]
    assert max(sizes) - min(sizes) <= 1
