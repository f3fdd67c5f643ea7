"""The torch-free sharding host (``opty_amd.shard_host``): partition with
unequal shards, evaluation of a node range from the GLOBAL free vector,
broadcast, gather-v with in-place Jacobian slices and equation-major
constraint placement, the shared host vector -- in plain processes that talk
through ``SocketTransport`` and never import torch.  The shard evaluator is
the oracle (test infrastructure); on GPUs it is ``opty_hip_eval_shard`` and
the exchange is the library's own RCCL communicator (``RcclTransport``)."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

from examples import problems
from opty_amd.sharded import partition_nodes, slab_of


class _OracleShard(object):
    """NumPy twin of ``tests/test_sharded_gloo._OracleShard``."""

    def __init__(self, kw, a, b):
        from oracle.collocation_oracle import OracleCollocator
        self.N = kw['num_collocation_nodes']
        self.a, self.b = a, b
        self.traj_map = kw.get('known_trajectory_map', {})
        cnt = b - a + 1
        known = {k: np.zeros(cnt) for k in self.traj_map}
        self.o = OracleCollocator(name='shard', **dict(
            kw, num_collocation_nodes=cnt, known_trajectory_map=known,
            instance_constraints=None))
        self.o.known_parameter_map = dict(self.o.known_parameter_map)
        self.con = self.o.generate_constraint_function()
        self.jac = self.o.generate_jacobian_function()
        self.rows = self.o.n + self.o.q
        self.full = None
        if kw.get('instance_constraints') is not None:
            self.full = OracleCollocator(name='shard_full', **kw)

    def __call__(self, free, con2d, jac1d, a, b, what='both'):
        assert (a, b) == (self.a, self.b)
        for k, v in self.traj_map.items():
            v = v(free) if callable(v) else v
            self.o.known_trajectory_map[k] = np.asarray(v)[a:b + 1]
        slab = slab_of(free, self.N, self.rows, a, b)
        if con2d is not None:
            con2d[...] = self.con(slab).reshape(self.o.M, b - a)
        if jac1d is not None:
            jac1d[...] = np.asarray(self.jac(slab))

    def instance(self, free, con_tail, jac_tail):
        if con_tail is not None:
            con_tail[...] = self.full.eval_instance_constraints(free)
        if jac_tail is not None:
            jac_tail[...] = \
                self.full.eval_instance_constraints_jacobian_values(free)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, N, out, root):
    from opty_amd.shard_host import NodeShard, SocketTransport
    from opty_amd.sharded import SharedHostVector
    t = SocketTransport(rank, world, '127.0.0.1', port)
    try:
        factory, fkw = problems.CONFIGS[name]
        kw = factory(**dict(fkw, num_nodes=N))
        a, b = partition_nodes(N - 1, world)[rank]
        ev = _OracleShard(dict(kw), a, b)
        sh = NodeShard(rank=rank, world_size=world, transport=t,
                       evaluator=ev, instance_evaluator=ev.instance, **kw)
        assert (sh.a, sh.b) == (a, b)
        free = problems.make_free(sh.num_free, seed=7,
                                  variable_duration=ev.o.variable_duration) \
            if rank == root else None
        # collective conveniences: values of the GLOBAL problem on `root`
        con = sh.constraints(free, root=root)
        jac = sh.jacobian(free, root=root)
        assert (con is None) == (rank != root) == (jac is None)
        # pieces: broadcast, local evaluation, gather of both at once to
        # another rank (its own shard written in place)
        other = (root + 1) % world
        sh.broadcast_free(free, root)
        sh.evaluate('both', in_place=(rank == other))
        got = sh.gather(other, 'both')
        assert (got is None) == (rank != other)
        # the node-wide host vector, created and agreed on over the same
        # side channel
        vec = SharedHostVector('opty_t_sh_%d' % port, sh.nnz, rank, pin=False,
                               transport=t)
        lc, lj = sh._local_buffers() if rank != other else (None, None)
        if rank == other:
            vec.array[:] = got[1]
        t.barrier()
        extra = {}
        if rank == root:
            extra = dict(con=con, jac=jac, free=free)
        if rank == other:
            extra.update(g_con=got[0].copy(), g_jac=got[1].copy())
        np.savez(out % rank, h_jac=np.array(vec.array),
                 torch_loaded='torch' in sys.modules, **extra)
    finally:
        t.close()


@pytest.mark.parametrize('name,N,world,root', [
    ('msd_be_small', 24, 2, 0),                     # 23 nodes: 12 + 11
    ('pend2_link_vardur_unkmass_small', 26, 3, 2),  # 9 + 8 + 8, root != hub
    ('gaitlike_3link_mid_small', 38, 2, 1),         # instance constraints
    ('chaplygin_be_small', 100, 3, 1),              # M > n, 8 instance cons
    ('implicit_traj_be_small', 40, 2, 0),           # callable known traj.
    ('pend3_link_midpoint_small', 44, 8, 5)])       # 43 nodes: 3 x 6 + 5 x 5
def test_torch_free_shards_reassemble_to_full(tmp_path, name, N, world, root):
    from oracle.collocation_oracle import OracleCollocator
    out = str(tmp_path/'rank%d.npz')
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, N, out,
                                               root)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    factory, fkw = problems.CONFIGS[name]
    full = OracleCollocator(name='shard', **factory(**dict(fkw,
                                                           num_nodes=N)))
    z = [np.load(out % r) for r in range(world)]
    free = z[root]['free']
    c_ref = full.generate_constraint_function()(free)
    j_ref = np.asarray(full.generate_jacobian_function()(free))
    kw = dict(rtol=1e-13, atol=1e-13)
    other = (root + 1) % world
    np.testing.assert_allclose(z[root]['con'], c_ref, **kw)
    np.testing.assert_allclose(z[root]['jac'], j_ref, **kw)
    np.testing.assert_allclose(z[other]['g_con'], c_ref, **kw)
    np.testing.assert_allclose(z[other]['g_jac'], j_ref, **kw)
    for r in range(world):
        np.testing.assert_allclose(z[r]['h_jac'], j_ref, **kw)
        assert not bool(z[r]['torch_loaded']), \
            'rank %d imported torch' % r


def test_side_channel_primitives():
    """``SocketTransport`` alone: object broadcast from every root, min
    all-reduce, gather with a root that is not the hub."""
    ctx = mp.get_context('spawn')
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_primitives, args=(r, 3, port, q))
             for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    assert [r[0] for r in res] == [0, 1, 2]
    for rank, objs, low, gathered in res:
        assert objs == [('from', 0), ('from', 1), ('from', 2)]
        assert low == 5
        assert gathered == ([b'r0', b'r1', b'r2'] if rank == 2 else None)


def _primitives(rank, world, port, q):
    from opty_amd.shard_host import SocketTransport
    t = SocketTransport(rank, world, '127.0.0.1', port)
    objs = [t.bcast_object(('from', rank) if rank == root else None, root)
            for root in range(world)]
    low = t.allreduce_min(5 + rank)
    gathered = t.gather_bytes(b'r%d' % rank, root=2)
    t.barrier()
    q.put((rank, objs, low, gathered))
    t.close()


@pytest.mark.gpu
def test_torch_free_shard_on_the_device_matches_the_golden():
    """World of one on the GPU, no torch objects anywhere: device memory
    from the C ABI, ``opty_hip_eval_shard``, values against the reference's
    golden record."""
    import golden_util as gu
    from opty_amd.shard_host import NodeShard
    name = 'config3_10link_small'
    meta, z = gu.load(name)
    kw = problems.build(name)
    sh = NodeShard(rank=0, world_size=1, **kw)
    con = sh.constraints(z['free'])
    jac = sh.jacobian(z['free'])
    np.testing.assert_allclose(con, z['con'], rtol=1e-10,
                               atol=1e-10*np.abs(z['con']).max())
    np.testing.assert_allclose(jac, z['jac'], rtol=1e-10,
                               atol=1e-10*np.abs(z['jac']).max())
    rows, cols = sh.jacobian_indices_local()
    np.testing.assert_array_equal(rows, z['rows'])
    np.testing.assert_array_equal(cols, z['cols'])
    # not in place + gather: the same vectors
    sh.set_free(z['free'])
    sh.evaluate('both')
    gc, gj = sh.gather(0, 'both')
    # (the fused kernel here, the separate ones above: equal to rounding)
    np.testing.assert_allclose(gj.numpy(), jac, rtol=1e-12,
                               atol=1e-12*np.abs(jac).max())
    np.testing.assert_allclose(gc.numpy(), con, rtol=1e-12,
                               atol=1e-12*np.abs(con).max())
    sh.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,world,root', [
    ('config2_pendulum_small', 2, 1),           # instance constraints
    ('config3_10link_small', 3, 0)])            # 40 nodes: shards 14/13/13
def test_torch_free_ranks_over_the_library_communicator(name, world, root,
                                                        tmp_path):
    """``NodeShard`` + ``RcclTransport`` with 2 and 3 ranks on one GPU: the
    unique id travels over the TCP side channel, the data through
    ``opty_hip_bcast_free`` / ``opty_hip_gather_v`` (pointed at the test
    transport with librccl's entry points, ``tests/fake_rccl`` -- RCCL
    refuses duplicate devices); checked on the root against the single-GPU
    collocator; no rank imports torch."""
    import subprocess
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    lib = str(tmp_path/'libfake_rccl.so')
    proc = subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared',
                           '-fPIC', '-O1', os.path.join(
                               repo, 'tests', 'fake_rccl', 'fake_rccl.cpp'),
                           '-o', lib], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    env = dict(os.environ, OPTY_HIP_RCCL_LIBRARY=lib)
    port = _free_port()
    procs = [subprocess.Popen(
        [sys.executable, os.path.join(repo, 'tests', 'shard_host_worker.py'),
         name, str(r), str(world), str(root), str(port)],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d:\n%s' % (r, out[-3000:])
        assert 'rank %d of %d ok' % (r, world) in out
    assert '4 gathers checked' in outs[root]
