"""One rank of ``tests/test_shard_host.py::test_torch_free_ranks_over_the_
library_communicator``: ``NodeShard`` + ``RcclTransport`` on ``cuda:0`` (the
library's RCCL entry points are a test transport there: several ranks on one
GPU), no torch in the process.

    shard_host_worker.py <problem> <rank> <world> <root> <port>
"""
import os
import sys

os.environ['OPTY_HIP_NO_TORCH'] = '1'
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np  # noqa: E402


def main():
    name, rank, world, root, port = sys.argv[1], int(sys.argv[2]), \
        int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import opty_amd
    from examples import problems
    from opty_amd.shard_host import NodeShard, RcclTransport, SocketTransport
    side = SocketTransport(rank, world, '127.0.0.1', port)
    rccl = RcclTransport(side, device=0)
    kw = problems.build(name)
    sh = NodeShard(rank=rank, world_size=world, transport=rccl, **kw)
    vd = sh.collocator._variable_duration
    free = problems.make_free(sh.num_free, seed=5, variable_duration=vd) \
        if rank == root else None
    con = sh.constraints(free, root=root)
    jac = sh.jacobian(free, root=root)
    # both at once, the root's shard NOT in place
    sh.broadcast_free(free, root)
    sh.evaluate('both')
    got = sh.gather(root, 'both')
    if rank == root:
        ref = opty_amd.ConstraintCollocator(**kw)
        c_ref = ref.generate_constraint_function()(free)
        j_ref = np.array(ref.generate_jacobian_function()(free))
        for a, b in ((con, c_ref), (jac, j_ref), (got[0].numpy(), c_ref),
                     (got[1].numpy(), j_ref)):
            np.testing.assert_allclose(a, b, rtol=1e-12,
                                       atol=1e-12*np.abs(b).max())
        print('4 gathers checked')
    side.barrier()
    assert 'torch' not in sys.modules, 'torch was imported'
    print('rank %d of %d ok' % (rank, world))
    sh.close()
    rccl.close()
    side.close()


if __name__ == '__main__':
    main()
