"""One rank of ``tests/test_sharded_gpu.py::test_gather_v_with_several_ranks``:
``python comm_worker.py <problem> <rank> <world> <root> <id file>`` -- all ranks
on ``cuda:0``, the library's communicator over the test transport
(``OPTY_HIP_RCCL_LIBRARY`` = the double of ``tests/fake_rccl``)."""
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from opty_amd.sharded import ShardedCollocator                # noqa: E402
from examples import problems                                 # noqa: E402


def main():
    name, rank, world, root, idfile = sys.argv[1], int(sys.argv[2]), \
        int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    if rank == 0:
        with open(idfile + '.tmp', 'wb') as f:
            f.write(hb.HipComm.unique_id())
        os.replace(idfile + '.tmp', idfile)
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    uid = open(idfile, 'rb').read()
    comm = hb.HipComm(uid, rank, world, device=0)
    kw = problems.build(name)
    sh = ShardedCollocator(rank=rank, world_size=world, device='cuda:0',
                           comm=comm, **kw)
    col = sh.collocator
    free_h = problems.make_free(col.num_free, seed=3,
                                variable_duration=col._variable_duration)
    # only the root knows the free vector: the others get it by broadcast
    free = torch.from_numpy(free_h if rank == root
                            else np.zeros_like(free_h)).cuda()
    sh.broadcast_free(free, root)
    torch.cuda.synchronize()
    assert np.array_equal(free.cpu().numpy(), free_h), 'broadcast'
    checked = 0
    for in_place in (False, True):
        for what in ('both', 'con', 'jac'):
            sh.evaluate(free, in_place=(in_place and rank == root), what=what)
            got = sh.gather(root, what)
            torch.cuda.synchronize()
            if rank != root:
                assert got is None
                continue
            ref = opty_amd.ConstraintCollocator(**kw)
            con0 = ref.generate_constraint_function()(free_h)
            jac0 = np.array(ref.generate_jacobian_function()(free_h))
            ref.hip.close()
            con, jac = got
            if what != 'jac':
                np.testing.assert_allclose(
                    con.cpu().numpy(), con0, rtol=1e-12,
                    atol=1e-12*np.abs(con0).max())
            if what != 'con':
                np.testing.assert_allclose(
                    jac.cpu().numpy(), jac0, rtol=1e-12,
                    atol=1e-12*np.abs(jac0).max())
            checked += 1
    comm.close()
    print('rank %d of %d ok (%d gathers checked, shard [%d, %d))'
          % (rank, world, checked, sh.a, sh.b), flush=True)


if __name__ == '__main__':
    main()
