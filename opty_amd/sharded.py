"""Node sharding of one collocation problem over the GPUs of a node
(one process per GPU, ``torch.distributed``; backend ``nccl`` = RCCL over xGMI
on GPUs, ``gloo`` in the CPU tests).

The path shards naturally: constraint node ``i`` reads only
time nodes ``i`` and ``i + 1`` of every trajectory row
(``opty/direct_collocation.py:2145, 2153-2155, 2411-2413``), so rank ``g``
with constraint nodes ``[a_g, b_g)`` needs the time-node columns
``[a_g, b_g]`` -- a one-node halo -- plus the node-invariant tail of ``free``.
That slab *is* the free vector of an ordinary collocation problem with
``b_g - a_g + 1`` nodes, so every rank simply runs an unmodified
:class:`opty_amd.ConstraintCollocator` on its slab:

* the Jacobian shard is the contiguous slice ``[a_g*P, b_g*P)`` of the global
  node-major value vector;
* the constraint shard is, per equation ``j``, the slice
  ``[j*(N-1) + a_g, j*(N-1) + b_g)`` of the global equation-major vector;
* the COO indices need no communication (closed form with the global ``N``).

No collective is needed to *evaluate*.  Re-assembling the full vectors (what a
single-process IPOPT wants) is one all-gather per output.
"""

import numpy as np

__all__ = ['partition_nodes', 'ShardedCollocator']


def partition_nodes(num_constraint_nodes, world_size):
    """Contiguous, balanced split of the constraint nodes ``[0, N-1)``:
    ``[(a_0, b_0), ...]``; sizes differ by at most one."""
    q, r = divmod(int(num_constraint_nodes), int(world_size))
    out, a = [], 0
    for g in range(world_size):
        b = a + q + (1 if g < r else 0)
        out.append((a, b))
        a = b
    return out


class ShardedCollocator(object):
    """One rank's view of a node-sharded collocation problem.

    Parameters are those of :class:`opty_amd.ConstraintCollocator` for the
    GLOBAL problem, plus ``rank`` / ``world_size`` (default: from
    ``torch.distributed``), and ``local_factory`` -- a callable that builds the
    local evaluator from the local keyword dict (default: the HIP
    ``ConstraintCollocator``; the CPU tests inject the oracle).  The local
    evaluator must offer ``generate_constraint_function()``,
    ``generate_jacobian_function()``, ``num_free``, ``num_states`` and
    ``num_unknown_input_trajectories``.

    Instance constraints are not node-sharded; they stay with the caller.
    """

    def __init__(self, equations_of_motion, state_symbols,
                 num_collocation_nodes, node_time_interval,
                 known_parameter_map={}, known_trajectory_map={},
                 instance_constraints=None, rank=None, world_size=None,
                 group=None, local_factory=None, **kwargs):
        if instance_constraints is not None:
            raise NotImplementedError('instance constraints are evaluated by '
                                      'the caller, not by the node shards')
        if rank is None or world_size is None:
            import torch.distributed as dist
            rank = dist.get_rank(group)
            world_size = dist.get_world_size(group)
        self.rank, self.world_size, self.group = rank, world_size, group
        self.N = int(num_collocation_nodes)
        self.ranges = partition_nodes(self.N - 1, world_size)
        self.a, self.b = self.ranges[rank]
        if self.b <= self.a:
            raise ValueError('more ranks than constraint nodes')
        for k, v in known_trajectory_map.items():
            if callable(v):
                raise NotImplementedError('callable known trajectories are '
                                          'not supported by the node shards')
            if len(v) != self.N:
                raise ValueError('The known parameter {} is not length {}.'
                                 .format(k, self.N))
        local_known = {k: np.ascontiguousarray(v[self.a:self.b + 1])
                       for k, v in known_trajectory_map.items()}
        local_kw = dict(equations_of_motion=equations_of_motion,
                        state_symbols=state_symbols,
                        num_collocation_nodes=self.b - self.a + 1,
                        node_time_interval=node_time_interval,
                        known_parameter_map=known_parameter_map,
                        known_trajectory_map=local_known, **kwargs)
        if local_factory is None:
            from .direct_collocation import ConstraintCollocator
            local_factory = lambda kw: ConstraintCollocator(**kw)
        self.local = local_factory(local_kw)
        self._con = self._jac = None

    # -- layout ---------------------------------------------------------------
    @property
    def num_local_nodes(self):
        """Constraint nodes owned by this rank."""
        return self.b - self.a

    def local_free(self, free_global, num_rows, num_tail):
        """The rank's slab of the global free vector: columns ``[a, b]`` of
        each of the ``num_rows`` trajectory rows (states, then unknown
        inputs), followed by the ``num_tail`` node-invariant entries."""
        free_global = np.asarray(free_global)
        N = self.N
        rows = free_global[:num_rows*N].reshape(num_rows, N)
        tail = free_global[num_rows*N:]
        assert len(tail) == num_tail
        return np.concatenate((rows[:, self.a:self.b + 1].ravel(), tail))

    def _rows_tail(self):
        loc = self.local
        num_rows = loc.num_states + loc.num_unknown_input_trajectories
        return num_rows, loc.num_free - num_rows*(self.b - self.a + 1)

    # -- evaluation -------------------------------------------------------------
    def constraints_local(self, free_global):
        """``(M, b - a)`` equation-major shard of ``constraints(free)``."""
        if self._con is None:
            self._con = self.local.generate_constraint_function()
        num_rows, num_tail = self._rows_tail()
        con = self._con(self.local_free(free_global, num_rows, num_tail))
        return np.asarray(con).reshape(-1, self.num_local_nodes)

    def jacobian_local(self, free_global):
        """Contiguous slice ``[a*P, b*P)`` of ``jacobian(free)``."""
        if self._jac is None:
            self._jac = self.local.generate_jacobian_function()
        num_rows, num_tail = self._rows_tail()
        return np.asarray(
            self._jac(self.local_free(free_global, num_rows, num_tail)))

    # -- re-assembly (the only collectives) --------------------------------------
    def _all_gather(self, local, device=None):
        """All-gather of variable-size 1-D shards (sizes differ by <= one
        node): pad to the largest shard, one ``all_gather_into_tensor``."""
        import torch
        import torch.distributed as dist
        sizes = [int(np.prod(s)) for s in self._shard_shapes(local)]
        width = max(sizes)
        t = torch.zeros(width, dtype=torch.float64, device=device)
        t[:local.size] = torch.as_tensor(np.ascontiguousarray(local).ravel(),
                                         device=device)
        out = torch.empty(self.world_size*width, dtype=torch.float64,
                          device=device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        out = out.cpu().numpy().reshape(self.world_size, width)
        return [out[g, :sizes[g]] for g in range(self.world_size)]

    def _shard_shapes(self, local):
        per_node = local.size//self.num_local_nodes
        return [((b - a)*per_node,) for a, b in self.ranges]

    def constraints(self, free_global, device=None):
        """Full equation-major ``constraints(free)`` on every rank."""
        loc = self.constraints_local(free_global)
        M = loc.shape[0]
        parts = self._all_gather(loc, device)
        blocks = [p.reshape(M, b - a) for p, (a, b) in zip(parts,
                                                           self.ranges)]
        return np.hstack(blocks).ravel()

    def jacobian(self, free_global, device=None):
        """Full node-major ``jacobian(free)`` on every rank."""
        return np.concatenate(self._all_gather(self.jacobian_local(
            free_global), device))

    def jacobian_indices_local(self):
        """Global int64 COO indices of this rank's Jacobian slice, from the
        device index kernel run with the global ``N`` and this rank's node
        offset (needs the HIP local evaluator)."""
        from . import hip_backend as hb
        hip = self.local.hip
        rows = np.empty(hip.nnz, dtype=np.int64)
        cols = np.empty(hip.nnz, dtype=np.int64)
        hip.jacobian_indices_shard(self.N, self.a, rows, cols, hb.HOST)
        return rows, cols
