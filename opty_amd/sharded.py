"""Node sharding of ONE collocation problem over the GPUs of a node
(BASELINE config 4; one process per GPU, ``torch.distributed``: backend
``nccl`` = RCCL over xGMI on GPUs, ``gloo`` in the CPU tests).

The path shards naturally: constraint node ``i`` reads only
time nodes ``i`` and ``i + 1`` of every trajectory row
(``opty/direct_collocation.py:2145, 2153-2155, 2411-2413``).  Rank ``g`` owns
the constraint nodes ``[a_g, b_g)`` and

* evaluates them with ``opty_hip_eval_shard`` straight from the GLOBAL free
  vector in its HBM -- no host slicing, the kernels take the node range as an
  argument and read the time-node columns ``[a_g, b_g]`` (one-node halo);
* its Jacobian shard is the contiguous slice ``[a_g*P, b_g*P)`` of the global
  node-major value vector (``:2885-2887``);
* its constraint shard is, per equation ``j``, the segment
  ``[j*(N-1) + a_g, j*(N-1) + b_g)`` of the global equation-major vector
  (``:2446``);
* the COO indices need no communication (closed form with the global ``N``).

Evaluation needs **no collective**.  What a single-process IPOPT wants -- the
whole vectors in one place -- is offered three ways, never folded into the
evaluation itself:

``gather(dst)``     point-to-point gather-v over RCCL: every rank sends its two
                    shards, ``dst`` receives the Jacobian slices *in place*
                    (views of the global vector, shard sizes may differ) and
                    the constraint blocks into a staging buffer that one
                    strided device copy scatters to ``j*(N-1) + a_g``;
``all_gather()``    the same exchange with every rank as a destination;
``to_host(vec)``    every rank copies its shard over its *own* PCIe link into
                    one page-locked host vector shared by all processes
                    (:class:`SharedHostVector`) -- 8 links in parallel instead
                    of funnelling 792 MB through one GPU.
"""

import os

import numpy as np

__all__ = ['partition_nodes', 'slab_of', 'ShardedCollocator',
           'SharedHostVector', 'ShardedCallbacks']


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


def slab_of(free_global, num_nodes, num_rows, a, b):
    """The free vector of the ``(b - a + 1)``-node problem that constraint
    nodes ``[a, b)`` of an ``num_nodes``-node problem form: columns
    ``[a, b]`` of each of the ``num_rows`` trajectory rows (states, then
    unknown inputs) followed by the node-invariant tail.  Host helper (tests,
    callers that keep ``free`` on the host); the device path never builds
    it."""
    free_global = np.asarray(free_global)
    rows = free_global[:num_rows*num_nodes].reshape(num_rows, num_nodes)
    return np.concatenate((rows[:, a:b + 1].ravel(),
                           free_global[num_rows*num_nodes:]))


class SharedHostVector(object):
    """One float64 host vector mapped by every rank of the node (a file in
    ``/dev/shm``), page-locked in each process so that device-to-host copies
    into it run at PCIe rate.  Rank ``owner`` creates it; it is what the
    process that runs IPOPT reads.

    ``torch_view(lo, hi)`` is a CPU tensor over ``[lo, hi)`` for
    ``copy_(device_tensor, non_blocking=True)``.  ``pin``: True page-locks
    the whole vector in this process, ``(lo, hi)`` only that element range
    (the part this rank writes; rounded out to whole pages), False nothing.
    """

    def __init__(self, name, count, rank, group=None, owner=0, pin=True):
        import torch.distributed as dist
        self.path = os.path.join('/dev/shm', name)
        self.count = int(count)
        self._pinned = False
        multi = dist.is_available() and dist.is_initialized()
        error = None
        if rank == owner:
            # reserve the pages now: a full /dev/shm raises here (ENOSPC)
            # instead of a SIGBUS at the first write
            try:
                fd = os.open(self.path, os.O_CREAT | os.O_RDWR | os.O_TRUNC,
                             0o600)
                try:
                    os.posix_fallocate(fd, 0, max(8, 8*self.count))
                finally:
                    os.close(fd)
                self.array = np.memmap(self.path, dtype=np.float64,
                                       mode='r+', shape=(self.count,))
            except OSError as err:
                error = err
                try:
                    os.unlink(self.path)
                except OSError:
                    pass
        if multi:
            # every rank learns whether the owner succeeded: a failure must
            # not leave the others waiting at a barrier
            import torch
            on_gpu = dist.get_backend(group) == 'nccl'
            ok = torch.tensor([0 if error else 1], dtype=torch.int32,
                              device='cuda' if on_gpu else 'cpu')
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if ok.item() == 0:
                raise OSError('could not create the shared host vector %s '
                              '(%d bytes): %s' % (self.path, 8*self.count,
                                                  error or 'owner failed'))
        elif error:
            raise error
        if rank != owner:
            self.array = np.memmap(self.path, dtype=np.float64, mode='r+',
                                   shape=(self.count,))
        if multi:
            dist.barrier(group)
        if rank == owner:
            os.unlink(self.path)        # the mappings keep the memory alive
        self._pin_view = None
        if pin:
            from . import hip_backend as hb
            lo, hi = (0, self.count) if pin is True else pin
            per_page = 4096//8
            lo = (int(lo)//per_page)*per_page
            hi = min(self.count, -(-int(hi)//per_page)*per_page)
            self._pin_view = self.array[lo:hi]
            hb.host_register(self._pin_view)
            self._pinned = True

    def torch_view(self, lo=0, hi=None):
        import torch
        return torch.from_numpy(self.array[lo:self.count if hi is None
                                           else hi])

    def close(self):
        """Unpins the vector; the mapping itself lives as long as any array
        handed out refers to it."""
        if self._pinned:
            from . import hip_backend as hb
            hb.host_unregister(self._pin_view)
            self._pinned = False
        self._pin_view = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedCollocator(object):
    """One rank's share of a node-sharded collocation problem.

    Parameters are those of :class:`opty_amd.ConstraintCollocator` for the
    GLOBAL problem, plus ``rank`` / ``world_size`` (default: from
    ``torch.distributed``), ``group`` and ``device`` (a ``torch.device``;
    default: the collocator's HIP device).

    ``evaluator``: ``f(free, con2d, jac1d, a, b)`` that fills the shard's
    ``(M, b - a)`` constraint block (a possibly strided view) and its
    ``(b - a)*P`` Jacobian values from the global ``free`` tensor.  Default:
    the HIP kernels (``opty_hip_eval_shard``); the CPU tests inject an
    oracle-backed one to exercise the partition and the exchange under
    ``gloo``.  With an evaluator, ``block_shape = (M, P)`` must be given.

    Instance constraints and the CSR layout are not node-sharded.
    """

    def __init__(self, equations_of_motion, state_symbols,
                 num_collocation_nodes, node_time_interval,
                 known_parameter_map={}, known_trajectory_map={},
                 instance_constraints=None, rank=None, world_size=None,
                 group=None, device=None, evaluator=None, block_shape=None,
                 **kwargs):
        import torch
        if instance_constraints is not None:
            raise NotImplementedError('instance constraints are evaluated by '
                                      'the caller, not by the node shards')
        if kwargs.get('jacobian_layout', 'coo') != 'coo':
            raise NotImplementedError(
                'the row-sorted (csr) layout is not node-sharded: a shard of '
                'it is not a contiguous slice of the global value vector')
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
        self.collocator = None
        if evaluator is None:
            from .direct_collocation import ConstraintCollocator
            if device is not None:
                kwargs.setdefault('device', torch.device(device).index or 0)
            # the handle is built for the GLOBAL problem (its kernels read
            # the global free vector); its strip count is chosen for the
            # shard's launch size
            self.collocator = ConstraintCollocator(
                equations_of_motion, state_symbols, num_collocation_nodes,
                node_time_interval, known_parameter_map, known_trajectory_map,
                launch_nodes=max(b - a for a, b in self.ranges), **kwargs)
            prog = self.collocator._build_program()
            self.M, self.P = prog.M, prog.P
            self.device = torch.device('cuda', self.collocator._device)
            evaluator = self._hip_evaluate
        else:
            if block_shape is None:
                raise ValueError('block_shape=(M, P) is needed with an '
                                 'evaluator')
            self.M, self.P = block_shape
            self.device = torch.device(device or 'cpu')
        self._evaluate = evaluator
        cnt = self.b - self.a
        f64 = dict(dtype=torch.float64, device=self.device)
        self.con_local = torch.empty((self.M, cnt), **f64)
        self.jac_local = torch.empty(cnt*self.P, **f64)
        self._global = None         # (con, jac) on ranks that receive
        self._stage = None          # constraint blocks of the other ranks
        self._in_place = False
        self._stream = None

    # -- layout ---------------------------------------------------------------
    @property
    def num_local_nodes(self):
        """Constraint nodes owned by this rank."""
        return self.b - self.a

    def _global_buffers(self, what='both'):
        """This rank's copies of the global vectors (allocated on first use,
        each on its own: a constraints-only destination never holds the
        Jacobian)."""
        import torch
        f64 = dict(dtype=torch.float64, device=self.device)
        ncn = self.N - 1
        if self._global is None:
            self._global = [None, None]
        if what != 'jac' and self._global[0] is None:
            self._global[0] = torch.empty(self.M*ncn, **f64)
            self._stage = {
                g: torch.empty((self.M, b - a), **f64)
                for g, (a, b) in enumerate(self.ranges) if g != self.rank}
        if what != 'con' and self._global[1] is None:
            self._global[1] = torch.empty(self.P*ncn, **f64)
        return self._global

    def _own_views(self, what='both'):
        """This rank's shard as views of the global vectors (None for the
        part ``what`` leaves out)."""
        con, jac = self._global_buffers(what)
        return (con.view(self.M, self.N - 1)[:, self.a:self.b]
                if what != 'jac' else None,
                jac[self.a*self.P:self.b*self.P] if what != 'con' else None)

    # -- evaluation (no collective) ----------------------------------------------
    def _hip_evaluate(self, free, con2d, jac1d, a, b, what='both'):
        import torch
        from . import hip_backend as hb
        # the kernels run on torch's current stream, so that the exchange and
        # the copies that follow (torch / RCCL ops) are ordered behind them
        stream = hb.torch_stream_pointer(
            torch.cuda.current_stream(self.device))
        if stream != self._stream:
            self.collocator.hip.set_stream(stream)
            self._stream = stream
        sel = {'both': hb.EVAL_FUSED, 'con': hb.EVAL_CON, 'jac': hb.EVAL_JAC}
        self.collocator.hip.eval_shard(
            sel[what], free, con2d if what != 'jac' else None,
            con2d.stride(0), jac1d if what != 'con' else None, a, b)

    def evaluate(self, free, in_place=False, what='both'):
        """Constraints and Jacobian (``what``: ``'both'``, ``'con'`` or
        ``'jac'``) of this rank's nodes from the global ``free`` tensor (on
        this rank's device).  Returns ``(con, jac)``:
        ``con`` is ``(M, b - a)`` (row ``j`` = equation ``j``), ``jac`` the
        slice ``[a*P, b*P)`` of the global value vector.  ``in_place``: write
        the shard directly into this rank's copy of the global vectors (what
        a gather destination does, so that its own share is never copied)."""
        if free.numel() != self._num_free():
            raise ValueError('free must have {} entries, got {}'.format(
                self._num_free(), free.numel()))
        con, jac = self._own_views() if in_place else \
            (self.con_local, self.jac_local)
        if what == 'both':
            self._evaluate(free, con, jac, self.a, self.b)
        else:
            self._evaluate(free, con, jac, self.a, self.b, what)
        self._in_place = bool(in_place)
        return con, jac

    def _num_free(self):
        if self.collocator is not None:
            return self.collocator.num_free
        return self._free_size

    def set_num_free(self, count):
        """Length of the global free vector (injected evaluators only)."""
        self._free_size = int(count)

    def broadcast_free(self, free, src=0):
        """RCCL broadcast of the global free vector from rank ``src`` (18 MB
        for config 4).  The alternative with ``free`` on the host: every rank
        loads it over its own PCIe link from a :class:`SharedHostVector`."""
        import torch.distributed as dist
        dist.broadcast(free, src, group=self.group)
        return free

    # -- re-assembly (the only communication) ---------------------------------------
    def _exchange(self, dsts, what='both'):
        """Every rank sends its shards (``what``: both, ``'con'`` or ``'jac'``)
        to every rank in ``dsts`` (but itself); destinations receive the
        Jacobian slices in place and the constraint blocks into staging.  One
        batch of point-to-point ops: shard sizes differ by up to one node,
        which an all-gather of equal pieces cannot express without padding
        copies."""
        import torch.distributed as dist
        want_con, want_jac = what != 'jac', what != 'con'
        recvs, sends = [], []               # (device tensor, peer)
        if self.rank in dsts:
            con_g, jac_g = self._global_buffers(what)
            if not self._in_place:
                own_con, own_jac = self._own_views(what)
                if want_con:
                    own_con.copy_(self.con_local)
                if want_jac:
                    own_jac.copy_(self.jac_local)
            for g, (a, b) in enumerate(self.ranges):
                if g != self.rank:
                    if want_jac:
                        recvs.append((jac_g[a*self.P:b*self.P], g))
                    if want_con:
                        recvs.append((self._stage[g], g))
        src_con, src_jac = (self._own_views(what) if self._in_place
                            else (self.con_local, self.jac_local))
        if self._in_place and want_con and \
                any(d != self.rank for d in dsts):
            # the constraint shard inside the global vector is strided
            self.con_local.copy_(src_con)
            src_con = self.con_local
        for d in dsts:
            if d != self.rank:
                if want_jac:
                    sends.append((src_jac, d))
                if want_con:
                    sends.append((src_con, d))
        # gloo moves host memory only: a GPU run that rendezvoused with gloo
        # (several ranks sharing one GPU on a development box -- RCCL refuses
        # duplicate devices) stages the messages through the host
        via_host = (self.device.type == 'cuda' and
                    dist.get_backend(self.group) == 'gloo')
        landing = [(t.cpu() if via_host else t) for t, _ in recvs]
        ops = [dist.P2POp(dist.irecv, buf, g, self.group)
               for buf, (_, g) in zip(landing, recvs)]
        ops += [dist.P2POp(dist.isend, t.cpu() if via_host else t, d,
                           self.group) for t, d in sends]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if via_host:
            for buf, (t, _) in zip(landing, recvs):
                t.copy_(buf)
        if self.rank in dsts:
            if want_con:
                con2d = con_g.view(self.M, self.N - 1)
                for g, (a, b) in enumerate(self.ranges):
                    if g != self.rank:
                        con2d[:, a:b].copy_(self._stage[g])
            return con_g, jac_g
        return None

    def gather(self, dst=0, what='both'):
        """Gather-v of the last :meth:`evaluate` to rank ``dst``: returns the
        full equation-major constraint vector and node-major Jacobian value
        vector there (device tensors owned by this object, overwritten by the
        next call; the one ``what`` leaves out is None), ``None`` on the
        other ranks."""
        return self._exchange([dst], what)

    def all_gather(self):
        """The full vectors on every rank."""
        return self._exchange(list(range(self.world_size)))

    def to_host(self, con_host, jac_host):
        """Copies this rank's shard into the node-wide host vectors
        (:class:`SharedHostVector` of ``M*(N-1)`` and ``P*(N-1)`` doubles; None
        skips one) over this rank's own PCIe link; asynchronous on the current
        stream."""
        con, jac = (self._own_views() if self._in_place
                    else (self.con_local, self.jac_local))
        if jac_host is not None:
            jac_host.torch_view(self.a*self.P, self.b*self.P).copy_(
                jac, non_blocking=True)
        if con_host is None:
            return
        ncn = self.N - 1
        dst = con_host.torch_view().view(self.M, ncn)[:, self.a:self.b]
        if self._in_place:
            self.con_local.copy_(con)
            con = self.con_local
        # M row segments of the equation-major vector
        dst.copy_(con, non_blocking=True)

    # -- host conveniences (NumPy in, NumPy out on every rank) -----------------------
    def _as_device(self, free_global):
        import torch
        return torch.as_tensor(np.ascontiguousarray(free_global,
                                                    dtype=np.float64),
                               device=self.device)

    def constraints_local(self, free_global):
        """``(M, b - a)`` equation-major shard of ``constraints(free)``."""
        con, _ = self.evaluate(self._as_device(free_global))
        return con.cpu().numpy()

    def jacobian_local(self, free_global):
        """Contiguous slice ``[a*P, b*P)`` of ``jacobian(free)``."""
        _, jac = self.evaluate(self._as_device(free_global))
        return jac.cpu().numpy()

    def constraints(self, free_global):
        """Full equation-major ``constraints(free)`` on every rank."""
        self.evaluate(self._as_device(free_global))
        return self.all_gather()[0].cpu().numpy()

    def jacobian(self, free_global):
        """Full node-major ``jacobian(free)`` on every rank."""
        self.evaluate(self._as_device(free_global))
        return self.all_gather()[1].cpu().numpy()

    def jacobian_indices_local(self):
        """Global int64 COO indices of this rank's Jacobian slice, from the
        closed-form index kernel restricted to this rank's nodes."""
        from . import hip_backend as hb
        hip = self.collocator.hip
        count = (self.b - self.a)*self.P
        rows = np.empty(count, dtype=np.int64)
        cols = np.empty(count, dtype=np.int64)
        hip.jacobian_indices_range(self.a, self.b, rows, cols, hb.HOST)
        return rows, cols


class ShardedCallbacks(object):
    """``constraints(free)`` / ``jacobian(free)`` for a host-side NLP solver,
    served by all ranks of a node-sharded problem.

    The solver (IPOPT) runs in ONE process, rank ``root``; the other ranks
    call :meth:`serve` and wait.  A callback on the root broadcasts a command
    and ``free``; every rank evaluates its node range and copies its Jacobian
    shard straight into a page-locked host vector shared by all processes --
    the host-visible rate scales with the number of PCIe links instead of
    funnelling the 792 MB Jacobian of BASELINE config 4 through one
    (``to_host``).  The small vectors take the cheapest
    way: over RCCL ``free`` is uploaded once and broadcast GPU to GPU, the
    constraint shards are gathered to the root's GPU; under gloo both go
    through shared host vectors as well.  Layouts are the reference's
    (``opty/direct_collocation.py:2446``, ``:2885-2887``): ``constraints``
    returns a fresh array, ``jacobian`` the persistent shared buffer.
    """

    _STOP, _CON, _JAC, _BOTH = 0, 1, 2, 3

    def __init__(self, sharded, name=None, root=0, pin=True,
                 fresh_constraints=True, jac_host=None):
        import torch
        import torch.distributed as dist
        sh = self.sh = sharded
        # False: constraints(free) may return a buffer the next call
        # overwrites instead of a fresh array (cyipopt copies the result
        # anyway)
        self.fresh_constraints = bool(fresh_constraints)
        self.root = root
        self.is_root = sh.rank == root
        name = name or 'opty_cb_%d' % os.getppid()
        ncn = sh.N - 1
        nfree = sh._num_free()
        gpu = sh.device.type == 'cuda'
        pin = bool(pin and gpu)
        # Over RCCL the small vectors travel GPU to GPU: `free` (18 MB for
        # config 4) goes up once on the root and is broadcast over xGMI, the
        # constraint shards (2 MB each) are gathered to the root's GPU and
        # come down as one fresh array -- a single-threaded host memcpy of
        # either costs more than that.  Only the Jacobian, the 792 MB that
        # matter, goes through the shared page-locked host vector, every
        # shard over its own PCIe link.  Under gloo (CPU tests, several ranks
        # on one GPU) everything goes through shared host vectors.
        self._rccl = gpu and dist.get_backend(sh.group) == 'nccl'
        self.free_host = self.con_host = None
        if not self._rccl:
            self.free_host = SharedHostVector(name + '_free', nfree, sh.rank,
                                              sh.group, root, pin)
            self.con_host = SharedHostVector(name + '_con', sh.M*ncn,
                                             sh.rank, sh.group, root, pin)
        # ``jac_host``: an existing shared vector of P*(N-1) doubles to use
        # (the caller keeps ownership)
        self._own_jac_host = jac_host is None
        self.jac_host = jac_host if jac_host is not None else \
            SharedHostVector(name + '_jac', sh.P*ncn, sh.rank, sh.group, root,
                             (sh.a*sh.P, sh.b*sh.P) if pin else False)
        self.num_free = nfree
        self.free_dev = torch.empty(nfree, dtype=torch.float64,
                                    device=sh.device)
        self._cmd = torch.zeros(1, dtype=torch.int64,
                                device=sh.device if self._rccl else 'cpu')
        self._pending = self._con_dev = None
        self._dist = dist

    # -- one evaluation, on every rank ------------------------------------------
    def _round(self, cmd):
        import torch
        what = {self._CON: 'con', self._JAC: 'jac', self._BOTH: 'both'}[cmd]
        if self._rccl:
            if self.is_root:
                self.free_dev.copy_(torch.from_numpy(self._pending))
            self._dist.broadcast(self.free_dev, self.root,
                                 group=self.sh.group)
        else:
            self.free_dev.copy_(self.free_host.torch_view(),
                                non_blocking=True)
        self.sh.evaluate(self.free_dev, what=what)
        if what != 'jac':
            if self._rccl:
                got = self.sh.gather(self.root, what='con')
                if self.is_root:
                    self._con_dev = got[0]
            else:
                self.sh.to_host(self.con_host, None)
        if what != 'con':
            self.sh.to_host(None, self.jac_host)
        if self.sh.device.type == 'cuda':
            torch.cuda.synchronize(self.sh.device)
        self._dist.barrier(self.sh.group)      # every shard has landed

    def _command(self, cmd):
        self._cmd.fill_(cmd)
        self._dist.broadcast(self._cmd, self.root, group=self.sh.group)
        return int(self._cmd.item())

    # -- the solver's side (rank `root`) -------------------------------------------
    def _call(self, free, cmd):
        assert self.is_root, 'callbacks run on the root rank; others serve()'
        free = np.ascontiguousarray(free, dtype=np.float64)
        if free.shape != (self.num_free,):
            raise ValueError('free must have shape ({},), got {}'.format(
                self.num_free, free.shape))
        if self._rccl:
            self._pending = free
        else:
            self.free_host.array[:] = free
        self._command(cmd)
        self._round(cmd)
        self._pending = None

    def _constraints_result(self):
        if self._rccl:
            return self._con_dev.cpu().numpy()       # fresh, as :2444
        if not self.fresh_constraints:
            return self.con_host.array
        return np.array(self.con_host.array)         # fresh, as :2444

    def constraints(self, free):
        self._call(free, self._CON)
        return self._constraints_result()

    def jacobian(self, free):
        self._call(free, self._JAC)
        return self.jac_host.array                   # persistent, as :2814

    def constraints_and_jacobian(self, free):
        self._call(free, self._BOTH)
        return self._constraints_result(), self.jac_host.array

    def shutdown(self):
        """Releases the serving ranks (root only; idempotent)."""
        if self.is_root and self._cmd is not None:
            self._command(self._STOP)
        self.close()

    # -- the other ranks ---------------------------------------------------------------
    def serve(self):
        """Evaluates on command until the root shuts the service down."""
        assert not self.is_root
        while True:
            cmd = self._command(0)
            if cmd == self._STOP:
                break
            self._round(cmd)
        self.close()

    def close(self):
        self._cmd = None
        for v in (self.free_host, self.con_host,
                  self.jac_host if self._own_jac_host else None):
            if v is not None:
                v.close()
