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
* the COO indices need no communication (closed form with the global ``N``);
* the ``o`` instance constraints (boundary conditions: every gallery problem
  has them, ``opty/direct_collocation.py:2158-2282``) are not collocation
  nodes: their values / partials follow the sharded part of the vectors
  (``:2985-2991``, ``:2686-2688``) and are evaluated once, from the global
  ``free`` every rank holds, by whichever rank assembles a vector
  (``opty_hip_eval_instance``: one lane, microseconds) -- no communication;
* known trajectories given as functions of ``free`` (``:2916-2917``) are
  evaluated on the host from the global ``free`` and uploaded before the launch.

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

    ``name`` is a prefix: the owner appends a random suffix, creates the file
    exclusively (``O_EXCL | O_NOFOLLOW``) and tells the other ranks the path
    (``broadcast_object_list``), so ranks started from different shells and
    stale or planted files of the same name are both harmless.  A failure on
    any rank (creation, mapping) reaches every rank as ``OSError``.

    ``torch_view(lo, hi)`` is a CPU tensor over ``[lo, hi)`` for
    ``copy_(device_tensor, non_blocking=True)``.  ``pin``: True page-locks
    the whole vector in this process, ``(lo, hi)`` only that element range
    (the part this rank writes; rounded out to whole pages), False nothing.
    ``device``: where the status all-reduce lives under RCCL (default: the
    current CUDA device).
    """

    _mappings = 0

    def __init__(self, name, count, rank, group=None, owner=0, pin=True,
                 device=None, transport=None):
        #: ``transport``: a torch-free side channel
        #: (``opty_amd.shard_host.SocketTransport``) instead of a
        #: ``torch.distributed`` group -- torch is not imported then
        if transport is not None:
            self._init_with_transport(name, count, rank, owner, pin,
                                      transport)
            return
        import torch.distributed as dist
        self.count = int(count)
        # identity of THIS mapping: a later vector can be mapped at the same
        # address, where a handle still believes its invariant entries to be
        # (ShardedCollocator.to_host passes `fresh` by this token)
        SharedHostVector._mappings += 1
        self.token = SharedHostVector._mappings
        self._pinned = False
        self.array = None
        multi = dist.is_available() and dist.is_initialized()
        error = None
        path = None
        if rank == owner:
            path = os.path.join('/dev/shm', '%s_%s' % (
                name, os.urandom(6).hex()))
            # reserve the pages now: a full /dev/shm raises here (ENOSPC)
            # instead of a SIGBUS at the first write
            try:
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR |
                             getattr(os, 'O_NOFOLLOW', 0), 0o600)
                try:
                    os.posix_fallocate(fd, 0, max(8, 8*self.count))
                finally:
                    os.close(fd)
                self.array = np.memmap(path, dtype=np.float64,
                                       mode='r+', shape=(self.count,))
            except OSError as err:
                error = err
                try:
                    os.unlink(path)
                except OSError:
                    pass

        def agree(err, what):
            # every rank learns whether all succeeded: a failure must not
            # leave the others waiting at a barrier
            import torch
            on_gpu = dist.get_backend(group) == 'nccl'
            where = 'cpu'
            if on_gpu:
                where = device if device is not None else torch.device(
                    'cuda', torch.cuda.current_device())
            ok = torch.tensor([0 if err else 1], dtype=torch.int32,
                              device=where)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if ok.item() == 0:
                raise OSError('could not %s the shared host vector %s '
                              '(%d bytes): %s' % (what, self.path,
                                                  8*self.count,
                                                  err or 'another rank '
                                                  'failed'))

        if multi:
            box = [path]
            dist.broadcast_object_list(
                box, src=dist.get_global_rank(group, owner)
                if group is not None else owner, group=group)
            self.path = box[0]
            agree(error, 'create')
            if rank != owner:
                try:
                    self.array = np.memmap(self.path, dtype=np.float64,
                                           mode='r+', shape=(self.count,))
                except OSError as err:
                    error = err
            agree(error, 'map')        # also: everyone has mapped the file
        else:
            self.path = path
            if error:
                raise error
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

    def _create(self, name):
        """Creates the file exclusively and maps it: ``(path, error)``."""
        path = os.path.join('/dev/shm', '%s_%s' % (name,
                                                   os.urandom(6).hex()))
        try:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR |
                         getattr(os, 'O_NOFOLLOW', 0), 0o600)
            try:
                os.posix_fallocate(fd, 0, max(8, 8*self.count))
            finally:
                os.close(fd)
            self.array = np.memmap(path, dtype=np.float64, mode='r+',
                                   shape=(self.count,))
            return path, None
        except OSError as err:
            try:
                os.unlink(path)
            except OSError:
                pass
            return path, err

    def _init_with_transport(self, name, count, rank, owner, pin, transport):
        self.count = int(count)
        SharedHostVector._mappings += 1
        self.token = SharedHostVector._mappings
        self._pinned = False
        self.array = None
        path, error = self._create(name) if rank == owner else (None, None)
        self.path = transport.bcast_object(path, owner)

        def agree(err, what):
            if transport.allreduce_min(0 if err else 1) == 0:
                raise OSError('could not %s the shared host vector %s (%d '
                              'bytes): %s' % (what, self.path, 8*self.count,
                                              err or 'another rank failed'))
        agree(error, 'create')
        if rank != owner:
            try:
                self.array = np.memmap(self.path, dtype=np.float64,
                                       mode='r+', shape=(self.count,))
            except OSError as err:
                error = err
        agree(error, 'map')
        if rank == owner:
            os.unlink(self.path)
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
    GLOBAL problem (instance constraints and known trajectories given as
    functions of ``free`` included), plus ``rank`` / ``world_size`` (default:
    from ``torch.distributed``), ``group`` and ``device`` (a ``torch.device``;
    default: the collocator's HIP device).

    ``evaluator``: ``f(free, con2d, jac1d, a, b, what='both')`` that fills the
    shard's ``(M, b - a)`` constraint block (a possibly strided view; None for
    ``what='jac'``) and its ``(b - a)*P`` Jacobian values (None for
    ``what='con'``) from the global ``free`` tensor.  Default: the HIP kernels
    (``opty_hip_eval_shard``); the CPU tests inject an oracle-backed one to
    exercise the partition and the exchange under ``gloo``.  With an
    evaluator, a problem with instance constraints also needs
    ``instance_evaluator``: ``g(free, con_tail, jac_tail)`` (either may be
    None); ``block_shape`` is checked against the problem's ``(M, P)``.

    The known maps are re-read on every :meth:`evaluate` (the reference reads
    them on every call, ``opty/direct_collocation.py:2891-2926``), callable
    known trajectories are evaluated on the host from ``free``.

    The CSR layout is not node-sharded.
    """

    def __init__(self, equations_of_motion, state_symbols,
                 num_collocation_nodes, node_time_interval,
                 known_parameter_map={}, known_trajectory_map={},
                 instance_constraints=None, rank=None, world_size=None,
                 group=None, device=None, evaluator=None, block_shape=None,
                 instance_evaluator=None, comm=None, **kwargs):
        import torch
        #: ``hip_backend.HipComm`` (or None): when given -- and the HIP
        #: evaluator is in use -- :meth:`broadcast_free` and :meth:`gather` go
        #: through the C ABI's own RCCL communicator (``opty_hip_bcast_free``
        #: / ``opty_hip_gather_v``) instead of ``torch.distributed``
        self.comm = comm
        if kwargs.get('jacobian_layout', 'coo') != 'coo':
            raise NotImplementedError(
                'only the node-major layout is node-sharded: a shard of the '
                'row-sorted (csr) or varying-first layout is not a '
                'contiguous slice of the global value vector')
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
        #: the rank that writes the instance tails of node-wide host vectors
        #: (:meth:`to_host`): the last one, whose node range is never the
        #: larger
        self.tail_rank = world_size - 1
        self._par_map = known_parameter_map
        self._traj_map = known_trajectory_map
        self.callable_known = any(callable(v)
                                  for v in known_trajectory_map.values())
        #: what injected evaluators read (set by :meth:`set_known`)
        self.known_parameters = self.known_trajectories = None
        from .direct_collocation import ConstraintCollocator
        if device is not None and torch.device(device).type == 'cuda':
            kwargs.setdefault('device', torch.device(device).index or 0)
        # The handle is built for the GLOBAL problem (its kernels read the
        # global free vector); its strip count is chosen for the shard's
        # launch size.  The symbolic side (sizes, orderings, instance index
        # map) is used with injected evaluators too; the device is not
        # touched before the first HIP evaluation.
        col = self.collocator = ConstraintCollocator(
            equations_of_motion, state_symbols, num_collocation_nodes,
            node_time_interval, known_parameter_map, known_trajectory_map,
            instance_constraints,
            launch_nodes=max(b - a for a, b in self.ranges), **kwargs)
        self.M = col.num_eom
        self.P = col._build_program().P if kwargs.get('prune_zeros') \
            else col.num_eom*col.num_block_columns
        self.o = col.num_instance_constraints
        self.nnz_inst = len(col._inst_rows)
        self._hip_mode = evaluator is None
        if self._hip_mode:
            self.device = torch.device('cuda', col._device)
            evaluator = self._hip_evaluate
            instance_evaluator = self._hip_instance
        else:
            if block_shape is not None and \
                    tuple(block_shape) != (self.M, self.P):
                raise ValueError('block_shape {} does not match the '
                                 'problem\'s (M, P) = {}'.format(
                                     tuple(block_shape), (self.M, self.P)))
            if self.o and instance_evaluator is None:
                raise ValueError('instance_evaluator is needed with an '
                                 'evaluator for a problem with instance '
                                 'constraints')
            self.device = torch.device(device or 'cpu')
        self._evaluate = evaluator
        self._evaluate_instance = instance_evaluator
        cnt = self.b - self.a
        f64 = dict(dtype=torch.float64, device=self.device)
        self.con_local = torch.empty((self.M, cnt), **f64)
        self.jac_local = torch.empty(cnt*self.P, **f64)
        self.inst_con = torch.empty(self.o, **f64)
        self.inst_jac = torch.empty(self.nnz_inst, **f64)
        self._global = None         # (con, jac) on ranks that receive
        self._stage = None          # constraint blocks of the other ranks
        self._in_place = False
        self._stream = None
        self._last_free = None

    # -- layout ---------------------------------------------------------------
    @property
    def num_local_nodes(self):
        """Constraint nodes owned by this rank."""
        return self.b - self.a

    @property
    def num_constraints(self):
        """``M*(N-1) + o``: length of the global constraint vector."""
        return self.M*(self.N - 1) + self.o

    @property
    def nnz(self):
        """``P*(N-1) + nnz_inst``: length of the global Jacobian vector."""
        return self.P*(self.N - 1) + self.nnz_inst

    def _global_buffers(self, what='both'):
        """This rank's copies of the global vectors, instance tails included
        (allocated on first use, each on its own: a constraints-only
        destination never holds the Jacobian)."""
        import torch
        f64 = dict(dtype=torch.float64, device=self.device)
        if self._global is None:
            self._global = [None, None]
        if what != 'jac' and self._global[0] is None:
            self._global[0] = torch.empty(self.num_constraints, **f64)
            self._stage = {
                g: torch.empty((self.M, b - a), **f64)
                for g, (a, b) in enumerate(self.ranges) if g != self.rank}
        if what != 'con' and self._global[1] is None:
            self._global[1] = torch.empty(self.nnz, **f64)
        return self._global

    def _own_views(self, what='both'):
        """This rank's shard as views of the global vectors (None for the
        part ``what`` leaves out, which is not allocated either)."""
        con, jac = self._global_buffers(what)
        ncn = self.N - 1
        return (con[:self.M*ncn].view(self.M, ncn)[:, self.a:self.b]
                if what != 'jac' else None,
                jac[self.a*self.P:self.b*self.P] if what != 'con' else None)

    def _tail_views(self, what='both'):
        """The instance tails of this rank's global vectors."""
        con, jac = self._global_buffers(what)
        ncn = self.N - 1
        return (con[self.M*ncn:] if what != 'jac' else None,
                jac[self.P*ncn:] if what != 'con' else None)

    # -- known maps ---------------------------------------------------------------
    def known_parameter_values(self):
        """Current values of ``known_parameter_map`` in its own (=
        ``known_parameters``) order."""
        return np.array([float(v) for v in self._par_map.values()],
                        dtype=np.float64)

    def known_trajectory_values(self, free_host):
        """``(m_known, N)`` array of ``known_trajectory_map`` in its own (=
        ``known_input_trajectories``) order; callables are evaluated at the
        host vector ``free_host`` (``opty/direct_collocation.py:2916-2917``)."""
        vals = np.array([v(free_host) if callable(v) else v
                         for v in self._traj_map.values()], dtype=np.float64)
        if vals.shape != (len(self._traj_map), self.N):
            raise ValueError('every known trajectory must have {} '
                             'values.'.format(self.N))
        return vals

    def _use_stream(self):
        """The kernels run on torch's current stream, so that the exchange and
        the copies that follow (torch / RCCL ops) are ordered behind them."""
        import torch
        from . import hip_backend as hb
        stream = hb.torch_stream_pointer(
            torch.cuda.current_stream(self.device))
        if stream != self._stream:
            self.collocator.hip.set_stream(stream)
            self._stream = stream

    def sync_known(self, free=None):
        """Brings the device copies of the known parameters / trajectories up
        to date with this rank's maps (HIP evaluator only; changed values are
        re-uploaded, callables re-evaluated from ``free``)."""
        col = self.collocator
        if not self._hip_mode:
            return
        hip = col.hip
        self._use_stream()
        host = None
        if self.callable_known and free is not None:
            host = free.detach().cpu().numpy() if hasattr(free, 'cpu') \
                else np.asarray(free)
        col._sync_known(hip, host)

    def set_known(self, params=None, traj=None):
        """Installs known-parameter values (``(p_known,)`` array) and / or
        known trajectories (``(m_known, N)`` array or device tensor) that came
        from elsewhere -- :class:`ShardedCallbacks` broadcasts the solver
        rank's values so that every rank evaluates with the same ones."""
        col = self.collocator
        if params is not None:
            params = np.array(params, dtype=np.float64)
            self.known_parameters = params
            if self._hip_mode and len(params) and col._specialize:
                # parameter-specialised kernels (requested, or chosen by
                # the collocator itself: specialize_parameters=None) carry
                # the OLD values as literals: the values go into this rank's
                # map and _sync_known prints, compiles and verifies the
                # kernels for them
                for sym, v in zip(col.known_parameters, params):
                    col.known_parameter_map[sym] = float(v)
                self._use_stream()
                col._sync_known(col.hip, None)
            elif self._hip_mode and len(params):
                self._use_stream()
                col.hip.set_known_parameters(params)
                col._uploaded_parameters = params
        if traj is not None:
            self.known_trajectories = traj
            if self._hip_mode:
                self._use_stream()
                col.hip.set_known_trajectories(traj)
                # unknown to _sync_known: the next direct evaluate() compares
                # against nothing and uploads this rank's own map again
                col._uploaded_trajectories = None

    # -- evaluation (no collective) ----------------------------------------------
    def _hip_evaluate(self, free, con2d, jac1d, a, b, what='both'):
        from . import hip_backend as hb
        self._use_stream()
        sel = {'both': hb.EVAL_FUSED, 'con': hb.EVAL_CON, 'jac': hb.EVAL_JAC}
        self.collocator.hip.eval_shard(
            sel[what], free, con2d if what != 'jac' else None,
            con2d.stride(0) if what != 'jac' else self.N - 1,
            jac1d if what != 'con' else None, a, b)

    def _hip_instance(self, free, con_tail, jac_tail):
        self._use_stream()
        self.collocator.hip.eval_instance(free, con_tail, jac_tail)

    def evaluate(self, free, in_place=False, what='both', sync=True):
        """Constraints and Jacobian (``what``: ``'both'``, ``'con'`` or
        ``'jac'``) of this rank's nodes from the global ``free`` tensor (on
        this rank's device).  Returns ``(con, jac)``:
        ``con`` is ``(M, b - a)`` (row ``j`` = equation ``j``), ``jac`` the
        slice ``[a*P, b*P)`` of the global value vector (None for the part
        ``what`` leaves out).  ``in_place``: write the shard directly into
        this rank's copy of the global vectors (what a gather destination
        does, so that its own share is never copied).  ``sync=False`` skips
        re-reading the known maps (the caller installed them,
        :meth:`set_known`).  The instance constraints are not part of a
        shard: :meth:`evaluate_instance`, or the re-assembly calls."""
        if free.numel() != self._num_free():
            raise ValueError('free must have {} entries, got {}'.format(
                self._num_free(), free.numel()))
        if sync:
            self.sync_known(free)
        con, jac = self._own_views(what) if in_place else \
            (self.con_local if what != 'jac' else None,
             self.jac_local if what != 'con' else None)
        if what == 'both':
            self._evaluate(free, con, jac, self.a, self.b)
        else:
            self._evaluate(free, con, jac, self.a, self.b, what)
        self._in_place = bool(in_place)
        self._last_free = free
        return con, jac

    def evaluate_instance(self, free=None, what='both', in_place=False):
        """The ``o`` instance-constraint values and their ``nnz_inst``
        partials from the global ``free`` (default: the vector of the last
        :meth:`evaluate`): ``(con_tail, jac_tail)``, into the tails of this
        rank's global vectors (``in_place``) or into local buffers."""
        if self.o == 0:
            return self.inst_con, self.inst_jac
        free = self._last_free if free is None else free
        if free is None:
            raise ValueError('no free vector: call evaluate() first or pass '
                             'one')
        con, jac = self._tail_views(what) if in_place else \
            (self.inst_con if what != 'jac' else None,
             self.inst_jac if what != 'con' else None)
        self._evaluate_instance(free, con, jac)
        return con, jac

    def _num_free(self):
        return self.collocator.num_free

    def broadcast_free(self, free, src=0):
        """RCCL broadcast of the global free vector from rank ``src`` (18 MB
        for config 4).  The alternative with ``free`` on the host: every rank
        loads it over its own PCIe link from a :class:`SharedHostVector`."""
        if self.comm is not None and self._hip_mode:
            self._use_stream()
            self.comm.bcast_free(self.collocator.hip, free, src)
            return free
        import torch.distributed as dist
        dist.broadcast(free, src, group=self.group)
        return free

    # -- re-assembly (the only communication) ---------------------------------------
    def _exchange(self, dsts, what='both'):
        """Every rank sends its shards (``what``: both, ``'con'`` or ``'jac'``)
        to every rank in ``dsts`` (but itself); destinations receive the
        Jacobian slices in place and the constraint blocks into staging, and
        evaluate the instance tails themselves (every rank holds the global
        ``free``).  One batch of point-to-point ops: shard sizes differ by up
        to one node, which an all-gather of equal pieces cannot express
        without padding copies."""
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
            if self.o:
                self.evaluate_instance(None, what, in_place=True)
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
                ncn = self.N - 1
                con2d = con_g[:self.M*ncn].view(self.M, ncn)
                for g, (a, b) in enumerate(self.ranges):
                    if g != self.rank:
                        con2d[:, a:b].copy_(self._stage[g])
            return con_g, jac_g
        return None

    def gather(self, dst=0, what='both'):
        """Gather-v of the last :meth:`evaluate` to rank ``dst``: returns the
        full equation-major constraint vector and node-major Jacobian value
        vector there, instance tails included (device tensors owned by this
        object, overwritten by the next call; the one ``what`` leaves out is
        None), ``None`` on the other ranks.  With a :attr:`comm` the exchange
        is the C ABI's ``opty_hip_gather_v`` (grouped ``ncclSend`` /
        ``ncclRecv`` issued by the library on the handle's stream), otherwise
        ``torch.distributed`` point-to-point."""
        if self.comm is not None and self._hip_mode:
            return self._gather_v(dst, what)
        return self._exchange([dst], what)

    def _gather_v(self, dst, what):
        from . import hip_backend as hb
        sel = {'both': hb.EVAL_PAIR, 'con': hb.EVAL_CON,
               'jac': hb.EVAL_JAC}[what]
        want_con, want_jac = what != 'jac', what != 'con'
        bounds = [a for a, _ in self.ranges] + [self.ranges[-1][1]]
        self._use_stream()
        hip = self.collocator.hip
        if self.rank != dst:
            src_con, src_jac = (self._own_views(what) if self._in_place
                                else (self.con_local, self.jac_local))
            if self._in_place and want_con:
                # the constraint shard inside the global vector is strided:
                # a message is dense
                self.con_local.copy_(src_con)
                src_con = self.con_local
            self.comm.gather_v(hip, bounds, src_con if want_con else None,
                               src_jac if want_jac else None, None, None,
                               dst, sel)
            return None
        con_g, jac_g = self._global_buffers(what)
        if self.o:
            self.evaluate_instance(None, what, in_place=True)
        own = (None, None) if self._in_place else (
            self.con_local if want_con else None,
            self.jac_local if want_jac else None)
        self.comm.gather_v(hip, bounds, own[0], own[1],
                           con_g if want_con else None,
                           jac_g if want_jac else None, dst, sel)
        return con_g, jac_g

    def all_gather(self):
        """The full vectors on every rank."""
        return self._exchange(list(range(self.world_size)))

    def to_host(self, con_host, jac_host):
        """Copies this rank's shard into the node-wide host vectors
        (:class:`SharedHostVector` of ``M*(N-1) + o`` and ``P*(N-1) +
        nnz_inst`` doubles; None skips one) over this rank's own PCIe link.
        Enqueued on the current stream; a large Jacobian shard (the HIP
        evaluator, ``_PACKED_MIN_VALUES``) is moved by
        ``opty_hip_shard_jac_to_host`` instead -- only its varying entries
        after the first time -- which returns when the shard has landed.
        Synchronise the stream before reading the host vectors either way.
        Rank :attr:`tail_rank` also evaluates and writes the instance
        tails."""
        what = 'both' if (con_host is not None and jac_host is not None) \
            else ('con' if con_host is not None else 'jac')
        con, jac = (self._own_views(what) if self._in_place
                    else (self.con_local, self.jac_local))
        ncn = self.N - 1
        tails = (None, None)
        if self.o and self.rank == self.tail_rank and \
                (con_host is not None or jac_host is not None):
            tails = self.evaluate_instance(None, what)
        if jac_host is not None:
            if self._hip_mode and jac.is_contiguous() and \
                    jac.numel() >= self._PACKED_MIN_VALUES:
                # large shards: after the first copy only the entries that
                # can change cross PCIe (opty_hip_shard_jac_to_host: packed
                # on the device, scattered into the shared vector by this
                # process's host threads); synchronous
                self._use_stream()
                token = getattr(jac_host, 'token', None)
                fresh = token is None or \
                    token != getattr(self, '_host_jac_token', None)
                self.collocator.hip.shard_jac_to_host(
                    jac, jac_host.array, self.a, self.b, fresh)
                self._host_jac_token = token
            else:
                jac_host.torch_view(self.a*self.P, self.b*self.P).copy_(
                    jac, non_blocking=True)
            if tails[1] is not None and self.nnz_inst:
                jac_host.torch_view(self.P*ncn, self.nnz).copy_(
                    tails[1], non_blocking=True)
        if con_host is None:
            return
        dst = con_host.torch_view(0, self.M*ncn).view(
            self.M, ncn)[:, self.a:self.b]
        if self._in_place:
            self.con_local.copy_(con)
            con = self.con_local
        # M row segments of the equation-major vector
        dst.copy_(con, non_blocking=True)
        if tails[0] is not None:
            con_host.torch_view(self.M*ncn, self.num_constraints).copy_(
                tails[0], non_blocking=True)

    #: Jacobian values of a shard from which :meth:`to_host` moves only the
    #: varying entries
    _PACKED_MIN_VALUES = 1 << 20

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
        """Full ``constraints(free)`` on every rank (equation-major, then the
        instance constraints)."""
        self.evaluate(self._as_device(free_global))
        return self.all_gather()[0].cpu().numpy()

    def jacobian(self, free_global):
        """Full ``jacobian(free)`` on every rank (node-major blocks, then the
        instance partials)."""
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

    def instance_indices(self):
        """Global int64 COO indices of the instance tail of the Jacobian
        (``opty/direct_collocation.py:2233-2251``, ``:2686-2688``); they
        follow the last rank's slice."""
        return self.collocator._instance_constraints_jacobian_indices()


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
    (``opty/direct_collocation.py:2446``, ``:2885-2887``, instance tails
    ``:2985-2991``): ``constraints`` returns a fresh array, ``jacobian`` the
    persistent shared buffer.

    The ROOT's ``known_parameter_map`` / ``known_trajectory_map`` are the ones
    that count (the solver's process is where a user changes them between
    solves, ``plot_human_gait.py``): they are re-read on every call as the
    reference does (``:2891-2926``), and values that changed -- and the values
    of callable known trajectories, evaluated on the root from ``free``
    (``:2916-2917``) -- are broadcast with the command.
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
        name = name or 'opty_cb'
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
        shm = dict(group=sh.group, owner=root,
                   device=sh.device if gpu else None)
        if not self._rccl:
            self.free_host = SharedHostVector(name + '_free', nfree, sh.rank,
                                              pin=pin, **shm)
            self.con_host = SharedHostVector(name + '_con',
                                             sh.num_constraints, sh.rank,
                                             pin=pin, **shm)
        # ``jac_host``: an existing shared vector of P*(N-1) + nnz_inst
        # doubles to use (the caller keeps ownership)
        self._own_jac_host = jac_host is None
        self.jac_host = jac_host if jac_host is not None else \
            SharedHostVector(name + '_jac', sh.nnz, sh.rank,
                             pin=(sh.a*sh.P, sh.b*sh.P) if pin else False,
                             **shm)
        if self.jac_host.count != sh.nnz:
            raise ValueError('jac_host must hold {} doubles, has {}'.format(
                sh.nnz, self.jac_host.count))
        self.num_free = nfree
        self.free_dev = torch.empty(nfree, dtype=torch.float64,
                                    device=sh.device)
        # where broadcasts live: the GPU under RCCL, host memory under gloo
        self._bdev = sh.device if self._rccl else torch.device('cpu')
        self._cmd = torch.zeros(3, dtype=torch.int64, device=self._bdev)
        self._par_buf = torch.empty(len(sh._par_map), dtype=torch.float64,
                                    device=self._bdev)
        self._traj_buf = torch.empty((len(sh._traj_map), sh.N),
                                     dtype=torch.float64, device=self._bdev)
        self._sent_params = self._sent_traj = None
        self._pending = self._con_dev = None
        self._dist = dist

    # -- one evaluation, on every rank ------------------------------------------
    def _known_updates(self, free):
        """Root: ``(params, traj)`` to broadcast with this call (None = the
        other ranks already hold the current values)."""
        sh = self.sh
        params = traj = None
        if len(sh._par_map):
            now = sh.known_parameter_values()
            if self._sent_params is None or \
                    not np.array_equal(now, self._sent_params):
                params = self._sent_params = now
        if len(sh._traj_map):
            now = sh.known_trajectory_values(free)
            if sh.callable_known or self._sent_traj is None or \
                    not np.array_equal(now, self._sent_traj):
                traj = self._sent_traj = now
        return params, traj

    def _share_known(self, flags, params, traj):
        import torch
        sh = self.sh
        for flag, buf, val in ((flags[0], self._par_buf, params),
                               (flags[1], self._traj_buf, traj)):
            if not flag:
                continue
            if self.is_root:
                buf.copy_(torch.from_numpy(val).view_as(buf))
            self._dist.broadcast(buf, self.root, group=sh.group)
        if flags[0]:
            sh.set_known(params=self._par_buf.cpu().numpy())
        if flags[1]:
            sh.set_known(traj=self._traj_buf if self._rccl
                         else self._traj_buf.numpy())

    def _round(self, cmd, flags, params=None, traj=None):
        import torch
        what = {self._CON: 'con', self._JAC: 'jac', self._BOTH: 'both'}[cmd]
        self._share_known(flags, params, traj)
        if self._rccl:
            if self.is_root:
                self.free_dev.copy_(torch.from_numpy(self._pending))
            self._dist.broadcast(self.free_dev, self.root,
                                 group=self.sh.group)
        else:
            self.free_dev.copy_(self.free_host.torch_view(),
                                non_blocking=True)
        self.sh.evaluate(self.free_dev, what=what, sync=False)
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

    def _command(self, cmd, flags=(0, 0)):
        """Broadcasts ``(cmd, parameters follow, trajectories follow)`` from
        the root; returns what arrived."""
        if self.is_root:
            self._cmd.copy_(self._cmd.new_tensor([cmd, flags[0], flags[1]]))
        self._dist.broadcast(self._cmd, self.root, group=self.sh.group)
        got = self._cmd.tolist()
        return int(got[0]), (int(got[1]), int(got[2]))

    # -- the solver's side (rank `root`) -------------------------------------------
    def _call(self, free, cmd):
        assert self.is_root, 'callbacks run on the root rank; others serve()'
        free = np.ascontiguousarray(free, dtype=np.float64)
        if free.shape != (self.num_free,):
            raise ValueError('free must have shape ({},), got {}'.format(
                self.num_free, free.shape))
        params, traj = self._known_updates(free)
        if self._rccl:
            self._pending = free
        else:
            self.free_host.array[:] = free
        flags = (int(params is not None), int(traj is not None))
        self._command(cmd, flags)
        self._round(cmd, flags, params, traj)
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
            cmd, flags = self._command(0)
            if cmd == self._STOP:
                break
            self._round(cmd, flags)
        self.close()

    def close(self):
        self._cmd = None
        for v in (self.free_host, self.con_host,
                  self.jac_host if self._own_jac_host else None):
            if v is not None:
                v.close()
