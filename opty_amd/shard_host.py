"""Node sharding of ONE collocation problem without PyTorch: the "thin ctypes"
host of BASELINE.json's ``north_star``.

:mod:`opty_amd.sharded` serves callers that live in PyTorch (device tensors,
``torch.distributed`` groups).  This module does the same job with nothing but
NumPy, ``ctypes`` and sockets:

* device memory through the C ABI (``opty_hip_device_alloc`` /
  ``opty_hip_memcpy``: :class:`opty_amd.hip_backend.DeviceVector`);
* evaluation of this rank's node range straight from the global ``free`` in
  its HBM (``opty_hip_eval_shard``), in place into the global vectors on the
  rank that assembles them -- no collective on the data path
  (``opty/direct_collocation.py:2145, 2153-2155, 2411-2413``: constraint node
  ``i`` reads time nodes ``i`` and ``i + 1`` only);
* re-assembly through the library's own RCCL communicator
  (``opty_hip_bcast_free`` / ``opty_hip_gather_v``: :class:`RcclTransport`),
  bootstrapped over a plain TCP side channel (:class:`SocketTransport`, which
  hands round ``opty_hip_comm_unique_id``) -- or, for host-memory shards (the
  CPU tests inject an oracle-backed evaluator), through that side channel
  itself;
* the node-wide page-locked host vector of :mod:`opty_amd.sharded`
  (``SharedHostVector(transport=...)``).

``launch_env()`` reads the rank layout the way ``torch.distributed.run`` (or
any launcher) exports it: ``RANK``, ``WORLD_SIZE``, ``LOCAL_RANK``,
``MASTER_ADDR``, ``MASTER_PORT``.
"""

import os
import pickle
import socket
import struct
import time

import numpy as np

from .sharded import partition_nodes

__all__ = ['SocketTransport', 'RcclTransport', 'NodeShard', 'launch_env']


def launch_env():
    """``(rank, world, local_rank, addr, port)`` from the environment a
    launcher exports (defaults: a world of one on 127.0.0.1:29500)."""
    env = os.environ
    return (int(env.get('RANK', 0)), int(env.get('WORLD_SIZE', 1)),
            int(env.get('LOCAL_RANK', env.get('RANK', 0))),
            env.get('MASTER_ADDR', '127.0.0.1'),
            int(env.get('MASTER_PORT', 29500)))


def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)))
    sock.sendall(payload)


def _recv(sock):
    head = _recv_exact(sock, 8)
    return _recv_exact(sock, struct.unpack('<Q', head)[0])


def _recv_exact(sock, count):
    buf = bytearray(count)
    view, got = memoryview(buf), 0
    while got < count:
        n = sock.recv_into(view[got:], count - got)
        if n == 0:
            raise ConnectionError('peer closed the side channel')
        got += n
    return bytes(buf)


class SocketTransport(object):
    """Ranks of one node (or a few) around a TCP hub on rank 0: the side
    channel of a torch-free launch -- it carries the RCCL unique id, status
    words and barriers -- and, for HOST-memory shards, the data itself
    (``bcast`` / ``gather_v`` over NumPy arrays, the same placement rules as
    ``opty_hip_gather_v``).  Every call is collective and blocking.

    ``port + 1`` is used (the launcher's ``MASTER_PORT`` itself may belong
    to a ``torch.distributed`` store of the same job).
    """

    def __init__(self, rank, world, addr='127.0.0.1', port=29500,
                 timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self._peers = {}
        self._hub = None
        if self.world == 1:
            return
        port = int(port) + 1
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            try:
                while len(self._peers) < self.world - 1:
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(timeout)
                    r = struct.unpack('<i', _recv_exact(conn, 4))[0]
                    self._peers[r] = conn
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack('<i', self.rank))
            self._hub = s

    # -- primitives: everything goes through the hub ---------------------------
    def _to_hub(self, payload):
        _send(self._hub, payload)

    def _from_hub(self):
        return _recv(self._hub)

    def bcast_bytes(self, payload, root=0):
        """``payload`` (bytes on ``root``, ignored elsewhere) on every rank."""
        if self.world == 1:
            return payload
        if self.rank == 0:
            if root != 0:
                payload = _recv(self._peers[root])
            for r, conn in self._peers.items():
                if r != root:
                    _send(conn, payload)
            return payload
        if self.rank == root:
            self._to_hub(payload)
            return payload
        return self._from_hub()

    def bcast_object(self, obj, root=0):
        return pickle.loads(self.bcast_bytes(
            pickle.dumps(obj) if self.rank == root else b'', root))

    def gather_bytes(self, payload, root=0):
        """``[payload of rank 0, ..., of rank world-1]`` on ``root``, None
        elsewhere."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = {0: payload}
            for r, conn in self._peers.items():
                parts[r] = _recv(conn)
            ordered = [parts[r] for r in range(self.world)]
            if root == 0:
                return ordered
            _send(self._peers[root], pickle.dumps(ordered))
            return None
        self._to_hub(payload)
        if self.rank == root:
            return pickle.loads(self._from_hub())
        return None

    def allreduce_min(self, value):
        parts = self.gather_bytes(struct.pack('<q', int(value)), 0)
        low = min(struct.unpack('<q', p)[0] for p in parts) \
            if self.rank == 0 else 0
        return struct.unpack('<q', self.bcast_bytes(
            struct.pack('<q', low), 0))[0]

    def barrier(self):
        self.allreduce_min(0)

    # -- data path for HOST shards (CPU tests; GPUs use RcclTransport) --------
    def bcast(self, array, root=0):
        """In-place broadcast of a C-contiguous NumPy array."""
        data = self.bcast_bytes(array.tobytes() if self.rank == root
                                else b'', root)
        if self.rank != root:
            array[...] = np.frombuffer(data, dtype=array.dtype).reshape(
                array.shape)
        return array

    def gather_v(self, bounds, con_shard, jac_shard, con_global, jac_global,
                 root, M, P):
        """Node shards to ``root`` with the placement of
        ``opty_hip_gather_v``: rank ``g``'s Jacobian values form the slice
        ``[bounds[g]*P, bounds[g+1]*P)`` of the node-major vector, its ``(M,
        b - a)`` constraint block the columns ``[a, b)`` of the
        equation-major ``(M, N-1)`` matrix (``opty/direct_collocation.py:
        2446, 2885-2887``).  A part that is None on every rank is skipped."""
        mine = pickle.dumps((
            None if con_shard is None else np.ascontiguousarray(con_shard),
            None if jac_shard is None else np.ascontiguousarray(jac_shard)))
        parts = self.gather_bytes(mine, root)
        if self.rank != root:
            return
        ncn = bounds[-1]
        for g, blob in enumerate(parts):
            con, jac = pickle.loads(blob)
            a, b = bounds[g], bounds[g + 1]
            if g == root and con is None and jac is None:
                continue                # evaluated in place
            if jac is not None and jac_global is not None:
                jac_global[a*P:b*P] = jac
            if con is not None and con_global is not None:
                con_global[:M*ncn].reshape(M, ncn)[:, a:b] = con

    def close(self):
        for conn in self._peers.values():
            conn.close()
        if self._hub is not None:
            self._hub.close()
        self._peers, self._hub = {}, None


class RcclTransport(object):
    """The data path on GPUs: the C ABI's own RCCL communicator
    (``opty_hip_comm_create``), its unique id handed round over ``side`` (a
    :class:`SocketTransport`).  ``bcast`` / ``gather_v`` take device pointers
    (``DeviceVector`` or integers) and run on the problem handle's stream."""

    def __init__(self, side, device=0):
        from . import hip_backend as hb
        self.side = side
        self.rank, self.world = side.rank, side.world
        uid = side.bcast_bytes(hb.HipComm.unique_id() if side.rank == 0
                               else b'', 0)
        self.comm = hb.HipComm(uid, side.rank, side.world, device)

    def barrier(self):
        self.side.barrier()

    def close(self):
        self.comm.close()


class NodeShard(object):
    """One rank's share of a node-sharded collocation problem, torch-free.

    Parameters are those of :class:`opty_amd.ConstraintCollocator` for the
    GLOBAL problem plus ``rank`` / ``world_size`` and

    ``transport`` : :class:`RcclTransport` (device shards) or
        :class:`SocketTransport` (host shards); may be None for a world of
        one.
    ``evaluator`` / ``instance_evaluator`` : as in
        :class:`opty_amd.sharded.ShardedCollocator`, over NumPy arrays --
        ``f(free, con2d, jac1d, a, b, what)`` fills the ``(M, b - a)`` block
        and the ``(b - a)*P`` values; with one, the shard lives in host
        memory (CPU tests).  Default: the HIP kernels on ``device``.

    The known maps are re-read on every :meth:`evaluate`, as the reference
    re-reads them on every call (``opty/direct_collocation.py:2891-2926``).
    """

    def __init__(self, equations_of_motion, state_symbols,
                 num_collocation_nodes, node_time_interval,
                 known_parameter_map={}, known_trajectory_map={},
                 instance_constraints=None, rank=0, world_size=1,
                 transport=None, device=0, evaluator=None,
                 instance_evaluator=None, **kwargs):
        from .direct_collocation import ConstraintCollocator
        if kwargs.get('jacobian_layout', 'coo') != 'coo':
            raise NotImplementedError('only the node-major layout is '
                                      'node-sharded')
        self.rank, self.world_size = int(rank), int(world_size)
        if self.world_size > 1 and transport is None:
            raise ValueError('a world of %d needs a transport'
                             % self.world_size)
        self.transport = transport
        self.N = int(num_collocation_nodes)
        self.ranges = partition_nodes(self.N - 1, self.world_size)
        self.bounds = [a for a, _ in self.ranges] + [self.ranges[-1][1]]
        self.a, self.b = self.ranges[self.rank]
        if self.b <= self.a:
            raise ValueError('more ranks than constraint nodes')
        self._hip_mode = evaluator is None
        if self._hip_mode:
            kwargs.setdefault('device', int(device))
        col = self.collocator = ConstraintCollocator(
            equations_of_motion, state_symbols, num_collocation_nodes,
            node_time_interval, known_parameter_map, known_trajectory_map,
            instance_constraints,
            launch_nodes=max(b - a for a, b in self.ranges), **kwargs)
        self.device = int(device)
        self.M = col.num_eom
        self.P = col._build_program().P
        self.o = col.num_instance_constraints
        self.nnz_inst = len(col._inst_rows)
        self.num_free = col.num_free
        self.callable_known = any(callable(v)
                                  for v in known_trajectory_map.values())
        if not self._hip_mode and self.o and instance_evaluator is None:
            raise ValueError('instance_evaluator is needed with an evaluator '
                             'for a problem with instance constraints')
        self._evaluate = evaluator
        self._evaluate_instance = instance_evaluator
        self._free = None               # this rank's copy of the free vector
        self._local = None              # (con (M, cnt), jac (cnt*P)) shards
        self._global = None             # (con, jac) on an assembling rank
        self._in_place = False

    # -- sizes ----------------------------------------------------------------
    @property
    def num_constraints(self):
        return self.M*(self.N - 1) + self.o

    @property
    def nnz(self):
        return self.P*(self.N - 1) + self.nnz_inst

    # -- memory -----------------------------------------------------------------
    def _alloc(self, count):
        if not self._hip_mode:
            return np.empty(count)
        from . import hip_backend as hb
        return hb.DeviceVector(np.empty(0), self.device) if count == 0 \
            else _device_empty(count, self.device)

    def _free_buffer(self):
        if self._free is None:
            self._free = self._alloc(self.num_free)
        return self._free

    def _local_buffers(self):
        if self._local is None:
            cnt = self.b - self.a
            self._local = (self._alloc(self.M*cnt), self._alloc(cnt*self.P))
        return self._local

    def _global_buffers(self):
        if self._global is None:
            self._global = (self._alloc(self.num_constraints),
                            self._alloc(self.nnz))
        return self._global

    # -- the free vector ----------------------------------------------------------
    def set_free(self, free_host):
        """Installs the global free vector on this rank (host array; uploaded
        over this rank's own PCIe link in device mode)."""
        free_host = np.ascontiguousarray(free_host, dtype=np.float64)
        if free_host.size != self.num_free:
            raise ValueError('free must have {} entries, got {}'.format(
                self.num_free, free_host.size))
        buf = self._free_buffer()
        if self._hip_mode:
            if self.collocator._hip is not None:
                # launches that still read the previous vector
                self.collocator._hip.synchronize()
            _upload(buf, free_host)
        else:
            buf[:] = free_host
        self._free_host = free_host

    def broadcast_free(self, free_host=None, root=0):
        """The global free vector from rank ``root`` to every rank: uploaded
        there, then ``opty_hip_bcast_free`` (RCCL) -- or the side channel for
        host shards."""
        if self.rank == root:
            self.set_free(free_host)
        buf = self._free_buffer()
        if self.world_size == 1:
            return
        if self._hip_mode:
            self.transport.comm.bcast_free(self.collocator.hip, buf, root)
            if self.callable_known:
                self.collocator.hip.synchronize()
                self._free_host = buf.numpy()
        else:
            self.transport.bcast(buf, root)
            self._free_host = buf

    # -- evaluation (no collective) -------------------------------------------------
    def evaluate(self, what='both', in_place=False, sync=True):
        """This rank's node range from the installed free vector.  Returns
        ``(con, jac)`` -- host mode: the ``(M, b - a)`` array and the ``(b -
        a)*P`` values; device mode: ``DeviceVector``s (or, ``in_place``, the
        addresses inside this rank's global vectors).  ``sync=False`` leaves
        the launch in flight on the handle's stream (timing loops;
        :meth:`gather` and ``collocator.hip.synchronize()`` wait for it)."""
        if self._free is None:
            raise ValueError('no free vector: set_free() / broadcast_free()')
        cnt, ncn = self.b - self.a, self.N - 1
        self._in_place = bool(in_place)
        if not self._hip_mode:
            if in_place:
                gc, gj = self._global_buffers()
                con = gc[:self.M*ncn].reshape(self.M, ncn)[:, self.a:self.b]
                jac = gj[self.a*self.P:self.b*self.P]
            else:
                lc, jac = self._local_buffers()
                con = lc.reshape(self.M, cnt)
            self._evaluate(self._free, con if what != 'jac' else None,
                           jac if what != 'con' else None, self.a, self.b,
                           what)
            return con, jac
        from . import hip_backend as hb
        col = self.collocator
        hip = col.hip
        col._sync_known(hip, self._free_host if self.callable_known
                        else None)
        sel = {'both': hb.EVAL_FUSED, 'con': hb.EVAL_CON,
               'jac': hb.EVAL_JAC}[what]
        if in_place:
            gc, gj = self._global_buffers()
            con, stride = gc.data_ptr() + 8*self.a, ncn
            jac = gj.data_ptr() + 8*self.a*self.P
        else:
            lc, lj = self._local_buffers()
            con, jac, stride = lc, lj, cnt
        hip.eval_shard(sel, self._free, con if what != 'jac' else None,
                       stride, jac if what != 'con' else None, self.a, self.b)
        if sync:
            hip.synchronize()
        return con, jac

    def _instance_tails(self, what):
        if not self.o:
            return
        gc, gj = self._global_buffers()
        ncn = self.N - 1
        if self._hip_mode:
            self.collocator.hip.eval_instance(
                self._free,
                gc.data_ptr() + 8*self.M*ncn if what != 'jac' else None,
                gj.data_ptr() + 8*self.P*ncn if what != 'con' else None)
        else:
            self._evaluate_instance(
                self._free, gc[self.M*ncn:] if what != 'jac' else None,
                gj[self.P*ncn:] if what != 'con' else None)

    # -- re-assembly (the only communication) -----------------------------------------
    def gather(self, root=0, what='both'):
        """Gather-v of the last :meth:`evaluate` to ``root``: the global
        constraint vector and Jacobian values there (instance tails
        included; host arrays in host mode, ``DeviceVector``s in device
        mode; the part ``what`` leaves out is None), None elsewhere."""
        want_con, want_jac = what != 'jac', what != 'con'
        is_root = self.rank == root
        lc, lj = (None, None) if (is_root and self._in_place) \
            else self._local_buffers()
        if is_root:
            gc, gj = self._global_buffers()
            self._instance_tails(what)
        if not self._hip_mode:
            cnt = self.b - self.a
            if self.world_size > 1 or not self._in_place:
                t = self.transport or SocketTransport(0, 1)
                t.gather_v(self.bounds,
                           lc.reshape(self.M, cnt) if want_con and
                           lc is not None else None,
                           lj if want_jac and lj is not None else None,
                           gc if is_root else None, gj if is_root else None,
                           root, self.M, self.P)
            return (gc if want_con else None,
                    gj if want_jac else None) if is_root else None
        from . import hip_backend as hb
        sel = {'both': hb.EVAL_PAIR, 'con': hb.EVAL_CON,
               'jac': hb.EVAL_JAC}[what]
        hip = self.collocator.hip
        if self.world_size == 1:
            hip.synchronize()
            if not self._in_place:
                # one rank: its shard IS the collocation part of the vectors
                ncn = self.N - 1
                if want_con:
                    _copy_dd(gc.data_ptr(), lc.data_ptr(), 8*self.M*ncn)
                if want_jac:
                    _copy_dd(gj.data_ptr(), lj.data_ptr(), 8*self.P*ncn)
            return (gc if want_con else None, gj if want_jac else None)
        self.transport.comm.gather_v(
            hip, self.bounds, lc if want_con else None,
            lj if want_jac else None,
            gc if (is_root and want_con) else None,
            gj if (is_root and want_jac) else None, root, sel)
        hip.synchronize()
        if is_root:
            return (gc if want_con else None, gj if want_jac else None)
        return None

    # -- host conveniences (collective) -----------------------------------------------
    def _collective(self, free_host, what, root):
        self.broadcast_free(free_host, root)
        self.evaluate(what, in_place=self.rank == root)
        out = self.gather(root, what)
        if out is None:
            return None
        pick = out[0] if what == 'con' else out[1]
        return pick.numpy() if self._hip_mode else pick.copy()

    def constraints(self, free_host=None, root=0):
        """``constraints(free)`` of the global problem on ``root`` (every
        rank calls this; ``free_host`` matters on ``root`` only)."""
        return self._collective(free_host, 'con', root)

    def jacobian(self, free_host=None, root=0):
        """``jacobian(free)`` of the global problem on ``root``."""
        return self._collective(free_host, 'jac', root)

    def jacobian_indices_local(self):
        """Global int64 COO indices of this rank's Jacobian slice."""
        from . import hip_backend as hb
        count = (self.b - self.a)*self.P
        rows = np.empty(count, dtype=np.int64)
        cols = np.empty(count, dtype=np.int64)
        self.collocator.hip.jacobian_indices_range(self.a, self.b, rows, cols,
                                                   hb.HOST)
        return rows, cols

    def close(self):
        for group in (self._local, self._global, (self._free,)):
            for v in group or ():
                if hasattr(v, 'close'):
                    v.close()
        self._local = self._global = self._free = None
        if self._hip_mode and self.collocator._hip is not None:
            self.collocator._hip.close()
            self.collocator._hip = None


def _device_empty(count, device):
    """An uninitialised ``DeviceVector`` of ``count`` doubles."""
    from . import hip_backend as hb
    v = hb.DeviceVector.__new__(hb.DeviceVector)
    v._lib = hb.load_library()
    v.size = int(count)
    v.ptr = v._lib.opty_hip_device_alloc(int(device), max(8, 8*int(count)))
    if not v.ptr:
        raise hb.HipBackendError(v._lib.opty_hip_last_error().decode())
    return v


def _upload(vector, host):
    from . import hip_backend as hb
    hb._check(vector._lib.opty_hip_memcpy(vector.ptr, host.ctypes.data,
                                          host.nbytes, 0))


def _copy_dd(dst, src, nbytes):
    from . import hip_backend as hb
    hb._check(hb.load_library().opty_hip_memcpy(dst, src, nbytes, 2))
