"""Prints a :class:`~opty_amd.codegen.program.CollocationProgram` as HIP source
for gfx950.

The printed module is the GPU counterpart of the ``_c.c`` / ``.pyx`` pair the
reference emits (``opty/utils.py:483-529``): instead of a scalar
``eval_matrix`` called from an OpenMP node loop it contains wave-per-64-nodes
kernels built on ``csrc/opty_device.h``:

``opty_uni``      one lane evaluates every *node-invariant* sub-expression
                  (anything that depends on parameters / h only -- the
                  reference recomputes those per node)
                  into a small table ``uni[]`` that the other kernels read
                  with scalar loads; it only has to run when parameters, h or
                  (for unknown parameters / variable duration) ``free`` change
``opty_con``      constraints, equation-major coalesced stores
``opty_jac``      Jacobian blocks, staged through an LDS tile in chunks of
                  ``KC`` entries and flushed node-major with 16-byte stores;
                  the P entries of the block are split into ``G`` contiguous
                  entry ranges, each evaluated by its own wave (``G`` waves per
                  64 nodes) so that a 10^5-node problem yields enough waves to
                  fill 256 CUs and each wave's live register set stays small
``opty_conjac``   both outputs from one launch (shared sub-expressions)
``opty_inst``     instance-constraint values and partials (one lane)

Register pressure is the hard part: a 10-link pendulum
Jacobian has thousands of temporaries.  The printer therefore

* keeps per-node *inputs* in an LDS slab for the whole kernel and re-reads
  them per chunk instead of holding ~2n+m doubles in VGPRs,
* re-loads node-invariant values per chunk through scalar loads (SGPRs),
* visits outputs in memory order and emits each output's not-yet-available
  operands depth-first right before it, so temporaries are born late.
"""

import hashlib

from . import ir

WAVE = 64
TS = 65
LINE_MODE_MIN_P = 64

KERNEL_PARAMS = (
    'const double *__restrict__ free_, const double *__restrict__ known_traj, '
    'const double *__restrict__ params, const double *__restrict__ uni_c, '
    'double *__restrict__ uni_w, const long long *__restrict__ inst_idx, '
    'double *__restrict__ con, double *__restrict__ jac, double h, '
    'long long N, long long con_stride, long long node_begin, '
    'long long node_end')


class EmitOptions(object):
    """Knobs of the printer.

    chunk : entries of the per-node block staged per LDS tile flush (even)
    groups : waves per 64-node block for the Jacobian kernels (None = auto)
    max_live : auto-grouping target for the number of simultaneously live
        float64 temporaries of one wave
    """

    def __init__(self, chunk=32, groups=None, max_live=100, ablate=None,
                 flush_unroll=4):
        self.flush_unroll = int(flush_unroll)
        self.chunk = int(chunk)
        assert self.chunk % 2 == 0 and self.chunk >= 2
        # blocks with P >= LINE_MODE_MIN_P use the line-aligned ring flush,
        # which needs chunk to be a multiple of the 16 doubles of a line
        self.groups = groups
        self.max_live = int(max_live)
        # profiling aids (never used by the product): 'store_only' writes a
        # lane-dependent dummy instead of evaluating the expressions,
        # 'compute_only' predicates every flush store off
        assert ablate in (None, 'store_only', 'compute_only')
        self.ablate = ablate

    def key(self):
        return 'chunk=%d groups=%s max_live=%d ablate=%s flush_unroll=%d' % (
            self.chunk, self.groups, self.max_live, self.ablate,
            self.flush_unroll)


def _lit(v):
    if v != v:
        return '__builtin_nan("")'
    if v in (float('inf'), float('-inf')):
        return ('-' if v < 0 else '') + '__builtin_inf()'
    s = repr(float(v))
    if 'e' not in s and '.' not in s and 'n' not in s:
        s += '.0'
    return s


class _Body(object):
    """Emits straight-line code for DAG nodes.

    ``leaf(i)`` decides whether node ``i`` is a *leaf* of this body -- a value
    that is fetched (LDS / scalar load) rather than computed -- and returns
    the C expression that fetches it, or None.  Leaves are re-fetched in every
    scope (``new_scope``), computed temporaries are emitted once.
    """

    def __init__(self, dag, needed, leaf):
        self.dag = dag
        self.lines = []
        self.done = {}
        self.scope = {}
        self.scope_id = 0
        self.needed = needed
        self.leaf = leaf

    def new_scope(self):
        self.scope = {}
        self.scope_id += 1

    def ref(self, i):
        d = self.dag
        if d.op[i] == ir.CONST:
            v = d.value(i)
            return _lit(v) if v >= 0 else '(%s)' % _lit(v)
        if i in self.scope:
            return self.scope[i]
        return self.done[i]

    def _have(self, i):
        return i in self.done or i in self.scope or \
            self.dag.op[i] == ir.CONST

    def _fetch(self, i):
        src = self.leaf(i)
        if src is None:
            return False
        name = 'f%d_%d' % (i, self.scope_id)
        self.lines.append('const double %s = %s;' % (name, src))
        self.scope[i] = name
        return True

    def emit(self, root):
        """Makes sure ``root`` is available; returns the C expression naming
        it."""
        d = self.dag
        stack = [(root, False)]
        while stack:
            i, ready = stack.pop()
            if self._have(i):
                continue
            if not ready:
                if self.leaf(i) is not None:
                    self._fetch(i)
                    continue
                stack.append((i, True))
                for j in reversed(d.operands(i)):
                    if not self._have(j):
                        stack.append((j, False))
                continue
            self._emit_node(i)
        return self.ref(root)

    def _emit_node(self, i):
        d = self.dag
        op = d.op[i]
        a = d.args[i]
        name = 'v%d' % i
        r = self.ref
        if op == ir.ADD:
            e = '%s + %s' % (r(a[0]), r(a[1]))
        elif op == ir.SUB:
            e = '%s - %s' % (r(a[0]), r(a[1]))
        elif op == ir.MUL:
            e = '%s*%s' % (r(a[0]), r(a[1]))
        elif op == ir.DIV:
            e = '%s/%s' % (r(a[0]), r(a[1]))
        elif op == ir.NEG:
            e = '-%s' % r(a[0])
        elif op == ir.POWI:
            x, n = r(a[0]), a[1]
            if n == 2:
                e = '%s*%s' % (x, x)
            elif n == 3:
                e = '%s*%s*%s' % (x, x, x)
            else:
                e = 'opty_powi<%d>(%s)' % (n, x)
        elif op == ir.POW:
            e = 'pow(%s, %s)' % (r(a[0]), r(a[1]))
        elif op == ir.MAX:
            e = 'fmax(%s, %s)' % (r(a[0]), r(a[1]))
        elif op == ir.MIN:
            e = 'fmin(%s, %s)' % (r(a[0]), r(a[1]))
        elif op == ir.ATAN2:
            e = 'atan2(%s, %s)' % (r(a[0]), r(a[1]))
        elif op in ('sin', 'cos'):
            other = 'cos' if op == 'sin' else 'sin'
            j = d._memo.get((other, a))
            if (j is not None and j in self.needed and not self._have(j)
                    and self.leaf(j) is None):
                s_id, c_id = (i, j) if op == 'sin' else (j, i)
                self.lines.append('double v%d, v%d; sincos(%s, &v%d, &v%d);'
                                  % (s_id, c_id, r(a[0]), s_id, c_id))
                self.done[s_id] = 'v%d' % s_id
                self.done[c_id] = 'v%d' % c_id
                return
            e = '%s(%s)' % (op, r(a[0]))
        elif op == 'abs':
            e = 'fabs(%s)' % r(a[0])
        elif op == 'sign':
            x = r(a[0])
            e = '(double)((%s > 0.0) - (%s < 0.0))' % (x, x)
        elif op == 'step':
            e = '(%s > 0.0 ? 1.0 : 0.0)' % r(a[0])
        elif op in ir.UNARY:
            e = '%s(%s)' % (op, r(a[0]))
        else:
            raise NotImplementedError(op)
        self.lines.append('const double %s = %s;' % (name, e))
        self.done[i] = name


def _max_live(dag, chunks, is_leaf):
    """Largest number of simultaneously live computed temporaries when the
    chunks' roots are emitted depth-first in order (what ``_Body`` does)."""
    need = set()
    stack = [r for c in chunks for r in c]
    while stack:
        i = stack.pop()
        if i in need or dag.op[i] == ir.CONST or is_leaf(i):
            continue
        need.add(i)
        stack.extend(dag.operands(i))
    uses = dict.fromkeys(need, 0)
    for i in need:
        for j in set(dag.operands(i)):
            if j in uses:
                uses[j] += 1
    for c in chunks:
        for r in c:
            if r in uses:
                uses[r] += 1
    done, live, peak = set(), 0, 0
    for c in chunks:
        for root in c:
            if root not in uses:
                continue
            stack = [(root, False)]
            while stack:
                i, ready = stack.pop()
                if i in done or i not in uses:
                    continue
                if not ready:
                    stack.append((i, True))
                    for j in reversed(dag.operands(i)):
                        if j in uses and j not in done:
                            stack.append((j, False))
                    continue
                done.add(i)
                live += 1
                peak = max(peak, live)
                for j in set(dag.operands(i)):
                    if j in uses:
                        uses[j] -= 1
                        if uses[j] == 0:
                            live -= 1
            uses[root] -= 1
            if uses[root] == 0:
                live -= 1
    return peak


class _ModuleWriter(object):

    def __init__(self, prog, opts):
        self.p = prog
        self.o = opts
        self.dag = prog.dag
        self.uni_slot = {}          # uniform frontier node -> slot in uni[]

    # -- leaves -------------------------------------------------------------
    def _is_vec_input(self, i):
        return self.dag.op[i] == ir.INPUT and \
            self.dag.args[i][0] in ('cur', 'adj')

    def _uniform_leaf(self, i):
        """Non-constant node-invariant node: lives in the ``uni`` table."""
        return self.dag.uni[i] and self.dag.op[i] != ir.CONST

    def _slot(self, i):
        s = self.uni_slot.get(i)
        if s is None:
            s = self.uni_slot[i] = len(self.uni_slot)
        return s

    def _row_ptr(self, r):
        src, k = self.p.rows[r]
        if src == 'free':
            return 'free_ + %dLL*N' % k
        return 'known_traj + %dLL*N' % k

    def _scalar_source(self, i):
        """C expression loading a node-invariant INPUT node from its home."""
        p = self.p
        kind, idx = self.dag.args[i]
        tail = 'free_[%dLL*N + %%d]' % (p.n + p.q)
        if kind == 'par':
            src, k = p.pars[idx]
            return ('params[%d]' % k) if src == 'known' else tail % k
        if kind == 'h':
            return 'h' if p.h[0] == 'fixed' else tail % p.h[1]
        if kind == 'free':
            return 'free_[inst_idx[%d]]' % idx
        raise AssertionError(kind)

    # -- grouping --------------------------------------------------------------
    def _chunks(self, e0, e1):
        K = self.o.chunk
        return [(c, min(c + K, e1)) for c in range(e0, e1, K)]

    def line_mode(self):
        """Line-aligned ring flush (see opty_device.h) for all but tiny
        blocks; needs the chunk width to be a multiple of a 16-double line."""
        return self.p.P >= LINE_MODE_MIN_P and self.o.chunk % 16 == 0

    def _virtual_end(self, e1):
        """Waves evaluate 15 entries past their range so that they own whole
        lines; the last range wraps into entries 0..14 (of the next node)."""
        return e1 + 15 if self.line_mode() else e1

    def group_ranges(self):
        """Splits the P entries of the block into G contiguous ranges whose
        boundaries are multiples of the chunk width (hence even).  With
        ``groups=None`` the number of groups is the smallest for which every
        group's estimated live temporaries stay below ``max_live``."""
        P, K = self.p.P, self.o.chunk
        unit = 16 if self.line_mode() else K
        nunits = max(1, P//unit if self.line_mode() else (P + K - 1)//K)

        def split(G):
            b = [((g*nunits)//G)*unit for g in range(G)] + [P]
            return [(b[g], b[g + 1]) for g in range(G)]

        if self.o.groups is not None:
            return split(max(1, min(int(self.o.groups), nunits)))
        leaf = lambda i: self._is_vec_input(i) or self._uniform_leaf(i)
        G = 1
        while True:
            ranges = split(G)
            worst = max(
                _max_live(self.dag,
                          [[self.p.jac_out[v % P] for v in range(a, b)]
                           for a, b in self._chunks(e0, self._virtual_end(e1))],
                          leaf)
                for e0, e1 in ranges)
            if worst <= self.o.max_live or G >= min(nunits, 16):
                return ranges
            G += 1

    # -- kernels ---------------------------------------------------------------
    def _group_body(self, e0, e1, con_rows):
        """Code for one wave evaluating Jacobian entries [e0, e1) and the
        constraint rows ``con_rows`` of its 64 nodes.  Returns (lines, number
        of slab rows)."""
        p, d = self.p, self.dag
        K = self.o.chunk
        vend = self._virtual_end(e1) if e1 > e0 else e1
        roots = [p.jac_out[v % p.P] for v in range(e0, vend)]
        roots += [p.con_out[j] for j in con_rows]
        needed = set(d.reachable(roots))
        rows = sorted({d.args[i][1] for i in needed if self._is_vec_input(i)})
        slab_of = {r: s for s, r in enumerate(rows)}
        line_mode = self.line_mode() and e1 > e0
        R = K + 16
        if line_mode:
            tile_rows = R
        else:
            tile_rows = min(K, e1 - e0) if e1 > e0 else 0
        slab0 = tile_rows*TS

        def leaf(i):
            if self._is_vec_input(i):
                kind, r = d.args[i]
                off = p.cur_offset if kind == 'cur' else p.adj_offset
                return 'lds[%d + lane + %d]' % (slab0 + slab_of[r]*TS, off)
            if self._uniform_leaf(i):
                return 'uni_c[%d]' % self._slot(i)
            return None

        # Slab fill: issue EVERY global load first (65 time nodes per row: one
        # per lane plus the halo node, which has a wave-uniform address), then
        # the LDS writes.  Load-by-load (`load; wait; ds_write; branch`) costs
        # one full memory round trip per row and dominated the wave's life.
        lines = []
        if rows:
            lines.append('const long long t_ld = node0 + lane < N - 1 ? '
                         'node0 + lane : N - 1;')
            lines.append('const long long t_halo = node0 + 64 < N - 1 ? '
                         'node0 + 64 : N - 1;')
        for r in rows:
            lines.append('const double sl%d = (%s)[t_ld];'
                         % (r, self._row_ptr(r)))
            lines.append('const double sh%d = (%s)[t_halo];'
                         % (r, self._row_ptr(r)))
        for r in rows:
            lines.append('lds[%d + lane] = sl%d;'
                         % (slab0 + slab_of[r]*TS, r))
        if rows:
            lines.append('if (lane == 0) {')
            for r in rows:
                lines.append('    lds[%d] = sh%d;'
                             % (slab0 + slab_of[r]*TS + WAVE, r))
            lines.append('}')
            lines.append('opty_wave_sync();')
        body = _Body(d, needed, leaf)
        for j in con_rows:
            ref = body.emit(p.con_out[j])
            body.lines.append('if (valid) con[%dLL*con_stride + node] = %s;'
                              % (j, ref))
        nv = '(N < 0 ? nvalid : 0)' if self.o.ablate == 'compute_only' \
            else 'nvalid'

        def value(e):
            if self.o.ablate == 'store_only':
                return '(double)(lane + %d)' % e
            return body.emit(p.jac_out[e])

        if line_mode:
            if e1 < p.P:
                assert e1 + 15 <= p.P, 'last entry range must be >= 16 wide'
            body.lines.append('const int b0 = opty_line_phase(jrow);')
            for c0, c1 in self._chunks(e0, e1 + 15):
                body.new_scope()
                for v in range(c0, c1):
                    body.lines.append('lds[%d + lane] = %s;'
                                      % ((v % R)*TS, value(v % p.P)))
                body.lines.append('opty_wave_sync();')
                body.lines.append(
                    'opty_flush_lines<%d, %d, %d>(lds, jrow, %d, b0, %d, %d, '
                    '%d, %d, %s, lane);' % ((c1 - c0 + 15)//16, R,
                                            self.o.flush_unroll, p.P,
                                            c0 - 15, e0, e1, c1, nv))
                if e0 == 0 and c0 == 0:
                    assert c1 >= 15
                    if self.o.ablate != 'compute_only':
                        body.lines.append('opty_head_piece<%d>(lds, jrow, '
                                          '%d, b0, lane);' % (R, p.P))
                body.lines.append('opty_wave_sync();')
            return lines + body.lines, tile_rows + len(rows)

        wide = (p.P % 2 == 0)
        for c0, c1 in self._chunks(e0, e1):
            body.new_scope()
            for e in range(c0, c1):
                body.lines.append('lds[%d + lane] = %s;'
                                  % ((e - c0)*TS, value(e)))
            body.lines.append('opty_wave_sync();')
            w = c1 - c0
            fl = 'opty_flush16' if (wide and w % 2 == 0 and c0 % 2 == 0) \
                else 'opty_flush8'
            body.lines.append('%s<%d>(lds, jrow + %d, %dLL, %s, lane);'
                              % (fl, w, c0, p.P, nv))
            body.lines.append('opty_wave_sync();')
        return lines + body.lines, tile_rows + len(rows)

    _PROLOGUE = '''\
    const int lane = threadIdx.x;
    const long long nblk = (node_end - node_begin + 63)/64;
    {map}
    if (blk >= nblk) return;
    const long long node0 = node_begin + blk*64;
    const long long node = node0 + lane;
    const bool valid = node < node_end;
    const long long rem = node_end - node0;
    const int nvalid = rem < 64 ? (int)rem : 64;
    double *jrow = jac + (node0 - node_begin)*{P}LL;
    (void)valid; (void)jrow; (void)nvalid; (void)node;
'''

    def kernel(self, name, groups, con_of_group):
        """One kernel; ``groups`` = list of (e0, e1); ``con_of_group[g]`` =
        constraint rows stored by group g."""
        G = len(groups)
        bodies, lds_rows = [], 1
        for g, (e0, e1) in enumerate(groups):
            lines, rows = self._group_body(e0, e1, con_of_group[g])
            bodies.append(lines)
            lds_rows = max(lds_rows, rows)
        if G == 1:
            mapping = 'const long long blk = blockIdx.x; const int grp = 0;'
        else:
            # XCD-aware: consecutive workgroup ids round-robin the 8 XCDs, so
            # give every XCD whole node blocks -- all G waves of a node block
            # (which write interleaved pieces of the same node rows and read
            # the same slab) then share one L2.
            mapping = ('const long long wid = blockIdx.x; '
                       'const long long xcd = wid & 7, slot = wid >> 3; '
                       'const long long blk = (slot/%d)*8 + xcd; '
                       'const int grp = (int)(slot %% %d);' % (G, G))
        src = ['extern "C" __global__ void __launch_bounds__(64)',
               '%s(%s)' % (name, KERNEL_PARAMS), '{',
               '    __shared__ double lds[%d];' % (lds_rows*TS),
               self._PROLOGUE.format(map=mapping, P=self.p.P)]
        if G == 1:
            src += ['    ' + ln for ln in bodies[0]]
        else:
            src.append('    switch (grp) {')
            for g, lines in enumerate(bodies):
                src.append('    case %d: {' % g)
                src += ['        ' + ln for ln in lines]
                src.append('    } break;')
            src.append('    default: break;')
            src.append('    }')
        src.append('}')
        return '\n'.join(src), dict(name=name, groups=G,
                                    lds_bytes=lds_rows*TS*8)

    def uniform_kernel(self):
        """Must be printed after every kernel that allocates ``uni`` slots."""
        d = self.dag
        slots = sorted(self.uni_slot.items(), key=lambda kv: kv[1])
        roots = [i for i, _ in slots]
        needed = set(d.reachable(roots))

        def leaf(i):
            if d.op[i] == ir.INPUT:
                return self._scalar_source(i)
            return None

        body = _Body(d, needed, leaf)
        for i, s in slots:
            ref = body.emit(i)
            body.lines.append('uni_w[%d] = %s;' % (s, ref))
        src = ['extern "C" __global__ void __launch_bounds__(64)',
               'opty_uni(%s)' % KERNEL_PARAMS, '{',
               '    if (threadIdx.x != 0 || blockIdx.x != 0) return;']
        src += ['    ' + ln for ln in body.lines] + ['}']
        dynamic = any(d.op[i] == ir.INPUT and
                      self._scalar_source(i).startswith('free_')
                      for i in needed)
        return '\n'.join(src), len(slots), dynamic

    def inst_kernel(self):
        p, d = self.p, self.dag
        roots = list(p.inst_con_out) + list(p.inst_jac_out)
        needed = set(d.reachable(roots))

        def leaf(i):
            if d.op[i] == ir.INPUT:
                return self._scalar_source(i)
            return None

        body = _Body(d, needed, leaf)
        for k, node in enumerate(p.inst_con_out):
            ref = body.emit(node)
            body.lines.append('if (con) con[%dLL*con_stride + %d] = %s;'
                              % (p.M, k, ref))
        for k, node in enumerate(p.inst_jac_out):
            ref = body.emit(node)
            body.lines.append('if (jac) jac[(node_end - node_begin)*%dLL + '
                              '%d] = %s;' % (p.P, k, ref))
        src = ['extern "C" __global__ void __launch_bounds__(64)',
               'opty_inst(%s)' % KERNEL_PARAMS, '{',
               '    if (threadIdx.x != 0 || blockIdx.x != 0) return;']
        src += ['    ' + ln for ln in body.lines] + ['}']
        return '\n'.join(src), dict(name='opty_inst', groups=1, lds_bytes=0)


def emit_module(prog, opts=None):
    """Returns ``(source, meta)``; ``meta`` describes the launch geometry the
    runtime needs (waves per node block, size of the ``uni`` table, whether
    the table depends on ``free``)."""
    opts = opts or EmitOptions()
    w = _ModuleWriter(prog, opts)
    groups = w.group_ranges()
    all_rows = list(range(prog.M))
    # constraint row j is stored by the wave that owns Jacobian row j's first
    # entry (their temporaries overlap the most)
    con_of = [[] for _ in groups]
    for j in all_rows:
        e = j*prog.C
        for g, (e0, e1) in enumerate(groups):
            if e0 <= e < e1:
                con_of[g].append(j)
    parts = []
    kernels = {}
    for key, name, grp, cons in (
            ('con', 'opty_con', [(0, 0)], [all_rows]),
            ('jac', 'opty_jac', groups, [[] for _ in groups]),
            ('conjac', 'opty_conjac', groups, con_of)):
        src, meta = w.kernel(name, grp, cons)
        parts += [src, '']
        kernels[key] = meta
    if prog.inst_con_out:
        src, meta = w.inst_kernel()
        parts += [src, '']
        kernels['inst'] = meta
    src, num_uniform, dynamic = w.uniform_kernel()
    parts += [src, '']
    head = ['// generated by opty_amd.codegen.emit_hip -- do not edit',
            '// %s' % opts.key(),
            '#include "opty_device.h"', '']
    source = '\n'.join(head + parts)
    meta = dict(kernels=kernels, groups=[list(g) for g in groups],
                chunk=opts.chunk, P=prog.P, M=prog.M, C=prog.C,
                num_uniform=num_uniform, uniform_dynamic=bool(dynamic),
                sha=hashlib.sha256(source.encode()).hexdigest())
    return source, meta
