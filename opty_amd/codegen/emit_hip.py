"""Prints a :class:`~opty_amd.codegen.program.CollocationProgram` as HIP source
for gfx950.

The printed module is the GPU counterpart of the ``_c.c`` / ``.pyx`` pair the
reference emits (``opty/utils.py:483-529``): instead of a scalar
``eval_matrix`` called from an OpenMP node loop it contains wave-per-64-nodes
kernels built on ``csrc/opty_device.h``:

``opty_con``      constraints, equation-major coalesced stores
``opty_jac``      Jacobian blocks, staged through an LDS tile in chunks of
                  ``KC`` entries and flushed node-major with 16-byte stores;
                  the P entries of the block are split into ``G`` contiguous
                  entry ranges, each evaluated by its own wave (``G`` waves per
                  64 nodes) so that a 10^5-node problem yields enough waves to
                  fill 256 CUs and each wave's live register set stays small
``opty_conjac``   both outputs from one launch (shared sub-expressions)
``opty_inst``     instance-constraint values and partials (one lane)

Scheduling: outputs are visited in memory order; each output's not-yet-emitted
operands are emitted depth-first just before it, so temporaries are created as
late as possible and LLVM's register allocator sees short live ranges.
"""

import hashlib

from . import ir

WAVE = 64
TS = 65


class EmitOptions(object):
    def __init__(self, chunk=32, groups=None, lds_slab=True):
        self.chunk = int(chunk)
        self.groups = groups
        self.lds_slab = bool(lds_slab)

    def key(self):
        return 'chunk=%d groups=%s slab=%d' % (self.chunk, self.groups,
                                               self.lds_slab)


def _lit(v):
    if v != v:
        return '__builtin_nan("")'
    if v in (float('inf'), float('-inf')):
        return ('-' if v < 0 else '') + '__builtin_inf()'
    s = repr(float(v))
    if 'e' not in s and '.' not in s:
        s += '.0'
    return s


class _Body(object):
    """Emits straight-line code for a set of DAG nodes, once each."""

    def __init__(self, dag, needed, name_of_input):
        self.dag = dag
        self.lines = []
        self.done = {}
        self.needed = needed            # set of node ids reachable in scope
        self.name_of_input = name_of_input

    def ref(self, i):
        d = self.dag
        if d.op[i] == ir.CONST:
            v = d.value(i)
            return _lit(v) if v >= 0 else '(%s)' % _lit(v)
        return self.done[i]

    def emit(self, root):
        """Makes sure ``root`` is computed; returns the C expression naming
        it."""
        d = self.dag
        if d.op[root] == ir.CONST:
            return self.ref(root)
        stack = [(root, False)]
        while stack:
            i, ready = stack.pop()
            if i in self.done or d.op[i] == ir.CONST:
                continue
            if d.op[i] == ir.INPUT:
                self.done[i] = self.name_of_input(i)
                continue
            if not ready:
                stack.append((i, True))
                for j in reversed(d.operands(i)):
                    if j not in self.done and d.op[j] != ir.CONST:
                        stack.append((j, False))
                continue
            self._emit_node(i)
        return self.done[root]

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
            if j is not None and j in self.needed and j not in self.done:
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


class _KernelWriter(object):

    def __init__(self, prog, opts):
        self.p = prog
        self.o = opts
        self.dag = prog.dag

    # -- input plumbing --------------------------------------------------------
    def _input_name(self, i):
        kind, idx = self.dag.args[i]
        return {'cur': 'xc%d', 'adj': 'xa%d', 'par': 'p%d', 'h': 'hh',
                'free': 'fa%d'}[kind] % ((idx,) if kind != 'h' else ())

    def _row_ptr(self, r):
        src, k = self.p.rows[r]
        if src == 'free':
            return 'a.free_ + %dLL*a.N' % k
        return 'a.known_traj + %dLL*a.N' % k

    def _scalar_loads(self, inputs):
        """Loads of the node-invariant inputs (uniform addresses)."""
        p = self.p
        out = []
        tail = 'a.free_[%dLL*a.N + %%d]' % (p.n + p.q)
        for i in inputs:
            kind, idx = self.dag.args[i]
            if kind == 'par':
                src, k = p.pars[idx]
                e = ('a.params[%d]' % k) if src == 'known' else tail % k
                out.append('const double p%d = %s;' % (idx, e))
            elif kind == 'h':
                e = 'a.h' if p.h[0] == 'fixed' else tail % p.h[1]
                out.append('const double hh = %s;' % e)
            elif kind == 'free':
                out.append('const double fa%d = a.free_[a.inst_idx[%d]];'
                           % (idx, idx))
        return out

    def _vector_loads(self, inputs):
        """Per-node inputs: LDS slab (one coalesced row load, both time
        offsets read from LDS) or direct global loads."""
        p = self.p
        rows = sorted({self.dag.args[i][1] for i in inputs})
        use = {}
        for i in inputs:
            kind, r = self.dag.args[i]
            use.setdefault(r, set()).add(kind)
        out = []
        if not rows:
            return out, 0
        if self.o.lds_slab:
            for s, r in enumerate(rows):
                out.append('opty_slab_load(lds, %d, %s, node0, a.N - 1, lane);'
                           % (s, self._row_ptr(r)))
            out.append('opty_wave_sync();')
            for s, r in enumerate(rows):
                if 'cur' in use[r]:
                    out.append('const double xc%d = lds[%d + lane + %d];'
                               % (r, s*TS, p.cur_offset))
                if 'adj' in use[r]:
                    out.append('const double xa%d = lds[%d + lane + %d];'
                               % (r, s*TS, p.adj_offset))
            out.append('opty_wave_sync();')
            return out, len(rows)
        for r in rows:
            if 'cur' in use[r]:
                out.append('const double xc%d = (%s)[tn + %d];'
                           % (r, self._row_ptr(r), p.cur_offset))
            if 'adj' in use[r]:
                out.append('const double xa%d = (%s)[tn + %d];'
                           % (r, self._row_ptr(r), p.adj_offset))
        return out, 0

    def _split_inputs(self, roots):
        need = self.dag.reachable(roots)
        vec, sca = [], []
        for i in need:
            if self.dag.op[i] == ir.INPUT:
                (vec if self.dag.args[i][0] in ('cur', 'adj')
                 else sca).append(i)
        return set(need), vec, sca

    # -- kernels ---------------------------------------------------------------
    def group_ranges(self):
        """Splits the P entries of the block into G contiguous ranges whose
        boundaries are multiples of the chunk width (hence even)."""
        P, K = self.p.P, self.o.chunk
        G = self.o.groups
        nchunks = (P + K - 1)//K
        if G is None:
            G = max(1, min(8, nchunks//8))
        G = max(1, min(G, nchunks))
        bounds = [((g*nchunks)//G)*K for g in range(G)] + [P]
        return [(bounds[g], bounds[g + 1]) for g in range(G)]

    def _group_body(self, e0, e1, con_rows):
        """Code for one wave evaluating Jacobian entries [e0, e1) and the
        constraint rows ``con_rows`` of its 64 nodes."""
        p = self.p
        K = self.o.chunk
        roots = [p.jac_out[e] for e in range(e0, e1)]
        roots += [p.con_out[j] for j in con_rows]
        needed, vec, sca = self._split_inputs(roots)
        loads, nslab = self._vector_loads(vec)
        body = _Body(self.dag, needed, self._input_name)
        lines = loads + self._scalar_loads(sca)
        wide = (p.P % 2 == 0)
        for j in con_rows:
            ref = body.emit(p.con_out[j])
            body.lines.append('if (valid) a.con[%dLL*a.con_stride + node] = '
                              '%s;' % (j, ref))
        c0 = e0
        while c0 < e1:
            c1 = min(c0 + K, e1)
            for e in range(c0, c1):
                ref = body.emit(p.jac_out[e])
                body.lines.append('lds[%d + lane] = %s;'
                                  % ((e - c0)*TS, ref))
            body.lines.append('opty_wave_sync();')
            w = c1 - c0
            fl = 'opty_flush16' if (wide and w % 2 == 0 and c0 % 2 == 0) \
                else 'opty_flush8'
            body.lines.append('%s<%d>(lds, jrow + %d, %dLL, nvalid, lane);'
                              % (fl, w, c0, p.P))
            body.lines.append('opty_wave_sync();')
            c0 = c1
        lines += body.lines
        lds_rows = max(nslab, min(K, e1 - e0) if e1 > e0 else 0)
        return lines, lds_rows

    _PROLOGUE = '''\
    const int lane = threadIdx.x;
    const long long nblk = (a.node_end - a.node_begin + 63)/64;
    {map}
    if (blk >= nblk) return;
    const long long node0 = a.node_begin + blk*64;
    const long long node = node0 + lane;
    const bool valid = node < a.node_end;
    const long long rem = a.node_end - node0;
    const int nvalid = rem < 64 ? (int)rem : 64;
    const long long tn = valid ? node : a.node_end - 1;
    double *jrow = a.jac + (node0 - a.node_begin)*{P}LL;
    (void)tn; (void)jrow; (void)nvalid;
'''

    def _kernel(self, name, groups, con_of_group):
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
               '%s(const OptyKernelArgs a)' % name, '{',
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

    def inst_kernel(self):
        p = self.p
        roots = list(p.inst_con_out) + list(p.inst_jac_out)
        needed, vec, sca = self._split_inputs(roots)
        assert not vec
        body = _Body(self.dag, needed, self._input_name)
        lines = self._scalar_loads(sca)
        for k, node in enumerate(p.inst_con_out):
            ref = body.emit(node)
            body.lines.append('if (a.con) a.con[%dLL*a.con_stride + %d] = %s;'
                              % (p.M, k, ref))
        for k, node in enumerate(p.inst_jac_out):
            ref = body.emit(node)
            body.lines.append('if (a.jac) a.jac[(a.node_end - a.node_begin)*'
                              '%dLL + %d] = %s;' % (p.P, k, ref))
        lines += body.lines
        src = ['extern "C" __global__ void __launch_bounds__(64)',
               'opty_inst(const OptyKernelArgs a)', '{',
               '    if (threadIdx.x != 0 || blockIdx.x != 0) return;']
        src += ['    ' + ln for ln in lines] + ['}']
        return '\n'.join(src), dict(name='opty_inst', groups=1, lds_bytes=0)


def emit_module(prog, opts=None):
    """Returns ``(source, meta)``; ``meta['kernels']`` describes the launch
    geometry the runtime needs (kernel name, waves per node block)."""
    opts = opts or EmitOptions()
    w = _KernelWriter(prog, opts)
    groups = w.group_ranges()
    G = len(groups)
    all_rows = list(range(prog.M))
    # constraint row j is stored by the wave that owns Jacobian row j's first
    # entry (their temporaries overlap the most)
    con_of = [[] for _ in groups]
    for j in all_rows:
        e = j*prog.C
        for g, (e0, e1) in enumerate(groups):
            if e0 <= e < e1:
                con_of[g].append(j)
    parts = ['// generated by opty_amd.codegen.emit_hip -- do not edit',
             '// %s' % opts.key(),
             '#include "opty_device.h"', '']
    kernels = {}
    src, meta = w._kernel('opty_con', [(0, 0)], [all_rows])
    parts += [src, '']
    kernels['con'] = meta
    src, meta = w._kernel('opty_jac', groups, [[] for _ in groups])
    parts += [src, '']
    kernels['jac'] = meta
    src, meta = w._kernel('opty_conjac', groups, con_of)
    parts += [src, '']
    kernels['conjac'] = meta
    if prog.inst_con_out:
        src, meta = w.inst_kernel()
        parts += [src, '']
        kernels['inst'] = meta
    source = '\n'.join(parts)
    meta = dict(kernels=kernels, groups=[list(g) for g in groups],
                chunk=opts.chunk, P=prog.P, M=prog.M, C=prog.C,
                sha=hashlib.sha256(source.encode()).hexdigest())
    return source, meta
