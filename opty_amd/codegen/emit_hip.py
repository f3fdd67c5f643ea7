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
import os

from . import ir

WAVE = 64
TS = 65
LINE_MODE_MIN_P = 64
#: csr layout: rows up to this many entries are staged whole and written
#: as one contiguous span per wave (tile = 65*8 bytes per entry)
CSR_MAX_ROW = 64
#: bytes of one evaluation up to which the runtime's host-buffer entry points
#: use mapped host memory (OPTY_LATENCY_PATH_BYTES in csrc/runtime.cpp)
LATENCY_PATH_BYTES = 2 << 20

#: node-invariant operations up to which a small problem evaluates them in
#: every lane instead of launching opty_uni before every evaluation
INLINE_UNIFORM_MAX_NODES = 64
#: node-invariant operations that depend on `free` (a free node time
#: interval, unknown parameters) up to which they are evaluated in the lanes,
#: so that the ``uni`` table does not have to be refilled by a launch of
#: opty_uni before every evaluation (_ModuleWriter._choose_inline_dynamic)
INLINE_DYNAMIC_MAX_OPS = 48

#: workgroups the runtime launches opty_uni with (OPTY_UNI_WORKGROUPS in
#: csrc/opty_internal.h)
UNI_WORKGROUPS = 16
#: waves an MI355X holds at once when a CU takes four of these kernels' waves
#: (256 CUs x 4 SIMDs; one wave per SIMD at their register footprint)
RESIDENT_WAVES = 1024
#: strip classes a list schedule can hold (OPTY_HIP_MAX_CLASSES)
MAX_CLASSES = 32
#: entries of a node's block one wave of opty_jac takes when registers do not
#: ask for fewer (~50 KB of output per wave and 64-node block), at most 32
#: waves.  Interleaved A/B timings on MI355X after coefficient collection made
#: the evaluation cheap (profiles/r02_strip_sweeps.txt), best strip count of
#: opty_jac for n-link pendulums: 8 links (P = 666) 8, 10 links (990) 10-12,
#: 14 links (1830) 18, 24 links (5100) >= 32.
STRIP_ENTRIES = 100
#: work-aware strip cuts (EmitOptions.cut='work'): blocks of up to this many
#: 16-entry units (the dynamic programme evaluates O(units^2) strip costs),
#: no strip wider than this multiple of an even share, and what one stored
#: entry weighs next to one weighted operation
WORK_CUT_MAX_UNITS = 128
WORK_CUT_MAX_SHARE = 3.0
WORK_CUT_MAX_WEIGHT = 2.0
#: ... and no strip whose estimated live temporaries (``_max_live``) exceed
#: what 512 registers hold without spilling (measured: 238 builds clean, 274
#: spills 30-60 registers)
WORK_CUT_MAX_LIVE = int(os.environ.get('OPTY_WORK_CUT_MAX_LIVE', 245))
#: what one more wave per block costs next to one weighted operation (slab
#: fill, dispatch; measured on the biped: +1 % of the launch per strip)
WORK_CUT_STRIP_OVERHEAD = 200
STORE_WEIGHT = 6
#: The fused kernel shares every block with the constraint waves and wants
#: fewer, wider strips the larger the block: best counts 6-7 / 9 / 10-12 / 20
#: for P = 666 / 990 / 1830 / 5100, i.e. about 0.286 sqrt(P) (an empirical fit
#: over that family; off by one strip costs 1-3 %, the r01 choice cost 7 %).
FUSED_STRIPS_PER_SQRT_ENTRY = 0.286
#: constraint vector of one launch above which the fused kernel streams its
#: constraint stores past the caches (between the 70 MB of N = 4*10^5, still
#: fine with plain stores, and the 176 MB of N = 10^6)
CON_CACHE_BYTES = 128 << 20

#: doubles between the end of a launch's Jacobian values and the wave records
#: of ``EmitOptions.trace``
TRACE_OFFSET = 4096

KERNEL_PARAMS = (
    'const double *__restrict__ free_, const double *__restrict__ known_traj, '
    'const double *__restrict__ params, const double *__restrict__ uni_c, '
    'double *__restrict__ uni_w, const long long *__restrict__ inst_idx, '
    'double *__restrict__ con, double *__restrict__ jac, double h, '
    'long long N, long long con_stride, long long node_begin, '
    'long long node_end')


_UNIFORM_TRIG_HELPERS = '''\
// sincos behind a wave-uniform test (EmitOptions.fast_trig = 2): the short path
// of opty_sincos (opty_device.h) when no lane of the wave needs the library's.
__device__ __forceinline__ void optyu_sincos(double x, double *s, double *c) {
    const bool odd = !(__builtin_fabs(x) <= 1048576.0);
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {
        sincos(x, s, c);
        return;
    }
    const double k = __builtin_rint(x*6.36619772367581382433e-01);
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    r = __builtin_fma(-k, -1.49738490485916983294e-33, r);
    const double z = r*r;
    double ps = z*1.58969099521155010221e-10 + -2.50507602534068634195e-08;
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(z*r, ps, r);
    double pc = z*-1.13596475577881948265e-11 + 2.08757232129817482790e-09;
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5*z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + (z*z)*pc);
    const int n = (int)k;
    const bool swap = n & 1;
    const double a = swap ? cr : sr;
    const double b = swap ? sr : cr;
    *s = (n & 2) ? -a : a;
    *c = ((n + 1) & 2) ? -b : b;
}
__device__ __forceinline__ double optyu_sin(double x) {
    double s, c;
    optyu_sincos(x, &s, &c);
    return s;
}
__device__ __forceinline__ double optyu_cos(double x) {
    double s, c;
    optyu_sincos(x, &s, &c);
    return c;
}'''


_LOOP_HELPERS = '''\
// Values the optimiser must take as new in every iteration of a persistent
// kernel's item loops (see the kernels).
template <typename T>
__device__ __forceinline__ T *opty_opaque(T *p) {
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ long long opty_opaque(long long x) {
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ double opty_opaque(double x) {
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ int opty_opaque_lane(int x) {
    asm volatile("" : "+v"(x));
    return x;
}'''


class EmitOptions(object):
    """Knobs of the printer.

    chunk : entries of the per-node block staged per LDS tile flush (even)
    groups : waves per 64-node block for the Jacobian kernels (None = auto)
    max_live : auto-grouping target for the number of simultaneously live
        float64 temporaries of one wave
    """

    def __init__(self, chunk=32, groups=None, max_live=125, ablate=None,
                 flush_unroll=4, waves=None, store_aux=18, con_rows_per_wave=0,
                 interleave=0, pad=0, occupancy=0, con_nt=None, fast_trig=0,
                 fused_groups=None, small_flush='flat', con_split='work',
                 fold_instance=None, inline_uniform=None, dear_first=0,
                 cut=None, con_attach=None, forget=0, rotate=None,
                 work_live=None, inline_dynamic=None, order=None, trace=0,
                 park=0, park_live=215, park_spread=0, strips=None,
                 fused_strips=None,
                 fused_order=None, deterministic=0, class_cost=None,
                 fused_class_cost=None, share_rcp=0, publish=0):
        # 1 (r06, launches that under-fill the chip -- node shards): the
        # waves of a block form ONE workgroup and evaluate the block's
        # isomorphic sub-models (codegen/isomorph.py: the musculotendon
        # actuators and activation dynamics of the muscle-driven leg) ONCE,
        # side by side -- one instance per wave, longest first -- before
        # their strips: every instance's interface values (the few values
        # the rest of the block reads: 13 per muscle) are published in LDS
        # rows behind the ring tiles, one barrier later every strip reads
        # them as it reads the input slab.  The heavy strips lose the work
        # they used to repeat (leg: 3465 / 3363 -> 1900 / 1798 weighted
        # operations after a 496-operation stage).  Costs LDS (82 rows for
        # the leg) and a workgroup as wide as the block has waves: only
        # where a CU holds one block anyway
        self.publish = int(publish)
        # 1: a denominator that divides two or more values of a wave is
        # inverted ONCE (one true division, 1.0/b) and the quotients become
        # products with that reciprocal: an f64 division is 11-13 vector
        # instructions (v_div_scale x2, v_rcp, Newton steps, v_div_fmas,
        # v_div_fixup), and a wave that holds a SIMD alone pays an issue
        # slot for every one of them (the muscle-driven leg: 78 divisions
        # over 32 denominators; the biped 32 over 8).  A quotient then
        # carries two roundings instead of one (a*(1/b) against a/b: <= 1.5
        # ulp); poles and signed zeros come out as the division's.  Off by
        # default: a launch plan turns it on where it was measured to pay
        self.share_rcp = int(share_rcp)
        # relative wave durations of the strip classes of a persistent kernel
        # ('20.5;11.6;4.7': measured by a traced launch) for its list
        # schedule; None: the printer's estimate
        self.class_cost = class_cost
        self.fused_class_cost = fused_class_cost
        # 1: every wave prints the same operations for the same DAG node:
        # sin / cos of an argument whose other half exists anywhere in the
        # DAG always come from one ``sincos`` (not from ``sin`` alone in the
        # waves that happen to need only one of the two).  Together with
        # ``-ffp-contract=off`` (hip_backend.DETERMINISTIC_FLAGS) a node's
        # values are bit-identical in every launch geometry
        self.deterministic = int(deterministic)
        # the same two choices for opty_conjac alone (None: as opty_jac's)
        self.fused_strips = fused_strips
        assert fused_order in (None, 'block', 'class', 'tail', 'list')
        self.fused_order = fused_order
        # explicit cut of a node-major block (experiments, plan files):
        # ``'96:160;160:348+0:96'`` = two waves, the second one with two
        # strips (boundaries on 16-entry lines; every entry covered once)
        self.strips = strips
        # LDS parking (line-mode Jacobian waves): a wave whose live
        # temporaries exceed the register file keeps up to ``park`` of them
        # in LDS rows of its own (64 doubles each, behind its ring tile)
        # instead of being cut into two strips that both evaluate what they
        # share.  The printer plans the wave (_WavePlan): the order in which
        # the entries' expressions are evaluated (memory order, or cheapest
        # next -- whichever keeps fewer values alive; the ring tile takes
        # them in memory order all the same), then a farthest-next-use
        # eviction over the straight-line code with ``park_live`` doubles in
        # registers.  0 = off
        self.park = int(park)
        self.park_live = int(park_live)
        # 1: the flushes of chunks whose values are ready early (constants,
        # cheap entries) are SPREAD over the wave's operation stream instead
        # of issued as soon as they can be: a wave that evaluates a heavy
        # strip and store-only strips (``'160:348+0:96+96:160'``) then issues
        # its stores in the shadow of its own arithmetic
        self.park_spread = int(park_spread)
        # profiling aid (never used by the product): every wave records its
        # start / end (wall_clock64, shader cycles), its strip and where it
        # ran behind the Jacobian values -- ``jac`` must hold
        # TRACE_OFFSET + 4 x (waves of the launch) more doubles
        # (tools/wave_timeline.py)
        self.trace = int(trace)
        # dispatch order of a launch's workgroups: 'block' = the workgroups
        # of one 64-node block back to back (they read the same slab and
        # write interleaved strips of the same rows: what a store-bound
        # block wants), 'class' = strip class by strip class, the longest
        # waves of ALL blocks first (longest-processing-time order: a launch
        # whose waves differ 5x in length is packed onto the SIMDs like a
        # list schedule instead of leaving the long waves of the last blocks
        # to run alone at the end).  None = 'block'
        # 'tail': block by block, except that the last blocks of a launch
        # (as many as put one resident set of long waves on the chip) are
        # dispatched class by class -- their long waves first, the short
        # ones last: a launch in 'block' order ends with the long waves of
        # its last blocks running alone
        assert order in (None, 'block', 'class', 'tail', 'list')
        self.order = order
        # 1: a strip's temporaries are dropped at every chunk boundary and
        # recomputed where needed again (bounded register pressure; the last
        # resort before a build that spills vector registers)
        self.forget = int(forget)
        # fused kernel: constraint rows evaluated by the Jacobian wave that
        # already computes most of their sub-expressions instead of by a
        # constraint wave of their own (emit_module): None = automatic (blocks
        # whose arithmetic, not their bytes, sets the pace), 0 / 1 = never /
        # whenever a row shares at least half of its work with a strip
        self.con_attach = None if con_attach is None else int(con_attach)
        # where the strips of a node-major block are cut: 'even' = equal
        # entry counts; 'work' = at the line boundaries that make the strips'
        # total evaluation work smallest (_ModuleWriter.group_ranges): a
        # block whose expensive entries share most of their sub-expressions
        # (the muscle-driven leg: 74 of 348 entries are not structural
        # zeros, every strip that touches rows 4-5 costs ~3400 of the
        # block's 4460 weighted operations; the seven-segment biped: 24 even
        # strips evaluate 29 600 weighted operations per node, 5 work-aware
        # ones 12 500) is evaluated once instead of once per strip that
        # happens to cross it.  None = automatic: the work-aware cut with the
        # fewest strips the registers allow when the even cut's
        # recomputation is what the block would wait for
        # (_ModuleWriter._automatic_work_cut; only while ``groups`` is None)
        # node-invariant sub-expressions that depend on `free` evaluated in
        # the lanes instead of by opty_uni before every evaluation: None =
        # automatic (when they are few), 0 / 1 = never / always
        self.inline_dynamic = None if inline_dynamic is None \
            else int(inline_dynamic)
        assert cut in (None, 'even', 'work')
        self.cut = cut
        # live temporaries a strip of a work-aware cut may have (None:
        # WORK_CUT_MAX_LIVE); what the spill-free builder lowers when such a
        # cut spills -- more strips would only split the store-only part
        self.work_live = None if work_live is None else int(work_live)
        # 1: the strip a block's first workgroup takes advances from block
        # to block (_ModuleWriter.kernel); None = automatic (the unequal
        # waves of a work-aware cut), 0 = every block in strip order
        self.rotate = None if rotate is None else int(rotate)
        # 1: the waves of a block are dispatched longest first -- constraint
        # waves, then the Jacobian strips by descending evaluation work --
        # instead of in entry order (a launch of a few rounds ends when its
        # last LONG wave ends; with the cheap, store-only strips last the
        # tail is short)
        self.dear_first = int(dear_first)
        # node-invariant sub-expressions evaluated by every lane instead of
        # read from the table opty_uni fills: None = automatic (small
        # problems whose table depends on `free` -- unknown parameters,
        # variable duration -- and would cost a launch of opty_uni in every
        # evaluation), 0 / 1 = never / always
        self.inline_uniform = None if inline_uniform is None \
            else int(bool(inline_uniform))
        # instance-constraint tails evaluated by one extra workgroup of the
        # main kernels instead of a launch of opty_inst: None = automatic
        # (problems small enough for the runtime's latency path, where a
        # launch costs as much as the evaluation: emit_module), 0 / 1 = never
        # / always
        self.fold_instance = None if fold_instance is None \
            else int(bool(fold_instance))
        # how the constraint rows are cut into waves: 'work' balances the
        # waves' operation counts (_constraint_waves), 'count' gives every
        # wave the same number of rows (the fallback when the balanced cut
        # makes a kernel spill vector registers: hip_backend.spill_free)
        assert con_split in ('work', 'count')
        self.con_split = con_split
        # strips of opty_conjac when they differ from opty_jac's ``groups``
        # (None: as ``groups``, or automatic); what a measured launch plan
        # sets (opty_amd/launch_plan.py)
        self.fused_groups = None if fused_groups is None \
            else int(fused_groups)
        # node-major blocks with P < 64: 'flat' stages the whole P x 64 tile
        # and sweeps it as one span (opty_flush_flat), 'chunk' flushes K-entry
        # pieces per node (smaller tile: more waves per CU, which a block of
        # expensive expressions prefers)
        assert small_flush in ('flat', 'chunk')
        self.small_flush = small_flush
        # 1: sin / cos through opty_sincos (opty_device.h: three-FMA reduction
        # + minimax kernels, library path in a cold branch) instead of the
        # library's inline sincos.  Measured on MI355X (profiles/
        # r03_ab_fast_trig.txt): no gain -- 10-link fused 0.1362 vs 0.1357 ms,
        # 24-link 0.3848 vs 0.3841 ms, node shards the same -- the kernels
        # wait on LDS / scalar loads / the store queue, not on the vector
        # ALU; kept as an option, off by default.
        # 2 (r05, experiment): the same kernels behind a WAVE-UNIFORM test --
        # if any lane of the wave has a large (> 2^20), infinite or NaN
        # argument, every lane takes the library's sincos, else every lane
        # the short path: no if / else that narrows EXEC is left on the hot
        # path of the generated kernels, which is where hipcc 7.2 misplaces
        # register copies (profiles/r05_exec_fault.txt)
        assert int(fast_trig) in (0, 1, 2)
        self.fast_trig = int(fast_trig)
        # non-temporal constraint stores: None = automatic (opty_con always;
        # opty_conjac when the constraint vector of a launch is too large to
        # stay in the caches, see emit_module), 0 / 1 = never / always
        self.con_nt = None if con_nt is None else int(con_nt)
        # experiment: ask the compiler for that many waves per SIMD
        # (amdgpu_waves_per_eu: 2 caps the kernels at 256 VGPRs)
        self.occupancy = int(occupancy)
        # experiment: that many empty waves appended to every block of
        # opty_jac (they fill the slab and exit)
        self.pad = int(pad)
        # 1: every wave gets a cheap strip and an expensive strip of the
        # block (see _ModuleWriter.group_ranges); measured slower than one
        # contiguous strip per wave on MI355X (more address streams), kept
        # for experiments
        self.interleave = int(interleave)
        self.con_rows_per_wave = int(con_rows_per_wave)
        # cache-policy bits of the flush stores: nt | sc1 -- the Jacobian is
        # written once and never re-read by the kernel; streaming it past the
        # L2 keeps the slab / uni-table reads (and the fused kernel's
        # constraint waves) L2-resident: +7 % on opty_jac, +27 % on
        # opty_conjac (profiles/r01_tuning.txt)
        self.store_aux = int(store_aux)
        self.flush_unroll = int(flush_unroll)
        # waves per workgroup (they share one input slab); None: as many as
        # it takes to keep 4+ waves resident per CU within its 160 KB of LDS
        self.waves = None if waves is None else int(waves)
        self.chunk = int(chunk)
        assert self.chunk % 2 == 0 and self.chunk >= 2
        assert self.chunk < 16 or self.chunk in (16, 32, 64), \
            'line-mode chunks are 16, 32 or 64 entries'
        # blocks with P >= LINE_MODE_MIN_P use the line-aligned ring flush,
        # which needs chunk to be a multiple of the 16 doubles of a line
        self.groups = groups
        self.max_live = int(max_live)
        # profiling aids (never used by the product): 'store_only' writes a
        # lane-dependent dummy instead of evaluating the expressions,
        # 'compute_only' predicates every flush store off
        assert ablate in (None, 'store_only', 'compute_only', 'only_cheap',
                          'only_dear', 'uni_lit', 'uni_free')
        self.ablate = ablate

    def key(self):
        return ('chunk=%d groups=%s max_live=%d ablate=%s flush_unroll=%d '
                'waves=%s store_aux=%d con_rows_per_wave=%d interleave=%d' % (
                    self.chunk, self.groups, self.max_live, self.ablate,
                    self.flush_unroll, self.waves, self.store_aux,
                    self.con_rows_per_wave, self.interleave) +
                (' pad=%d' % self.pad if self.pad else '') +
                (' occupancy=%d' % self.occupancy if self.occupancy else '') +
                (' con_nt=%d' % self.con_nt if self.con_nt is not None
                 else '') +
                (' fast_trig=%d' % self.fast_trig if self.fast_trig else '') +
                (' fused_groups=%d' % self.fused_groups
                 if self.fused_groups is not None else '') +
                ('' if self.small_flush == 'flat' else ' small_flush=chunk') +
                ('' if self.con_split == 'work' else ' con_split=count') +
                ('' if self.cut is None else ' cut=%s' % self.cut) +
                ('' if self.con_attach is None
                 else ' con_attach=%d' % self.con_attach) +
                (' forget=1' if self.forget else '') +
                ('' if self.rotate is None else ' rotate=%d' % self.rotate) +
                ('' if self.work_live is None
                 else ' work_live=%d' % self.work_live) +
                ('' if self.inline_dynamic is None
                 else ' inline_dynamic=%d' % self.inline_dynamic) +
                ('' if self.fold_instance is None
                 else ' fold_instance=%d' % self.fold_instance) +
                ('' if self.inline_uniform is None
                 else ' inline_uniform=%d' % self.inline_uniform) +
                (' dear_first=1' if self.dear_first else '') +
                ('' if self.order is None else ' order=%s' % self.order) +
                (' trace=1' if self.trace else '') +
                (' park=%d/%d' % (self.park, self.park_live)
                 if self.park else '') +
                (' park_spread=1' if self.park_spread else '') +
                ('' if not self.strips else ' strips=%s' % self.strips) +
                ('' if not self.fused_strips
                 else ' fused_strips=%s' % self.fused_strips) +
                ('' if self.fused_order is None
                 else ' fused_order=%s' % self.fused_order) +
                (' deterministic=1' if self.deterministic else '') +
                (' share_rcp=1' if self.share_rcp else '') +
                (' publish=1' if self.publish else ''))


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

    def __init__(self, dag, needed, leaf, fast_trig=True, deterministic=False,
                 share_rcp=False):
        self.dag = dag
        # denominators that divide two or more needed values (share_rcp)
        self.rcp = {}
        self.shared_den = set()
        if share_rcp:
            count = {}
            for i in needed:
                if dag.op[i] == ir.DIV and dag.op[dag.args[i][1]] != ir.CONST:
                    b = dag.args[i][1]
                    count[b] = count.get(b, 0) + 1
            self.shared_den = {b for b, c in count.items() if c >= 2}
        self.deterministic = bool(deterministic)
        self.trig = {0: '', 1: 'opty_', 2: 'optyu_'}[int(fast_trig)]
        self.lines = []
        self.done = {}
        self.gen = 0
        self.scope = {}
        self.scope_id = 0
        self.scope_start = 0
        self.fetches = []
        self.needed = needed
        self.leaf = leaf

    def new_scope(self):
        """Starts a new scope (chunk): leaves are re-fetched in it."""
        self.end_scope()
        self.scope = {}
        self.scope_id += 1

    def end_scope(self):
        """Ends a fetch batch.  The scalar (node-invariant) loads issued
        since the last call are hoisted to the point where the batch began,
        so that they are issued back to back and merge into wide ``s_load``
        instructions instead of one load + wait in front of every first use.
        Batches are kept small (one output entry): hoisting a whole chunk's
        loads needs hundreds of SGPRs for a 24-link system and spills.
        (Tried and rejected: an LDS table of these values -- the compiler
        prefetches the LDS reads far ahead and the VGPR count explodes;
        scheduling / memory barriers between entries -- no effect on the
        allocation.)"""
        self.lines[self.scope_start:self.scope_start] = self.fetches
        self.fetches = []
        self.scope_start = len(self.lines)

    begin_entry = end_scope

    def forget(self):
        """Drops every computed temporary: what is needed again is computed
        again (under a new name).  Bounds the live values of a long strip at
        the price of re-evaluating what its chunks share."""
        self.done = {}
        self.rcp = {}
        self.gen += 1

    def _name(self, i):
        return 'v%d' % i if self.gen == 0 else 'v%d_%d' % (i, self.gen)

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
        line = 'const double %s = %s;' % (name, src)
        # only scalar loads are hoisted; per-lane LDS reads stay at their
        # first use (hoisted, every input of the batch is live at once)
        if self.dag.uni[i]:
            self.fetches.append(line)
        else:
            self.lines.append(line)
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
        name = self._name(i)
        r = self.ref
        if op == ir.ADD:
            e = '%s + %s' % (r(a[0]), r(a[1]))
        elif op == ir.SUB:
            e = '%s - %s' % (r(a[0]), r(a[1]))
        elif op == ir.MUL:
            e = '%s*%s' % (r(a[0]), r(a[1]))
        elif op == ir.DIV and a[1] in self.shared_den:
            inv = self.rcp.get(a[1])
            if inv is None:
                inv = 'r%d_%d' % (a[1], len(self.lines))
                self.lines.append('const double %s = 1.0/%s;'
                                  % (inv, r(a[1])))
                self.rcp[a[1]] = inv
            e = '%s*%s' % (r(a[0]), inv)
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
        elif op == ir.SELECT:
            rel = {'lt': '<', 'le': '<=', 'eq': '==', 'ne': '!='}[a[0]]
            e = '(%s %s %s ? %s : %s)' % (r(a[1]), rel, r(a[2]), r(a[3]),
                                          r(a[4]))
        elif op in ('sin', 'cos'):
            other = 'cos' if op == 'sin' else 'sin'
            j = d._memo.get((other, a))
            if self.deterministic and j is not None:
                # one form in every wave: the pair, whether this wave needs
                # the other half or not (it is dead code then)
                tag = len(self.lines)
                sn, cn = 'sc%d_%ds' % (i, tag), 'sc%d_%dc' % (i, tag)
                self.lines.append('double %s, %s; %ssincos(%s, &%s, &%s);'
                                  % (sn, cn, self.trig, r(a[0]), sn, cn))
                self.done[i] = sn if op == 'sin' else cn
                if j in self.needed and not self._have(j) and \
                        self.leaf(j) is None:
                    self.done[j] = cn if op == 'sin' else sn
                return
            if (j is not None and j in self.needed and not self._have(j)
                    and self.leaf(j) is None):
                s_id, c_id = (i, j) if op == 'sin' else (j, i)
                sn, cn = self._name(s_id), self._name(c_id)
                self.lines.append('double %s, %s; %ssincos(%s, &%s, &%s);'
                                  % (sn, cn, self.trig, r(a[0]), sn, cn))
                self.done[s_id] = sn
                self.done[c_id] = cn
                return
            e = '%s%s(%s)' % (self.trig, op, r(a[0]))
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


#: LDS parking of a planned wave (``EmitOptions.park``): plain 8-byte LDS
#: accesses, lane-linear (conflict-free)
_PARK_HELPERS = '''\
__device__ __forceinline__ void opty_park(double *park, int slot, int lane,
                                          double v) {
    park[slot*OPTY_WAVE + lane] = v;
}
__device__ __forceinline__ double opty_unpark(const double *park, int slot,
                                              int lane_z) {
    return park[slot*OPTY_WAVE + lane_z];
}'''


class _WavePlan(object):
    """Straight-line schedule of ONE wave with LDS parking
    (``EmitOptions.park``).

    ``targets``: what the wave has to produce, in memory order --
    ``('c', j, root)`` a constraint row (stored when computed) or
    ``('e', k, root)`` the k-th ring write of the wave; ``chunk_of[k]`` = the
    flush after which ring write k's row may be overwritten, i.e. the ring
    writes of chunk c are issued, in order, once ALL its targets are
    computed, followed by the chunk's flush.  ``is_leaf(i)``: node ``i`` is
    fetched, not computed (constants, slab inputs, node-invariant values).

    The plan is a list of events ``('op', node, partner)`` (``partner``: the
    cos / sin of the same argument produced by the same ``sincos``),
    ``('con', j)``, ``('chunk', c)``; ``parks[t]`` = values stored to their
    LDS slot after event ``t`` (``[(node, slot)]``), ``slot_of[node]`` where a
    parked value is reloaded from.  A value sits in a register from its
    computation (or reload) to its eviction or last use; when more than
    ``budget`` are alive the one whose next use is farthest away goes
    (Belady): computed once, stored once, reloaded once per later use
    cluster."""

    def __init__(self, dag, targets, chunks, is_leaf, budget, order=None,
                 spread=False):
        self.spread = bool(spread)
        self.dag = d = dag
        self.targets = targets
        self.chunks = chunks            # chunk c -> list of target indices
        self.is_leaf = is_leaf
        roots = [t[2] for t in targets]
        self.need = [i for i in d.reachable(
            {r for r in roots if not is_leaf(r)}) if not is_leaf(i)]
        self.bit = {i: k for k, i in enumerate(self.need)}
        self._cone = {}
        best = None
        for how in ((order,) if order else ('memory', 'cheapest')):
            events = self._schedule(how)
            peak = self._belady(events, None)[3]
            if best is None or peak < best[0]:
                best = (peak, how, events)
        self.peak, self.order, self.events = best
        self.parks, self.slot_of, self.slots, _ = self._belady(self.events,
                                                               budget)

    # -- order ---------------------------------------------------------------
    def _cone_mask(self, root):
        m = self._cone.get(root)
        if m is None:
            m = 0
            if not self.is_leaf(root):
                for i in self.dag.reachable([root]):
                    k = self.bit.get(i)
                    if k is not None:
                        m |= 1 << k
            self._cone[root] = m
        return m

    def _partner(self, i, done):
        d = self.dag
        if d.op[i] not in ('sin', 'cos'):
            return None
        other = 'cos' if d.op[i] == 'sin' else 'sin'
        j = d._memo.get((other, d.args[i]))
        if j is None or j not in self.bit or j in done:
            return None
        return j

    def _schedule(self, how):
        d = self.dag
        T = len(self.targets)
        work = [k for k in range(T)
                if not self.is_leaf(self.targets[k][2])]
        mask = {k: self._cone_mask(self.targets[k][2]) for k in work}
        done, donemask = set(), 0
        events = []
        finished = [False]*T
        for k in range(T):
            if k not in mask:
                finished[k] = True      # nothing to compute
        chunk_left = [sum(1 for k in c if not finished[k])
                      for c in self.chunks]
        chunk_of = {}
        for c, ks in enumerate(self.chunks):
            for k in ks:
                chunk_of[k] = c
        next_chunk = 0

        # (spread: at most one flush per ``pace`` operations while there are
        # operations left; everything that is still due at the end)
        total_ops = len(self.need)
        pace = max(1, total_ops//max(1, len(self.chunks))) \
            if self.spread else 0
        since = [pace]

        def flush_ready(final=False):
            nonlocal next_chunk
            while next_chunk < len(self.chunks) and \
                    chunk_left[next_chunk] == 0:
                if pace and not final and since[0] < pace:
                    return
                events.append(('chunk', next_chunk))
                next_chunk += 1
                since[0] = 0

        flush_ready()
        rest = list(work)
        while rest:
            if how == 'memory':
                k = rest[0]
            else:
                # cheapest next: the target that needs the fewest values
                # that are not there yet (constraint rows before entries,
                # then memory order)
                k = min(rest, key=lambda k: (
                    (mask[k] & ~donemask).bit_count(), k))
            rest.remove(k)
            events.append(('begin', k))
            root = self.targets[k][2]
            stack = [(root, False)]
            while stack:
                i, ready = stack.pop()
                if i in done or self.is_leaf(i):
                    continue
                if not ready:
                    stack.append((i, True))
                    for j in reversed(d.operands(i)):
                        if j not in done and not self.is_leaf(j):
                            stack.append((j, False))
                    continue
                pair = self._partner(i, done)
                events.append(('op', i, pair))
                for v in (i, pair):
                    if v is not None:
                        done.add(v)
                        donemask |= 1 << self.bit[v]
                since[0] += 1
                if pace:
                    flush_ready()
            if self.targets[k][0] == 'c':
                events.append(('con', k))
            finished[k] = True
            if k in chunk_of:
                chunk_left[chunk_of[k]] -= 1
            flush_ready()
        flush_ready(final=True)
        assert next_chunk == len(self.chunks)
        return events

    # -- registers -----------------------------------------------------------
    def _uses(self, events):
        """``{node: [event indices that read it]}`` (ascending)."""
        d = self.dag
        uses = {}
        for t, ev in enumerate(events):
            if ev[0] == 'op':
                for j in set(d.operands(ev[1])):
                    if not self.is_leaf(j):
                        uses.setdefault(j, []).append(t)
            elif ev[0] == 'con':
                r = self.targets[ev[1]][2]
                if not self.is_leaf(r):
                    uses.setdefault(r, []).append(t)
            elif ev[0] == 'chunk':
                for k in self.chunks[ev[1]]:
                    r = self.targets[k][2]
                    if not self.is_leaf(r):
                        u = uses.setdefault(r, [])
                        if not u or u[-1] != t:
                            u.append(t)
        return uses

    def _belady(self, events, budget):
        """``(parks, slot_of, slots, peak)``; ``budget=None``: only the peak
        number of simultaneously live values."""
        import bisect
        d = self.dag
        uses = self._uses(events)

        def next_use(v, t):
            u = uses.get(v, ())
            k = bisect.bisect_right(u, t)
            return u[k] if k < len(u) else None

        inreg, slot_of, taken, stored = set(), {}, set(), set()
        parks, slots, peak = {}, 0, 0
        self.drops = drops = {}
        for t, ev in enumerate(events):
            if ev[0] == 'op':
                reads = {j for j in d.operands(ev[1])
                         if not self.is_leaf(j)}
                made = [v for v in ev[1:] if v is not None]
            elif ev[0] == 'con':
                r = self.targets[ev[1]][2]
                reads, made = ({r} if not self.is_leaf(r) else set()), []
            else:
                # the ring writes of a chunk are issued one after the other:
                # a parked value is reloaded for its own store only
                reads, made = set(), []
            inreg |= reads
            inreg.update(made)
            peak = max(peak, len(inreg))
            for v in list(inreg):
                if next_use(v, t) is None:
                    inreg.discard(v)
            for v in [v for v in stored if next_use(v, t) is None]:
                # (slot_of[v] stays recorded: its reloads lie behind)
                stored.discard(v)
                taken.discard(slot_of[v])
            if budget is None:
                continue
            while len(inreg) > budget:
                cand = [v for v in inreg if v not in reads and v not in made]
                if not cand:
                    break
                v = max(cand, key=lambda v: (next_use(v, t), v))
                inreg.discard(v)
                drops.setdefault(t, []).append(v)
                if v not in stored:
                    sl = 0
                    while sl in taken:
                        sl += 1
                    taken.add(sl)
                    slot_of[v] = sl
                    stored.add(v)
                    slots = max(slots, sl + 1)
                    parks.setdefault(t, []).append((v, sl))
        return parks, slot_of, slots, peak


class _ModuleWriter(object):

    def __init__(self, prog, opts, inline_uniform=False, literals=None):
        self.p = prog
        self.o = opts
        self.dag = prog.dag
        #: node-invariant nodes whose VALUES are printed (node -> float;
        #: ConstraintCollocator(specialize_parameters=True)) instead of read
        #: from the table opty_uni fills
        self.literals = literals or {}
        # True: no uni[] table -- node-invariant INPUTs are scalar loads from
        # their homes and what depends on them is computed in every lane
        self.inline_uni = bool(inline_uniform)
        self._dyn = {}
        self.inline_dynamic = False
        self.inline_dynamic = self._choose_inline_dynamic()
        self.uni_slot = {}          # uniform frontier node -> slot in uni[]
        self._auto = None           # (G_live, G) of group_ranges()
        self._auto_work = None      # strips of the automatic work-aware cut
        self._work_cut_objective = None
        self._dense = {}            # unit range -> estimated live values
        self._wcost = {}            # weighted work of entry ranges
        self._con_nt = False        # constraint stores of the kernel in print
        self._park_rows = 0         # LDS rows the planned waves park values in
        self._plans = []
        self._pub = None            # publication plan (EmitOptions.publish)
        self._pub_rows = {}         # published node -> LDS row, kernel in print

    # -- leaves -------------------------------------------------------------
    def _is_vec_input(self, i):
        return self.dag.op[i] == ir.INPUT and \
            self.dag.args[i][0] in ('cur', 'adj')

    def _uniform_leaf(self, i):
        """Non-constant node-invariant node: lives in the ``uni`` table
        (without a table: only the scalar inputs are leaves).  Node-invariant
        values that depend on `free` (``_dynamic``) are not table entries when
        they are few (``inline_dynamic``): the scalars they start from are
        leaves, loaded from their homes, and the handful of operations behind
        them are evaluated in the lanes."""
        d = self.dag
        if not d.uni[i] or d.op[i] == ir.CONST:
            return False
        if i in self.literals:
            return True
        if self.inline_uni or (self.inline_dynamic and self._dynamic(i)):
            return d.op[i] == ir.INPUT
        return True

    def _dynamic(self, i):
        """Does node-invariant node ``i`` depend on a value that lives in
        `free` (a free node time interval, an unknown parameter)?  Such
        values change with every evaluation."""
        hit = self._dyn.get(i)
        if hit is None:
            d = self.dag
            if d.op[i] == ir.INPUT:
                hit = self._scalar_source(i).startswith('free_')
            else:
                hit = any(self._dynamic(j) for j in d.operands(i))
            self._dyn[i] = hit
        return hit

    def _choose_inline_dynamic(self):
        """Node-invariant sub-expressions that depend on `free` make the whole
        ``uni`` table dynamic: ``opty_uni`` has to run before EVERY evaluation
        -- a launch of its own (2-5 us next to a 50-70 us evaluation) for, in
        a variable-duration problem, ``1/h`` and a handful of products with
        it (biped: 6 of 78 table entries, 16 operations; muscle-driven leg:
        5 of 132, 15 operations).  When at most ``INLINE_DYNAMIC_MAX_OPS``
        operations are behind them, they are evaluated in the lanes instead
        and the table holds only what changes with the parameters."""
        d, p = self.dag, self.p
        if self.inline_uni or self.o.inline_dynamic == 0:
            return False
        need = [i for i in d.reachable(set(p.con_out) | set(p.jac_out))
                if d.uni[i] and d.op[i] != ir.CONST and self._dynamic(i)]
        if not need:
            return False
        ops = sum(1 for i in need if d.op[i] != ir.INPUT)
        return self.o.inline_dynamic == 1 or ops <= INLINE_DYNAMIC_MAX_OPS

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
    def csr(self):
        return getattr(self.p, 'layout', 'coo') == 'csr'

    def _row_of(self, e):
        """Equation whose row holds stored entry ``e`` (csr layout)."""
        rs = self.p.row_start
        j = 0
        while rs[j + 1] <= e:
            j += 1
        return j

    def _chunks(self, e0, e1):
        K = self.o.chunk
        if self.csr():
            # never across a row; a row that fits the tile is one chunk
            out, rs = [], self.p.row_start
            for j in range(self.p.M):
                a, b = max(e0, rs[j]), min(e1, rs[j + 1])
                step = CSR_MAX_ROW if b - a <= CSR_MAX_ROW else K
                out += [(c, min(c + step, b)) for c in range(a, b, step)]
            return out
        return [(c, min(c + K, e1)) for c in range(e0, e1, K)]

    def line_mode(self):
        """Line-aligned ring flush (see opty_device.h) for all but tiny
        blocks; needs the chunk width to be a multiple of a 16-double line."""
        return self.p.P >= LINE_MODE_MIN_P and self.o.chunk % 16 == 0 \
            and not self.csr()

    def _virtual_end(self, e1):
        """Waves evaluate 15 entries past their range so that they own whole
        lines; the last range wraps into entries 0..14 (of the next node)."""
        return e1 + 15 if self.line_mode() else e1

    def _strip_cost(self, e0, e1):
        """Evaluation work of a strip: DAG nodes that have to be computed per
        node (node-invariant values and constants are free)."""
        d, p = self.dag, self.p
        roots = [p.jac_out[v % p.P] for v in range(e0, self._virtual_end(e1))]
        return sum(1 for i in d.reachable(roots)
                   if d.op[i] not in (ir.CONST, ir.INPUT) and not d.uni[i])

    def _weighted_cost(self, e0, e1):
        """Weighted per-node operations of the strip ``[e0, e1)`` (plus the
        15 entries past its end that its wave evaluates as well)."""
        key = (e0, e1)
        hit = self._wcost.get(key)
        if hit is None:
            d, p = self.dag, self.p
            roots = {p.jac_out[v % p.P]
                     for v in range(e0, self._virtual_end(e1))}
            hit = self._wcost[key] = sum(
                _node_weight(d, i) for i in d.reachable(roots))
        return hit

    def _work_cut(self, S, unit, nunits):
        """``S`` contiguous strips with boundaries on multiples of ``unit``
        entries that minimise ``sum(cost) + WORK_CUT_MAX_WEIGHT*max(cost)``
        over the strips, ``cost`` = weighted evaluation work + store work.

        The sum is what a launch of many blocks pays (every SIMD is busy:
        sub-expressions shared by neighbouring entries should be evaluated by
        ONE wave, not by every wave whose strip crosses them); the largest
        strip is the critical path of a block and what its wave needs in
        registers.  Exact for a given bound on the largest strip (dynamic
        programme over the unit boundaries); the bound is swept over the
        quantiles of all strip costs."""
        P = self.p.P
        wmax = max(1, int(-(-WORK_CUT_MAX_SHARE*nunits//S)))
        ends = [u*unit for u in range(nunits)] + [P]    # boundary k -> entry
        costs = {}
        limit = self.o.work_live or WORK_CUT_MAX_LIVE
        leaf = lambda i: self._is_vec_input(i) or self._uniform_leaf(i)
        dense = set()       # strips that do not fit the register file
        for a in range(nunits):
            for b in range(a + 1, min(nunits, a + wmax) + 1):
                width = ends[b] - ends[a]
                work = self._weighted_cost(ends[a], ends[b])
                # (the small convex term breaks ties between cuts of a
                # store-only region: even pieces, not slivers)
                costs[a, b] = work + \
                    STORE_WEIGHT*width + 0.25*STORE_WEIGHT*width*width/P
                # a wave whose live temporaries exceed the register file
                # spills (the muscle-driven leg: both dynamic rows in one
                # strip, 274 live doubles, 30-60 spilled registers, slower
                # than two strips of 220): such strips are not offered.  (A
                # strip cannot keep more values alive than it computes.)
                if work <= limit:
                    continue
                if (a, b - 1) in dense or (a + 1, b) in dense:
                    dense.add((a, b))
                    continue
                live = self._dense.get((a, b))
                if live is None:
                    live = self._dense[a, b] = _max_live(self.dag, [
                        [self.p.jac_out[v % P] for v in range(c0, c1)]
                        for c0, c1 in self._chunks(
                            ends[a], self._virtual_end(ends[b]))], leaf)
                if live > limit:
                    dense.add((a, b))
        if len(dense) < len(costs):
            # (a single line that is too dense stays: it cannot be cut)
            for ab in dense:
                if ab[1] - ab[0] > 1:
                    del costs[ab]

        def solve(bound):
            INF = float('inf')
            best = [{0: 0.0}]
            back = []
            for s in range(1, S + 1):
                cur, frm = {}, {}
                for b in range(s, nunits - (S - s) + 1):
                    for a in range(max(s - 1, b - wmax), b):
                        prev = best[s - 1].get(a)
                        c = costs.get((a, b))
                        if prev is None or c is None or c > bound:
                            continue
                        if prev + c < cur.get(b, INF):
                            cur[b], frm[b] = prev + c, a
                best.append(cur)
                back.append(frm)
            if nunits not in best[S]:
                return None
            cuts, b = [nunits], nunits
            for s in range(S, 0, -1):
                b = back[s - 1][b]
                cuts.append(b)
            cuts.reverse()
            return cuts

        levels = sorted(set(costs.values()))
        picks = sorted({levels[min(len(levels) - 1, (k*len(levels))//24)]
                        for k in range(25)} | {levels[-1]})
        best_cuts, best_obj = None, None
        for bound in picks:
            cuts = solve(bound)
            if cuts is None:
                continue
            cs = [costs[cuts[g], cuts[g + 1]] for g in range(S)]
            obj = sum(cs) + WORK_CUT_MAX_WEIGHT*max(cs)
            if best_obj is None or obj < best_obj:
                best_cuts, best_obj = cuts, obj
        if best_cuts is None:
            return None             # no S strips within the bounds
        self._work_cut_objective = best_obj
        return [(ends[best_cuts[g]], ends[best_cuts[g + 1]])
                for g in range(S)]

    def _automatic_work_cut(self, even):
        """The work-aware cut of a block whose EVEN cut (``even``: the
        automatic strips) would make the launch wait for recomputed
        arithmetic, or None.

        Every strip evaluates what its entries need, so sub-expressions
        shared by entries of different strips are evaluated once per strip.
        For a store-bound block (the n-link pendulums: 17 x weighted
        operations / bytes of the block well below 1) that hides behind the
        stores and finer strips only help the store stream.  The rows of a
        gait-like system share most of a long computation (contact forces,
        musculotendon curves): the seven-segment biped's 24 even strips
        evaluate 29 600 weighted operations per node where the block has 6
        300, 13 of the 24 waves are long, and the launch takes 0.135 ms for
        326 MB; cut where the work is (``_work_cut``) 4 waves carry the
        arithmetic (12 500 operations in all) and ONE writes the 224 entries
        of the kinematic rows: 0.073 ms (profiles/r04_work_cut.txt).  Fewer
        strips are better as long as no wave outgrows the register file
        (``WORK_CUT_MAX_LIVE``) or has to write too much besides: the count
        is the one with the smallest ``sum + 2 max`` of the strips' costs
        plus ``WORK_CUT_STRIP_OVERHEAD`` per wave (biped: 4 / 5 / 6 / 8 / 10
        strips 0.078 / 0.073 / 0.075 / 0.078 / 0.080 ms fused).  Chosen when
        the even cut's summed work per byte reaches ``ATTACH_MIN_INTENSITY``
        and the work-aware cut saves at least a tenth of it."""
        P = self.p.P
        if not self.line_mode() or self.csr() or self.o.interleave or \
                self.o.ablate is not None:
            return None
        nunits = P//16
        if not 2 < nunits <= WORK_CUT_MAX_UNITS or len(even) < 3:
            return None
        work_even = sum(self._weighted_cost(e0, e1)
                        for grp in even for e0, e1 in grp)
        nbytes = 8.0*64*(P + self.p.M + len(self.p.rows))
        if 17.0*work_even/nbytes < ATTACH_MIN_INTENSITY:
            return None
        first = max(2, (P + 255)//256)
        best = None
        for S in range(first, len(even) + 1):
            cuts = self._work_cut(S, 16, nunits)
            if cuts is None:
                continue
            obj = self._work_cut_objective + WORK_CUT_STRIP_OVERHEAD*S
            if best is None or obj < best[0]:
                best = (obj, cuts)
            elif S > len(best[1]) + 2:
                break               # more strips only add waves from here on
        if best is None:
            return None
        work = sum(self._weighted_cost(e0, e1) for e0, e1 in best[1])
        return best[1] if work <= 0.9*work_even else None

    def group_ranges(self, count=None):
        """Assigns the P entries of the block to G waves (``count`` of them,
        or the printer option ``groups``, or the automatic choice).  The block is cut
        into strips (contiguous entry ranges whose boundaries are multiples of
        a 16-double line, or of the chunk width for tiny blocks); a wave gets
        one strip, or -- ``interleave`` -- two: the cheapest remaining and the
        most expensive remaining one, so that every wave carries the same mix
        of store-only entries (structural zeros, constants) and evaluation
        work instead of half the waves only storing and half only computing.
        With ``groups=None`` G is ``auto_groups()[1]``.  Returns a list of
        groups, each a list of ``(e0, e1)`` strips in evaluation order."""
        P, K = self.p.P, self.o.chunk
        if self.o.strips and self.line_mode() and count is None:
            groups = self.explicit_strips(self.o.strips)
            self._auto = (len(groups), len(groups))
            return groups
        unit = 16 if self.line_mode() else K
        nunits = max(1, P//unit if self.line_mode() else (P + K - 1)//K)
        two = bool(self.o.interleave) and self.line_mode()
        if self.csr():
            # strips are whole rows: cut where a row starts
            starts = sorted(set(self.p.row_start[:-1]) - {P})
            nunits = max(1, len(starts))

        def cut(S):
            if self.csr():
                # the row start nearest to an even share of the entries
                b = [0]
                for g in range(1, S):
                    rest = [x for x in starts if x > b[-1]]
                    keep = S - g - 1        # starts the later waves need
                    rest = rest[:len(rest) - keep] if keep else rest
                    b.append(min(rest, key=lambda x: abs(x - g*P/S)))
                b.append(P)
                return [(b[g], b[g + 1]) for g in range(S)]
            if self.o.cut == 'work' and self.line_mode() and \
                    1 < S < nunits <= WORK_CUT_MAX_UNITS:
                strips = self._work_cut(S, unit, nunits)
                if strips is not None:
                    return strips
            b = [((g*nunits)//S)*unit for g in range(S)] + [P]
            return [(b[g], b[g + 1]) for g in range(S)]

        def split(G):
            if not two or 2*G > nunits//2:
                return [[rg] for rg in cut(G)]
            strips = cut(2*G)
            order = sorted(range(2*G),
                           key=lambda k: (self._strip_cost(*strips[k]), k))
            groups = []
            for g in range(G):
                cheap, dear = strips[order[g]], strips[order[2*G - 1 - g]]
                # alternate the order so that at any time some waves store
                # while others evaluate
                groups.append([cheap, dear] if g % 2 == 0 else [dear, cheap])
            return groups

        if count is None and self.o.groups is not None:
            count = int(self.o.groups)
        if count is not None:
            return split(max(1, min(int(count), nunits)))
        if self._auto is None:
            leaf = lambda i: self._is_vec_input(i) or self._uniform_leaf(i)

            def live(G):
                return max(
                    _max_live(self.dag,
                              [[self.p.jac_out[v % P] for v in range(a, b)]
                               for e0, e1 in grp
                               for a, b in self._chunks(
                                   e0, self._virtual_end(e1))],
                              leaf)
                    for grp in split(G))

            # at least one wave per 256 entries (2 KB of every node row) so
            # that even an all-constant block yields enough waves
            first = max(1, min(nunits, (P + 255)//256))
            top = min(nunits, 32)
            # Registers: the fewest strips whose waves' estimated live
            # temporaries stay below ``max_live`` -- or, for systems whose
            # shared per-node values alone exceed that (a 24-link pendulum
            # keeps ~235 alive however finely the block is cut), within 10 %
            # of what the finest cut achieves: finer strips no longer help
            # the registers then.
            cand = sorted({g for g in list(range(first, 9)) +
                           [10, 12, 16, 20, 24, top] if first <= g <= top})
            budget = self.o.max_live
            worst = {cand[0]: live(cand[0])}
            if worst[cand[0]] > budget and len(cand) > 1:
                worst[top] = live(top)
                budget = max(budget, 1.1*worst[top])
            G = cand[-1]
            for g in cand:
                if g not in worst:
                    worst[g] = live(g)
                if worst[g] <= budget:
                    G = g
                    break
            fine = G
            if self.line_mode():
                fine = max(G, min(nunits, 32, -(-P//STRIP_ENTRIES)))
            self._auto = (G, fine)
            if self.o.cut is None:
                strips = self._automatic_work_cut(split(fine))
                if strips is not None:
                    self._auto_work = strips
                    self._auto = (min(G, len(strips)), len(strips))
        if self._auto_work is not None:
            return [[rg] for rg in self._auto_work]
        return split(self._auto[1])

    def explicit_strips(self, spec):
        """``'96:160;160:348+0:96'`` -> ``[[(96, 160)], [(160, 348), (0,
        96)]]``, checked: boundaries on 16-entry lines, every entry once."""
        P = self.p.P
        groups = [[tuple(int(x) for x in rg.split(':'))
                   for rg in grp.split('+')] for grp in spec.split(';')]
        cover = sorted(rg for grp in groups for rg in grp)
        assert cover[0][0] == 0 and cover[-1][1] == P and all(
            a[1] == b[0] for a, b in zip(cover, cover[1:])) and all(
                e0 % 16 == 0 and e1 > e0 for e0, e1 in cover), cover
        return groups

    def auto_groups(self):
        """``(G_live, G)``: the fewest strips whose waves' estimated live
        temporaries stay below ``max_live`` (registers), and the automatic
        choice -- at least that, and about one wave per ``STRIP_ENTRIES``
        entries (node-major line-mode blocks; at most 32)."""
        if self._auto is None:
            self.group_ranges()
        return self._auto

    # -- publication of isomorphic sub-models (EmitOptions.publish) -----------
    def _publication(self):
        """``(rows, tasks)``: ``rows`` = {published node: LDS row}, ``tasks``
        = [(interface nodes of one instance, its weight)], from the
        isomorphic instance groups of the block (``codegen/isomorph.py``);
        None when there is nothing worth a barrier."""
        if self._pub is None:
            from . import isomorph
            d, p = self.dag, self.p
            is_leaf = isomorph.default_leaf(d)
            roots = list(p.con_out) + list(p.jac_out)
            need = set(i for i in d.reachable(roots) if not is_leaf(i))
            users = {}
            for i in need:
                for j in d.operands(i):
                    if j in need:
                        users.setdefault(j, []).append(i)
            outs = set(roots)
            rows, tasks = {}, []
            for g in isomorph.instance_groups(
                    d, roots, lambda i: _node_weight(d, i), is_leaf):
                for r in g['roots']:
                    c = isomorph.cone(d, r, is_leaf)
                    iface = sorted(v for v in c if v in outs or any(
                        u not in c for u in users.get(v, ())))
                    for v in iface:
                        rows.setdefault(v, len(rows))
                    tasks.append((iface, g['weight']))
            self._pub = (rows, tasks) if tasks else False
        return self._pub or None

    def _publish_stage(self, W, slab_of):
        """Code of the stage between the slab fill and the strips: wave ``w``
        of the workgroup evaluates its share of the instances (longest
        first onto the least loaded wave) and stores their interface values
        into the ``pub`` rows; one barrier."""
        rows, tasks = self._publication()
        d = self.dag
        loads, mine = [0]*W, [[] for _ in range(W)]
        for iface, weight in sorted(tasks, key=lambda t: -t[1]):
            w = loads.index(min(loads))
            loads[w] += weight
            mine[w].append(iface)
        leaf = self._leaf_fn(slab_of, {})
        lines = ['switch (wave) {']
        stored = set()
        for w in range(W):
            if not mine[w]:
                continue
            need = set(d.reachable([v for iface in mine[w] for v in iface]))
            body = _Body(d, need, leaf, self.o.fast_trig,
                         self.o.deterministic, self.o.share_rcp)
            for iface in mine[w]:
                body.new_scope()
                for v in iface:
                    ref = body.emit(v)
                    if v not in stored:
                        stored.add(v)
                        body.lines.append('pub[%d + lane] = %s;'
                                          % (rows[v]*TS, ref))
            body.end_scope()
            lines.append('case %d: {' % w)
            lines += ['    ' + ln for ln in body.lines]
            lines.append('} break;')
        lines += ['default: break;', '}', '__syncthreads();']
        return lines, max(loads)

    def _leaf_fn(self, slab_of, published):
        """``leaf(i)`` of a wave's straight-line code: how a value that is
        fetched rather than computed is fetched (input slab, node-invariant
        table / literal / scalar home, published row), or None."""
        p, d = self.p, self.dag

        def leaf(i):
            if i in published:
                return 'pub[%d + lane]' % (published[i]*TS)
            if self._is_vec_input(i):
                kind, r = d.args[i]
                off = p.cur_offset if kind == 'cur' else p.adj_offset
                return 'lds[%d + lane + %d]' % (slab_of[r]*TS, off)
            if self._uniform_leaf(i):
                if i in self.literals:
                    return _lit(self.literals[i])
                if self.inline_uni or (self.inline_dynamic and
                                       self._dynamic(i)):
                    return self._scalar_source(i)
                # diagnostics (wrong values): what would the kernel cost if
                # node-invariant operands were literals / free?
                if self.o.ablate == 'uni_lit':
                    return _lit(1.0 + 1.0/(3.0 + self._slot(i)))
                if self.o.ablate == 'uni_free':
                    self._slot(i)
                    return 'h'
                return 'uni_c[%d]' % self._slot(i)
            return None
        return leaf

    # -- kernels ---------------------------------------------------------------
    def _kernel_rows(self, groups, con_of_group):
        """Trajectory rows any wave of the kernel reads (the shared slab)."""
        p, d = self.p, self.dag
        roots = list(self._pub_rows)
        for grp, cons in zip(groups, con_of_group):
            for e0, e1 in grp:
                vend = self._virtual_end(e1) if e1 > e0 else e1
                roots += [p.jac_out[v % p.P] for v in range(e0, vend)]
            roots += [p.con_out[j] for j in cons]
        needed = d.reachable(roots)
        return sorted({d.args[i][1] for i in needed if self._is_vec_input(i)})

    def _ring_rows(self, grp):
        K = self.o.chunk
        width = max(e1 - e0 for e0, e1 in grp)
        if width <= 0:
            return 0
        if self.line_mode():
            return K + 16
        if self.csr():
            rs = self.p.row_start
            wide = any(rs[j + 1] - rs[j] > CSR_MAX_ROW and K % 16 == 0
                       for e0, e1 in grp for j in range(self.p.M)
                       if e0 <= rs[j] < e1)
            return max([K + 16 if wide else 0] +
                       [b - a for e0, e1 in grp
                        for a, b in self._chunks(e0, e1)])
        if self._whole_block_strip(grp):
            return width
        return min(K, width)

    def _group_body(self, grp, con_rows, slab_of):
        """Code for one wave evaluating the Jacobian entry strips ``grp``
        (``[(e0, e1), ...]``, in this order) and the constraint rows
        ``con_rows`` of its 64 nodes.  Inputs come from the workgroup's shared
        slab (``lds``), outputs are staged in the wave's private ring tile
        (``ring``)."""
        p, d = self.p, self.dag
        K = self.o.chunk
        roots = [p.con_out[j] for j in con_rows]
        for e0, e1 in grp:
            vend = self._virtual_end(e1) if e1 > e0 else e1
            roots += [p.jac_out[v % p.P] for v in range(e0, vend)]
        needed = set(d.reachable(roots))
        R = K + 16

        leaf = self._leaf_fn(slab_of, self._pub_rows)

        body = _Body(d, needed, leaf, self.o.fast_trig,
                     self.o.deterministic, self.o.share_rcp)
        # LDS parking: the wave is planned as a whole (constraint rows
        # included) when it is worth it -- its temporaries in memory order
        # would not fit the registers
        planned = bool(self.o.park and self.line_mode() and
                       self.o.ablate is None and not self.o.forget and
                       any(e1 > e0 for e0, e1 in grp))
        if planned:
            est = _max_live(d, [[p.con_out[j]] for j in con_rows] + [
                [p.jac_out[v % p.P] for v in range(a, b)]
                for e0, e1 in grp if e1 > e0
                for a, b in self._chunks(e0, self._virtual_end(e1))],
                lambda i: leaf(i) is not None)
            planned = est > self.o.park_live
        for j in ([] if planned else con_rows):
            body.new_scope()      # fetches hoisted per row, not per kernel
            ref = body.emit(p.con_out[j])
            if self._con_nt:
                body.lines.append(
                    'if (valid) __builtin_nontemporal_store(%s, '
                    '&con[%dLL*con_stride + node]);' % (ref, j))
            else:
                body.lines.append(
                    'if (valid) con[%dLL*con_stride + node] = %s;'
                    % (j, ref))
        nv = '(N < 0 ? nvalid : 0)' if self.o.ablate == 'compute_only' \
            else 'nvalid'

        def value(e):
            if self.o.ablate == 'store_only':
                return '(double)(lane + %d)' % e
            return body.emit(p.jac_out[e])

        strips = [rg for rg in grp if rg[1] > rg[0]]
        if strips and self.line_mode():
            body.lines.append('const int b0 = opty_line_phase(jrow);')
        if strips and planned:
            self._planned_strips(body, strips, con_rows, nv, R)
            strips = []
        for e0, e1 in strips:
            body.lines.append('// strip %d %d' % (e0, e1))
            if self.line_mode():
                self._strip_lines(body, e0, e1, value, nv, R)
            elif self.csr():
                self._strip_csr(body, e0, e1, value, nv)
            elif self._whole_block_strip(grp):
                self._strip_flat(body, value, nv)
            else:
                self._strip_simple(body, e0, e1, value, nv)
        body.end_scope()
        return body.lines

    def _planned_strips(self, body, strips, con_rows, nv, R):
        """The wave's code from a :class:`_WavePlan`: constraint rows and
        entry expressions in the planned order, long-lived values parked in
        the wave's LDS rows (``park``), ring writes and flushes chunk by
        chunk in memory order."""
        p, d = self.p, self.dag
        K = self.o.chunk
        targets = [('c', j, p.con_out[j]) for j in con_rows]
        chunks, flush = [], []
        for e0, e1 in strips:
            for c0 in range(e0, e1 + 15, K):
                c1 = min(c0 + K, e1 + 15)
                ks = []
                for v in range(c0, c1):
                    ks.append(len(targets))
                    targets.append(('e', v, p.jac_out[v % p.P]))
                chunks.append(ks)
                flush.append((e0, e1, c0, c1))

        def is_leaf(i):
            return d.op[i] == ir.CONST or body.leaf(i) is not None

        plan = _WavePlan(d, targets, chunks, is_leaf, self.o.park_live,
                         spread=self.o.park_spread)
        self._park_rows = max(self._park_rows, plan.slots)
        self._plans.append(dict(strips=strips, order=plan.order,
                                peak=plan.peak, slots=plan.slots,
                                ops=sum(1 for e in plan.events
                                        if e[0] == 'op')))
        lines = body.lines
        reloads = [0]

        def operand(j):
            if d.op[j] == ir.CONST or j in body.scope or j in body.done:
                return
            if body.leaf(j) is not None:
                body._fetch(j)
                return
            reloads[0] += 1
            name = 'v%d_r%d' % (j, reloads[0])
            lines.append('const double %s = opty_unpark(park, %d, lane_z);'
                         % (name, plan.slot_of[j]))
            body.done[j] = name

        for t, ev in enumerate(plan.events):
            if ev[0] == 'begin':
                body.new_scope()
            elif ev[0] == 'op':
                i, pair = ev[1], ev[2]
                for j in d.operands(i):
                    operand(j)
                if d.op[i] in ('sin', 'cos') and body.deterministic:
                    body._emit_node(i)
                    if pair is None:
                        # (the plan did not ask for the other half here: it
                        # is computed again where it is needed)
                        other = 'cos' if d.op[i] == 'sin' else 'sin'
                        j = d._memo.get((other, d.args[i]))
                        if j is not None and j in body.done and \
                                body.done[j].startswith('sc%d_' % i):
                            del body.done[j]
                elif d.op[i] in ('sin', 'cos'):
                    a = body.ref(d.args[i][0])
                    if pair is not None:
                        s_id, c_id = (i, pair) if d.op[i] == 'sin' \
                            else (pair, i)
                        sn, cn = body._name(s_id), body._name(c_id)
                        lines.append('double %s, %s; %ssincos(%s, &%s, &%s);'
                                     % (sn, cn, body.trig, a, sn, cn))
                        body.done[s_id], body.done[c_id] = sn, cn
                    else:
                        name = body._name(i)
                        lines.append('const double %s = %s%s(%s);'
                                     % (name, body.trig, d.op[i], a))
                        body.done[i] = name
                else:
                    body._emit_node(i)
            elif ev[0] == 'con':
                j, root = targets[ev[1]][1], targets[ev[1]][2]
                operand(root)
                ref = body.ref(root)
                if self._con_nt:
                    lines.append('if (valid) __builtin_nontemporal_store(%s, '
                                 '&con[%dLL*con_stride + node]);' % (ref, j))
                else:
                    lines.append('if (valid) con[%dLL*con_stride + node] = '
                                 '%s;' % (j, ref))
            else:
                c = ev[1]
                e0, e1, c0, c1 = flush[c]
                if c0 == e0:
                    lines.append('// strip %d %d' % (e0, e1))
                body.new_scope()
                for k in chunks[c]:
                    v, root = targets[k][1], targets[k][2]
                    if is_leaf(root):
                        val = body.emit(root)
                    elif root in body.done:
                        val = body.done[root]
                    else:       # parked: straight from its row to the ring
                        val = 'opty_unpark(park, %d, lane_z)' \
                            % plan.slot_of[root]
                    lines.append('ring[%d + lane] = %s;' % ((v % R)*TS, val))
                self._flush_chunk(body, e0, e1, c0, c1, nv, R, p.P)
            for v, sl in plan.parks.get(t, ()):
                lines.append('opty_park(park, %d, lane, %s);'
                             % (sl, body.done[v]))
            for v in plan.drops.get(t, ()):
                body.done.pop(v, None)

    def _whole_block_strip(self, grp):
        """A wave that evaluates the WHOLE block of a small (P < 64)
        node-major system: its 64 nodes' values are one contiguous span."""
        strips = [rg for rg in grp if rg[1] > rg[0]]
        return (not self.line_mode() and not self.csr() and
                self.o.small_flush == 'flat' and
                strips == [(0, self.p.P)] and self.p.P <= CSR_MAX_ROW)

    def _strip_flat(self, body, value, nv):
        """Small blocks (P < 64), node-major: the P values of the wave's 64
        nodes are ONE contiguous span of 64*P doubles, ``jrow[nd*P + e]``.  The
        whole P x 64 tile is staged and swept front to back by
        ``opty_flush_flat`` -- every store instruction writes eight whole
        128-byte lines -- instead of K-entry pieces at arbitrary 16-byte
        offsets per node (``_strip_simple``: 256-byte segments, the pattern
        that runs at half the line-aligned rate, profiles/r01_store_bench.txt)."""
        P, K = self.p.P, self.o.chunk
        for e in range(P):
            if e % K == 0:
                body.new_scope()
            body.begin_entry()
            body.lines.append('ring[%d + lane] = %s;' % (e*TS, value(e)))
        body.lines.append('opty_wave_sync();')
        body.lines.append('opty_flush_flat<%d>(ring, jrow, %s, lane);'
                          % (P, nv))
        body.lines.append('opty_wave_sync();')

    def _strip_lines(self, body, e0, e1, value, nv, R, width=None,
                     jrow='jrow', b0='b0'):
        """Ring tile + line-aligned flush of one strip (see opty_device.h).
        ``width`` doubles separate two nodes in the output (the block width P,
        or one equation's row length in the row-sorted layout), ``jrow`` points
        at the wave's first node there, ``b0`` is its line phase."""
        P = self.p.P if width is None else width
        K = self.o.chunk
        if e1 < P:
            assert e1 + 15 <= P, 'last entry range must be >= 16 wide'
        for c0 in range(e0, e1 + 15, K):
            c1 = min(c0 + K, e1 + 15)
            body.new_scope()
            if self.o.forget and c0 > e0:
                body.forget()
            for v in range(c0, c1):
                body.begin_entry()
                body.lines.append('ring[%d + lane] = %s;'
                                  % ((v % R)*TS, value(v % P)))
            self._flush_chunk(body, e0, e1, c0, c1, nv, R, P, jrow, b0)

    def _flush_chunk(self, body, e0, e1, c0, c1, nv, R, P, jrow='jrow',
                     b0='b0'):
        """The flush after the ring writes of the chunk ``[c0, c1)`` of the
        strip ``[e0, e1)``."""
        body.lines.append('opty_wave_sync();')
        nlp = 1
        while 16*nlp < c1 - c0:
            nlp *= 2
        body.lines.append(
            'opty_flush_lines<%d, %d, %d>(ring, %s, %d, %s, %d, %d,'
            ' %d, %d, %d, %s, lane);' % (
                nlp, R, self.o.flush_unroll, jrow, P, b0, c0 - 15,
                (c0 - 15) % R, e0, e1, c1, nv))
        if e0 == 0 and c0 == 0:
            assert c1 >= 15
            if self.o.ablate != 'compute_only':
                body.lines.append('opty_head_piece<%d>(ring, %s, '
                                  '%d, %s, lane);' % (R, jrow, P, b0))
        body.lines.append('opty_wave_sync();')

    def _strip_csr(self, body, e0, e1, value, nv):
        """Row-sorted layout: equation j's L entries of the wave's 64 nodes
        are one contiguous span ``jac[S_j*ncn + i*L + pos]`` (``S_j`` entries
        precede row j in a block, ``ncn`` constraint nodes in this launch).
        Rows up to CSR_MAX_ROW entries are staged whole and written front to
        back (opty_flush_flat); wider rows go through the ring tile with the
        row as the "block" (line-aligned flush, as the node-major layout)."""
        p = self.p
        K = self.o.chunk
        for j in range(p.M):
            S, L = p.row_start[j], p.row_start[j + 1] - p.row_start[j]
            if L == 0 or S < e0 or S >= e1:
                continue
            assert S + L <= e1, 'strips hold whole rows'
            dst = 'jac + %dLL*ncn + nloc*%dLL' % (S, L)
            if L > CSR_MAX_ROW and K % 16 == 0:
                body.end_scope()
                body.lines.append('double *const jrow%d = %s;' % (j, dst))
                body.lines.append('const int b0_%d = opty_line_phase(jrow%d);'
                                  % (j, j))
                body.scope_start = len(body.lines)
                self._strip_lines(body, 0, L, lambda v: value(S + v), nv,
                                  K + 16, L, 'jrow%d' % j, 'b0_%d' % j)
                continue
            for c0, c1 in self._chunks(S, S + L):
                body.new_scope()
                for e in range(c0, c1):
                    body.begin_entry()
                    body.lines.append('ring[%d + lane] = %s;'
                                      % ((e - c0)*TS, value(e)))
                body.lines.append('opty_wave_sync();')
                if c0 == S and c1 == S + L:
                    body.lines.append('opty_flush_flat<%d>(ring, %s, %s, '
                                      'lane);' % (L, dst, nv))
                else:
                    body.lines.append('opty_flush8<%d>(ring, %s + %d, %dLL, '
                                      '%s, lane);' % (c1 - c0, dst, c0 - S,
                                                      L, nv))
                body.lines.append('opty_wave_sync();')

    def _strip_simple(self, body, e0, e1, value, nv):
        """Per-chunk tile + flush for tiny blocks (P < 64)."""
        p = self.p
        wide = (p.P % 2 == 0)
        for c0, c1 in self._chunks(e0, e1):
            body.new_scope()
            for e in range(c0, c1):
                body.begin_entry()
                body.lines.append('ring[%d + lane] = %s;'
                                  % ((e - c0)*TS, value(e)))
            body.lines.append('opty_wave_sync();')
            w = c1 - c0
            fl = 'opty_flush16' if (wide and w % 2 == 0 and c0 % 2 == 0) \
                else 'opty_flush8'
            body.lines.append('%s<%d>(ring, jrow + %d, %dLL, %s, lane);'
                              % (fl, w, c0, p.P, nv))
            body.lines.append('opty_wave_sync();')

    def _slab_fill(self, rows, slab_of, W):
        """Cooperative slab fill: the workgroup's W waves split the rows.  A
        wave issues EVERY global load of its share first (65 time nodes per
        row: one per lane plus the halo node, whose address is wave-uniform),
        then the LDS writes.  Load-by-load (`load; wait; ds_write; branch`)
        costs one memory round trip per row and dominated the wave's life."""
        if not rows:
            return []
        lines = ['const long long t_ld = node0 + lane < N - 1 ? '
                 'node0 + lane : N - 1;',
                 'const long long t_halo = node0 + 64 < N - 1 ? '
                 'node0 + 64 : N - 1;']

        def share(w):
            mine = rows[w::W]
            out = []
            for r in mine:
                out.append('const double sl%d = (%s)[t_ld];'
                           % (r, self._row_ptr(r)))
                out.append('const double sh%d = (%s)[t_halo];'
                           % (r, self._row_ptr(r)))
            for r in mine:
                out.append('lds[%d + lane] = sl%d;' % (slab_of[r]*TS, r))
            if mine:
                out.append('if (lane == 0) {')
                for r in mine:
                    out.append('    lds[%d] = sh%d;'
                               % (slab_of[r]*TS + WAVE, r))
                out.append('}')
            return out

        if W == 1:
            return lines + share(0) + ['opty_wave_sync();']
        lines.append('switch (wave) {')
        for w in range(W):
            lines.append('case %d: {' % w)
            lines += ['    ' + ln for ln in share(w)]
            lines.append('} break;')
        lines.append('default: break;')
        lines.append('}')
        lines.append('__syncthreads();')
        return lines

    _PROLOGUE = '''\
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nblk = (node_end - node_begin + 63)/64;
    // XCD-aware placement: consecutive workgroup ids round-robin the 8 XCDs;
    // all workgroups of one 64-node block (same input slab, interleaved
    // strips of the same output rows) go to the SAME XCD / L2, back to back.
    const long long xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    {mapping}
    if (blk >= nblk) return;
    const long long node0 = node_begin + blk*64;
    const long long node = node0 + lane;
    const bool valid = node < node_end;
    const long long rem = node_end - node0;
    const int nvalid = rem < 64 ? (int)rem : 64;
    const long long ncn = node_end - node_begin, nloc = node0 - node_begin;
    double *jrow = jac + nloc*{P}LL;
    double *const ring = lds + {slab} + wave*{ring};
    (void)valid; (void)jrow; (void)nvalid; (void)node; (void)ring; (void)grp;
    (void)ncn;
'''

    # order == 'list': a persistent kernel.  At most RESIDENT_WAVES one-wave
    # workgroups (one per SIMD: these waves hold one alone), each evaluates
    # the list of (block, strip class) items that the runtime's list schedule
    # gives it (``sched``, one more kernel parameter: csrc/runtime.cpp
    # build_schedule) -- instead of one workgroup per item, dispatched by the
    # hardware as SIMDs fall idle.
    _LIST_HEAD = '''\
    const int lane_o = threadIdx.x & 63;
    const int wave = 0;
    const long long nblk = (node_end - node_begin + 63)/64;
    const long long xcd = blockIdx.x & 7;
    const int npw = sched[0];
'''
    _LIST_LOOP = '''\
    const int *const items = sched + 2 + npw;
    int it = sched[1 + blockIdx.x];
    const int it_end = sched[2 + blockIdx.x];
    int item = it < it_end ? items[it] : -1;
    const double h_o = h;
    const long long N_o = N, stride_o = con_stride, begin_o = node_begin,
        end_o = node_end;
'''
    # Inside the loops every kernel argument and the lane number are re-defined
    # through an empty ``asm volatile``: to the compiler nothing an item
    # computes is loop-invariant.  Otherwise it hoists what the strips derive
    # from them -- scalar loads of parameters and table values, row addresses,
    # per-lane LDS addresses -- out of the loop, where it stays live through
    # every item: hundreds of spilled SGPRs in VGPR lanes and spilled VGPRs in
    # kernels that have no register to spare (biped: 58 -> 390 SGPR spills,
    # 0 -> 77 spilled VGPRs).  The modules are built without MachineLICM for
    # the same reason (materialised constants).  (Pointers: the kernel's own
    # ``__restrict__`` parameters plus an opaque zero -- a pointer that went
    # through the asm itself would lose what lets the parameter loads be
    # scalar loads.)
    _LIST_ITEM = '''\
    while (item >= 0) {{
    const long long blk = (long long)(item & 0xffffff)*8 + xcd;
    const int grp = item >> 24;
    ++it;
    const int item_next = it < it_end ? items[it] : -1;
    const long long opq = opty_opaque(0LL);
    const double *const free_i = free_ + opq, *const known_i = known_traj + opq;
    const double *const params_i = params + opq, *const uni_i = uni_c + opq;
    const long long *const inst_i = inst_idx + opq;
    double *const con_i = con + opq, *const jac_i = jac + opq;
    {{
    const double *const free_ = free_i, *const known_traj = known_i;
    const double *const params = params_i, *const uni_c = uni_i;
    const long long *const inst_idx = inst_i;
    double *const con = con_i, *const jac = jac_i;
    const double h = opty_opaque(h_o);
    const long long N = opty_opaque(N_o), con_stride = opty_opaque(stride_o);
    const long long node_begin = opty_opaque(begin_o);
    const long long node_end = opty_opaque(end_o);
    const int lane = opty_opaque_lane(lane_o);
    (void)known_traj; (void)params; (void)uni_c; (void)inst_idx; (void)con;
    (void)h; (void)con_stride;
    if (blk < nblk) {{
    const long long node0 = node_begin + blk*64;
    const long long node = node0 + lane;
    const bool valid = node < node_end;
    const long long rem = node_end - node0;
    const int nvalid = rem < 64 ? (int)rem : 64;
    const long long ncn = node_end - node_begin, nloc = node0 - node_begin;
    double *jrow = jac + nloc*{P}LL;
    double *const ring = lds + {slab} + wave*{ring};
    (void)valid; (void)jrow; (void)nvalid; (void)node; (void)ring; (void)grp;
    (void)ncn;
'''
    _LIST_NEXT = '''\
    }
    }
    opty_wave_sync();
    item = item_next;
    }
'''

    def kernel(self, name, groups, con_of_group, W=1, con_nt=False,
               inst_lines=None, first_group=0, order=None):
        """One kernel.  ``groups`` = one list of entry strips ``(e0, e1)`` per
        wave; ``con_of_group[g]`` = constraint rows stored by wave g.  A
        workgroup is ``W`` consecutive groups of one 64-node block: they share
        one input slab (filled cooperatively) and each owns a ring tile.
        ``inst_lines``: body of the instance-constraint tails, run by lane 0
        of the first workgroup AFTER the node blocks' (the runtime launches
        it only with whole-problem evaluations).  ``first_group``: the
        workgroup of a block that holds this wave is dispatched first, the
        others follow cyclically."""
        G = len(groups)
        self._con_nt = bool(con_nt)
        keep = [True]*G
        if self.o.ablate in ('only_cheap', 'only_dear'):
            cheap = [sum(self._strip_cost(*rg) for rg in grp if rg[1] > rg[0])
                     <= 16 and not con_of_group[g]
                     for g, grp in enumerate(groups)]
            keep = [c == (self.o.ablate == 'only_cheap') for c in cheap]
        # publication of the block's isomorphic sub-models (one workgroup
        # per block, a stage and a barrier before the strips): Jacobian
        # kernels of line-mode blocks with several waves, hardware dispatch
        self._pub_rows = {}
        pub = None
        if self.o.publish and name in ('opty_jac', 'opty_conjac') and \
                G > 1 and G <= 16 and self.line_mode() and \
                (order or self.o.order) != 'list' and not self.o.park and \
                self.o.ablate is None and not self.o.trace:
            pub = self._publication()
        if pub is not None:
            self._pub_rows = pub[0]
        rows = self._kernel_rows([g for g, k in zip(groups, keep) if k],
                                 [c for c, k in zip(con_of_group, keep) if k])
        ring_rows = max([self._ring_rows(g) for g in groups] + [0])
        if pub is not None:
            nring = sum(any(e1 > e0 for e0, e1 in grp) for grp in groups)
            if (len(rows) + nring*ring_rows + len(pub[0]))*TS*8 > 160*1024:
                # does not fit a CU's LDS next to the ring tiles
                pub, self._pub_rows = None, {}
                rows = self._kernel_rows(
                    [g for g, k in zip(groups, keep) if k],
                    [c for c, k in zip(con_of_group, keep) if k])
            else:
                W = G
        slab_of = {r: k for k, r in enumerate(rows)}
        if W is None:
            if G <= 4 and any(con_of_group) and not self.o.strips and \
                    not self.o.fused_strips and \
                    self.o.cut != 'work' and self._auto_work is None and \
                    any(e1 > e0 for grp in groups for e0, e1 in grp):
                # the fused kernel of a small block: its few waves form ONE
                # workgroup -- one slab fill, and the constraint wave rides in
                # the Jacobian wave's LDS instead of reserving a ring tile it
                # never uses in a workgroup of its own.  (Not the unequal
                # waves of a work-aware cut: a workgroup keeps its CU's LDS
                # until its longest wave is done.)
                W = G
            else:
                W = self._waves_per_workgroup(len(rows), ring_rows)
        W = max(1, min(W, G))
        order = order or self.o.order
        listed = order == 'list' and G <= MAX_CLASSES and any(
            e1 > e0 for grp in groups for e0, e1 in grp)
        if listed:
            W = 1
        # a hand-set workgroup width that does not fit the CU's LDS (64-entry
        # chunks of a 4-wave workgroup next to a large slab) is narrowed
        # instead of failing in hipcc ("local memory exceeds limit")
        while W > 1 and (len(rows) + W*ring_rows)*TS*8 > 160*1024:
            W -= 1
        sets = (G + W - 1)//W
        self._park_rows = 0
        bodies = [self._group_body(grp, con_of_group[g], slab_of)
                  if keep[g] else []
                  for g, grp in enumerate(groups)]
        park_rows = self._park_rows
        self.uses_park = getattr(self, 'uses_park', False) or park_rows > 0
        # Ring tiles only for the waves that stage Jacobian entries: the
        # constraint waves come last, so in every workgroup the ring users
        # are its first waves (tile index == wave index) and a workgroup of
        # [Jacobian wave, constraint wave] -- the fused kernel of a small
        # block -- holds one tile, not two.
        users = [any(e1 > e0 for e0, e1 in grp) for grp in groups]
        rings = 0
        for w0 in range(0, G, W):
            mine = users[w0:w0 + W]
            assert mine == sorted(mine, reverse=True), 'ring users first'
            rings = max(rings, sum(mine))
        if W == 1:
            rings = 1           # one size per kernel: nothing to share
        pub_base = (len(rows) + rings*ring_rows)*TS
        lds_doubles = max(1, pub_base + len(self._pub_rows)*TS +
                          W*park_rows*WAVE)
        occ = ' __attribute__((amdgpu_waves_per_eu(%d, %d)))' % (
            self.o.occupancy, self.o.occupancy) if self.o.occupancy else ''
        src = ['extern "C" __global__ void __launch_bounds__(%d)%s'
               % (64*W, occ),
               '%s(%s)' % (name, KERNEL_PARAMS + (
                   ', const int *__restrict__ sched' if listed else '')), '{',
               '    __shared__ double lds[%d];' % lds_doubles]
        if listed:
            src += [self._LIST_HEAD]
        if inst_lines:
            src += ['    if (blockIdx.x >= npw) {' if listed else
                    '    if (blockIdx.x >= ((node_end - node_begin + 63)/64 + '
                    '7)/8*8*%d) {' % sets,
                    '        if (threadIdx.x == 0) {']
            src += ['            ' + ln for ln in inst_lines]
            src += ['        }', '        return;', '    }']
        rot = 'slot' if (first_group//W) % sets == 0 else \
            '(slot + %d)' % ((first_group//W) % sets)
        rotate = self.o.rotate
        if rotate is None:
            rotate = self.o.cut == 'work' or self._auto_work is not None
        if rotate and sets > 1:
            # Every block starts one strip further: the hardware hands
            # consecutive workgroups of an XCD to consecutive CUs, so with a
            # strip count that divides the CU count (4 strips, 32 CUs) every
            # CU would see the same strip of every block -- a quarter of
            # them the store-only one, the others only long waves (the
            # biped's Jacobian kernel with 4 work-aware strips: 0.104 ms,
            # with 5: 0.066 ms).
            rot = '(%s + slot/%d)' % (rot, sets)
        if order == 'tail' and sets > 1:
            # block by block, then the last ``tail`` blocks of every XCD
            # class by class (sets sorted longest first)
            tail = max(1, RESIDENT_WAVES//(8*max(1, sets - 1)))
            mapping = (
                'const long long nslot = (nblk + 7) >> 3;\n'
                '    const long long ntail = nslot < {tail} ? nslot : {tail};\n'
                '    const long long nhead = nslot - ntail;\n'
                '    const long long stail = slot - nhead*{sets};\n'
                '    const long long blk = (stail < 0 ? slot/{sets} : '
                'nhead + stail % ntail)*8 + xcd;\n'
                '    const int grp = (int)(stail < 0 ? slot % {sets} : '
                'stail/ntail)*{W} + wave;').format(W=W, sets=sets, tail=tail)
        elif order in ('class', 'tail'):
            # strip class by strip class: workgroup set s of every block
            # before set s + 1 of any (the printer sorted the sets longest
            # first); a block's workgroups stay on one XCD
            mapping = ('const long long nslot = (nblk + 7) >> 3;\n'
                       '    const long long blk = (slot % nslot)*8 + xcd;\n'
                       '    const int grp = (int)(slot/nslot)*{W} + wave;'
                       ).format(W=W)
        else:
            mapping = ('const long long blk = (slot/{sets})*8 + xcd;\n'
                       '    const int grp = (int)({rot} % {sets})*{W} + wave;'
                       ).format(sets=sets, W=W, rot=rot)
        if not listed:
            src += [self._PROLOGUE.format(P=self.p.P, mapping=mapping,
                                          slab=len(rows)*TS,
                                          ring=ring_rows*TS)]
        if park_rows and listed:
            src.append('    double *const park_o = lds + %d;'
                       % ((len(rows) + rings*ring_rows)*TS))
        elif park_rows:
            # the wave's parking rows (behind the ring tiles); reloads go
            # through ``lane_z`` == lane, which the compiler cannot prove, so
            # that it neither forwards the parked value from its register
            # (which would keep it alive) nor moves a reload above its store
            src += ['    double *const park = lds + %d + wave*%d;'
                    % ((len(rows) + rings*ring_rows)*TS, park_rows*WAVE),
                    '    const int lane_z = lane + (int)(N >> 62);']
        trace_head = [
            '    const long long tr_w0 = wall_clock64();',
            '    const long long tr_c0 = __builtin_readcyclecounter();'] \
            if self.o.trace else []
        trace_tail = [
            '    if (lane == 0 && jac) {',
            '        long long *tr = reinterpret_cast<long long *>(jac + '
            'ncn*%dLL + %d) + ((long long)%s*%d + wave)*4;'
            % (self.p.P, TRACE_OFFSET,
               '(((long long)(item & 0xffffff)*%d + grp)*8 + xcd)' % sets
               if listed else 'blockIdx.x', W),
            '        tr[0] = tr_w0; tr[1] = wall_clock64();',
            '        tr[2] = ((long long)grp << 40) | blk;',
            '        tr[3] = ((long long)(__builtin_readcyclecounter() - '
            'tr_c0) << 24) | (long long)(__builtin_amdgcn_s_getreg('
            'GETREG_IMMED(3, 0, 20)) << 16) | (long long)'
            '__builtin_amdgcn_s_getreg(GETREG_IMMED(15, 0, 4));',
            '    }'] if self.o.trace else []
        fill = ['    ' + ln for ln in self._slab_fill(rows, slab_of, W)]
        stage_weight = 0
        if pub is not None:
            assert W == G and not listed and not park_rows
            stage, stage_weight = self._publish_stage(W, slab_of)
            fill += ['    double *const pub = lds + %d;' % pub_base] + \
                ['    ' + ln for ln in stage]
        if listed:
            # one loop around the switch: items in the order of the schedule
            src += [self._LIST_LOOP,
                    self._LIST_ITEM.format(P=self.p.P, slab=len(rows)*TS,
                                           ring=ring_rows*TS)]
            if park_rows:
                src += ['    double *const park = park_o;',
                        '    const int lane_z = lane + (int)(N >> 62);',
                        '    (void)park; (void)lane_z;']
            src += trace_head + fill
            src.append('    switch (grp) {')
            for g, lines in enumerate(bodies):
                src.append('    case %d: {' % g)
                src += ['        ' + ln for ln in lines]
                src.append('    } break;')
            src.append('    default: break;')
            src.append('    }')
            src += trace_tail + [self._LIST_NEXT]
        else:
            src += trace_head + fill
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
            src += trace_tail
        src.append('}')
        text = '\n'.join(src)
        # sha of this kernel's own source: profiles/traffic.json keys the PMC
        # counters on it, so that they go stale with the kernel they were
        # collected on and not with a change elsewhere in the module
        return text, dict(name=name, groups=G, waves_per_wg=W,
                          wgs_per_block=sets, lds_bytes=lds_doubles*8,
                          park_rows=park_rows,
                          published_rows=len(self._pub_rows),
                          publish_stage_weight=stage_weight,
                          persist=RESIDENT_WAVES if listed else 0,
                          class_cost=self._given_cost(name, G) or [
                              float(sum(self._weighted_cost(e0, e1) +
                                        STORE_WEIGHT*(e1 - e0)
                                        for e0, e1 in grp if e1 > e0) +
                                    60*len(con_of_group[g]) + 100)
                              for g, grp in enumerate(groups)]
                          if listed else [],
                          sha=hashlib.sha256(text.encode()).hexdigest())

    def _given_cost(self, name, G):
        given = self.o.fused_class_cost if name == 'opty_conjac' \
            else self.o.class_cost
        if not given:
            return None
        cost = [float(c) for c in str(given).split(';')]
        return cost if len(cost) == G else None

    @staticmethod
    def _waves_per_workgroup(slab_rows, ring_rows, lds_per_cu=160*1024):
        """A workgroup holds one input slab and one ring tile per wave; the
        CU's 160 KB of LDS bound how many workgroups are resident.  A
        50-state system (27 KB slab + 25 KB ring) fits three single-wave
        workgroups per CU -- one SIMD idles -- but two two-wave workgroups.
        Returns the smallest width that keeps a wave on each of the CU's four
        SIMDs (kernels this large rarely fit a second wave's registers), or
        else the one with most resident waves."""
        best, best_w = 0, 1
        for W in (1, 2, 3, 4):
            lds = max(1, (slab_rows + W*ring_rows)*TS)*8
            waves = min(4, (lds_per_cu//lds)*W)
            if waves > best:
                best, best_w = waves, W
        return best_w

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

        # The table is filled by up to UNI_WORKGROUPS single-lane workgroups
        # (problems whose table depends on `free` -- variable duration,
        # unknown parameters -- pay for this kernel in every evaluation); each
        # takes a contiguous share of the slots and recomputes what it shares
        # with the others.
        nparts = max(1, min(UNI_WORKGROUPS, len(slots)//48))
        src = ['extern "C" __global__ void __launch_bounds__(64)',
               'opty_uni(%s)' % KERNEL_PARAMS, '{',
               '    if (threadIdx.x != 0) return;',
               '    switch (blockIdx.x) {']
        for b in range(nparts):
            part = slots[b*len(slots)//nparts:(b + 1)*len(slots)//nparts]
            body = _Body(d, set(d.reachable([i for i, _ in part])), leaf,
                         self.o.fast_trig, self.o.deterministic)
            for i, s in part:
                ref = body.emit(i)
                body.lines.append('uni_w[%d] = %s;' % (s, ref))
            body.end_scope()
            src.append('    case %d: {' % b)
            src += ['        ' + ln for ln in body.lines]
            src.append('    } break;')
        src += ['    default: break;', '    }', '}']
        dynamic = any(d.op[i] == ir.INPUT and
                      self._scalar_source(i).startswith('free_')
                      for i in needed)
        return '\n'.join(src), len(slots), dynamic

    def inst_lines(self):
        """Statements storing the instance-constraint values and partials
        (one lane; ``con`` / ``jac`` / ``con_stride`` / the node range of a
        whole-problem launch)."""
        p, d = self.p, self.dag
        roots = list(p.inst_con_out) + list(p.inst_jac_out)
        needed = set(d.reachable(roots))

        def leaf(i):
            if d.op[i] == ir.INPUT:
                return self._scalar_source(i)
            return None

        body = _Body(d, needed, leaf, self.o.fast_trig,
                     self.o.deterministic)
        for k, node in enumerate(p.inst_con_out):
            ref = body.emit(node)
            body.lines.append('if (con) con[%dLL*con_stride + %d] = %s;'
                              % (p.M, k, ref))
        for k, node in enumerate(p.inst_jac_out):
            ref = body.emit(node)
            body.lines.append('if (jac) jac[(node_end - node_begin)*%dLL + '
                              '%d] = %s;' % (p.P, k, ref))
        body.end_scope()
        return list(body.lines)

    def inst_kernel(self):
        src = ['extern "C" __global__ void __launch_bounds__(64)',
               'opty_inst(%s)' % KERNEL_PARAMS, '{',
               '    if (threadIdx.x != 0 || blockIdx.x != 0) return;']
        src += ['    ' + ln for ln in self.inst_lines()] + ['}']
        return '\n'.join(src), dict(name='opty_inst', groups=1, lds_bytes=0)


#: constraint rows move into a Jacobian wave of the fused kernel when the
#: block's weighted operations per byte reach this (x 17: one weighted
#: operation is ~1.75 instructions of 4 cycles on one of 1024 SIMDs, a byte
#: 1/6 ps of the chip's store stream -- at 1.0 the two sides are even; the
#: measured kernels run one wave per SIMD and issue at half that rate), and a
#: row goes to the strip that shares at least this much of its work
ATTACH_MIN_INTENSITY = 0.5
ATTACH_MIN_SHARE = 0.5
ATTACH_LEFTOVER_WEIGHT = 64


def _attach_constraint_rows(prog, w, opts, fused_jac, con_sets):
    """``(rows per Jacobian wave, remaining constraint waves)`` for the fused
    kernel.

    The constraint rows and the partials of the same equations share most of
    their sub-expressions (the muscle-driven leg: all constraint rows alone
    2370 weighted operations, on top of the Jacobian's 4460 only 212).  A
    store-bound block hides a separate constraint wave behind its stores (and
    was measured slower with the rows attached: 10-link 0.143 -> 0.153 ms,
    DESIGN.md 4.4); a block bound by its arithmetic pays for every operation
    twice.  So for blocks above ``ATTACH_MIN_INTENSITY`` each row moves to
    the strip that computes at least ``ATTACH_MIN_SHARE`` of it already
    (greedy, most expensive row first); rows without such a strip keep their
    constraint waves."""
    none = [[] for _ in fused_jac]
    if opts.con_attach == 0 or opts.con_rows_per_wave or not fused_jac or \
            opts.ablate is not None:
        return none, con_sets
    d = prog.dag

    def weight(roots):
        return sum(_node_weight(d, i) for i in d.reachable(set(roots)))

    rows = [j for rs in con_sets for j in rs]
    if opts.con_attach is None:
        work = weight(prog.jac_out) + weight(prog.con_out)
        nbytes = 8.0*64*(prog.P + prog.M + len(prog.rows))
        if 17.0*work/nbytes < ATTACH_MIN_INTENSITY:
            return none, con_sets
    strip_roots = []
    for grp in fused_jac:
        roots = set()
        for e0, e1 in grp:
            if e1 > e0:
                roots |= {prog.jac_out[v % prog.P]
                          for v in range(e0, w._virtual_end(e1))}
        strip_roots.append(roots)
    base = [weight(r) for r in strip_roots]
    alone = {j: weight([prog.con_out[j]]) for j in rows}
    attached = [[] for _ in fused_jac]
    left = []
    for j in sorted(rows, key=lambda j: -alone[j]):
        if alone[j] == 0:
            # a row of constants / inputs: cheapest strip
            g = min(range(len(base)), key=lambda g: base[g])
            attached[g].append(j)
            continue
        extra = [weight(strip_roots[g] | {prog.con_out[j]}) - base[g]
                 for g in range(len(fused_jac))]
        g = min(range(len(extra)), key=lambda g: (extra[g], base[g]))
        if extra[g] <= (1.0 - ATTACH_MIN_SHARE)*alone[j]:
            attached[g].append(j)
            strip_roots[g].add(prog.con_out[j])
            base[g] += extra[g]
        else:
            left.append(j)
    if not any(attached):
        return none, con_sets
    if left and sum(alone[j] for j in left) <= ATTACH_LEFTOVER_WEIGHT:
        # what is left would be a wave of its own for next to nothing (the
        # kinematic rows q' - u of a multibody system): cheapest strip
        g = min(range(len(base)), key=lambda g: base[g])
        attached[g] += left
        left = []
    keep = set(left)
    remaining = [[j for j in rs if j in keep] for rs in con_sets]
    return [sorted(a) for a in attached], [rs for rs in remaining if rs]


def _fit_one_round(auto_groups, con_waves, node_blocks, live_groups=None):
    """Strip count for a launch of ``node_blocks`` 64-node blocks
    (``auto_groups`` for large launches; never fewer than 60 % of
    ``live_groups``, what the registers allow).

    A launch whose waves are all resident at once finishes in one round; one
    that needs 1.3 rounds takes almost as long as two.  Measured on MI355X
    (10-link pendulum, 12 500 nodes = 196 blocks -- one of 8 node shards of
    BASELINE config 4, profiles/r02_strip_sweeps.txt): 6 strips + 1
    constraint wave per block = 1372 waves, 0.0253 ms; 4 strips + 1 = 980
    waves <= 1024, 0.0233 ms.  Coarser strips cost registers (4 strips: 298
    VGPRs, still no scratch), so the count only goes down to 60 % of what the
    register-pressure rule picked, and only when that makes the launch fit."""
    if not node_blocks or node_blocks*(auto_groups + con_waves) \
            <= RESIDENT_WAVES:
        return auto_groups
    live = auto_groups if live_groups is None else live_groups
    for g in range(auto_groups - 1, max(2, -(-3*live//5)) - 1, -1):
        if node_blocks*(g + con_waves) <= RESIDENT_WAVES:
            return g
    return auto_groups


def _dual_occupancy_cut(prog, writer, opts, live_groups, con_waves,
                        node_blocks):
    """Launch geometry for SMALL launches: two waves per SIMD.

    A node shard of BASELINE config 4 (12 500 nodes = 196 blocks) is too
    short to amortise a wave's own critical path (slab fill, then its chunks
    one after the other): with one wave per SIMD either every wave is long (4
    strips: one round of 980 waves, 0.0221 ms on some boxes, 0.0260 ms on
    others) or the launch takes two rounds (9 strips: 0.0239 ms).  With
    16-entry chunks (16.6 KB ring tiles), four waves per workgroup sharing
    one slab and the kernels capped at 256 VGPRs (amdgpu_waves_per_eu(2)) a CU
    holds 8 waves, the chip 2048: 7 strips + 1 constraint wave = 2 full
    workgroups per block, 1568 waves in ONE round of short waves --
    0.0200-0.0202 ms where 4 strips took 0.0221, 0.0222 ms where they took
    0.0260 (profiles/r02_strip_sweeps.txt).  It only pays when it makes the
    launch fit: at 25 000 nodes it ties with the default, at 50 000+ it loses
    (0.081 vs 0.072 ms), as it does at full size (r01: no gain).

    Returns ``(options, strips)`` or None when it does not apply: the printer
    options were set by hand, the launch fits one round as it is or does not
    fit 2048 waves, the slab + four ring tiles exceed half a CU's LDS, or the
    strips would be fewer than the register minimum."""
    import copy
    if (opts.chunk != 32 or opts.waves is not None or opts.occupancy or
            not writer.line_mode() or not node_blocks):
        return None
    total = 4*((2*RESIDENT_WAVES//int(node_blocks))//4)   # waves per block
    strips = total - con_waves
    if strips < max(2, live_groups):
        return None
    strips = min(strips, writer.auto_groups()[1])
    # whole workgroups only: a partly filled one still takes a slot of four
    strips -= (strips + con_waves) % 4
    if strips < max(2, live_groups):
        return None
    lds = (len(prog.rows) + 4*(16 + 16))*TS*8
    if 2*lds > 160*1024:
        return None
    dual = copy.copy(opts)
    dual.chunk, dual.waves, dual.occupancy = 16, 4, 2
    return dual, strips


#: rough per-lane instruction weights of DAG operations (double precision on
#: gfx950; opty_sincos is ~35 for the pair), for balancing work between waves
_OP_WEIGHT = {'sin': 18, 'cos': 18, 'tan': 60, 'exp': 30, 'log': 35,
              'sqrt': 12, ir.DIV: 12, ir.POW: 90, ir.ATAN2: 80, 'asin': 60,
              'acos': 60, 'atan': 50, 'sinh': 60, 'cosh': 60, 'tanh': 60,
              'erf': 60, 'erfc': 60, 'asinh': 70, 'acosh': 70, 'atanh': 70,
              'log1p': 40, 'expm1': 40, 'log2': 35, 'log10': 35, 'exp2': 30,
              'cbrt': 40, 'tgamma': 120, 'lgamma': 120}


def _node_weight(dag, i):
    if dag.op[i] in (ir.CONST, ir.INPUT) or dag.uni[i]:
        return 0
    return _OP_WEIGHT.get(dag.op[i], 1)


def _constraint_waves(prog, w, opts, scale=1.0):
    """Splits the M constraint rows over as few waves as the register budget
    allows, balanced by WORK, not by count.

    Every constraint wave re-reads the slab and recomputes the sub-expressions
    its rows share with the others (all the sin / cos of a multibody system),
    so one wave for all rows is fastest when it fits (10-link pendulum).  These
    waves run one per SIMD whatever they need, so their budget is the whole
    register file: 1.5 x ``max_live`` estimated temporaries is where spilling
    starts (24-link, 50 rows: 10 / 5 / 4 / 2 waves of equal row COUNT have
    estimates 123 / 174 / 227 / 532 and take 0.102 / 0.071 / 0.101 / 0.290 ms,
    profiles/r02_strip_sweeps.txt).  Equal counts put the 25 trivial kinematic
    rows ``q' - u`` of such a system into 2.5 waves and the 25 dynamic rows
    into the other 2.5; instead the rows are cut into k contiguous ranges that
    minimise the most expensive wave (operation weights, shared nodes counted
    once per wave), for the smallest k whose ranges all fit the budget."""
    dag, M = prog.dag, prog.M
    if M <= 1:
        return [list(range(M))]
    leaf = lambda i: w._is_vec_input(i) or w._uniform_leaf(i)
    # cost[a][b]: work of one wave evaluating rows a..b-1
    cost = [[0]*(M + 1) for _ in range(M)]
    for a in range(M):
        seen, total = set(), 0
        for b in range(a, M):
            stack = [prog.con_out[b]]
            while stack:
                i = stack.pop()
                if i in seen:
                    continue
                seen.add(i)
                if dag.op[i] in (ir.CONST, ir.INPUT) or dag.uni[i]:
                    continue
                total += _node_weight(dag, i)
                stack.extend(dag.operands(i))
            cost[a][b + 1] = total + 2*(b + 1 - a)       # + the row's store
    # (``scale``: 0.5 when the kernels are capped at 256 VGPRs, two waves
    # per SIMD)
    budget = 1.5*opts.max_live*scale
    best = None
    for k in range(1, M + 1):
        # linear partition: minimise the largest range cost
        INF = float('inf')
        dp = [[INF]*(M + 1) for _ in range(k + 1)]
        cut = [[0]*(M + 1) for _ in range(k + 1)]
        dp[0][0] = 0
        for parts in range(1, k + 1):
            for b in range(parts, M + 1):
                for a in range(parts - 1, b):
                    c = max(dp[parts - 1][a], cost[a][b])
                    if c < dp[parts][b]:
                        dp[parts][b], cut[parts][b] = c, a
        bounds, b = [M], M
        for parts in range(k, 0, -1):
            b = cut[parts][b]
            bounds.append(b)
        bounds.reverse()
        sets = [list(range(bounds[t], bounds[t + 1])) for t in range(k)]
        best = sets
        worst = max(_max_live(dag, [[prog.con_out[j]] for j in rs], leaf)
                    for rs in sets)
        if worst <= budget:
            break
    return best


def emit_matrix_module(prog, opts=None):
    """Module of a *matrix program* (``program.matrix_program``: a plain
    ``(rows x cols)`` matrix of expressions evaluated for ``n`` independent
    argument rows -- the call shape of the reference's ``ufuncify_matrix``,
    ``opty/utils.py:639-640``): only ``opty_jac`` (the matrix entries, staged
    and flushed like a Jacobian block) and ``opty_uni``."""
    opts = opts or EmitOptions()
    w = _ModuleWriter(prog, opts)
    groups = w.group_ranges()
    src, kmeta = w.kernel('opty_jac', groups, [[] for _ in groups],
                          opts.waves)
    usrc, num_uniform, dynamic = w.uniform_kernel()
    assert not dynamic
    head = ['// generated by opty_amd.codegen.emit_hip (matrix program) -- '
            'do not edit', '// %s' % opts.key(),
            '#define OPTY_STORE_AUX %d' % opts.store_aux,
            '#include "opty_device.h"', '']
    source = '\n'.join(head + [src, '', usrc, ''])
    meta = dict(kernels={'jac': kmeta}, chunk=opts.chunk, P=prog.P,
                num_uniform=num_uniform, uniform_dynamic=False,
                sha=hashlib.sha256(source.encode()).hexdigest())
    return source, meta


def emit_module(prog, opts=None, node_blocks=None, literals=None):
    """Returns ``(source, meta)``; ``meta`` describes the launch geometry the
    runtime needs (waves per node block, size of the ``uni`` table, whether
    the table depends on ``free``).  ``node_blocks``: 64-node blocks of the
    launches this module is built for (picks the strip count of small
    launches, see ``_fit_one_round``); None = large launches."""
    opts = opts or EmitOptions()
    # Small problems (see ``fold`` below) whose node-invariant table would be
    # refilled by opty_uni before every evaluation do without the table.
    inline = opts.inline_uniform
    if inline is None:
        rows = len(prog.rows)
        inline = bool(node_blocks) and \
            8*64*int(node_blocks)*(prog.P + prog.M + rows) \
            <= LATENCY_PATH_BYTES and \
            (prog.h[0] != 'fixed' or
             any(src != 'known' for src, _ in prog.pars)) and \
            sum(1 for i in range(len(prog.dag.op)) if prog.dag.uni[i] and
                prog.dag.op[i] not in (ir.CONST, ir.INPUT)) \
            <= INLINE_UNIFORM_MAX_NODES
    w = _ModuleWriter(prog, opts, inline, literals)
    groups = w.group_ranges()
    # Constraint rows may be split over several waves (contiguous row ranges):
    # one wave evaluating all M defects of a big system runs out of registers.
    def row_sets(rpw):
        return [list(range(a, min(a + rpw, prog.M)))
                for a in range(0, prog.M, rpw)]

    def by_count():
        # equal row counts, as few waves as the register estimate allows
        leaf = lambda i: w._is_vec_input(i) or w._uniform_leaf(i)
        parts = 1
        while True:
            sets = row_sets(-(-prog.M//parts))
            worst = max(_max_live(prog.dag, [[prog.con_out[j]] for j in rs],
                                  leaf) for rs in sets)
            if worst <= 1.5*opts.max_live or len(sets) >= prog.M:
                return sets
            parts += 1

    # opty_con's waves are balanced by work; the fused kernel's constraint
    # waves as well unless that made it spill (``con_split='count'``: one
    # register allocation serves all waves of a kernel, and next to Jacobian
    # strips at the 512-VGPR limit the balanced cut of a 24-link system --
    # 33 + 9 + 8 rows -- tipped it into scratch where 5 x 10 rows did not)
    if opts.con_rows_per_wave:
        alone_sets = con_sets = row_sets(max(1, int(opts.con_rows_per_wave)))
    else:
        alone_sets = _constraint_waves(
            prog, w, opts, 0.5 if opts.occupancy == 2 else 1.0)
        con_sets = alone_sets if opts.con_split == 'work' else by_count()
    fused_jac = groups
    if opts.groups is None:
        live, auto = w.auto_groups()
        if w._auto_work is not None and opts.fused_groups is None:
            # the automatic work-aware cut (blocks that would wait for
            # recomputed arithmetic) serves both Jacobian kernels as it is
            fused_jac = groups
        else:
            # the fused kernel carries the constraint waves as well
            fused = max(live, min(auto, int(round(
                FUSED_STRIPS_PER_SQRT_ENTRY*prog.P**0.5)))) if w.line_mode() \
                else auto
            dual = None
            if node_blocks and int(node_blocks)*(fused + len(con_sets)) > \
                    RESIDENT_WAVES and not opts.con_rows_per_wave:
                # at two waves per SIMD every wave has half the registers:
                # the constraint rows are cut for that budget (one wave for
                # the 22 rows of the 10-link system needs 258 VGPRs, two
                # spills under the 256 cap)
                dual_sets = _constraint_waves(prog, w, opts, 0.5)
                dual = _dual_occupancy_cut(prog, w, opts, live,
                                           len(dual_sets), int(node_blocks))
            if dual is not None:
                opts, fused = dual
                w = _ModuleWriter(prog, opts, inline, literals)
                groups = w.group_ranges(fused)
                alone_sets = con_sets = dual_sets
            elif node_blocks:
                fit = _fit_one_round(auto, len(con_sets), int(node_blocks),
                                     live)
                if fit != auto:
                    groups = w.group_ranges(fit)
                fused = _fit_one_round(fused, len(con_sets), int(node_blocks),
                                       live)
            if opts.fused_groups is not None:
                fused = opts.fused_groups
            if fused != len(groups):
                fused_jac = w.group_ranges(fused)
            else:
                fused_jac = groups
    elif opts.fused_groups is not None:
        fused_jac = w.group_ranges(opts.fused_groups)
    if opts.fused_strips and w.line_mode():
        fused_jac = w.explicit_strips(opts.fused_strips)
    seeds = dict(jac=len(groups), fused=len(fused_jac),
                 con_waves=len(alone_sets), chunk=opts.chunk,
                 waves=opts.waves, occupancy=opts.occupancy,
                 line_mode=bool(w.line_mode()),
                 cut='work' if (w._auto_work is not None or
                                opts.cut == 'work') else 'even',
                 live=w.auto_groups()[0] if opts.groups is None else None)
    if opts.dear_first:
        def work(grp):
            return sum(w._strip_cost(e0, e1) for e0, e1 in grp if e1 > e0)
        groups = sorted(groups, key=work, reverse=True)
        fused_jac = sorted(fused_jac, key=work, reverse=True)
    else:
        # longest strips first (what the dispatch orders 'class' / 'tail'
        # hand out first)
        def work(grp):
            return sum(w._weighted_cost(e0, e1) + STORE_WEIGHT*(e1 - e0)
                       for e0, e1 in grp if e1 > e0)
        if opts.order in ('class', 'tail', 'list'):
            groups = sorted(groups, key=work, reverse=True)
        if (opts.fused_order or opts.order) in ('class', 'tail', 'list'):
            fused_jac = sorted(fused_jac, key=work, reverse=True)
    # The fused kernel is the Jacobian kernel plus the constraint waves (empty
    # entry ranges): the Jacobian waves keep their register budget, the extra
    # waves ride in the shadow of the store-bound Jacobian waves.  Where the
    # ARITHMETIC sets the pace instead, a constraint row is evaluated by the
    # Jacobian wave that computes most of its sub-expressions anyway.
    attached, con_sets = _attach_constraint_rows(prog, w, opts, fused_jac,
                                                 con_sets)
    con_groups = [[(0, 0)]]*len(con_sets)
    fused_groups = list(fused_jac) + con_groups
    con_of = attached + con_sets
    parts = []
    kernels = {}
    # Constraint stores.  Written once, never re-read by the kernels: on their
    # own (opty_con) they are fastest streamed past the caches (10-link,
    # N = 10^5: 0.0118 vs 0.0137 ms).  In the fused kernel a constraint vector
    # that fits the caches is better left to them (N = 10^5, 17.6 MB: 0.1359
    # vs 0.1465 ms), one that does not evicts into the Jacobian stream
    # (N = 10^6, 176 MB: 1.598 ms plain, 1.366 ms non-temporal) --
    # profiles/r02_strip_sweeps.txt.
    con_bytes = 8*prog.M*64*int(node_blocks or 0)
    nt_alone = opts.con_nt != 0
    nt_fused = opts.con_nt == 1 or (opts.con_nt is None and
                                    con_bytes > CON_CACHE_BYTES)
    # The instance tails ride in the main kernels' launch: one more workgroup
    # instead of one more kernel.  Small problems (BASELINE config 2) cost
    # what their launches cost; and next to a 50-70 us evaluation of a
    # gait-like problem the launch of opty_inst is still 5-10 % (biped:
    # 0.0767 -> 0.068 ms per evaluation with its 16 periodicity constraints
    # folded).  (Node shards evaluate the tails through opty_inst, once, on
    # the rank that assembles a vector.)
    fold = opts.fold_instance
    if fold is None:
        fold = bool(node_blocks and prog.inst_con_out)
    folded = w.inst_lines() if (fold and prog.inst_con_out) else None
    for key, name, grp, cons, wpw, nt in (
            ('con', 'opty_con', [[(0, 0)]]*len(alone_sets), alone_sets, 1,
             nt_alone),
            ('jac', 'opty_jac', list(groups) + [[(0, 0)]]*opts.pad,
             [[] for _ in range(len(groups) + opts.pad)], opts.waves, False),
            ('conjac', 'opty_conjac', fused_groups, con_of, opts.waves,
             nt_fused)):
        src, meta = w.kernel(
            name, grp, cons, wpw, nt, inst_lines=folded,
            first_group=len(fused_jac) if (opts.dear_first and
                                           key == 'conjac') else 0,
            order=opts.fused_order if key == 'conjac' else None)
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
            '#define OPTY_STORE_AUX %d' % opts.store_aux,
            '#include "opty_device.h"', '']
    if getattr(w, 'uses_park', False):
        head += [_PARK_HELPERS, '']
    if any(k.get('persist') for k in kernels.values()):
        head += [_LOOP_HELPERS, '']
    if opts.fast_trig == 2:
        head += [_UNIFORM_TRIG_HELPERS, '']
    source = '\n'.join(head + parts)
    meta = dict(kernels=kernels,
                groups=[[list(rg) for rg in grp] for grp in groups],
                fused_groups=[[list(rg) for rg in grp] for grp in fused_jac],
                chunk=opts.chunk, P=prog.P, M=prog.M, C=prog.C,
                geometry=seeds, layout=getattr(prog, 'layout', 'coo'),
                num_uniform=num_uniform, uniform_dynamic=bool(dynamic),
                inst_folded=bool(folded),
                con_attached=bool(any(attached)),
                sha=hashlib.sha256(source.encode()).hexdigest())
    if w._plans:
        meta['plans'] = w._plans
    if literals:
        meta['literals'] = len(literals)
    return source, meta
