"""Measured launch plans: the strip counts of a problem's kernels chosen by
timing them on the device instead of by the printer's rules of thumb.

The printer (``codegen/emit_hip.py``) picks the number of waves that share a
64-node block from register estimates and two constants fitted to one family
of systems (``STRIP_ENTRIES``, ``FUSED_STRIPS_PER_SQRT_ENTRY``: n-link
pendulums on the boxes round 2 happened to lease).  Those stay as SEEDS.
:func:`tune` builds the neighbouring geometries of a problem (in parallel:
``hipcc`` is a subprocess), times them interleaved on the GPU through
``opty_hip_time_eval_shard`` at the launch size the handle will be used at, and
records the winners; :func:`lookup` is consulted by
``ConstraintCollocator`` whenever the caller did not pass printer options.

Plan file (JSON, ``opty_amd/launch_plans.json``, tracked; ``OPTY_LAUNCH_PLANS``
overrides the path, ``OPTY_LAUNCH_PLANS=off`` disables lookups)::

    {"<problem sha>:<launch bucket>:<arch>": {
        "options": {"groups": 10, "fused_groups": 9, ...},   # EmitOptions kwargs
        "seed": {"jac": 10, "fused": 9},                      # what the rules said
        "measured_ms": {"fused": {"8": 0.1374, "9": 0.1359, ...},
                        "jac": {...}},
        "nodes": 99999, "device": "AMD Instinct MI355X", "problem": "..."}}

* problem sha: structural sha-256 of the collocation program (the expression
  DAG of every output, input homes, discretisation, layout: ``problem_sha``)
  -- it changes with the equations, the discretisation and the lowering, so a
  stale plan is never applied; ``PLAN_EPOCH`` is bumped when a printer change
  alters what recorded options MEAN;
* launch bucket: ``round(log2(64-node blocks of one launch))``, capped at 12
  (launches of 4096 blocks and more fill the chip many times over and share
  one optimum; a 196-block node shard does not);
* arch: the code-object target (``gfx950``).
"""

import json
import math
import os

from . import hip_backend as hb
from .codegen.emit_hip import EmitOptions, emit_module

_PKG = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FILE = os.path.join(_PKG, 'launch_plans.json')
_cache = {}


def plan_path():
    path = os.environ.get('OPTY_LAUNCH_PLANS', DEFAULT_FILE)
    return None if path == 'off' else path


def _load(path):
    if path is None:
        return {}
    try:
        mtime = os.path.getmtime(path)
    except OSError:
        return {}
    hit = _cache.get(path)
    if hit is None or hit[0] != mtime:
        try:
            with open(path) as f:
                _cache[path] = (mtime, json.load(f))
        except (OSError, ValueError):
            _cache[path] = (mtime, {})
    return _cache[path][1]


def bucket(node_blocks):
    """Launch-size class of a launch of ``node_blocks`` 64-node blocks."""
    return min(12, int(round(math.log2(max(1, int(node_blocks))))))


#: bumped when the MEANING of recorded options changes (a plan measured for
#: one printer's "groups=5" must not be applied by a printer that cuts five
#: strips elsewhere); printer changes that keep the meaning keep the plans
PLAN_EPOCH = 1


def problem_sha(prog):
    """Identity of a collocation program for the plan file: a structural
    hash of what the printer is given -- the expression DAG behind every
    output, the trajectory rows / parameters / interval homes, the
    discretisation offsets and the output layout.  It changes with the
    equations, the discretisation and the lowering, NOT with the printer (up
    to r04 the key was the sha of the emitted module: every printer change
    orphaned every measured plan, and computing it meant printing the module,
    seconds for a biped).  Memoised on the program."""
    hit = getattr(prog, '_plan_sha', None)
    if hit is not None:
        return hit
    import hashlib
    d = prog.dag
    memo = {}

    def node(root):
        # iterative post-order: digest of (op, operand digests / literals)
        stack = [root]
        while stack:
            i = stack[-1]
            if i in memo:
                stack.pop()
                continue
            todo = [j for j in d.operands(i) if j not in memo]
            if todo:
                stack.extend(todo)
                continue
            stack.pop()
            op, args = d.op[i], d.args[i]
            if op in ('const', 'in'):
                parts = [op] + [repr(a) for a in args]
            elif op == 'powi':
                parts = [op, memo[args[0]], repr(args[1])]
            elif op == 'select':
                parts = [op, repr(args[0])] + [memo[a] for a in args[1:]]
            else:
                parts = [str(op)] + [memo[a] for a in args]
            memo[i] = hashlib.sha256('|'.join(parts).encode()).hexdigest()[:32]
        return memo[root]

    h = hashlib.sha256()
    h.update(('epoch=%d' % PLAN_EPOCH).encode())
    for tag in ('con_out', 'jac_out', 'inst_con_out', 'inst_jac_out'):
        h.update(('\0%s:' % tag).encode())
        for r in getattr(prog, tag, ()) or ():
            h.update(node(r).encode())
    for tag in ('rows', 'pars', 'h', 'cur_offset', 'adj_offset', 'M', 'C',
                'n', 'q', 'layout', 'row_start', 'pruned'):
        h.update(('\0%s=%r' % (tag, getattr(prog, tag, None))).encode())
    sha = 'p' + h.hexdigest()[:19]
    try:
        prog._plan_sha = sha
    except AttributeError:
        pass
    return sha


def emitted_sha(prog):
    """The r01-r04 key: sha of the module the printer emits with default
    options (only ``tools/migrate_plan_keys.py`` still needs it)."""
    _, meta = emit_module(prog, EmitOptions(), node_blocks=None)
    return meta['sha'][:20]


def key_of(prog, node_blocks, sha=None):
    return '%s:%d:%s' % (sha or problem_sha(prog), bucket(node_blocks),
                         hb.ARCH)


def lookup(prog, node_blocks):
    """``EmitOptions`` of the measured plan for this program and launch size,
    or None."""
    plans = _load(plan_path())
    if not plans:
        return None
    entry = plans.get(key_of(prog, node_blocks))
    if not entry:
        return None
    try:
        return EmitOptions(**entry['options'])
    except (TypeError, AssertionError, KeyError):
        return None                     # written by another printer version


def lookup_entry(prog, node_blocks):
    """The raw plan entry for this program and launch size, or None."""
    plans = _load(plan_path())
    return plans.get(key_of(prog, node_blocks)) if plans else None


def options_kwargs(opts):
    """The keyword arguments that rebuild ``opts``: its attributes that
    differ from the printer's defaults."""
    default = vars(EmitOptions())
    return {k: v for k, v in vars(opts).items() if default.get(k) != v}


def record(key, entry, path=None):
    """Merges one entry into the plan file: read-modify-write under an
    exclusive ``flock`` of ``<path>.lock`` (the ranks of a sharded problem
    may all replace a refused build at the same moment), atomic replace.  The
    provenance of a build that replaced one the verification refused
    (``pinned`` / ``refused``) survives a later :func:`tune` of the same key
    unless the new entry brings its own."""
    path = path or plan_path() or DEFAULT_FILE
    import fcntl
    with open(path + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            _cache.pop(path, None)
            plans = dict(_load(path))
            old = plans.get(key) or {}
            entry = dict(entry)
            if 'pinned' in old and 'pinned' not in entry and \
                    old.get('options') == entry.get('options'):
                for tag in ('pinned', 'refused'):
                    if tag in old:
                        entry.setdefault(tag, old[tag])
            plans[key] = entry
            tmp = '%s.%d.tmp' % (path, os.getpid())
            with open(tmp, 'w') as f:
                json.dump(plans, f, indent=1, sort_keys=True)
            os.replace(tmp, path)
            _cache.pop(path, None)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _neighbours(seed, low, high):
    out = {seed, seed - 1, seed + 1, seed + 2}
    out |= {int(round(seed*f)) for f in (0.75, 0.88, 1.12, 1.25, 1.5)}
    return sorted(g for g in out if low <= g <= high)


#: LDS of one workgroup up to which a CU holds four of them (the generated
#: kernels run one wave per SIMD at their register footprint)
LDS_FOUR_PER_CU = 40*1024


def _merged_cuts(prog, seed, g):
    """Cuts of an arithmetic-bound block with FEWER long strips than the
    register file allows without help: the work-aware cut under a relaxed
    bound on a strip's live values, the waves that exceed the registers
    planned with LDS parking (``EmitOptions.park``).  ``[(tag, fused_strips
    spec)]``: every strip a wave of its own, and the short strips merged into
    one wave."""
    from .codegen.emit_hip import _ModuleWriter
    out, seen = [], set()
    nunits = prog.P//16
    if not 2 < nunits <= 128:
        return out
    chunk = seed.get('chunk', 32)
    w = _ModuleWriter(prog, EmitOptions(cut='work', work_live=400,
                                        chunk=chunk))
    for S in range(max(2, g['fused'] - 2), g['fused'] + 1):
        strips = w._work_cut(S, 16, nunits)
        if not strips:
            continue
        cost = [w._weighted_cost(e0, e1) for e0, e1 in strips]
        long_ = [k for k in range(len(strips))
                 if cost[k] >= 0.25*max(cost)]
        short = [k for k in range(len(strips)) if k not in long_]
        spec = ';'.join('%d:%d' % strips[k] for k in range(len(strips)))
        if spec not in seen:
            seen.add(spec)
            out.append(('merge%d' % S, spec))
        if len(short) > 1:
            # (the strip that ends the block wraps into entries 0..14: last)
            runs = []
            for k in sorted(short):
                if runs and runs[-1][1] == strips[k][0]:
                    runs[-1] = (runs[-1][0], strips[k][1])
                else:
                    runs.append(strips[k])
            runs.sort(key=lambda rg: (rg[1] != prog.P, rg[0]))
            spec = ';'.join(['%d:%d' % strips[k] for k in long_] +
                            ['+'.join('%d:%d' % rg for rg in runs)])
            if spec not in seen:
                seen.add(spec)
                out.append(('merge%d+' % S, spec))
    return out


def candidates(prog, node_blocks):
    """Printer-option variants worth timing around the seed geometry:
    ``[(label, options kwargs)]``, the seed first.  Labels ``fused...`` /
    ``jac...`` only compete for that kernel."""
    _, meta = emit_module(prog, EmitOptions(), node_blocks=node_blocks)
    g = meta['geometry']
    seed = dict(groups=g['jac'], fused_groups=g['fused'])
    if g.get('cut') == 'work':
        # the printer chose the work-aware cut (emit_hip._automatic_work_cut):
        # its neighbours are work-aware cuts with other strip counts
        seed['cut'] = 'work'
    if g['chunk'] != 32 or g['occupancy']:
        # the small-launch geometry (two waves per SIMD) came with its own
        # chunk / workgroup width: keep them with the seed
        seed.update(chunk=g['chunk'], waves=g['waves'],
                    occupancy=g['occupancy'])
    out = [('seed', dict(seed))]
    if not g['line_mode']:
        if prog.P <= 64:
            out.append(('chunk', dict(small_flush='chunk')))
        return out, g
    unit = 16
    top = max(1, min(32, prog.P//unit))
    low = max(1, -(-3*(g['live'] or 1)//5))
    fused_counts = [f for f in _neighbours(g['fused'], low, top)
                    if f != g['fused']]
    for f in fused_counts:
        out.append(('fused=%d' % f, dict(seed, fused_groups=f)))
    # the Jacobian-only kernel competes with the fused kernel's counts too
    # (r04: the 24-link opty_jac with its own 32 strips was slower than the
    # fused kernel that does strictly more with 20)
    jac_counts = _neighbours(g['jac'], low, top) + [g['fused']] + fused_counts
    for j in sorted(set(jac_counts)):
        if j != g['jac']:
            out.append(('jac=%d' % j, dict(seed, groups=j)))
    # one empty wave more per block: r02 measured the Jacobian kernel of the
    # 10-link system faster with it (dispatch order, not arithmetic)
    for j in sorted({g['jac'], g['fused']}):
        out.append(('jac:pad%d' % j, dict(seed, groups=j, pad=1)))
    if g['chunk'] == 32 and not g['occupancy']:
        # shorter ring tiles (16-entry chunks) leave LDS for wider workgroups
        # that share one input slab: fewer slab fills per block, which large
        # slabs (50-state systems) and short launches (node shards) may prefer
        out.append(('c16', dict(seed, chunk=16)))
        out.append(('c16w4', dict(seed, chunk=16, waves=4)))
    if g.get('cut') == 'work' or meta.get('con_attached'):
        # blocks bound by their arithmetic (unequal waves, one per SIMD): the
        # dispatch order of a launch's workgroups, and cuts with fewer long
        # strips whose waves park values in LDS
        shapes = [('', {})]
        if g['chunk'] == 32 and not g['occupancy']:
            shapes.append(('c16:', dict(chunk=16)))     # 17 KB less ring
        for pre, shape in shapes:
            base = dict(seed, **shape)
            for order in ('class', 'tail', 'list'):
                out.append(('jac:%s%s' % (pre, order),
                            dict(base, order=order)))
                out.append(('fused:%s%s' % (pre, order),
                            dict(base, fused_order=order)))
            for tag, spec in _merged_cuts(prog, base, g):
                for live in (235, 225):
                    kw = dict(base, fused_strips=spec, park=48,
                              park_live=live)
                    _, m = emit_module(prog, EmitOptions(**kw),
                                       node_blocks=node_blocks)
                    if m['kernels']['conjac']['lds_bytes'] > \
                            LDS_FOUR_PER_CU or not m.get('plans'):
                        continue    # would cost resident waves / no parking
                    out.append(('fused:%s%s/%d' % (pre, tag, live), kw))
                    out.append(('fused:%s%s/%d:class' % (pre, tag, live),
                                dict(kw, fused_order='class')))
                    out.append(('fused:%s%s/%d:list' % (pre, tag, live),
                                dict(kw, fused_order='list')))
                    break
    return out, g


def tune(collocator, iters=60, rounds=5, save=True, path=None, log=None,
         max_workers=6):
    """Times the candidate geometries of ``collocator``'s problem at its
    launch size on its device and records the winners.  Needs ``torch`` (for
    the device buffers) and a GPU.  Returns the plan entry."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    col = collocator
    prog = col._build_program()
    nodes = int(col._launch_nodes or col.num_collocation_nodes - 1)
    blocks = (nodes + 63)//64
    cands, geo = candidates(prog, blocks)
    dev = torch.device('cuda', col._device)
    built = []

    def build(item):
        # a candidate whose kernels spill vector registers is re-cut once
        # (constraint rows by count); one that still spills does not compete
        # for the kernel that spills (hip_backend.vgpr_spills)
        label, kw = item
        for attempt in (kw, dict(kw, con_split='count')):
            source, meta = emit_module(prog, EmitOptions(**attempt),
                                       node_blocks=blocks)
            hsaco = col._compile(source)
            spills = hb.vgpr_spills(hsaco)
            if 'opty_conjac' not in spills:
                break
        return label, attempt, meta, hsaco, spills

    with ThreadPoolExecutor(max_workers) as pool:
        built = list(pool.map(build, cands))
    f64 = dict(dtype=torch.float64, device=dev)
    N = col.num_collocation_nodes
    rng = np.random.default_rng(1)
    free = rng.uniform(-1.0, 1.0, col.num_free)
    if col._variable_duration:
        free[-1] = 0.01
    free = torch.from_numpy(free).to(dev)
    a = max(0, (N - 1 - nodes)//2)
    b = a + nodes
    con = torch.empty((prog.M, nodes), **f64)
    jac = torch.empty(nodes*prog.P, **f64)
    handles = []
    # candidates at the register limit whose values are wrong (hipcc faults,
    # ConstraintCollocator._verify_build) do not compete
    rejected = set()
    for k, (label, kw, meta, hsaco, spills) in enumerate(built):
        try:
            col._verify_build(hsaco, meta)
        except hb.BuildRejected as err:
            rejected.add(k)
            if log:
                log('%-10s refused: %s' % (label, err.verdict['errors']))
    for label, kw, meta, hsaco, spills in built:
        h = hb.HipProblem(col._descriptor(meta), hsaco)
        col._install_tables(h)
        h.use_torch_stream()
        handles.append(h)
    times = {k: [[] for _ in built] for k in ('fused', 'jac')}
    con_ms = []
    sel = {'fused': hb.EVAL_FUSED_KERNEL, 'jac': hb.EVAL_JAC}
    # clock ramp
    for _ in range(3):
        handles[0].time_eval_shard(hb.EVAL_FUSED, free, con, nodes, jac, a, b,
                                   iters)
    for _ in range(rounds):
        # (opty_con does not depend on the geometry of the Jacobian waves)
        con_ms.append(handles[0].time_eval_shard(
            hb.EVAL_CON, free, con, nodes, jac, a, b, iters))
        for k, h in enumerate(handles):
            for what in ('fused', 'jac'):
                label = built[k][0]
                if what == 'fused' and label.startswith('jac'):
                    continue
                if what == 'jac' and label.startswith('fused'):
                    continue
                if {'fused': 'opty_conjac', 'jac': 'opty_jac'}[what] in \
                        built[k][4]:
                    continue            # spills vector registers
                if k in rejected:
                    continue
                times[what][k].append(h.time_eval_shard(
                    sel[what], free, con, nodes, jac, a, b, iters))
    for h in handles:
        h.close()
    best = {}
    measured = {'fused': {}, 'jac': {}}
    for what in ('fused', 'jac'):
        for k, (label, kw, meta, _, _) in enumerate(built):
            if not times[what][k]:
                continue
            ms = float(np.median(times[what][k]))
            tag = label.split('=')[1] if '=' in label else label
            if label == 'seed':
                tag = str(geo['fused'] if what == 'fused' else geo['jac'])
            measured[what][tag] = ms
            # a challenger must beat the seed by more than run-to-run noise;
            # one that changes the shape of ALL kernels of the module (chunk
            # width / workgroup width) by more than box-to-box spread
            margin = 0.0 if label == 'seed' else (
                0.01 if ('=' in label or label.startswith(('jac', 'fused')))
                else 0.03)
            # ... and by more than the resolution of a microsecond-scale
            # launch
            if what not in best or (ms < best[what][0]*(1.0 - margin) and
                                    ms < best[what][0] - 3e-4):
                best[what] = (ms, label, kw)
            if log:
                log('%-10s %-6s %.4f ms' % (label, what, ms))
    if geo['line_mode']:
        # the fused kernel (what a solver's pair and the benchmark's step
        # cost) picks the shape (chunk / workgroup width / its strips); the
        # Jacobian-only kernel then takes the best strip count measured with
        # that shape
        options = dict(best['fused'][2])
        shape = lambda kw: (kw.get('chunk', 32), kw.get('waves'))
        same = [(float(np.median(times['jac'][k])), kw)
                for k, (label, kw, _, _, _) in enumerate(built)
                if times['jac'][k] and shape(kw) == shape(options)]
        if same:
            # what only the Jacobian kernel reads: its strip count and the
            # dispatch order of its workgroups
            jac_ms, jac_kw = min(same, key=lambda t: t[0])
            for tag in ('groups', 'order', 'pad'):
                options.pop(tag, None)
                if tag in jac_kw:
                    options[tag] = jac_kw[tag]
            best['jac'] = (jac_ms, 'jac', jac_kw)
    else:
        # small blocks: the only choice is how the tile is flushed; the fused
        # kernel decides
        options = dict(small_flush='chunk') \
            if best['fused'][1] == 'chunk' else {}
    sha = problem_sha(prog)
    measured['con'] = float(np.median(con_ms))
    # does ONE fused launch beat the two it replaces?  (opty_hip_desc.
    # fused_loses: opty_hip_eval_con_jac / EVAL_FUSED issue opty_con and
    # opty_jac when it does not -- by more than the resolution of the timer)
    fused_pays = best['fused'][0] <= \
        (best['jac'][0] + measured['con'])*1.01 + 3e-4
    # ... and is the fused kernel even faster than the Jacobian-only one
    # (10-link pendulum: 0.136 vs 0.139 ms -- the constraint wave rides in
    # the shadow of the store stream and the 10 waves per block dispatch
    # better than 12 strips)?  Then EVAL_JAC launches it
    # (opty_hip_desc.jac_via_fused), by more than run-to-run noise
    jac_via_fused = fused_pays and \
        best['fused'][0] < best['jac'][0]*0.99 - 3e-4
    entry = dict(options=options, fused_pays=bool(fused_pays),
                 jac_via_fused=bool(jac_via_fused),
                 seed=dict(jac=geo['jac'], fused=geo['fused']),
                 measured_ms=measured, nodes=nodes,
                 device=torch.cuda.get_device_name(dev), problem_sha=sha)
    if save:
        record(key_of(prog, blocks, sha), entry, path)
    return entry
