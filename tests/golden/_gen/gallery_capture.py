#!/usr/bin/env python
"""Records reference goldens for the problems of the reference's own example
gallery (``/root/reference/examples-gallery/*/plot_*.py``).  Build container
only -- nothing here travels to the GPU box but the ``.npz`` data it writes.

    python tests/golden/_gen/gallery_capture.py [script-stem ...]
    GALLERY_FLIP=1 python tests/golden/_gen/gallery_capture.py [...]

(``GALLERY_FLIP=1``: every script once more under the OTHER discretisation
rule -- ``gallery_<stem>__flipped.npz`` -- because 24 of the 28 use backward
Euler.)

Every script is executed (in a child process) with ``opty.Problem`` replaced
by a subclass that (1) records the arguments the script constructs its
problem with, (2) lets the REAL reference build its compiled (Cython + C)
constraint and Jacobian functions, and (3) stops the script at the first
``solve`` / ``plot_*`` call, keeping the script's initial guess.  The reference
functions are then evaluated at a deterministic ``free`` vector and

* the problem's inputs -- equations of motion, state symbols, maps, instance
  constraints, as SymPy expression tables (``tests/sympy_codec.py``) -- and
* the reference's outputs -- ``constraints(free)``, ``jacobian(free)``,
  ``jacobian_indices()`` (full for small problems, strided node samples +
  order-insensitive sums for large ones)

are written to ``tests/golden/gallery_<stem>.npz`` (+ an entry in
``tests/golden/GALLERY.json``).  ``tests/gallery_cases.py`` rebuilds the
keyword arguments from that data; ``tests/test_gallery_parity.py`` holds the
HIP path to it.
"""
import inspect
import json
import os
import runpy
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..', '..'))
OUT = os.path.abspath(os.path.join(HERE, '..'))
GALLERY = '/root/reference/examples-gallery'

#: scripts that need packages this container lacks (pygait2d, pydy, yeadon)
UNBUILDABLE = {'plot_human_gait', 'plot_park2004', 'plot_sit_to_stand'}

#: full fixtures up to this many Jacobian values, sampled ones beyond
FULL_MAX_NNZ = 150_000
SAMPLES = 12


class _Stop(Exception):
    pass


def _scripts():
    found = {}
    for level in sorted(os.listdir(GALLERY)):
        d = os.path.join(GALLERY, level)
        if not os.path.isdir(d):
            continue
        for f in sorted(os.listdir(d)):
            if f.startswith('plot_') and f.endswith('.py'):
                found[f[:-3]] = os.path.join(d, f)
    return found


def _child(stem):
    sys.path.insert(0, os.path.join(HERE, 'stubs'))
    sys.path.insert(0, '/root/reference')
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MPLBACKEND'] = 'Agg'
    import numpy as np
    import sympy as sm
    import opty
    import opty.direct_collocation as dc
    from examples import problems
    import sympy_codec

    path = _scripts()[stem]
    captured = []
    RefProblem = dc.Problem
    sig = inspect.signature(RefProblem.__init__)

    # (patched in place: the reference's ``_DocInherit`` descriptor recurses
    # without end on a subclass of ``Problem``)
    ref_init = RefProblem.__init__

    flip = os.environ.get('GALLERY_FLIP') == '1'

    def init(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        if flip:
            # the same problem under the OTHER discretisation rule (24 of
            # the 28 scripts use backward Euler): a second reference record
            # per script, for the midpoint columns u_i, u_n of real problems
            other = {'backward euler': 'midpoint',
                     'midpoint': 'backward euler'}
            bound.arguments['integration_method'] = other[
                bound.arguments['integration_method']]
            args, kwargs = bound.args[1:], bound.kwargs
        self._captured_args = dict(bound.arguments)
        t0 = time.time()
        ref_init(self, *args, **kwargs)
        self._build_seconds = time.time() - t0
        self._captured_guess = None
        captured.append(self)

    def stop_with(self, vector=None, *args, **kwargs):
        if vector is not None and np.ndim(vector) == 1:
            self._captured_guess = np.array(vector, dtype=float)
        raise _Stop()

    # objectives built by the reference's own helper are recorded too
    # (opty/utils.py:329: 9 of the scripts use it)
    import opty.utils as ou
    ref_cof = ou.create_objective_function
    cof_sig = inspect.signature(ref_cof)
    objectives = []

    def cof(*args, **kwargs):
        bound = cof_sig.bind(*args, **kwargs)
        bound.apply_defaults()
        out = ref_cof(*args, **kwargs)
        objectives.append((dict(bound.arguments), out))
        return out

    ou.create_objective_function = cof
    opty.create_objective_function = cof
    if hasattr(dc, 'create_objective_function'):
        dc.create_objective_function = cof

    RefProblem.__init__ = init
    for name in ('solve', 'plot_trajectories', 'plot_constraint_violations',
                 'plot_objective_value', 'plot_jacobian_sparsity'):
        if hasattr(RefProblem, name):
            setattr(RefProblem, name, stop_with)
    os.chdir(os.path.dirname(path))
    sys.argv = [path]
    t0 = time.time()
    try:
        runpy.run_path(path, run_name='__main__')
    except _Stop:
        pass
    except BaseException as err:            # noqa: BLE001
        if not captured:
            raise
        print('script stopped after the capture: %r' % (err,))
    if not captured:
        raise RuntimeError('%s never constructed a Problem' % stem)
    prob = captured[0]
    a = prob._captured_args
    col = prob.collocator
    N, M, n = col.num_collocation_nodes, col.num_eom, col.num_states
    vd = bool(col._variable_duration)
    q = col.num_unknown_input_trajectories
    be = col.integration_method == 'backward euler'
    C = 2*n + (q if be else 2*q) + col.num_unknown_parameters + int(vd)
    P = M*C

    # -- the free vector ---------------------------------------------------
    guess = prob._captured_guess
    if guess is None or guess.shape != (prob.num_free,):
        guess = None
    hashed = problems.make_free(prob.num_free, seed=0, variable_duration=vd,
                                interval=0.01)
    candidates = [('hash', hashed)]
    if guess is not None:
        candidates.append(('guess+hash', guess + 0.0625*hashed*(
            1.0 + np.abs(guess))))
        candidates.append(('guess', guess))
    candidates.append(('hash01', 0.25 + 0.5*np.abs(hashed)))
    for recipe, free in candidates:
        if vd and recipe != 'guess':
            free = free.copy()
            free[-1] = 0.01 if guess is None or not guess[-1] > 0 \
                else float(guess[-1])
        with np.errstate(all='ignore'):
            cv = np.array(prob.con(free), dtype=float)
            jv = np.array(prob.con_jac(free), dtype=float)
        if np.all(np.isfinite(cv)) and np.all(np.isfinite(jv)):
            break
    else:
        raise RuntimeError('%s: no candidate free vector gives finite values'
                           % stem)
    rows, cols = col.jacobian_indices()
    rows, cols = np.array(rows), np.array(cols)
    assert rows.dtype == np.int64 and cols.dtype == np.int64

    # instance tail in canonical order (by row, then column): the reference
    # iterates a set of atoms
    base = P*(N - 1)
    assert len(jv) == len(rows) and len(rows) >= base
    order = base + np.lexsort((cols[base:], rows[base:]))
    rows[base:], cols[base:], jv[base:] = rows[order], cols[order], jv[order]

    # -- the inputs, as data -------------------------------------------------
    par_map = a['known_parameter_map'] or {}
    traj_map = a['known_trajectory_map'] or {}
    inst = a['instance_constraints']
    callable_known = [str(k) for k, v in traj_map.items() if callable(v)]
    traj_vals = [np.asarray(v(free) if callable(v) else v, dtype=float)
                 for v in traj_map.values()]
    interval = a['node_time_interval']
    exprs = [sm.ImmutableDenseMatrix(a['equations_of_motion'])]
    exprs += list(a['state_symbols'])
    exprs += list(par_map.keys())
    exprs += list(traj_map.keys())
    exprs += list(inst or ())
    exprs += [interval if isinstance(interval, sm.Basic) else sm.Integer(0)]
    # always explicit: with ``time_symbol=None`` the reference takes
    # mechanics' global ``dynamicsymbols._t``, i.e. whatever the process set
    # last (``opty/direct_collocation.py:1490-1494``)
    exprs += [col.time_symbol]
    blob = sympy_codec.encode(exprs)
    back = sympy_codec.decode(json.loads(json.dumps(blob)))
    for x, y in zip(exprs, back):
        if sm.sympify(x) != y:
            raise RuntimeError('%s: expression does not survive the codec: %s'
                               % (stem, x))
    layout = dict(
        num_states=len(a['state_symbols']), num_par=len(par_map),
        num_traj=len(traj_map), num_inst=len(inst or ()),
        interval_is_symbol=isinstance(interval, sm.Basic),
        has_time_symbol=True)

    # -- the NLP facade's bound arrays (opty/direct_collocation.py:370-440) --
    extra = {}
    bounds = a.get('bounds') or {}
    eom_bounds = a.get('eom_bounds') or {}
    bound_keys = list(bounds.keys())
    extra['bounds_lo'] = np.array([np.broadcast_to(
        np.asarray(bounds[k][0], dtype=float), (N,)) for k in bound_keys]) \
        if bound_keys else np.zeros((0, N))
    extra['bounds_hi'] = np.array([np.broadcast_to(
        np.asarray(bounds[k][1], dtype=float), (N,)) for k in bound_keys]) \
        if bound_keys else np.zeros((0, N))
    extra['eom_bounds'] = np.array([[float(k), float(v[0]), float(v[1])]
                                    for k, v in eom_bounds.items()]) \
        if eom_bounds else np.zeros((0, 3))
    extra['lower_bound'] = np.asarray(prob.lower_bound, dtype=float)
    extra['upper_bound'] = np.asarray(prob.upper_bound, dtype=float)
    extra['low_con'] = np.asarray(prob._low_con_bounds, dtype=float)
    extra['upp_con'] = np.asarray(prob._upp_con_bounds, dtype=float)
    bounds_blob = sympy_codec.encode(bound_keys)
    assert sympy_codec.decode(json.loads(json.dumps(bounds_blob))) == \
        [sm.sympify(k) for k in bound_keys]
    extra['bounds_keys'] = np.array(json.dumps(bounds_blob))

    # -- the objective, when the script built it with the reference's helper -
    has_objective = False
    for oargs, (obj_f, grad_f) in objectives:
        if prob.obj is not obj_f:
            continue
        h_obj = oargs['node_time_interval']
        if isinstance(h_obj, sm.Basic):
            break                       # (not evaluable in the reference)
        oexprs = [sm.sympify(oargs['objective'])]
        groups = [list(oargs['state_symbols']),
                  list(oargs['unknown_input_trajectories']),
                  list(oargs['unknown_parameters'])]
        for g in groups:
            oexprs += g
        oexprs += [oargs['time_symbol']]
        oblob = sympy_codec.encode(oexprs)
        oback = sympy_codec.decode(json.loads(json.dumps(oblob)))
        assert all(sm.sympify(x) == y for x, y in zip(oexprs, oback))
        n_obj = (len(groups[0]) + len(groups[1]))*int(
            oargs['num_collocation_nodes']) + len(groups[2])
        ofree = free[:n_obj]
        extra['objective'] = np.array(json.dumps(oblob))
        extra['objective_layout'] = np.array(
            [len(g) for g in groups] +
            [int(oargs['num_collocation_nodes'])])
        extra['objective_interval'] = np.array([float(h_obj)])
        extra['objective_method'] = np.array(oargs['integration_method'])
        extra['objective_value'] = np.array([float(obj_f(ofree))])
        extra['objective_grad'] = np.asarray(grad_f(ofree), dtype=float)
        has_objective = True
        break

    meta = dict(
        name='gallery_' + stem[5:] + ('__flipped' if flip else ''),
        script=os.path.relpath(path, '/root/reference'),
        N=N, M=M, n=n, q=q, r=col.num_unknown_parameters, s=int(vd),
        o=col.num_instance_constraints, C=int(C),
        num_free=int(prob.num_free), num_constraints=int(col.num_constraints),
        nnz=int(len(rows)), nnz_inst=int(len(rows) - base),
        method=col.integration_method, free_recipe=recipe,
        callable_known=callable_known, layout=layout,
        has_objective=has_objective and not flip,
        flipped=flip, num_bounds=len(bound_keys),
        num_eom_bounds=len(eom_bounds),
        states=[str(x) for x in col.state_symbols],
        known_parameters=[str(x) for x in col.known_parameters],
        unknown_parameters=[str(x) for x in col.unknown_parameters],
        known_trajectories=[str(x) for x in col.known_input_trajectories],
        unknown_trajectories=[str(x) for x in
                              col.unknown_input_trajectories],
        sympy=sm.__version__, numpy=np.__version__,
        reference='csu-hmc/opty v1.6.0.dev0 (compiled cython backend)',
        build_s=round(prob._build_seconds, 1),
        wall_s=round(time.time() - t0, 1))
    arrays = dict(
        free=free,
        par_values=np.array([float(v) for v in par_map.values()]),
        traj_values=(np.array(traj_vals) if traj_vals
                     else np.zeros((0, N))),
        interval=np.array([0.0 if layout['interval_is_symbol']
                           else float(interval)]),
        problem=np.array(json.dumps(blob)), **extra)
    if len(jv) <= FULL_MAX_NNZ:
        arrays.update(con=cv, jac=jv, rows=rows, cols=cols)
        meta['kind'] = 'full'
    else:
        stride = max(1, (N - 1)//SAMPLES)
        nodes = set(range(0, N - 1, stride)) | {0, 1, N - 3, N - 2}
        nodes = np.array(sorted(k for k in nodes if 0 <= k < N - 1),
                         dtype=np.int64)
        blk = jv[:base].reshape(N - 1, P)
        cb = cv[:M*(N - 1)].reshape(M, N - 1)
        arrays.update(
            nodes=nodes, jac_nodes=blk[nodes], con_nodes=cb[:, nodes],
            rows_nodes=rows[:base].reshape(N - 1, P)[nodes],
            cols_nodes=cols[:base].reshape(N - 1, P)[nodes],
            jac_entry_sums=blk.sum(axis=0), con_eq_sums=cb.sum(axis=1),
            jac_abs_sum=np.array([np.abs(blk).sum()]),
            con_abs_sum=np.array([np.abs(cb).sum()]),
            con_tail=cv[M*(N - 1):], jac_tail=jv[base:],
            rows_tail=rows[base:], cols_tail=cols[base:])
        meta['kind'] = 'sampled'
    np.savez_compressed(os.path.join(OUT, meta['name'] + '.npz'), **arrays)
    print('META ' + json.dumps(meta))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--child':
        _child(sys.argv[2])
        return
    stems = sys.argv[1:] or [s for s in _scripts() if s not in UNBUILDABLE]
    manifest_path = os.path.join(OUT, 'GALLERY.json')
    manifest = {}
    if os.path.exists(manifest_path):
        with open(manifest_path) as f:
            manifest = json.load(f)
    from concurrent.futures import ThreadPoolExecutor

    def one(stem):
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.abspath(__file__),
                            '--child', stem], capture_output=True, text=True)
        meta = None
        for line in p.stdout.splitlines():
            if line.startswith('META '):
                meta = json.loads(line[5:])
        return stem, meta, p, time.time() - t0

    with ThreadPoolExecutor(int(os.environ.get('GALLERY_JOBS', '4'))) as ex:
        for stem, meta, p, secs in ex.map(one, stems):
            if meta is None:
                print('FAILED %s (%.0f s)\n%s' % (stem, secs,
                                                  p.stderr[-1500:]))
                continue
            manifest[meta['name']] = meta
            print('%-48s %s N=%d M=%d n=%d q=%d r=%d s=%d o=%d nnz=%d %s '
                  '(%.0f s)' % (meta['name'], meta['kind'], meta['N'],
                                meta['M'], meta['n'], meta['q'], meta['r'],
                                meta['s'], meta['o'], meta['nnz'],
                                meta['free_recipe'], secs))
            with open(manifest_path, 'w') as f:
                json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
