"""The problems of the reference's example gallery as test inputs.

``tests/golden/gallery_<name>.npz`` (written by
``tests/golden/_gen/gallery_capture.py`` in the build container) holds, per
gallery script, the arguments the script constructs its ``Problem`` with --
as data: SymPy expression tables (:mod:`sympy_codec`), parameter values and
known-trajectory arrays -- and what the REAL reference (compiled Cython
backend) returned for ``constraints(free)``, ``jacobian(free)`` and
``jacobian_indices()`` at the recorded ``free`` vector.  :func:`load` turns the
data back into keyword arguments that :class:`opty_amd.ConstraintCollocator`
(and the oracle) accept.

Known trajectories that a script supplies as functions of ``free``
(``plot_hilly_race.py``) are recorded as the arrays they return at the
fixture's ``free`` -- the reference evaluates them once per call and does not
differentiate through them (``opty/direct_collocation.py:2916-2917``).
"""
import json
import os

import numpy as np

import sympy_codec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

with open(os.path.join(GOLDEN, 'GALLERY.json')) as _f:
    MANIFEST = json.load(_f)

NAMES = sorted(MANIFEST)
FULL = [k for k in NAMES if MANIFEST[k]['kind'] == 'full']
SAMPLED = [k for k in NAMES if MANIFEST[k]['kind'] == 'sampled']


def load(name):
    """-> ``(meta, arrays, kwargs)``."""
    meta = MANIFEST[name]
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    lay = meta['layout']
    objs = sympy_codec.decode(json.loads(str(z['problem'])))
    k = 0
    eom = objs[k]
    k += 1
    states = tuple(objs[k:k + lay['num_states']])
    k += lay['num_states']
    par_syms = objs[k:k + lay['num_par']]
    k += lay['num_par']
    traj_syms = objs[k:k + lay['num_traj']]
    k += lay['num_traj']
    inst = tuple(objs[k:k + lay['num_inst']])
    k += lay['num_inst']
    interval_sym, time_sym = objs[k], objs[k + 1]
    kwargs = dict(
        equations_of_motion=eom, state_symbols=states,
        num_collocation_nodes=meta['N'],
        node_time_interval=(interval_sym if lay['interval_is_symbol']
                            else float(z['interval'][0])),
        known_parameter_map=dict(zip(par_syms,
                                     (float(v) for v in z['par_values']))),
        known_trajectory_map=dict(zip(traj_syms,
                                      (np.array(v) for v in
                                       z['traj_values']))),
        instance_constraints=inst if lay['num_inst'] else None,
        time_symbol=time_sym if lay['has_time_symbol'] else None,
        integration_method=meta['method'])
    return meta, z, kwargs


def facade(name):
    """Bounds of the script's ``Problem`` and what the reference made of
    them: ``(bounds, eom_bounds, expected arrays)``; ``bounds`` values are
    per-node arrays (a scalar bound is recorded broadcast)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    keys = sympy_codec.decode(json.loads(str(z['bounds_keys'])))
    # (recorded broadcast to N values: a parameter or the node time interval
    # has one slot)
    bounds = {k: ((np.array(lo), np.array(hi)) if k.args
                  else (float(lo[0]), float(hi[0])))
              for k, lo, hi in zip(keys, z['bounds_lo'], z['bounds_hi'])}
    eom_bounds = {int(k): (float(lo), float(hi))
                  for k, lo, hi in z['eom_bounds']}
    expected = {k: z[k] for k in ('lower_bound', 'upper_bound', 'low_con',
                                  'upp_con')}
    return bounds, (eom_bounds or None), expected


def objective(name):
    """Arguments of the script's ``create_objective_function`` call (as the
    reference's signature takes them) and the reference's value / gradient at
    the fixture's ``free``; None when the script wrote its objective by
    hand."""
    if not MANIFEST[name].get('has_objective'):
        return None
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    objs = sympy_codec.decode(json.loads(str(z['objective'])))
    n, q, r, N = (int(v) for v in z['objective_layout'])
    expr, rest = objs[0], objs[1:]
    args = dict(objective=expr, state_symbols=tuple(rest[:n]),
                unknown_input_trajectories=tuple(rest[n:n + q]),
                unknown_parameters=tuple(rest[n + q:n + q + r]),
                num_collocation_nodes=N,
                node_time_interval=float(z['objective_interval'][0]),
                integration_method=str(z['objective_method']),
                time_symbol=rest[n + q + r])
    num = (n + q)*N + r
    return args, z['free'][:num], float(z['objective_value'][0]), \
        z['objective_grad']


def rescale(kwargs, num_nodes):
    """The same problem on ``num_nodes`` collocation nodes (full-size bench /
    parity workloads from a gallery problem): the duration is kept for a
    fixed interval, instance-constraint times ``k*h`` of a variable-duration
    problem keep their relative position (``k -> round(k (N'-1)/(N-1))``, so
    ``(N-1)*h`` stays the last node), known trajectories are interpolated
    onto the new grid."""
    import sympy as sm
    kw = dict(kwargs)
    old = kw['num_collocation_nodes']
    if num_nodes == old:
        return kw
    kw['num_collocation_nodes'] = int(num_nodes)
    h = kw['node_time_interval']
    if isinstance(h, sm.Symbol):
        def move(con):
            rep = {}
            for f in con.atoms(sm.Function):
                if not isinstance(f, sm.core.function.AppliedUndef):
                    continue
                k = f.args[0]/h
                if k.is_Number and k != 0:
                    rep[f] = f.func(int(round(float(k)*(num_nodes - 1) /
                                              (old - 1)))*h)
            return con.xreplace(rep)
        if kw['instance_constraints'] is not None:
            kw['instance_constraints'] = tuple(
                move(sm.sympify(c)) for c in kw['instance_constraints'])
    else:
        kw['node_time_interval'] = float(h)*(old - 1)/(num_nodes - 1)
    grid_old = np.linspace(0.0, 1.0, old)
    grid_new = np.linspace(0.0, 1.0, num_nodes)
    kw['known_trajectory_map'] = {
        f: np.interp(grid_new, grid_old, v)
        for f, v in kw['known_trajectory_map'].items()}
    return kw
