#!/usr/bin/env python
"""Generates the golden vectors under ``tests/golden/`` by running the REAL
reference (``/root/reference/opty``, compiled Cython + C path) in the build
container.  Run from the repo root::

    python tests/golden/_gen/make_golden.py [name ...]

The reference cannot travel to the GPU box, so only its inputs/outputs are
committed (as ``.npz`` data), together with this script.  ``cyipopt`` is absent
here; ``stubs/cyipopt.py`` is a ten-line stand-in that lets
``opty/direct_collocation.py:10`` import.

For every problem in ``examples.problems.CONFIGS`` that is listed below it
stores: the symbol ordering the reference derived, sizes, the deterministic
``free`` recipe (seed) and

* small N: the full ``constraints(free)``, ``jacobian(free)`` and
  ``jacobian_indices()`` arrays;
* large N: a strided node sample of the con/jac blocks, per-equation /
  per-entry sums (order-insensitive checksums) and the index arrays only at the
  sampled nodes (the closed form is validated on the small cases).
"""

import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..', '..'))
sys.path.insert(0, os.path.join(HERE, 'stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, REPO)

import numpy as np                                            # noqa: E402
import sympy as sm                                            # noqa: E402
from opty.direct_collocation import ConstraintCollocator     # noqa: E402
from examples import problems                                 # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, '..'))

SMALL = ['config1_vyasarayani', 'config2_pendulum_small',
         'config3_10link_small', 'pend3_link_midpoint_small',
         'pend2_link_vardur_unkmass_small', 'msd_be_small', 'msd_mid_small',
         'vardur_pendulum_small', 'config5_standin_24link_small',
         'chaplygin_be_small', 'chaplygin_mid_small', 'one_eom_be_small',
         'one_eom_mid_small', 'implicit_traj_be_small',
         'implicit_traj_mid_small', 'c99_be_small', 'c99_mid_small',
         'elementary_be_small',
         'elementary_mid_small', 'delay_be_small', 'delay_mid_small',
         'odd_block_be_small', 'odd_block_mid_small',
         'piecewise_be_small', 'piecewise_mid_small',
         'states_only_mid_small', 'gaitlike_3link_be_small',
         'gaitlike_3link_mid_small', 'config5_gaitlike_24link_small',
         'one_legged_small', 'biped_small', 'biped_mid_small']
LARGE = {'config2_pendulum': 499, 'config3_10link': 4999,
         'config5_standin_24link': 4999, 'config5_gaitlike_24link': 4999,
         'config5_one_legged': 4999, 'config5_biped': 4999}


def sample_nodes(num_con_nodes, stride):
    idx = set(range(0, num_con_nodes, stride))
    idx |= {0, 1, num_con_nodes - 2, num_con_nodes - 1}
    return np.array(sorted(idx), dtype=np.int64)


def run(name):
    kw = problems.build(name)
    t0 = time.time()
    # the reference's generated C does not compile for Piecewise (its symbol
    # renaming breaks the printer's multi-line ternaries): those fixtures come
    # from its NumPy backend (lambdify; same wrappers, layouts and indices)
    numpy_backend = name.startswith('piecewise')
    col = ConstraintCollocator(
        parallel=not numpy_backend,
        backend='numpy' if numpy_backend else 'cython', **kw)
    con = col.generate_constraint_function()
    jac = col.generate_jacobian_function()
    rows, cols = col.jacobian_indices()
    N = col.num_collocation_nodes
    M = col.num_eom
    vd = col._variable_duration
    seed = 0
    free = problems.make_free(col.num_free, seed=seed, variable_duration=vd,
                              interval=0.01)
    cv = con(free).copy()
    jv = jac(free).copy()
    o = col.num_instance_constraints
    # The reference orders the partials of an instance constraint with two or
    # more function atoms by iterating a set; fixtures
    # store that tail in canonical order: by row, then column.
    base = len(rows) - sum(len(c.atoms(sm.Function))
                           for c in (col.instance_constraints or ()))
    order = base + np.lexsort((cols[base:], rows[base:]))
    rows[base:], cols[base:], jv[base:] = rows[order], cols[order], jv[order]
    qn = col.num_unknown_input_trajectories
    C = (2*col.num_states + (1 if col.integration_method ==
                             'backward euler' else 2)*qn +
         col.num_unknown_parameters + int(vd))
    assert (len(jv) - (len(rows) - (N - 1)*M*C)) == (N - 1)*M*C
    meta = dict(
        name=name, N=N, M=M, n=col.num_states,
        q=col.num_unknown_input_trajectories,
        r=col.num_unknown_parameters, s=int(vd), o=o, C=int(C),
        num_free=col.num_free, num_constraints=col.num_constraints,
        nnz=len(rows), nnz_inst=int(len(rows) - (N - 1)*M*C),
        method=col.integration_method, seed=seed,
        states=[str(x) for x in col.state_symbols],
        known_parameters=[str(x) for x in col.known_parameters],
        unknown_parameters=[str(x) for x in col.unknown_parameters],
        known_trajectories=[str(x) for x in col.known_input_trajectories],
        unknown_trajectories=[str(x) for x in
                              col.unknown_input_trajectories],
        sympy=sm.__version__, numpy=np.__version__,
        reference='csu-hmc/opty v1.6.0.dev0 (' + (
            'numpy backend)' if numpy_backend else
            'compiled cython backend, parallel=True)'),
        wall_s=round(time.time() - t0, 1))
    assert rows.dtype == np.int64 and cols.dtype == np.int64
    P = M*C
    arrays = {}
    if name in LARGE:
        nodes = sample_nodes(N - 1, LARGE[name])
        blk = jv[:P*(N - 1)].reshape(N - 1, P)
        cb = cv[:M*(N - 1)].reshape(M, N - 1)
        arrays.update(
            nodes=nodes, jac_nodes=blk[nodes], con_nodes=cb[:, nodes],
            rows_nodes=rows[:P*(N - 1)].reshape(N - 1, P)[nodes],
            cols_nodes=cols[:P*(N - 1)].reshape(N - 1, P)[nodes],
            jac_entry_sums=blk.sum(axis=0), con_eq_sums=cb.sum(axis=1),
            jac_abs_sum=np.array([np.abs(blk).sum()]),
            con_tail=cv[M*(N - 1):], jac_tail=jv[P*(N - 1):],
            rows_tail=rows[P*(N - 1):], cols_tail=cols[P*(N - 1):])
        meta['kind'] = 'sampled'
    else:
        arrays.update(free=free, con=cv, jac=jv, rows=rows, cols=cols)
        meta['kind'] = 'full'
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    return meta


def run_problem_facade():
    """Bound arrays and extraction helpers of the reference's ``Problem``
    (``opty/direct_collocation.py:370-440``, ``:972-1054``) for the pendulum
    swing-up with state / input / eom bounds."""
    from opty.direct_collocation import Problem
    kw = problems.pendulum_swing_up(num_nodes=31)
    theta, omega = kw['state_symbols']
    T = [f for f in kw['equations_of_motion'].atoms(sm.Function)
         if f.func.__name__ == 'T'][0]
    N = kw['num_collocation_nodes']
    bounds = {T: (-2.0, 2.0), omega: (-np.linspace(1.0, 3.0, N), 10.0)}
    prob = Problem(lambda f: 0.0, lambda f: f, bounds=bounds,
                   eom_bounds={1: (-0.5, 0.25)}, **kw)
    free = problems.make_free(prob.num_free, seed=3)
    np.savez_compressed(
        os.path.join(OUT, 'problem_facade.npz'),
        lower_bound=prob.lower_bound, upper_bound=prob.upper_bound,
        low_con=prob._low_con_bounds, upp_con=prob._upp_con_bounds,
        free=free, extract_T_theta=prob.extract_values(free, T, theta),
        time_vector=prob.time_vector(), INF=np.array([prob.INF]))
    return dict(kind='facade', name='problem_facade', N=N,
                num_free=prob.num_free,
                num_constraints=prob.num_constraints)


def run_ufuncify():
    """The reference's own ``test_ufuncify_matrix`` case
    (``opty/tests/test_utils.py:244-336``: a 2 x 2 matrix of transcendental
    expressions over symbols named ``a``, ``b``, ``if`` / ``I`` / ``i``,
    n = 10 000) on deterministic inputs, evaluated by the reference's
    ``ufuncify_matrix`` with ``c`` as a vector argument and as ``const``."""
    from opty.utils import ufuncify_matrix
    a, b, c, I, i = sm.symbols('a, b, if, I, i')
    mat = sm.Matrix([[a**2*sm.cos(sm.pi*b)**c, sm.tan(b)/sm.sin(a + b) + c**4],
                     [a**2 + b**2 - sm.sqrt(c),
                      ((a + b + c)*(a + b))/a*sm.sin(b)]])
    n = 10000
    # (0, 1) like the reference's np.random.random; c > 10 to stay real
    a_vals = 0.5*(problems.make_free(n, seed=21) + 1.0) + 2.0**-21
    b_vals = 0.5*(problems.make_free(n, seed=22) + 1.0) + 2.0**-21
    c_vals = 0.5*(problems.make_free(n, seed=23) + 1.0) + 10.0
    c_val = float(c_vals[17])
    out = {}
    f = ufuncify_matrix((a, b, c), mat)
    out['vec'] = f(np.empty((n, 4)), a_vals, b_vals, c_vals).copy()
    f = ufuncify_matrix((a, b, c), mat, const=(c,))
    out['const'] = f(np.empty((n, 4)), a_vals, b_vals, c_val).copy()
    # the variants below compile different C (OpenMP flags, symbol order in
    # cse): equal to the last bits, not bit for bit
    same = dict(rtol=1e-14, atol=0.0)
    f = ufuncify_matrix((a, b, c), mat, const=(c,), parallel=True)
    np.testing.assert_allclose(f(np.empty((n, 4)), a_vals, b_vals, c_val),
                               out['const'], **same)
    for other in (I, i):
        f = ufuncify_matrix((a, b, other), mat.xreplace({c: other}))
        np.testing.assert_allclose(
            f(np.empty((n, 4)), a_vals, b_vals, c_vals), out['vec'], **same)
    # cse pair input (opty/utils.py:677-682)
    f = ufuncify_matrix((a, b, c), sm.cse(mat))
    np.testing.assert_allclose(f(np.empty((n, 4)), a_vals, b_vals, c_vals),
                               out['vec'], **same)
    # inputs are a recipe (seeds above); outputs are kept for every 7th row
    rows = np.unique(np.append(np.arange(0, n, 7), n - 1))
    np.savez_compressed(os.path.join(OUT, 'ufuncify_matrix.npz'),
                        rows=rows, seeds=np.array([21, 22, 23]),
                        c_const=np.array([c_val]),
                        a_rows=a_vals[rows], result_vec=out['vec'][rows],
                        result_const=out['const'][rows])
    return dict(kind='ufuncify', name='ufuncify_matrix', n=n,
                sympy=sm.__version__,
                reference='csu-hmc/opty v1.6.0.dev0 ufuncify_matrix '
                          '(opty/utils.py:639)')


def run_objective():
    """Objective value and gradient of ``tests/objective_cases.py:
    reference_cases`` from the reference's ``create_objective_function``
    (``opty/utils.py:329-470``)."""
    from opty.utils import create_objective_function
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import objective_cases
    t, cases = objective_cases.reference_cases()
    arrays = {}
    for case in cases:
        states, inputs, unknowns = case['args']
        obj, grad = create_objective_function(
            case['expr'], states, inputs, unknowns, case['N'], case['h'],
            integration_method=case['method'], time_symbol=t)
        free = problems.make_free(case['num_free'], seed=case['seed'])
        arrays[case['name'] + '_value'] = np.array([obj(free)])
        arrays[case['name'] + '_grad'] = np.asarray(grad(free), dtype=float)
        print(case['name'], arrays[case['name'] + '_value'])
    np.savez_compressed(os.path.join(OUT, 'objective.npz'), **arrays)
    return dict(kind='objective', name='objective',
                cases=[c['name'] for c in cases], sympy=sm.__version__,
                reference='csu-hmc/opty v1.6.0.dev0 '
                          'create_objective_function (opty/utils.py:329)')


def main():
    if sys.argv[1:] == ['objective']:
        manifest_path = os.path.join(OUT, 'MANIFEST.json')
        with open(manifest_path) as f:
            manifest = json.load(f)
        manifest['objective'] = run_objective()
        with open(manifest_path, 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if sys.argv[1:] == ['ufuncify_matrix']:
        manifest_path = os.path.join(OUT, 'MANIFEST.json')
        with open(manifest_path) as f:
            manifest = json.load(f)
        manifest['ufuncify_matrix'] = run_ufuncify()
        with open(manifest_path, 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if sys.argv[1:] == ['problem_facade']:
        manifest_path = os.path.join(OUT, 'MANIFEST.json')
        with open(manifest_path) as f:
            manifest = json.load(f)
        manifest['problem_facade'] = run_problem_facade()
        with open(manifest_path, 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    names = sys.argv[1:] or (SMALL + list(LARGE))
    manifest_path = os.path.join(OUT, 'MANIFEST.json')
    manifest = {}
    if os.path.exists(manifest_path):
        with open(manifest_path) as f:
            manifest = json.load(f)
    for name in names:
        meta = run(name)
        manifest[name] = meta
        print(name, {k: meta[k] for k in ('N', 'M', 'n', 'q', 'r', 's', 'o',
                                          'C', 'nnz', 'wall_s')})
        with open(manifest_path, 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
