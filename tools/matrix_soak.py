#!/usr/bin/env python
"""Developer tool: ``opty_amd.ufuncify_matrix`` (the reference's plugin call
shape) for random matrices of random expressions -- shapes up to 7 x 9, up to
8 arguments some of them constants, with and without ``cse()`` -- at argument
counts around the 64-lane wave, against ``sympy.lambdify``.

    python tools/matrix_soak.py build 60    # CPU container: compile
    python tools/matrix_soak.py run 60      # GPU box
"""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import sympy as sm
import opty_amd
import random_problems as rp
import oracle_bounds


def case(seed):
    rng = np.random.default_rng(1000 + seed)
    nargs = int(rng.integers(1, 9))
    syms = list(sm.symbols('x0:%d' % nargs, real=True))
    nconst = int(rng.integers(0, min(3, nargs)))
    const = tuple(syms[nargs - nconst:])
    vec = syms[:nargs - nconst]
    rows, cols = int(rng.integers(1, 8)), int(rng.integers(1, 10))
    atoms = vec or syms
    mat = sm.Matrix(rows, cols, lambda i, j: rp._expr(rng, atoms,
                                                     list(const)))
    use_cse = bool(rng.integers(0, 2))
    return syms, const, mat, use_cse, rng


def main():
    mode, count = sys.argv[1], int(sys.argv[2])
    bad = 0
    worst = 0.0
    for seed in range(count):
        syms, const, mat, use_cse, rng = case(seed)
        expr = sm.cse(mat) if use_cse else mat
        f = opty_amd.ufuncify_matrix(syms, expr, const=const)
        if mode == 'build':
            continue
        ref = sm.lambdify(syms, mat, oracle_bounds._MODULES)
        for n in (1, 63, 64, 65, 1000):
            vals = [float(rng.uniform(-1, 1)) if s in const
                    else rng.uniform(-1, 1, n) for s in syms]
            out = np.full((n, mat.shape[0]*mat.shape[1]), np.nan)
            got = f(out, *vals)
            assert got.shape == (n,) + mat.shape
            want = np.empty_like(got)
            for i in range(n):
                want[i] = np.array(ref(*[v if np.isscalar(v) else v[i]
                                         for v in vals]), dtype=float)
            # a random composition may leave a function's domain (negative
            # base of a fractional power): NaN where the reference has NaN
            fin = np.isfinite(want)
            same_domain = np.array_equal(fin, np.isfinite(got))
            scale = np.maximum(np.abs(np.where(fin, want, 0.0)).max(axis=0),
                               1.0)
            err = float((np.abs(np.where(fin, got - want, 0.0))/scale).max())
            worst = max(worst, err)
            if not (err <= 1e-10 and same_domain):
                bad += 1
                print('MISMATCH seed', seed, 'n', n, 'err', err, flush=True)
    if mode == 'run':
        print('matrix soak: %d random matrix programs x 5 argument counts, '
              'worst error %.2e, %d mismatches' % (count, worst, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
