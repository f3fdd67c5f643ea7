"""Host-path byte levers that are pure host logic: which varying block
entries need not cross PCIe (``codegen.program.scaled_copies``), and the
host-side evaluation of node-invariant values (``codegen.evaluate``)."""
import math

import numpy as np
import pytest

import dag_interp
from examples import problems
from opty_amd import ConstraintCollocator
from opty_amd.codegen import ir
from opty_amd.codegen.evaluate import evaluate_uniform
from opty_amd.codegen.program import (chain_value, scaled_copies,
                                      varying_copies, varying_entries)


@pytest.mark.parametrize('name,moved', [
    ('config3_10link_small', (275, 269)), ('biped_small', (159, 156)),
    ('one_legged_small', (60, 59)), ('pend3_link_midpoint_small', (36, 35)),
    ('msd_be_small', None)])
def test_scaled_copies_reconstruct_their_entries(name, moved):
    """Every copy ``(dst, src, num, den)`` satisfies ``dst = prod(num) /
    prod(den) * src`` at random node values (the DAG through the test
    interpreter), the moved entries plus the copies are exactly the varying
    entries, and the moved count is the documented one."""
    col = ConstraintCollocator(**problems.build(name))
    prog = col._build_program()
    d = prog.dag
    unique0, copies0 = varying_copies(prog)
    unique, copies = scaled_copies(prog)
    assert sorted(unique + [c[0] for c in copies]) == varying_entries(prog)
    assert set(c[1] for c in copies) <= set(unique)
    assert len(unique) <= len(unique0)
    if moved is not None:
        assert (len(unique0), len(unique)) == moved
    rng = np.random.default_rng(3)
    K = 40
    cache = {}

    def inputs(kind, idx):
        key = (kind, idx)
        if key not in cache:
            cache[key] = rng.uniform(0.5, 1.5, K) if kind in ('cur', 'adj') \
                else (0.01 if kind == 'h' else rng.uniform(0.8, 1.3))
        return cache[key]
    nodes = sorted({st[1] for c in copies for ch in (c[2], c[3])
                    for st in ch if len(st) > 1})
    roots = [prog.jac_out[e] for c in copies for e in c[:2]] + nodes
    vals = dag_interp.evaluate(d, roots, inputs)
    byroot = dict(zip(roots, vals))
    fac = {n: float(np.asarray(byroot[n]).reshape(-1)[0]) for n in nodes}
    for dst, src, num, den in copies:
        scale = 1.0 if num == den else \
            chain_value(num, fac)/chain_value(den, fac)
        want = np.broadcast_to(byroot[prog.jac_out[dst]], (K,))
        got = scale*np.broadcast_to(byroot[prog.jac_out[src]], (K,))
        np.testing.assert_allclose(got, want, rtol=4e-16*(2 + len(num) +
                                                           len(den)),
                                   atol=0.0)
    # the collocator's own factors (known parameter map, fixed interval)
    if copies and not col._variable_duration:
        scales = col._copy_scales(copies)
        assert scales is not None and len(scales) == len(copies)
        assert all(s == 1.0 for s, c in zip(scales, copies)
                   if c[2] == c[3])


def test_uniform_evaluation_follows_ieee_at_the_edges():
    """Where Python raises, the host evaluation returns what C's libm and
    the device return: signed infinities at poles and overflows, NaN for
    domain errors."""
    d = ir.DAG()
    zero, big, mone = (d.input('par', k) for k in range(3))
    cases = {
        'log0': (d.unary('log', zero), -math.inf),
        'log_neg': (d.unary('log', big), math.nan),
        'sinh_neg': (d.unary('sinh', big), -math.inf),
        'exp_over': (d.unary('exp', d.neg(big)), math.inf),
        'pow0_m1.5': (d.pow(zero, d.const(-1.5)), math.inf),
        'pow_neg_odd': (d._node(ir.POW, (big, d.const(301.0))), -math.inf),
        'pow_neg_frac': (d._node(ir.POW, (big, d.const(0.3))), math.nan),
        'atanh_m1': (d.unary('atanh', mone), -math.inf),
        'sqrt_neg': (d.unary('sqrt', big), math.nan),
        'div0': (d.div(big, zero), -math.inf),
    }
    vals = evaluate_uniform(d, [n for n, _ in cases.values()],
                            lambda kind, k: (0.0, -1000.0, -1.0)[k])
    for label, (node, want) in cases.items():
        got = vals[node]
        assert (math.isnan(got) if math.isnan(want) else got == want), \
            (label, got, want)
