"""The instruction tape (``opty_amd.codegen.tape``) and the device kernel that
executes it (``opty_hip_tape_run``): the referee that code objects at the
register limit are held to before a handle exists
(``ConstraintCollocator._verify_build``)."""
import numpy as np
import pytest

import dag_interp
import tape_host
from examples import problems
from opty_amd.codegen import ir
from opty_amd.codegen.tape import Tape, TAPE_WORDS, T_UNARY0

NAMES = ['elementary_be_small', 'piecewise_mid_small', 'c99_be_small',
         'pend3_link_midpoint_small', 'vardur_pendulum_small',
         'implicit_traj_be_small', 'one_legged_small']


def _program_and_inputs(name, nodes=37, seed=3):
    import opty_amd
    col = opty_amd.ConstraintCollocator(**problems.build(name))
    prog = col._build_program()
    rng = np.random.default_rng(seed)
    cache = {}

    def inputs(kind, idx):
        key = (kind, idx)
        if key not in cache:
            cache[key] = rng.uniform(0.2, 0.9, nodes) \
                if kind in ('cur', 'adj') else float(rng.uniform(0.2, 0.9))
        return cache[key]
    return prog, inputs


@pytest.mark.parametrize('name', NAMES)
def test_tape_semantics_equal_the_dag_interpreter(name):
    """Encoding: running the tape (NumPy semantics) gives what the DAG
    interpreter gives for the same roots, bit for bit."""
    prog, inputs = _program_and_inputs(name)
    roots = list(prog.con_out) + list(prog.jac_out)
    tape = Tape(prog.dag, roots)
    assert tape.code.dtype == np.int32 and tape.code.shape[1] == TAPE_WORDS
    assert tape.code[:, 1:6].max() < tape.nslots
    vals = tape_host.run_on_host(tape, tape.table(37, inputs))
    want = dag_interp.evaluate(prog.dag, roots, inputs)
    for r, w in zip(roots, want):
        np.testing.assert_array_equal(vals[tape.slot[r]],
                                      np.broadcast_to(w, (37,)))


def test_every_operation_of_the_ir_has_an_opcode():
    dag = ir.DAG()
    x, y = dag.input('cur', 0), dag.input('cur', 1)
    roots = [dag.unary(u, x) for u in ir.UNARY]
    roots += [dag.add(x, y), dag.sub(x, y), dag.mul(x, y), dag.div(x, y),
              dag.neg(x), dag.powi(x, 5), dag.pow(x, y),
              dag.binary(ir.MAX, x, y), dag.binary(ir.MIN, x, y),
              dag.binary(ir.ATAN2, x, y)]
    roots += [dag.select(rel, x, y, x, y) for rel in ir.RELATIONS]
    tape = Tape(dag, roots)
    ops = set(tape.code[:, 0].tolist())
    assert {T_UNARY0 + k for k in range(len(ir.UNARY))} <= ops
    assert set(range(11)) <= ops


def _all_ops_tape(nodes):
    dag = ir.DAG()
    x, y = dag.input('cur', 0), dag.input('cur', 1)
    big = dag.add(x, dag.const(1.5))            # > 1: acosh, tgamma
    roots = []
    for u in ir.UNARY:
        roots.append(dag.unary(u, big if u == 'acosh' else x))
    roots += [dag.add(x, y), dag.sub(x, y), dag.mul(x, y), dag.div(x, y),
              dag.neg(x), dag.pow(x, y), dag.binary(ir.MAX, x, y),
              dag.binary(ir.MIN, x, y), dag.binary(ir.ATAN2, x, y)]
    roots += [dag.powi(x, n) for n in (2, 3, 4, 5, 6, 7, 11, 64)]
    roots += [dag.select(rel, x, y, dag.mul(x, x), dag.neg(y))
              for rel in ir.RELATIONS]
    rng = np.random.default_rng(5)
    data = {0: rng.uniform(0.05, 0.95, nodes), 1: rng.uniform(0.05, 0.95,
                                                              nodes)}
    data[1][::7] = data[0][::7]                 # eq / ne / le see ties
    return dag, roots, lambda kind, idx: data[idx]


@pytest.mark.gpu
def test_device_tape_kernel_every_operation():
    """Every opcode of the device kernel against the NumPy semantics (the
    device math library is within a few ulp of the host's)."""
    from opty_amd import hip_backend as hb
    nodes = 131
    dag, roots, inputs = _all_ops_tape(nodes)
    tape = Tape(dag, roots)
    want = tape_host.run_on_host(tape, tape.table(nodes, inputs))
    got = hb.tape_run(tape, tape.table(nodes, inputs))
    for r in roots:
        np.testing.assert_allclose(got[tape.slot[r]], want[tape.slot[r]],
                                   rtol=1e-12, atol=1e-15,
                                   err_msg=str((dag.op[r], dag.args[r])))


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_device_tape_kernel_on_the_zoo(name):
    from opty_amd import hip_backend as hb
    prog, inputs = _program_and_inputs(name, nodes=131)
    roots = list(prog.con_out) + list(prog.jac_out)
    tape = Tape(prog.dag, roots)
    want = tape_host.run_on_host(tape, tape.table(131, inputs))
    got = hb.tape_run(tape, tape.table(131, inputs))
    scale = max(float(np.abs(want[tape.slot[r]]).max()) for r in roots)
    for r in roots:
        np.testing.assert_allclose(got[tape.slot[r]], want[tape.slot[r]],
                                   rtol=1e-10, atol=1e-12*scale)


@pytest.mark.gpu
def test_tape_run_rejects_a_malformed_tape():
    from opty_amd import hip_backend as hb
    dag, roots, inputs = _all_ops_tape(8)
    tape = Tape(dag, roots)
    vals = tape.table(8, inputs)
    for column, value in ((0, 99), (2, tape.nslots), (1, -1)):
        bad = Tape(dag, roots)
        bad.code[3, column] = value
        with pytest.raises(hb.HipBackendError, match='instruction 3'):
            hb.tape_run(bad, vals.copy())


def test_row_error_treats_shared_non_finite_values_as_agreement():
    from opty_amd import ConstraintCollocator
    err = ConstraintCollocator._row_error
    row = np.array([0, 0, 1, 1])
    want = np.array([1.0, np.nan, np.inf, 2.0])
    assert err(want.copy(), want, row) == 0.0
    assert err(np.array([1.0, np.nan, np.inf, 2.0 + 2e-9]), want, row) == \
        pytest.approx(1e-9)
    assert err(np.array([1.0, 0.0, np.inf, 2.0]), want, row) == np.inf
    assert err(np.array([1.0, np.nan, -np.inf, 2.0]), want, row) == np.inf


@pytest.mark.gpu
def test_verification_moves_to_positive_inputs_when_needed(monkeypatch,
                                                           tmp_path):
    """Equations that are not finite on the seeded inputs from (-1, 1)
    (square roots / logarithms of states) are verified on a positive
    range."""
    import sympy as sm
    import sympy.physics.mechanics as me
    import opty_amd
    monkeypatch.setenv('OPTY_CROSS_CHECK', 'all')
    t = me.dynamicsymbols._t
    x, v, f = me.dynamicsymbols('x v f')
    eom = sm.Matrix([x.diff(t) - v,
                     v.diff(t) + sm.sqrt(x) + sm.log(x)*v - f])
    col = opty_amd.ConstraintCollocator(eom, (x, v), 50, 0.01,
                                        tmp_dir=str(tmp_path))
    assert col.hip is not None
    verdict = col._build_verdict
    assert verdict['ok'] and verdict['span'] == [0.1, 0.9], verdict
