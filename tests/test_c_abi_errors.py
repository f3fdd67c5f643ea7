"""GPU tests of the C ABI's error behaviour: every misuse returns a non-zero
status with a message (raised as ``HipBackendError``) -- nothing is silently
ignored, nothing falls back to a CPU path."""
import numpy as np
import pytest

from examples import problems

pytestmark = pytest.mark.gpu


def _fresh(name='config1_vyasarayani', **kw):
    """An un-configured handle for ``name`` (no parameters / interval set)."""
    import opty_amd
    from opty_amd import hip_backend as hb
    col = opty_amd.ConstraintCollocator(**kw, **problems.build(name))
    source, meta = col.generate_source()
    hsaco = hb.compile_module(source)
    return col, hb.HipProblem(col._descriptor(meta), hsaco), hb


def test_evaluation_before_configuration_fails():
    col, hip, hb = _fresh('config2_pendulum_small')
    free = problems.make_free(col.num_free)
    con = np.empty(col.num_constraints)
    with pytest.raises(hb.HipBackendError, match='parameter'):
        hip.eval_con(free, con, hb.HOST)
    hip.set_known_parameters([1.0, 1.0, 1.0, 9.81])
    with pytest.raises(hb.HipBackendError):
        hip.eval_con(free, con, hb.HOST)          # interval / instance map
    hip.close()


def test_wrong_counts_and_kinds_fail():
    col, hip, hb = _fresh('config2_pendulum_small')
    with pytest.raises(hb.HipBackendError, match='expected'):
        hip.set_known_parameters([1.0, 2.0])
    with pytest.raises(hb.HipBackendError):
        hip.set_interval(0.0)
    rows = np.empty(hip.nnz, dtype=np.int64)
    cols = np.empty(hip.nnz, dtype=np.int64)
    with pytest.raises(hb.HipBackendError, match='instance'):
        hip.jacobian_indices(rows, cols, hb.HOST)
    with pytest.raises(hb.HipBackendError, match='memory kind'):
        col.hip.eval_con(problems.make_free(col.num_free),
                         np.empty(col.num_constraints), 7)
    hip.close()


def test_bad_descriptor_and_code_object_fail():
    import opty_amd
    from opty_amd import hip_backend as hb
    col = opty_amd.ConstraintCollocator(
        **problems.build('config1_vyasarayani'))
    source, meta = col.generate_source()
    hsaco = hb.compile_module(source)
    good = col._descriptor(meta)
    with pytest.raises(hb.HipBackendError, match='device'):
        hb.HipProblem(dict(good, device=99), hsaco)
    with pytest.raises(hb.HipBackendError, match='layout'):
        hb.HipProblem(dict(good, layout=5), hsaco)
    with pytest.raises(hb.HipBackendError):
        hb.HipProblem(dict(good, N=1), hsaco)
    with pytest.raises(hb.HipBackendError):
        hb.HipProblem(good, '/nonexistent/module.hsaco')
    # a code object of another problem lacks nothing by name but a pruned
    # pattern that was never supplied is refused
    pruned = opty_amd.ConstraintCollocator(
        prune_zeros=True, **problems.build('config3_10link_small'))
    src2, meta2 = pruned.generate_source()
    hip = hb.HipProblem(pruned._descriptor(meta2), hb.compile_module(src2))
    rows = np.empty(hip.nnz, dtype=np.int64)
    with pytest.raises(hb.HipBackendError, match='pattern'):
        hip.jacobian_indices(rows, rows.copy(), hb.HOST)
    hip.close()


def test_csr_layout_is_not_node_sharded():
    col, hip, hb = _fresh('config3_10link_small', jacobian_layout='csr')
    hip.set_block_pattern(col._build_program().pattern)
    rows = np.empty(hip.nnz, dtype=np.int64)
    with pytest.raises(hb.HipBackendError, match='shard'):
        hip.jacobian_indices_shard(1000, 5, rows, rows.copy(), hb.HOST)
    hip.close()


def test_handles_release_their_device_memory():
    """Creating, using and destroying handles in a loop returns the device
    memory (staging buffers, tables, module) every time."""
    import torch
    import opty_amd
    kw = problems.build('config3_10link_small')
    col = opty_amd.ConstraintCollocator(**kw)
    source, meta = col.generate_source()
    from opty_amd import hip_backend as hb
    hsaco = hb.compile_module(source)
    free = problems.make_free(col.num_free)
    con = np.empty(col.num_constraints)

    def cycle():
        hip = hb.HipProblem(col._descriptor(meta), hsaco)
        hip.set_known_parameters([float(col.known_parameter_map[p])
                                  for p in col.known_parameters])
        hip.set_interval(col.node_time_interval)
        jac = np.empty(hip.nnz)
        hip.eval_con_jac(free, con, jac, hb.HOST)
        hip.close()

    cycle()
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    after = torch.cuda.mem_get_info()[0]
    assert before - after < 8 << 20, (before, after)


def test_matrix_function_misuse_fails():
    """``opty_hip_matrix_*``: bad descriptors, wrong code objects, empty
    evaluations and null arguments return a status, never garbage."""
    import sympy as sm
    from opty_amd import hip_backend as hb, ufuncify_matrix
    x, y = sm.symbols('x, y')
    f = ufuncify_matrix((x, y), sm.Matrix([[x*y, sm.sin(x)]]), const=(y,))
    good = dict(f.hip.desc)
    lib = hb.load_library()
    # a collocation module has no matrix geometry problem, but a matrix module
    # lacks opty_con / opty_conjac: creating a *problem* from it must fail
    import opty_amd
    col = opty_amd.ConstraintCollocator(
        **problems.build('config1_vyasarayani'))
    src, meta = col.generate_source()
    hsaco_matrix = hb.compile_module(f.source)
    with pytest.raises(hb.HipBackendError, match='missing'):
        hb.HipProblem(col._descriptor(meta), hsaco_matrix)
    for bad in (dict(good, rows=0), dict(good, wgs_per_block=0),
                dict(good, device=99), dict(good, num_vec=-1)):
        with pytest.raises(hb.HipBackendError):
            hb.HipMatrix(bad, hsaco_matrix)
    with pytest.raises(hb.HipBackendError, match='hipModuleLoad'):
        hb.HipMatrix(good, '/nonexistent.hsaco')
    res = np.empty((4, 2))
    xs = np.linspace(0.0, 1.0, 4)
    with pytest.raises(hb.HipBackendError, match='at least one'):
        f.hip.evaluate(res, [xs], [2.0], 0, hb.HOST)
    with pytest.raises(hb.HipBackendError, match='memory kind'):
        f.hip.evaluate(res, [xs], [2.0], 4, 5)
    with pytest.raises(hb.HipBackendError, match='null'):
        f.hip.evaluate(res, [None], [2.0], 4, hb.HOST)
    assert lib.opty_hip_matrix_eval(None, None, None, None, 4, 0) != 0
    # and the handle still works afterwards
    out = f(res, xs, 2.0)
    np.testing.assert_allclose(out[:, 0, 0], 2.0*xs, rtol=1e-15)
    np.testing.assert_allclose(out[:, 0, 1], np.sin(xs), rtol=1e-14)


def test_host_register_misuse_fails():
    from opty_amd import hip_backend as hb
    lib = hb.load_library()
    assert lib.opty_hip_host_register(None, 8) != 0
    a = np.zeros(1024)
    assert lib.opty_hip_host_register(a.ctypes.data, 0) != 0
    hb.host_register(a)
    hb.host_unregister(a)
    assert lib.opty_hip_host_unregister(None) == 0


def test_round3_entry_points_reject_misuse():
    """``opty_hip_eval_instance``, ``opty_hip_set_varying_entries``,
    ``opty_hip_eval_jac_persistent``, ``opty_hip_shard_jac_to_host``."""
    import torch
    col, hip, hb = _fresh('config2_pendulum_small')
    P = hip.desc['P']
    with pytest.raises(hb.HipBackendError, match='ascend'):
        hip.set_varying_entries([3, 2])
    with pytest.raises(hb.HipBackendError, match='ascend'):
        hip.set_varying_entries([0, P])
    with pytest.raises(hb.HipBackendError, match='ascend'):
        hip.set_varying_entries([1, 1])
    hip.set_varying_entries([])                     # a block of constants
    hip.set_varying_entries([0, 5])
    # opty_hip_set_entry_copies: destinations ascend, are not moved
    # themselves, and copy from an entry that is
    with pytest.raises(hb.HipBackendError, match='not a varying entry'):
        hip.set_entry_copies([(3, 2)])
    with pytest.raises(hb.HipBackendError, match='cannot be a copy'):
        hip.set_entry_copies([(5, 0)])
    with pytest.raises(hb.HipBackendError, match='ascend'):
        hip.set_entry_copies([(4, 0), (3, 5)])
    with pytest.raises(hb.HipBackendError, match='ascend'):
        hip.set_entry_copies([(P, 0)])
    hip.set_entry_copies([(3, 0), (7, 5)])
    hip.set_entry_copies([])
    free = problems.make_free(col.num_free)
    jac = hb.pinned_empty(hip.nnz)
    with pytest.raises(hb.HipBackendError, match='parameter'):
        hip.eval_jac_persistent(free, jac, True)    # not configured yet
    with pytest.raises(hb.HipBackendError, match='null'):
        hip.eval_jac_persistent(None, jac, True)
    dfree = torch.from_numpy(free).cuda()
    tail = torch.empty(4, dtype=torch.float64, device='cuda')
    with pytest.raises(hb.HipBackendError, match='parameter'):
        hip.eval_instance(dfree, tail, None)
    with pytest.raises(hb.HipBackendError, match='null'):
        hip.eval_instance(None, tail, None)
    djac = torch.empty(10*P, dtype=torch.float64, device='cuda')
    ncn = col.num_collocation_nodes - 1
    with pytest.raises(hb.HipBackendError, match='outside'):
        hip.shard_jac_to_host(djac, jac, ncn - 5, ncn + 5, True)
    with pytest.raises(hb.HipBackendError, match='null'):
        hip.shard_jac_to_host(djac, None, 0, 10, True)
    hip.close()
    # problems without instance constraints: eval_instance is a no-op
    col2, hip2, _ = _fresh('config1_vyasarayani')
    hip2.eval_instance(torch.zeros(col2.num_free, dtype=torch.float64,
                                   device='cuda'), None, None)
    hip2.close()
    # the row-sorted layout has no varying-entry table
    col3, hip3, _ = _fresh('msd_be_small', jacobian_layout='csr')
    with pytest.raises(hb.HipBackendError, match='node-major'):
        hip3.set_varying_entries([0])
    hip3.close()
    with pytest.raises(hb.HipBackendError):
        hb.set_host_threads(-1)


def test_tune_launch_records_a_plan_and_keeps_the_results(tmp_path,
                                                          monkeypatch):
    """``ConstraintCollocator.tune_launch``: the candidate geometries of a
    small problem are built, timed and a well-formed plan entry recorded; the
    collocator rebuilt from the plan returns the same values."""
    import json
    import opty_amd
    from opty_amd import launch_plan as lp
    path = tmp_path/'plans.json'
    monkeypatch.setenv('OPTY_LAUNCH_PLANS', str(path))
    factory, fkw = problems.CONFIGS['pend3_link_midpoint_small']
    kw = factory(**dict(fkw, num_nodes=20001))
    col = opty_amd.ConstraintCollocator(**kw)
    free = problems.make_free(col.num_free, seed=5)
    before = col.generate_jacobian_function()(free).copy()
    entry = col.tune_launch(iters=10, rounds=2)
    plans = json.loads(path.read_text())
    key = lp.key_of(col._build_program(), (20000 + 63)//64)
    assert plans[key]['options'] == entry['options']
    assert set(entry['measured_ms']) == {'fused', 'jac', 'con'}
    assert isinstance(entry['fused_pays'], bool)
    assert all(v > 0 for v in entry['measured_ms']['fused'].values())
    assert str(entry['seed']['fused']) in entry['measured_ms']['fused']
    opts = lp.lookup(col._build_program(), (20000 + 63)//64)
    assert opts is not None
    after = col.generate_jacobian_function()(free)
    np.testing.assert_allclose(after, before, rtol=1e-12, atol=1e-9)
