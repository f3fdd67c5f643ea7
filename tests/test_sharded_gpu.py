"""BASELINE config 4 on the GPU: node shards evaluated by ``opty_hip_eval_shard``
from the global free vector, re-assembled by the point-to-point gather-v, and
held to the reference's golden vectors of config 3 (sampled nodes +
checksums).  A 1-GPU box runs the ranks oversubscribed on ``cuda:0`` with a
``gloo`` rendezvous (RCCL refuses duplicate devices); the evaluation, the
partition and the exchange logic are the ones ``bench.py --gpus N`` runs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import golden_util as gu
from examples import problems

pytestmark = pytest.mark.gpu

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
RTOL = 1e-10


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('name', ['pend3_link_midpoint_small',
                                  'pend2_link_vardur_unkmass_small'])
def test_ranges_written_in_place_equal_the_whole_evaluation(name):
    """Three unequal node ranges, each written straight into the global
    equation-major / node-major vectors (con_stride = N - 1), give bit for bit
    what one whole-problem launch of the same handle gives; a dense
    (M x nodes) block of one range equals the same slice.  Midpoint with
    fixed parameters, and backward Euler with unknown parameters + variable
    duration (the node-invariant table is then refilled from the tail of the
    global ``free`` before every launch)."""
    import torch
    import opty_amd
    from opty_amd import hip_backend as hb
    factory, fkw = problems.CONFIGS[name]
    kw = factory(**dict(fkw, num_nodes=301))
    col = opty_amd.ConstraintCollocator(**kw)
    hip = col.hip
    dev = torch.device('cuda:0')
    hip.use_torch_stream()
    prog = col._build_program()
    M, P, ncn = prog.M, prog.P, col.num_collocation_nodes - 1
    free = torch.from_numpy(problems.make_free(
        col.num_free, seed=4,
        variable_duration=col._variable_duration)).to(dev)
    con = torch.empty(M*ncn, dtype=torch.float64, device=dev)
    jac = torch.empty(P*ncn, dtype=torch.float64, device=dev)
    hip.eval_con_jac(free, con, jac, hb.DEVICE)
    con2 = torch.full_like(con, np.nan)
    jac2 = torch.full_like(jac, np.nan)
    for a, b in ((0, 100), (100, 237), (237, 300)):
        hip.eval_shard(hb.EVAL_FUSED, free, con2[a:], ncn, jac2[a*P:], a, b)
    torch.cuda.synchronize()
    assert torch.equal(con, con2) and torch.equal(jac, jac2)
    # separate kernels, dense local block
    a, b = 64, 191
    blk = torch.full((M, b - a), np.nan, dtype=torch.float64, device=dev)
    jl = torch.full(((b - a)*P,), np.nan, dtype=torch.float64, device=dev)
    hip.eval_shard(hb.EVAL_PAIR, free, blk, b - a, jl, a, b)
    c_sep = torch.empty_like(con)
    j_sep = torch.empty_like(jac)
    hip.eval_con(free, c_sep, hb.DEVICE)
    hip.eval_jac(free, j_sep, hb.DEVICE)
    torch.cuda.synchronize()
    assert torch.equal(blk, c_sep.view(M, ncn)[:, a:b])
    assert torch.equal(jl, j_sep[a*P:b*P])
    # indices of a range == that slice of the global enumeration
    rows, cols = col.jacobian_indices()
    r = np.empty((b - a)*P, dtype=np.int64)
    c = np.empty_like(r)
    hip.jacobian_indices_range(a, b, r, c, hb.HOST)
    np.testing.assert_array_equal(r, rows[a*P:b*P])
    np.testing.assert_array_equal(c, cols[a*P:b*P])
    # misuse
    with pytest.raises(hb.HipBackendError):
        hip.eval_shard(hb.EVAL_FUSED, free, con2, ncn, jac2, 10, ncn + 1)
    with pytest.raises(hb.HipBackendError):
        hip.eval_shard(hb.EVAL_FUSED, free, con2, 5, jac2, 0, 10)
    hip.set_stream(None)


@pytest.mark.parametrize('rank', [0, 3, 7])
def test_one_of_eight_shards_against_the_reference(rank):
    """A 12 500-node shard of config 4 (world size 8) runs the small-launch
    geometry -- two waves per SIMD, 16-entry chunks, 4-wave workgroups
    (``emit_hip._dual_occupancy_cut``): values at the reference's sampled
    nodes that fall into the shard, indices of the shard."""
    import torch
    from opty_amd.sharded import ShardedCollocator
    meta, z = gu.load('config3_10link')
    N, M, C = meta['N'], meta['M'], meta['C']
    P = M*C
    sh = ShardedCollocator(rank=rank, world_size=8,
                           **problems.build('config3_10link'))
    km = sh.collocator.hip.desc
    assert km['fused_waves_per_wg'] == 4       # the small-launch geometry
    free = problems.make_free(sh.collocator.num_free, seed=meta['seed'])
    con, jac = sh.evaluate(torch.from_numpy(free).cuda())
    torch.cuda.synchronize()
    con, jac = con.cpu().numpy(), jac.cpu().numpy().reshape(-1, P)
    nodes = z['nodes']
    pick = (nodes >= sh.a) & (nodes < sh.b)
    assert pick.sum() >= 2
    sel = nodes[pick]
    cbn, jbn, _, _ = gu.error_bounds(sh.collocator, free, sel)
    gu.assert_close(jac[sel - sh.a], z['jac_nodes'][pick], RTOL,
                    what='1/8 shard jac nodes', bound=jbn)
    gu.assert_close(con[:, sel - sh.a], z['con_nodes'][:, pick], RTOL,
                    what='1/8 shard con nodes', bound=cbn)
    # every value of the shard was written (no NaN / stale zero blocks)
    assert np.isfinite(jac).all() and np.abs(jac).sum(axis=1).min() > 0
    rows, cols = sh.jacobian_indices_local()
    np.testing.assert_array_equal(
        rows.reshape(-1, P)[sel - sh.a], z['rows_nodes'][pick])
    np.testing.assert_array_equal(
        cols.reshape(-1, P)[sel - sh.a], z['cols_nodes'][pick])


@pytest.mark.parametrize('name', ['config2_pendulum_small',
                                  'gaitlike_3link_be_small',
                                  'gaitlike_3link_mid_small',
                                  'chaplygin_mid_small', 'delay_be_small'])
def test_instance_tails_next_to_node_ranges(name):
    """A problem WITH instance constraints evaluated as node ranges
    (``opty_hip_eval_shard``: collocation part only) plus
    ``opty_hip_eval_instance`` for the tails gives bit for bit what the
    whole-problem launch gives, and the reference's golden values."""
    import torch
    import opty_amd
    from opty_amd import hip_backend as hb
    meta, z = gu.load(name)
    col = opty_amd.ConstraintCollocator(**problems.build(name))
    hip = col.hip
    dev = torch.device('cuda:0')
    hip.use_torch_stream()
    M, P, ncn = meta['M'], meta['M']*meta['C'], meta['N'] - 1
    free = torch.from_numpy(z['free']).to(dev)
    con = torch.empty(col.num_constraints, dtype=torch.float64, device=dev)
    jac = torch.empty(hip.nnz, dtype=torch.float64, device=dev)
    hip.eval_con_jac(free, con, jac, hb.DEVICE)
    con2 = torch.full_like(con, np.nan)
    jac2 = torch.full_like(jac, np.nan)
    cuts = [0, ncn//3, ncn//3 + 1, ncn]
    for a, b in zip(cuts[:-1], cuts[1:]):
        hip.eval_shard(hb.EVAL_FUSED, free, con2[a:], ncn, jac2[a*P:], a, b)
    assert torch.isnan(con2[M*ncn:]).all() and torch.isnan(jac2[P*ncn:]).all()
    hip.eval_instance(free, con2[M*ncn:], jac2[P*ncn:])
    torch.cuda.synchronize()
    assert torch.equal(con, con2) and torch.equal(jac, jac2)
    # either tail alone
    con3 = torch.full((meta['o'],), np.nan, dtype=torch.float64, device=dev)
    hip.eval_instance(free, con3, None)
    jac3 = torch.full((meta['nnz_inst'],), np.nan, dtype=torch.float64,
                      device=dev)
    hip.eval_instance(free, None, jac3)
    torch.cuda.synchronize()
    assert torch.equal(con3, con[M*ncn:]) and torch.equal(jac3, jac[P*ncn:])
    cb, jb = gu.error_bounds(col, z['free'])
    gu.assert_close(con2.cpu().numpy(), z['con'], RTOL,
                    what=name + ' ranges+tails con', bound=cb)
    gu.assert_close(jac2.cpu().numpy(), z['jac'], RTOL,
                    what=name + ' ranges+tails jac', bound=jb)
    hip.set_stream(None)


@pytest.mark.parametrize('name,rank', [('config5_standin_24link', 0),
                                       ('config5_standin_24link', 7),
                                       ('config5_gaitlike_24link', 0),
                                       ('config5_gaitlike_24link', 4),
                                       ('config5_gaitlike_24link', 7)])
def test_one_of_eight_shards_of_the_config5_stand_ins(name, rank):
    """A 6 250-node shard (world size 8) of the 50-state, variable-duration
    stand-ins at N = 50 000 against the reference's sampled nodes; for the
    gait-like one (known trajectory, exp terms, 12 instance constraints) also
    the instance tails, which the assembling rank evaluates from the global
    free vector, and the int64 indices of shard and tail."""
    import torch
    from opty_amd.sharded import ShardedCollocator
    meta, z = gu.load(name)
    N, M, C = meta['N'], meta['M'], meta['C']
    P = M*C
    sh = ShardedCollocator(rank=rank, world_size=8, **problems.build(name))
    assert (sh.o, sh.nnz_inst) == (meta['o'], meta['nnz_inst'])
    free = problems.make_free(sh.collocator.num_free, seed=meta['seed'],
                              variable_duration=True)
    dfree = torch.from_numpy(free).cuda()
    con, jac = sh.evaluate(dfree)
    ic, ij = sh.evaluate_instance()
    torch.cuda.synchronize()
    con, jac = con.cpu().numpy(), jac.cpu().numpy().reshape(-1, P)
    nodes = z['nodes']
    pick = (nodes >= sh.a) & (nodes < sh.b)
    assert pick.sum() >= 1
    sel = nodes[pick]
    cbn, jbn, icb, ijb = gu.error_bounds(sh.collocator, free, sel)
    gu.assert_close(jac[sel - sh.a], z['jac_nodes'][pick], RTOL,
                    what=name + ' 1/8 shard jac nodes', bound=jbn)
    gu.assert_close(con[:, sel - sh.a], z['con_nodes'][:, pick], RTOL,
                    what=name + ' 1/8 shard con nodes', bound=cbn)
    assert np.isfinite(jac).all() and np.abs(jac).sum(axis=1).min() > 0
    rows, cols = sh.jacobian_indices_local()
    np.testing.assert_array_equal(
        rows.reshape(-1, P)[sel - sh.a], z['rows_nodes'][pick])
    np.testing.assert_array_equal(
        cols.reshape(-1, P)[sel - sh.a], z['cols_nodes'][pick])
    if meta['o']:
        gu.assert_close(ic.cpu().numpy(), z['con_tail'], RTOL,
                        what=name + ' shard con tail', bound=icb)
        gu.assert_close(ij.cpu().numpy(), z['jac_tail'], RTOL,
                        what=name + ' shard jac tail', bound=ijb)
        irows, icols = sh.instance_indices()
        np.testing.assert_array_equal(irows, z['rows_tail'])
        np.testing.assert_array_equal(icols, z['cols_tail'])


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from opty_amd.sharded import ShardedCollocator, SharedHostVector
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        kw = problems.build('config3_10link')
        sh = ShardedCollocator(device=dev, **kw)
        sh.collocator.hip.use_torch_stream()
        meta, _ = gu.load('config3_10link')
        free = torch.from_numpy(problems.make_free(
            sh.collocator.num_free, seed=meta['seed'])).to(dev)
        if rank != 0:
            free.zero_()
        # gloo broadcasts host memory; on a multi-GPU node this is RCCL
        host = free.cpu()
        dist.broadcast(host, 0)
        free.copy_(host)
        sh.evaluate(free, in_place=(rank == 0))
        got = sh.gather(0)
        ncn = sh.N - 1
        con_host = SharedHostVector('opty_t_con_%d' % port, sh.M*ncn, rank)
        jac_host = SharedHostVector('opty_t_jac_%d' % port, sh.P*ncn, rank)
        sh.to_host(con_host, jac_host)
        torch.cuda.synchronize()
        dist.barrier()
        rows, cols = sh.jacobian_indices_local()
        if rank == 0:
            con, jac = got
            np.savez(out, con=con.cpu().numpy(), jac=jac.cpu().numpy(),
                     h_con=np.array(con_host.array),
                     h_jac=np.array(jac_host.array))
        np.savez(out + '.idx%d' % rank, rows=rows, cols=cols,
                 ab=np.array([sh.a, sh.b]))
        con_host.close()
        jac_host.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_config3_and_gather_to_the_reference(tmp_path):
    """N - 1 = 99 999 nodes over 2 ranks (50 000 + 49 999): gathered vectors
    and the shared host vectors vs the reference's golden samples."""
    import torch.multiprocessing as mp
    out = str(tmp_path/'gathered.npz')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    meta, z = gu.load('config3_10link')
    N, M, C = meta['N'], meta['M'], meta['C']
    P = M*C
    got = np.load(out)
    nodes = z['nodes']
    import opty_amd
    col = opty_amd.ConstraintCollocator(**problems.build('config3_10link'))
    free = problems.make_free(col.num_free, seed=meta['seed'])
    cbn, jbn, _, _ = gu.error_bounds(col, free, nodes)   # per-entry floors
    for tag in ('', 'h_'):
        con, jac = got[tag + 'con'], got[tag + 'jac']
        assert con.shape == (M*(N - 1),) and jac.shape == (P*(N - 1),)
        blk = jac.reshape(N - 1, P)
        cb = con.reshape(M, N - 1)
        gu.assert_close(blk[nodes], z['jac_nodes'], RTOL,
                        what='sharded ' + tag + 'jac nodes', bound=jbn)
        gu.assert_close(cb[:, nodes], z['con_nodes'], RTOL,
                        what='sharded ' + tag + 'con nodes', bound=cbn)
        scale = float(z['jac_abs_sum'][0])
        gu.assert_close(blk.sum(axis=0), z['jac_entry_sums'], 1e-9,
                        scale=scale/P, what=tag + 'jac entry sums')
        gu.assert_close(cb.sum(axis=1), z['con_eq_sums'], 1e-9,
                        scale=float(np.abs(cb).sum())/M,
                        what=tag + 'con sums')
    # shard indices at the sampled nodes
    for rank in range(2):
        zi = np.load(out + '.idx%d.npz' % rank)
        a, b = zi['ab']
        sel = nodes[(nodes >= a) & (nodes < b)]
        pick = np.isin(nodes, sel)
        np.testing.assert_array_equal(
            zi['rows'].reshape(b - a, P)[sel - a], z['rows_nodes'][pick])
        np.testing.assert_array_equal(
            zi['cols'].reshape(b - a, P)[sel - a], z['cols_nodes'][pick])


@pytest.mark.parametrize('world', [2, 8])
def test_bench_strong_scaling_oversubscribed(world):
    """``bench.py --gpus 2`` / ``--gpus 8`` as the driver launches it
    (torch.distributed.run, all ranks on the one GPU here): strong scaling of
    ONE problem, all three re-assembly variants in the line.  Eight ranks is
    the dress rehearsal of BASELINE config 4 -- 99 999 = 7*12 500 + 12 499
    constraint nodes, eight processes registering one shared host vector,
    ``ShardedCallbacks`` with seven serving ranks -- so that the first run on
    eight devices cannot die of a rank-count bug."""
    env = dict(os.environ, OPTY_BENCH_OVERSUBSCRIBE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'bench.py'),
           '--gpus', str(world), '--steps', '5', '--warmup', '2',
           '--prewarm-ms', '20', '--no-cpu-baseline']
    proc = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO,
                          env=env, timeout=1500)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['scaling'] == 'strong' and res['n_gpus'] == world
    assert res['config']['nodes_per_launch'] == -(-99999//world)
    assert set(res['config']['variants']) == {'gather', 'to_host',
                                              'callbacks'}
    assert res['config']['variants']['callbacks']['evals_per_s'] > 0
    assert res['value'] > 0 and res['roofline']['frac'] > 0
    # the line certifies itself: the benched launches of both ranks and every
    # re-assembly variant were checked against the reference's golden record
    ver = res['config']['verify']
    assert ver['ok'] is True and ver['ranks'] == world, ver
    assert ver['worst_rel'] <= 1e-10
    labels = ' '.join(ver['checked'])
    for tag in ('benched launch', 'gather', 'to_host', 'callbacks'):
        assert tag in labels, ver


def test_bench_watchdog_prints_the_headline():
    """The re-assembly variants are the only part of ``bench.py`` that
    exchanges data between ranks; if they hang (here: a limit they cannot
    meet) the measured and verified headline line still appears, from rank
    0, and every rank ends with status 0."""
    env = dict(os.environ, OPTY_BENCH_OVERSUBSCRIBE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0',
               OPTY_BENCH_VARIANTS_TIMEOUT='0.2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'bench.py'),
           '--gpus', '2', '--steps', '5', '--warmup', '2', '--prewarm-ms',
           '20', '--no-cpu-baseline']
    proc = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO,
                          env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['value'] > 0
    assert 'watchdog' in res['config']['variants_error']
    assert res['config']['verify']['ok'] is True
    assert 'benched launch' in ' '.join(res['config']['verify']['checked'])


def test_bench_single_gpu_line_verifies_itself():
    """``bench.py`` as the driver runs it at N = 1: the JSON line carries
    ``config.verify.ok`` from checking the timed launch against the
    reference's golden record, and the exit code is 0."""
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '5',
           '--warmup', '2', '--prewarm-ms', '20', '--no-cpu-baseline',
           '--no-extras']
    proc = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO,
                          timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    res = json.loads([ln for ln in proc.stdout.splitlines()
                      if ln.startswith('{')][-1])
    ver = res['config']['verify']
    assert ver['ok'] is True and ver['worst_rel'] <= 1e-10, ver
    assert any('checksums over all nodes' in c for c in ver['checked'])
    assert any('sampled nodes' in c for c in ver['checked'])


def _problem_worker(rank, world, port, out):
    import sys
    import types
    import torch
    import torch.distributed as dist
    import sympy as sm
    import opty_amd
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        factory, fkw = problems.CONFIGS['pend3_link_midpoint_small']
        kw = factory(**dict(fkw, num_nodes=3002))
        F = [f for f in kw['equations_of_motion'].atoms(sm.Function)
             if f.func.__name__ == 'F'][0]
        N = kw['num_collocation_nodes']
        obj, grad = opty_amd.create_objective_function(
            sm.Integral(F**2, kw['time_symbol']), kw['state_symbols'], (F,),
            (), N, kw['node_time_interval'], integration_method='midpoint',
            time_symbol=kw['time_symbol'])
        prob = opty_amd.ShardedProblem(obj, grad, bounds={F: (-50.0, 50.0)},
                                       **kw)
        if rank != 0:
            prob.serve()
            return

        class FakeProblem(object):          # stands in for cyipopt.Problem
            def __init__(self, n, m, problem_obj=None, lb=None, ub=None,
                         cl=None, cu=None):
                self.o, self.m = problem_obj, m

            def solve(self, x, lagrange=[], zl=[], zu=[]):
                g = self.o.constraints(x)
                vals = np.array(self.o.jacobian(x))
                rows, cols = self.o.jacobianstructure()
                assert len(g) == self.m and len(vals) == len(rows)
                return x, {'g': g, 'jac': vals, 'obj': self.o.objective(x)}

        sys.modules['cyipopt'] = types.SimpleNamespace(Problem=FakeProblem)
        frees = [problems.make_free(prob.num_free, seed=s) for s in (5, 6)]
        _, info = prob.solve(frees[0])
        c2 = prob.constraints(frees[1])           # a second, different point
        rows, cols = prob.jacobianstructure()
        np.savez(out, f0=frees[0], f1=frees[1], g=info['g'], jac=info['jac'],
                 c2=c2, rows=rows, cols=cols, obj=np.array([info['obj']]),
                 ab=np.array([prob.sharded.a, prob.sharded.b]))
        prob.shutdown()
    finally:
        dist.destroy_process_group()


def test_sharded_problem_callbacks_from_two_ranks(tmp_path):
    """``opty_amd.ShardedProblem``: rank 0 drives a (stand-in) IPOPT whose
    callbacks are evaluated by both ranks through the shared page-locked host
    vectors; compared with the single-GPU ``Problem`` on the same inputs."""
    import sympy as sm
    import torch.multiprocessing as mp
    import opty_amd
    out = str(tmp_path/'root.npz')
    mp.spawn(_problem_worker, args=(2, _free_port(), out), nprocs=2,
             join=True)
    z = np.load(out)
    assert tuple(z['ab']) == (0, 1501)              # 3001 nodes: 1501 + 1500
    factory, fkw = problems.CONFIGS['pend3_link_midpoint_small']
    kw = factory(**dict(fkw, num_nodes=3002))
    col = opty_amd.ConstraintCollocator(**kw)
    con, jac = (col.generate_constraint_function(),
                col.generate_jacobian_function())
    rows, cols = col.jacobian_indices()
    np.testing.assert_array_equal(z['rows'], rows)
    np.testing.assert_array_equal(z['cols'], cols)
    cb, jb = gu.error_bounds(col, z['f0'])
    gu.assert_close(z['g'], con(z['f0']), 1e-12, what='sharded problem con',
                    bound=cb)
    gu.assert_close(z['jac'], jac(z['f0']), 1e-12,
                    what='sharded problem jac', bound=jb)
    cb, _ = gu.error_bounds(col, z['f1'])
    gu.assert_close(z['c2'], con(z['f1']), 1e-12,
                    what='sharded problem con 2', bound=cb)
    assert np.isfinite(z['obj'][0])


def _config2_worker(rank, world, port, out, backend):
    import sys
    import types
    import torch
    import torch.distributed as dist
    import opty_amd
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    kw = dict(rank=rank, world_size=world)
    if backend == 'nccl':
        kw['device_id'] = torch.device('cuda', 0)
    dist.init_process_group(backend, **kw)
    try:
        torch.cuda.set_device(0)
        pkw = problems.build('config2_pendulum')
        prob = opty_amd.ShardedProblem(lambda f: 0.0, lambda f: 0.0*f, **pkw)
        assert prob.sharded.o == 4 and prob.callbacks._rccl == (
            backend == 'nccl')
        if rank != 0:
            prob.serve()
            return
        meta, _ = gu.load('config2_pendulum')
        free = problems.make_free(prob.num_free, seed=meta['seed'])
        con = prob.constraints(free)
        jac = np.array(prob.jacobian(free))
        rows, cols = prob.jacobianstructure()
        # the solver changes a known parameter between solves
        # (plot_human_gait.py): the serving ranks follow
        g = [k for k in pkw['known_parameter_map'] if str(k) == 'g'][0]
        prob.collocator.known_parameter_map[g] = 3.7
        con_moon = prob.constraints(free)
        np.savez(out, con=con, jac=jac, rows=rows, cols=cols,
                 con_moon=con_moon,
                 ab=np.array([prob.sharded.a, prob.sharded.b]))
        prob.shutdown()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,backend', [(2, 'gloo'), (3, 'gloo'),
                                           (1, 'nccl')])
def test_sharded_problem_with_instance_constraints_vs_reference(
        tmp_path, world, backend):
    """BASELINE config 2 (N = 10 000, midpoint, FOUR instance constraints) as
    an ``opty_amd.ShardedProblem`` -- ranks oversubscribed on one GPU under
    gloo, and the RCCL code path with the one rank a 1-GPU box gives it --
    against the reference's golden: sampled nodes, checksums over all nodes,
    the instance tails of both vectors, int64 indices."""
    import torch.multiprocessing as mp
    import opty_amd
    out = str(tmp_path/'root.npz')
    mp.spawn(_config2_worker, args=(world, _free_port(), out, backend),
             nprocs=world, join=True)
    got = np.load(out)
    meta, z = gu.load('config2_pendulum')
    N, M, C = meta['N'], meta['M'], meta['C']
    P = M*C
    assert tuple(got['ab']) == (0, -(-(N - 1)//world))
    con, jac = got['con'], got['jac']
    assert len(con) == meta['num_constraints'] and len(jac) == meta['nnz']
    col = opty_amd.ConstraintCollocator(**problems.build('config2_pendulum'))
    free = problems.make_free(col.num_free, seed=meta['seed'])
    nodes = z['nodes']
    cbn, jbn, icb, ijb = gu.error_bounds(col, free, nodes)
    blk = jac[:P*(N - 1)].reshape(N - 1, P)
    cb = con[:M*(N - 1)].reshape(M, N - 1)
    gu.assert_close(blk[nodes], z['jac_nodes'], RTOL,
                    what='sharded config2 jac nodes', bound=jbn)
    gu.assert_close(cb[:, nodes], z['con_nodes'], RTOL,
                    what='sharded config2 con nodes', bound=cbn)
    gu.assert_close(blk.sum(axis=0), z['jac_entry_sums'], 1e-9,
                    scale=float(z['jac_abs_sum'][0])/P,
                    what='sharded config2 jac entry sums')
    gu.assert_close(cb.sum(axis=1), z['con_eq_sums'], 1e-9,
                    scale=float(np.abs(cb).sum())/M,
                    what='sharded config2 con sums')
    gu.assert_close(con[M*(N - 1):], z['con_tail'], RTOL,
                    what='sharded config2 con tail', bound=icb)
    gu.assert_close(jac[P*(N - 1):], z['jac_tail'], RTOL,
                    what='sharded config2 jac tail', bound=ijb)
    rows, cols = got['rows'], got['cols']
    assert rows.dtype == np.int64 and len(rows) == meta['nnz']
    np.testing.assert_array_equal(
        rows[:P*(N - 1)].reshape(N - 1, P)[nodes], z['rows_nodes'])
    np.testing.assert_array_equal(
        cols[:P*(N - 1)].reshape(N - 1, P)[nodes], z['cols_nodes'])
    np.testing.assert_array_equal(rows[P*(N - 1):], z['rows_tail'])
    np.testing.assert_array_equal(cols[P*(N - 1):], z['cols_tail'])
    # known-parameter change on the root reached every rank: the single-GPU
    # collocator with the same change
    g = [k for k in col.known_parameter_map if str(k) == 'g'][0]
    col.known_parameter_map[g] = 3.7
    want = col.generate_constraint_function()(free)
    assert not np.allclose(want, con)
    np.testing.assert_allclose(got['con_moon'], want, rtol=1e-12, atol=1e-12)


def _rccl_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from opty_amd.sharded import ShardedCollocator, ShardedCallbacks
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            device_id=torch.device('cuda', 0))
    try:
        torch.cuda.set_device(0)
        factory, fkw = problems.CONFIGS['config3_10link']
        kw = factory(**dict(fkw, num_nodes=5001))
        sh = ShardedCollocator(device='cuda:0', **kw)
        cb = ShardedCallbacks(sh, name='opty_t_rccl_%d' % port)
        assert cb._rccl and cb.free_host is None
        frees = [problems.make_free(sh.collocator.num_free, seed=s)
                 for s in (1, 2)]
        c1 = cb.constraints(frees[0])
        j1 = cb.jacobian(frees[0]).copy()
        c2, j2 = cb.constraints_and_jacobian(frees[1])
        np.savez(out, f1=frees[0], f2=frees[1], c1=c1, j1=j1, c2=c2,
                 j2=j2.copy())
        cb.shutdown()
    finally:
        dist.destroy_process_group()


def test_callbacks_over_rccl_single_rank(tmp_path):
    """The RCCL branch of ``ShardedCallbacks`` (``free`` broadcast GPU to GPU,
    constraints gathered on the root's GPU, Jacobian through the shared
    page-locked vector) -- with the one rank a 1-GPU box can give RCCL."""
    import torch.multiprocessing as mp
    import opty_amd
    out = str(tmp_path/'rccl.npz')
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    z = np.load(out)
    factory, fkw = problems.CONFIGS['config3_10link']
    col = opty_amd.ConstraintCollocator(**factory(**dict(fkw,
                                                         num_nodes=5001)))
    con, jac = (col.generate_constraint_function(),
                col.generate_jacobian_function())
    # the separate kernels of the same code object: bit for bit
    np.testing.assert_array_equal(z['c1'], con(z['f1']))
    np.testing.assert_array_equal(z['j1'], jac(z['f1']))
    # constraints_and_jacobian runs the fused kernel: same expressions,
    # scheduled (FMA-contracted) on their own
    np.testing.assert_allclose(z['c2'], con(z['f2']), rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(z['j2'], jac(z['f2']), rtol=1e-11, atol=1e-9)


def test_shard_to_host_moves_only_what_changed():
    """``ShardedCollocator.to_host`` on a large shard: the first copy moves the
    whole slice, later ones only the varying entries
    (``opty_hip_shard_jac_to_host``); this rank's slice of the shared host
    vector always equals the shard's values in device memory bit for bit --
    also after a known-parameter change and for an in-place shard -- and
    nothing outside the slice is touched."""
    import torch
    from opty_amd.sharded import ShardedCollocator, SharedHostVector
    factory, fkw = problems.CONFIGS['config3_10link']
    kw = factory(**dict(fkw, num_nodes=30001))
    sh = ShardedCollocator(rank=1, world_size=3, **kw)
    assert sh.jac_local.numel() >= sh._PACKED_MIN_VALUES
    jac_host = SharedHostVector('opty_t_pack', sh.nnz, 0,
                                pin=(sh.a*sh.P, sh.b*sh.P))
    jac_host.array[:] = -7.0
    lo, hi = sh.a*sh.P, sh.b*sh.P
    from opty_amd.codegen.program import varying_copies
    copies = varying_copies(sh.collocator._build_program())[1]
    assert len(copies) == 55
    seen = []
    for k, seed in enumerate((1, 2, 3, 4)):
        if k == 3:
            key = list(kw['known_parameter_map'])[-1]
            kw['known_parameter_map'][key] = 2.5
        free = problems.make_free(sh.collocator.num_free, seed=seed)
        _, jac = sh.evaluate(torch.from_numpy(free).cuda(),
                             in_place=(k == 2))
        sh.to_host(None, jac_host)
        torch.cuda.synchronize()
        want = jac.cpu().numpy()
        blk = want.reshape(-1, sh.P)
        for d, s in copies:     # repeated expressions: the source's value
            blk[:, d] = blk[:, s]
        np.testing.assert_array_equal(jac_host.array[lo:hi], want)
        assert (jac_host.array[:lo] == -7.0).all()
        assert (jac_host.array[hi:] == -7.0).all()
        seen.append(want.copy())
    assert not np.array_equal(seen[0], seen[1])
    # the parameter change moved node-invariant entries too
    P = sh.P
    from opty_amd.codegen.program import varying_entries
    var = set(varying_entries(sh.collocator._build_program()))
    static = [e for e in range(P) if e not in var]
    blk2, blk3 = seen[2].reshape(-1, P), seen[3].reshape(-1, P)
    assert (blk2[:, static] == blk2[0, static]).all()
    assert not np.array_equal(blk2[0, static], blk3[0, static])
    jac_host.close()


@pytest.mark.parametrize('name', ['config2_pendulum_small',
                                  'pend2_link_vardur_unkmass_small'])
def test_sharded_collocator_through_the_library_communicator(name):
    """``ShardedCollocator(comm=HipComm(...))`` in a world of one rank, inside a
    process that also holds PyTorch (one RCCL, one HIP runtime):
    ``broadcast_free`` and ``gather`` go through ``opty_hip_bcast_free`` /
    ``opty_hip_gather_v`` -- own shard copied from the shard buffers (strided
    constraint copy) or evaluated in place, instance tails by the root -- and
    return the single-GPU collocator's vectors."""
    import torch
    import opty_amd
    from opty_amd import hip_backend as hb
    from opty_amd.sharded import ShardedCollocator
    kw = problems.build(name)
    comm = hb.HipComm(hb.HipComm.unique_id(), 0, 1, device=0)
    assert (comm.rank, comm.world) == (0, 1)
    sh = ShardedCollocator(rank=0, world_size=1, device='cuda:0', comm=comm,
                           **kw)
    ref = opty_amd.ConstraintCollocator(**kw)
    free_h = problems.make_free(ref.num_free, seed=3,
                                variable_duration=ref._variable_duration)
    con0 = ref.generate_constraint_function()(free_h)
    jac0 = np.array(ref.generate_jacobian_function()(free_h))
    free = torch.from_numpy(free_h).cuda()
    assert sh.broadcast_free(free, 0) is free
    for in_place in (False, True):
        for what in ('both', 'con', 'jac'):
            sh.evaluate(free, in_place=in_place, what=what)
            con, jac = sh.gather(0, what)
            torch.cuda.synchronize()
            if what != 'jac':
                np.testing.assert_allclose(
                    con.cpu().numpy(), con0, rtol=1e-12,
                    atol=1e-12*np.abs(con0).max())
            if what != 'con':
                np.testing.assert_allclose(
                    jac.cpu().numpy(), jac0, rtol=1e-12,
                    atol=1e-12*np.abs(jac0).max())
    # misuse is reported through the C ABI's error channel
    with pytest.raises(hb.HipBackendError, match='root'):
        comm.bcast_free(sh.collocator.hip, free, root=5)
    comm.close()


def _fake_rccl(tmp_path):
    """Builds the test transport (``tests/fake_rccl``: librccl's entry points
    over /dev/shm, so that several ranks can share one GPU)."""
    lib = str(tmp_path/'libfake_rccl.so')
    src = os.path.join(REPO, 'tests', 'fake_rccl', 'fake_rccl.cpp')
    proc = subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared',
                           '-fPIC', '-O1', src, '-o', lib],
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    return lib


@pytest.mark.parametrize('name,world,root', [
    ('config2_pendulum_small', 2, 0),           # instance constraints
    ('pend2_link_vardur_unkmass_small', 3, 1),  # unknown parameters, free h
    ('config3_10link_small', 3, 2)])            # 40 nodes: shards 14/13/13
def test_gather_v_with_several_ranks(name, world, root, tmp_path):
    """``opty_hip_bcast_free`` / ``opty_hip_gather_v`` with 2 and 3 ranks --
    what RCCL cannot be asked for on a 1-GPU box (it refuses duplicate
    devices): the library is pointed at a test transport with librccl's
    entry points (``OPTY_HIP_RCCL_LIBRARY``), every rank runs
    ``ShardedCollocator(comm=HipComm(...))`` on ``cuda:0``.  Checked on the
    root against the single-GPU collocator: the broadcast free vector,
    gathers of both / constraints only / Jacobian only, own shard copied or
    evaluated in place, unequal shards, instance tails, a root that is not
    rank 0."""
    lib = _fake_rccl(tmp_path)
    env = dict(os.environ, OPTY_HIP_RCCL_LIBRARY=lib)
    idfile = str(tmp_path/'comm.id')
    procs = [subprocess.Popen(
        [sys.executable, os.path.join(REPO, 'tests', 'comm_worker.py'), name,
         str(r), str(world), str(root), idfile],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d:\n%s' % (r, out[-3000:])
        assert 'rank %d of %d ok' % (r, world) in out
    assert '6 gathers checked' in outs[root]
