#!/usr/bin/env python
"""Benchmark of the hot path: constraint + Jacobian evaluations per second of
the 10-link pendulum on a cart at N = 100 000 collocation nodes (BASELINE.json
metric, ``configs[2]``), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One *step* = one ``constraints(free)`` + one ``jacobian(free)`` on a free
vector that differs from the previous step's (rotating set of synthetic
vectors already in HBM), evaluated through the C ABI (``libopty_hip.so``) with
device pointers on torch's current stream.  By default both outputs of a step
come from ONE launch (``opty_hip_eval_con_jac``: the Jacobian waves plus one
constraint wave per 64-node block); ``--serial`` issues ``opty_hip_eval_con``
then ``opty_hip_eval_jac`` instead.

Multi-GPU (``--gpus N`` under ``torch.distributed.run``): the collocation
nodes are sharded, one contiguous node range per rank with a one-node halo
(SURVEY.md 8(e)); every rank owns 100 000 nodes of an N x 100 000-node problem
(weak scaling) and leaves its slice of the outputs in its own HBM -- there is
no data-path collective.  ``--gather`` adds the RCCL all-gather that
reassembles the full constraint/Jacobian vectors (reported separately in
``config``).

Timing: an untimed clock-ramp phase (``--prewarm-ms`` of the same step; an
idle GPU starts at its lowest clock), W untimed warm-up steps, then exactly K
steps between barrier + ``torch.cuda.synchronize()`` pairs, max over ranks.
``--to-host`` adds the per-rank device-to-host copies (PCIe-inclusive rate).

Prints ONE JSON line (rank 0).
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
WORKLOAD = 'config3_10link'


def cpu_baseline(kw, budget_s=14.0):
    """The oracle's C/OpenMP restatement of the reference's generated code,
    timed on this box's host cores on a bounded number of repetitions of the
    same N = 100 000 workload, for a few OpenMP team sizes; the best one is
    reported (``cores`` = its thread count)."""
    from oracle.collocation_oracle import OracleCollocator
    from opty_amd import problems
    ncpu = os.cpu_count() or 1
    orc = OracleCollocator(name='config3_10link', parallel=True, **kw)
    con = orc.generate_constraint_function()
    jac = orc.generate_jacobian_function()
    frees = [problems.make_free(orc.num_free, seed=s) for s in range(3)]
    con(frees[0]), jac(frees[0])                   # warm-up / page-in
    teams = sorted({t for t in (1, 16, ncpu//4, ncpu//2, ncpu) if t >= 1})
    results = {}
    for threads in teams:
        orc._c_con.parallel = orc._c_jac.parallel = threads
        con(frees[1]), jac(frees[1])
        t0 = time.perf_counter()
        reps = 0
        while True:
            f = frees[reps % 3]
            con(f)
            jac(f)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s/len(teams) or reps >= 100:
                break
        results[threads] = (reps/el, reps)
    best = max(results, key=lambda t: results[t][0])
    detail = ', '.join('%d thr: %.1f/s (%d reps)' % (t, results[t][0],
                                                      results[t][1])
                       for t in teams)
    return dict(value=results[best][0], unit='evals/s', cores=best,
                kind='port',
                sample='constraint+Jacobian evaluations of the full '
                       'N=%d 10-link problem, OpenMP over nodes ' % orc.N +
                       '(gcc -O2 -fopenmp), reference-shaped wrappers '
                       '(fresh con array + transpose copy); ' + detail)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nodes', type=int, default=100000)
    ap.add_argument('--gather', action='store_true',
                    help='all-gather the sharded outputs over RCCL each step')
    ap.add_argument('--serial', action='store_true',
                    help='two launches per step (opty_hip_eval_con, then '
                         'opty_hip_eval_jac) instead of opty_hip_eval_con_jac')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: --nodes is the GLOBAL node count, '
                         'sharded over the ranks (default: weak scaling, '
                         '--nodes per rank)')
    ap.add_argument('--to-host', action='store_true',
                    help='every rank also copies its con/jac shard into its '
                         'own page-locked host buffer each step (the '
                         'PCIe-inclusive rate a host-side IPOPT sees; never '
                         'the headline value)')
    ap.add_argument('--streams', type=int, default=1,
                    help='issue consecutive steps round-robin on this many '
                         'HIP streams with separate output buffers, so that '
                         'the drain of one evaluation overlaps the fill of '
                         'the next (default 1: strictly one after the other, '
                         'as an NLP solver calls them)')
    ap.add_argument('--prewarm-ms', type=float, default=150.0,
                    help='untimed clock-ramp phase before the warm-up steps '
                         '(wall milliseconds of the same step; 0 disables)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    if args.streams > 1 and (args.gather or args.to_host):
        ap.error('--streams > 1 is not combined with --gather / --to-host')

    import torch
    import torch.distributed as dist
    from opty_amd import problems, hip_backend as hb
    import opty_amd

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # Development aid: OPTY_BENCH_OVERSUBSCRIBE=1 lets several ranks share one
    # GPU (gloo rendezvous; RCCL refuses duplicate devices) so that the
    # multi-process launch path can be exercised on a 1-GPU box.  The result
    # is flagged in `config` and is not a scaling measurement.
    oversub = os.environ.get('OPTY_BENCH_OVERSUBSCRIBE') == '1' and \
        world > torch.cuda.device_count()
    if oversub:
        local_rank %= torch.cuda.device_count()
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, 'launch with torch.distributed.run'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if oversub:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device(
                'cuda', local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    factory, fkw = problems.CONFIGS[WORKLOAD]
    local_nodes = args.nodes
    if args.strong and world > 1:
        # contiguous node ranges with a one-node halo (opty_amd.sharded)
        from opty_amd.sharded import partition_nodes
        a, b = partition_nodes(args.nodes - 1, world)[rank]
        local_nodes = b - a + 1
    fkw = dict(fkw, num_nodes=local_nodes)
    kw = factory(**fkw)
    # every rank evaluates its own 100 000-node shard (nodes rank*N .. +N,
    # with its one-node halo, is exactly an N-node collocation problem)
    col = opty_amd.ConstraintCollocator(device=local_rank, **kw)
    hip = col.hip
    hip.set_stream(torch.cuda.current_stream().cuda_stream)

    nfree, ncon, nnz = col.num_free, col.num_constraints, hip.nnz
    frees = [torch.from_numpy(problems.make_free(nfree, seed=1000*rank + s))
             .to(dev) for s in range(4)]
    con = torch.empty(ncon, dtype=torch.float64, device=dev)
    jac = torch.empty(nnz, dtype=torch.float64, device=dev)
    lanes = [(torch.cuda.current_stream(), con, jac)]
    for _ in range(max(1, args.streams) - 1):
        lanes.append((torch.cuda.Stream(device=dev),
                      torch.empty(ncon, dtype=torch.float64, device=dev),
                      torch.empty(nnz, dtype=torch.float64, device=dev)))
    gathered = None
    if args.gather and world > 1:
        gathered = (torch.empty(world*ncon, dtype=torch.float64, device=dev),
                    torch.empty(world*nnz, dtype=torch.float64, device=dev))

    hosted = None
    if args.to_host:
        hosted = (torch.empty(ncon, dtype=torch.float64).pin_memory(),
                  torch.empty(nnz, dtype=torch.float64).pin_memory())

    def step(k):
        f = frees[k % len(frees)]
        stream, con, jac = lanes[k % len(lanes)]
        if len(lanes) > 1:
            hip.set_stream(stream.cuda_stream)
        if args.serial:
            hip.eval_con(f, con, hb.DEVICE)
            hip.eval_jac(f, jac, hb.DEVICE)
        else:
            hip.eval_con_jac(f, con, jac, hb.DEVICE)
        if gathered is not None:
            dist.all_gather_into_tensor(gathered[0], con)
            dist.all_gather_into_tensor(gathered[1], jac)
        if hosted is not None:
            hosted[0].copy_(con, non_blocking=True)
            hosted[1].copy_(jac, non_blocking=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock ramp: an idle MI355X sits at its lowest shader clock and needs a
    # few tens of milliseconds of work to reach the sustained one (the first
    # ~150 steps run ~10 % slower).  The same step is therefore run untimed
    # for `--prewarm-ms` of wall time before the W warm-up steps, so that the
    # K timed steps measure the steady state whatever K and W are.
    t_ramp = time.perf_counter()
    k = 0
    if gathered is not None and args.prewarm_ms > 0:
        # steps contain collectives: every rank must run the same number
        for k in range(256):
            step(k)
        torch.cuda.synchronize()
    while gathered is None and \
            (time.perf_counter() - t_ramp)*1e3 < args.prewarm_ms:
        for _ in range(16):
            step(k)
            k += 1
        torch.cuda.synchronize()
    for k in range(args.warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # Dominant kernel (opty_jac) duration, measured live with HIP events on
    # the stream the kernels are launched on.
    barrier()
    hip.set_stream(torch.cuda.current_stream().cuda_stream)
    jac_ms = hip.time_eval(hb.EVAL_JAC, frees[0], None, jac, args.steps)
    con_ms = hip.time_eval(hb.EVAL_CON, frees[1], con, None, args.steps)
    fused_ms = hip.time_eval(hb.EVAL_FUSED, frees[2], con, jac, args.steps)
    barrier()

    # Informational (rank 0, single GPU, not `value`): the same evaluations
    # issued round-robin on two streams with two sets of output buffers, so
    # that one launch's drain overlaps the next one's fill -- what a caller
    # with independent points to evaluate gets, not what an NLP solver sees.
    pipelined = None
    if world == 1 and len(lanes) == 1 and not (args.gather or args.to_host):
        s2 = torch.cuda.Stream(device=dev)
        con2, jac2 = torch.empty_like(con), torch.empty_like(jac)
        both = [(torch.cuda.current_stream(), con, jac), (s2, con2, jac2)]
        def two_streams(count):
            for k in range(count):
                st, c_, j_ = both[k % 2]
                hip.set_stream(st.cuda_stream)
                hip.eval_con_jac(frees[k % len(frees)], c_, j_, hb.DEVICE)
            torch.cuda.synchronize()

        two_streams(max(args.warmup, 20))       # first touch of the buffers
        tp = time.perf_counter()
        two_streams(args.steps)
        pipelined = args.steps/(time.perf_counter() - tp)
        hip.set_stream(torch.cuda.current_stream().cuda_stream)
        del con2, jac2

    if rank == 0:
        prog = col._build_program()
        P, M, N = prog.P, prog.M, local_nodes
        # algorithmic bytes of one Jacobian launch: read
        # `free` once, write the dense blocks once
        jac_bytes = 8.0*nfree + 8.0*P*(N - 1)
        con_bytes = 8.0*nfree + 8.0*M*(N - 1)
        if args.serial:
            dom, dom_ms, dom_bytes = 'opty_jac', jac_ms, jac_bytes
        else:   # one launch reads `free` once and writes both outputs
            dom, dom_ms = 'opty_conjac', fused_ms
            dom_bytes = 8.0*nfree + 8.0*M*(N - 1) + 8.0*P*(N - 1)
        achieved = dom_bytes/(dom_ms*1e-3)/1e9
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # collected separately with rocprofv3 as the microarch guide
        # prescribes; summaries under profiles/)
        traffic = None
        try:
            # the counters were collected on the 100 000-node launch
            if N == 100000:
                with open(os.path.join(REPO, 'profiles',
                                       'traffic.json')) as f:
                    traffic = json.load(f)[dom]['hbm_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        # weak scaling: every rank evaluates a full N-node problem per step;
        # strong scaling: all ranks together evaluate one
        value = args.steps*(1 if args.strong else world)/elapsed
        out = {
            'metric': 'constraint+Jacobian evals/sec at N=100k nodes '
                      '(10-link pendulum on cart, backward Euler)',
            'value': value, 'unit': 'evals/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3*elapsed/args.steps,
            'higher_is_better': True,
            'scaling': 'strong' if args.strong else 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': '10-link inverted pendulum on cart, %d nodes per '
                            'GPU, backward Euler, n=M=22, q=1, C=45, nnz=%d '
                            'per GPU' % (N, nnz),
                'step': ('opty_hip_eval_con + opty_hip_eval_jac (two '
                         'launches)' if args.serial else
                         'opty_hip_eval_con_jac (constraints and Jacobian of '
                         'one free vector from one launch)'),
                'sharding': 'nodes sharded one contiguous range per GPU, '
                            'outputs left distributed' +
                            (', + RCCL all-gather of con and jac'
                             if gathered is not None else '') +
                            (', + device-to-host copy of every shard into '
                             'page-locked memory' if hosted is not None
                             else ''),
                'oversubscribed': bool(oversub),
                'prewarm_ms': args.prewarm_ms,
                'streams': len(lanes),
                'two_stream_pipelined_evals_per_s': pipelined,
                'jac_waves_per_block': hip.desc['jac_wgs_per_block'] *
                hip.desc['jac_waves_per_wg'],
                'jac_GBps_nnz_written': 8.0*P*(N - 1)/(jac_ms*1e-3)/1e9,
                'kernel_ms': {'opty_jac': jac_ms, 'opty_con': con_ms,
                              'opty_conjac': fused_ms},
                'pair_algorithmic_GB': (jac_bytes + con_bytes)/1e9,
            },
            'roofline': {
                'bound': 'hbm', 'kernel': dom, 'achieved': achieved,
                'algorithmic_bytes_per_launch': dom_bytes,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved/HBM_PEAK_GBS, 'traffic': traffic},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(kw)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
