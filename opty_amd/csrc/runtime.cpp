// runtime.cpp -- the core of libopty_hip.so behind include/opty_hip.h.
//
// Owns, per problem handle: the loaded gfx950 code object with the generated
// collocation kernels, the device copies of the node-invariant data (known
// parameters, known trajectories, instance index tables), staging buffers for
// callers that hand over host memory, and one HIP stream.  Evaluations are
// plain kernel launches on that stream; nothing here computes on the CPU.
//
// Reference counterparts: the closures `constraints` / `constraints_jacobian`
// (opty/direct_collocation.py:2382-2446, :2816-2887), the wrapper of
// _wrap_constraint_funcs (:2928-3001) and jacobian_indices (:2450-2690).
#include "opty_internal.h"

namespace opty {

thread_local std::string g_error;

int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return 1;
}

}  // namespace opty

namespace {

// ---------------------------------------------------------------------------
// jacobian_indices as a closed form; one lane per entry
// pair, 16-byte stores.  Integer, HBM-write bound: 16 bytes per entry.
// ---------------------------------------------------------------------------
struct IndexDims {
    long long N;      // time nodes of the GLOBAL problem
    long long ncon;   // N - 1 (global)
    long long offset; // first global constraint node of this shard
    long long count;  // constraint nodes of this shard
    int n, q, M, C, tail, method;
    int P;                 // stored entries per node block
    const int *pattern;    // (j, k) per stored entry, or null: e -> (e/C, e%C)
    const int *rowinfo;    // CSR layout: (S_j, L_j) per stored entry, or null
};

__device__ __forceinline__ void index_of(const IndexDims &d, long long i,
                                         int j, int k, long long &row,
                                         long long &col) {
    row = (long long)j*d.ncon + i;
    const int n = d.n, q = d.q;
    const long long N = d.N;
    if (d.method == OPTY_HIP_BACKWARD_EULER) {
        if (k < n)              col = (long long)k*N + i + 1;
        else if (k < 2*n)       col = (long long)(k - n)*N + i;
        else if (k < 2*n + q)   col = (long long)(n + k - 2*n)*N + i + 1;
        else                    col = (long long)(n + q)*N + (k - 2*n - q);
    } else {
        if (k < n)              col = (long long)k*N + i;
        else if (k < 2*n)       col = (long long)(k - n)*N + i + 1;
        else if (k < 2*n + q)   col = (long long)(n + k - 2*n)*N + i;
        else if (k < 2*n + 2*q) col = (long long)(n + k - 2*n - q)*N + i + 1;
        else                    col = (long long)(n + q)*N + (k - 2*n - 2*q);
    }
}

// grid.x covers the nodes in blocks of `nodes_per_block`; the threads of a
// block sweep the P entries of each of its nodes, so consecutive lanes write
// consecutive int64s.
__global__ void __launch_bounds__(256)
opty_indices_kernel(IndexDims d, long long *rows, long long *cols,
                    int nodes_per_block) {
    const int P = d.P;
    const long long i0 = (long long)blockIdx.x*nodes_per_block;
    for (int s = 0; s < nodes_per_block; ++s) {
        const long long i = i0 + s;           // local constraint node
        if (i >= d.count) return;
        long long *r = rows + i*P;
        long long *c = cols + i*P;
        for (int e = threadIdx.x; e < P; e += blockDim.x) {
            int j, k;
            if (d.pattern) {
                j = d.pattern[2*e];
                k = d.pattern[2*e + 1];
            } else {
                j = e/d.C;
                k = e - j*d.C;
            }
            long long row, col;
            index_of(d, i + d.offset, j, k, row, col);
            if (d.rowinfo) {
                // row-sorted layout: S entries of a block precede row j,
                // the row holds L of them
                const long long S = d.rowinfo[2*e], L = d.rowinfo[2*e + 1];
                const long long dst = S*d.count + i*L + (e - S);
                rows[dst] = row;
                cols[dst] = col;
            } else {
                r[e] = row;
                c[e] = col;
            }
        }
    }
}

}  // namespace

namespace opty {

int use_device(const opty_hip_problem *p) {
    HIP_TRY(hipSetDevice(p->d.device));
    return 0;
}

int check_ready(const opty_hip_problem *p) {
    if (p->d.p_known > 0 && !p->have_params)
        return fail("known parameters were never set "
                    "(opty_hip_set_known_parameters)");
    if (p->d.m_known > 0 && !p->have_known)
        return fail("known trajectories were never set "
                    "(opty_hip_set_known_trajectories)");
    if (p->d.num_inst > 0 && !p->have_inst)
        return fail("instance indices were never set "
                    "(opty_hip_set_instance_indices)");
    if (p->d.s == 0 && !p->have_h)
        return fail("the node time interval was never set "
                    "(opty_hip_set_interval)");
    return 0;
}


NodeRange whole(const opty_hip_problem *p) {
    return NodeRange{0, p->ncon_nodes(), p->ncon_nodes()};
}

// hipStreamSynchronize target of a handle's stream: the legacy handle is
// synchronised through the null stream it stands for.
hipStream_t sync_target(hipStream_t s) {
    return s == (hipStream_t)OPTY_HIP_STREAM_LEGACY ? nullptr : s;
}

// wgs_per_block: workgroups per 64-node block (0: a single one-wave launch,
// < 0: that many workgroups regardless of the node count); threads: workgroup
// size.
// The launch evaluates the constraint nodes [begin, end) of the handle's
// problem: `con` points at the shard's first value of equation 0 (equations
// are `con_stride` doubles apart), `jac` at the shard's first block.
// List schedule of a persistent kernel (dispatch order 'list' of the printer):
// `npw` one-wave workgroups -- workgroup w runs on XCD w % 8 and holds a SIMD
// alone -- share the (node block, strip class) items of a launch; class g of
// every block takes cost[g] (any unit).  Per XCD: longest processing time
// first onto the least loaded workgroup, so that a launch costs about
// sum(durations) / npw instead of what the hardware's one-wave-per-item
// dispatch leaves idle between and after the waves.  Table:
//   [0] npw   [1 .. npw + 1] item offsets per workgroup   [npw + 2 ..] items,
// an item = (class << 24) | block slot s of the XCD (block = 8 s + XCD), in the
// order in which the workgroup evaluates them.
std::vector<int> build_schedule(int persist, long long nblk, int sets,
                                const float *cost) {
    static const bool rotate = !getenv("OPTY_HIP_LIST_NO_ROTATE");
    const long long nslot = (nblk + 7)/8;
    long long total = nslot*8*sets;
    const int npw = (int)(total < persist ? total : persist);
    const int bins = npw/8;
    std::vector<std::vector<int>> mine((size_t)npw);
    // classes, longest first (stable: the printer sorted them already)
    std::vector<int> order((size_t)sets);
    for (int g = 0; g < sets; ++g) order[(size_t)g] = g;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return cost[a] > cost[b];
    });
    for (int x = 0; x < 8 && bins > 0; ++x) {
        typedef std::pair<double, int> Load;        // (load, workgroup slot)
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> q;
        for (int j = 0; j < bins; ++j) q.push(Load(0.0, j));
        for (int g : order) {
            const double c = cost[g] > 0.f ? cost[g] : 1.0;
            for (long long s = 0; s < nslot; ++s) {
                if (s*8 + x >= nblk) break;
                Load l = q.top();
                q.pop();
                mine[(size_t)(l.second*8 + x)].push_back((g << 24) | (int)s);
                l.first += c;
                q.push(l);
            }
        }
    }
    std::vector<int> table;
    table.push_back(npw);
    int at = 0;
    for (int w = 0; w < npw; ++w) {
        table.push_back(at);
        // Every workgroup starts somewhere else in its list (longest first,
        // rotated by its number): the short, store-heavy strips of a launch
        // then run spread over its whole duration, next to the long ones --
        // all of them at its end, they queue for the memory system (biped:
        // 700 concurrent store-only waves took 13.9 us instead of 8.4)
        std::vector<int> &m = mine[(size_t)w];
        if (rotate && m.size() > 1)
            std::rotate(m.begin(), m.begin() + (w/8) % (int)m.size(), m.end());
        at += (int)m.size();
    }
    table.push_back(at);
    for (int w = 0; w < npw; ++w)
        table.insert(table.end(), mine[(size_t)w].begin(),
                     mine[(size_t)w].end());
    return table;
}

// persist > 0: a persistent kernel with a list schedule (`sched`: the
// handle's cache of tables, one per launch size); at most `persist`
// workgroups.
int launch(opty_hip_problem *p, hipFunction_t f, int wgs_per_block,
           int threads, const double *free_, double *con, double *jac,
           const NodeRange &rg, bool inst_block = false, int persist = 0,
           std::vector<Schedule> *sched = nullptr,
           const float *cost = nullptr) {
    KernelArgs a;
    a.free_ = free_;
    a.known_traj = p->d_known;
    a.params = p->d_params;
    a.uni_c = p->d_uni;
    a.uni_w = p->d_uni;
    a.inst_idx = p->d_inst_idx;
    // the kernels index con with the global node number
    a.con = con ? con - rg.begin : nullptr;
    a.jac = jac;
    a.h = p->h;
    a.N = p->d.N;
    a.con_stride = rg.con_stride;
    a.node_begin = rg.begin;
    a.node_end = rg.end;
    a.sched = nullptr;
    size_t size = offsetof(KernelArgs, sched);
    int npw = 0;
    if (persist > 0 && wgs_per_block > 0) {
        const long long nblk = (rg.end - rg.begin + 63)/64;
        if (nblk == 0) return 0;
        Schedule *hit = nullptr;
        for (Schedule &sc : *sched)
            if (sc.nblk == nblk) hit = &sc;
        if (!hit) {
            // first launch of this size: build and upload (synchronous)
            std::vector<int> table =
                build_schedule(persist, nblk, wgs_per_block, cost);
            Schedule sc;
            sc.nblk = nblk;
            sc.npw = table[0];
            HIP_TRY(hipMalloc((void **)&sc.d_table,
                              table.size()*sizeof(int)));
            HIP_TRY(hipMemcpy(sc.d_table, table.data(),
                              table.size()*sizeof(int),
                              hipMemcpyHostToDevice));
            if (sched->size() >= 16) {          // shard sizes come and go
                (void)hipStreamSynchronize(sync_target(p->stream));
                (void)hipFree(sched->front().d_table);
                sched->erase(sched->begin());
            }
            sched->push_back(sc);
            hit = &sched->back();
        }
        a.sched = hit->d_table;
        npw = hit->npw;
        size = sizeof a;
    }
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a,
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &size,
                      HIP_LAUNCH_PARAM_END};
    unsigned grid = 1;
    if (wgs_per_block < 0) {
        grid = (unsigned)(-wgs_per_block);      // plain grid, no node blocks
    } else if (wgs_per_block > 0) {
        // node blocks padded to a multiple of the 8 XCDs (see the kernels'
        // prologue: block -> XCD placement); surplus workgroups exit at once
        const long long nblk = ((rg.end - rg.begin + 63)/64 + 7)/8*8;
        // inst_block: one more workgroup, which evaluates the instance-
        // constraint tails (modules built with desc.inst_folded)
        grid = (unsigned)(nblk*wgs_per_block);
        // (a persistent kernel reads the same number from its table: the
        // workgroup behind them evaluates the instance tails)
        if (persist > 0) grid = (unsigned)npw;
        grid += inst_block ? 1u : 0u;
        if (grid == 0) return 0;
    }
    HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, threads, 1, 1, 0, p->stream,
                                  nullptr, config));
    return 0;
}

// The instance-constraint tails (opty/direct_collocation.py:2985-2991): `con`
// / `jac` point at the first of the o values / nnz_inst partials (either may
// be null).  One lane; reads the global free vector through the atom table.
int launch_instance(opty_hip_problem *p, const double *free_, double *con_tail,
                    double *jac_tail) {
    // opty_inst stores con[M*con_stride + k] and jac[(end - begin)*P + k]
    return launch(p, p->k_inst, 0, 64, free_, con_tail, jac_tail,
                  NodeRange{0, 0, 0});
}

// what: OPTY_HIP_EVAL_*; device pointers only.  `with_inst`: the launch
// covers the whole problem and the instance tails follow the last node's
// values (node shards leave them to opty_hip_eval_instance).

bool routing_enabled() {
    // OPTY_HIP_ROUTING=plan: the launch plan's flags as they are (A/B runs)
    const char *e = getenv("OPTY_HIP_ROUTING");
    return !(e && !strcmp(e, "plan"));
}

// Average duration (ms) of one issue of `fn` on the handle's stream: one
// untimed issue, a short batch to size the others, then the best of three
// timed batches of ~2 ms each (hipEvents).  Batches of a few launches are not
// enough: the biped's opty_jac took 0.0667 ms in bursts of four and 0.0723 ms
// sustained, next to a fused kernel at 0.0710 -- the decision flipped on the
// length of the measurement (BENCH of r06's last day).
template <typename Fn>
int time_issue(opty_hip_problem *p, Fn fn, float *ms_out) {
    if (int rc = fn()) return rc;
    int n = 2;
    float best = 1e30f;
    for (int round = 0; round < 4; ++round) {
        HIP_TRY(hipEventRecord(p->ev_cal0, p->stream));
        for (int i = 0; i < n; ++i)
            if (int rc = fn()) return rc;
        HIP_TRY(hipEventRecord(p->ev_cal1, p->stream));
        HIP_TRY(hipEventSynchronize(p->ev_cal1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev_cal0, p->ev_cal1));
        if (round == 0) {
            const float per = ms/n > 1e-4f ? ms/n : 1e-4f;
            const int want = (int)(2.0f/per) + 1;
            n = want > 256 ? 256 : (want < 4 ? 4 : want);
            continue;                   // sizing batch: not a measurement
        }
        if (ms/n < best) best = ms/n;
    }
    *ms_out = best;
    return 0;
}

// Measures, for the launch size of `rg`, the three kernels an entry point
// can be served by -- opty_conjac, opty_con, opty_jac -- on the handle's own
// device and stream, into the caller's buffers (the evaluation is a pure
// function of `free`: writing a result twice is harmless; a missing
// constraint vector is replaced by the handle's scratch), and decides
//   fused_loses   : opty_con + opty_jac beat opty_conjac,
//   jac_via_fused : opty_conjac beats opty_jac,
// each against the launch plan's flag with 1 % + 0.3 us in favour of the flag
// (two kernels within the resolution of the timer must not flip from handle
// to handle).  A few launches, once per handle and launch size (VERDICT r05
// item 2: the plan file's flags were measured on another box, and were wrong
// on the driver's for two problems).
int calibrate_route(opty_hip_problem *p, const double *free_, double *con,
                    double *jac, const NodeRange &rg, Route *out) {
    if (!p->ev_cal0) {
        HIP_TRY(hipEventCreate(&p->ev_cal0));
        HIP_TRY(hipEventCreate(&p->ev_cal1));
    }
    NodeRange cr = rg;
    if (!con) {
        if (int rc = ensure(&p->d_con_scratch, (size_t)p->num_con()))
            return rc;
        con = p->d_con_scratch + rg.begin;
        cr.con_stride = p->ncon_nodes();
    }
    auto fused = [&] {
        return launch(p, p->k_conjac, p->d.fused_wgs_per_block,
                      64*p->d.fused_waves_per_wg, free_, con, jac, cr, false,
                      p->d.fused_persist, &p->sched_fused,
                      p->d.fused_class_cost);
    };
    auto conk = [&] {
        return launch(p, p->k_con, p->d.con_wgs_per_block,
                      64*p->d.con_waves_per_wg, free_, con, nullptr, cr);
    };
    auto jack = [&] {
        return launch(p, p->k_jac, p->d.jac_wgs_per_block,
                      64*p->d.jac_waves_per_wg, free_, nullptr, jac, cr,
                      false, p->d.jac_persist, &p->sched_jac,
                      p->d.jac_class_cost);
    };
    Route r;
    r.nblk = (rg.end - rg.begin + 63)/64;
    if (int rc = time_issue(p, fused, &r.ms_fused)) return rc;
    if (int rc = time_issue(p, conk, &r.ms_con)) return rc;
    if (int rc = time_issue(p, jack, &r.ms_jac)) return rc;
    const float pair = r.ms_con + r.ms_jac;
    const bool plan_loses = p->d.fused_loses != 0;
    r.fused_loses = plan_loses ? !(r.ms_fused < pair*0.99f - 3e-4f)
                               : (pair < r.ms_fused*0.99f - 3e-4f);
    const bool plan_via = p->d.jac_via_fused != 0;
    r.jac_via_fused = !r.fused_loses &&
        (plan_via ? !(r.ms_jac < r.ms_fused*0.99f - 3e-4f)
                  : (r.ms_fused < r.ms_jac*0.99f - 3e-4f));
    static const bool trace = getenv("OPTY_HIP_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "opty_hip: routing of %lld-block launches: opty_conjac "
                "%.4f ms, opty_con %.4f + opty_jac %.4f = %.4f ms -> "
                "fused_loses %d (plan %d), jac_via_fused %d (plan %d)\n",
                r.nblk, r.ms_fused, r.ms_con, r.ms_jac, pair,
                (int)r.fused_loses, (int)plan_loses, (int)r.jac_via_fused,
                (int)plan_via);
    *out = r;
    return 0;
}

// The route of the launch size of `rg`; measured at its first use.  *out
// stays null when it cannot be measured (no Jacobian buffer: never asked).
int route_for(opty_hip_problem *p, const double *free_, double *con,
              double *jac, const NodeRange &rg, const Route **out) {
    const long long nblk = (rg.end - rg.begin + 63)/64;
    for (const Route &r : p->routes)
        if (r.nblk == nblk) {
            *out = &r;
            return 0;
        }
    if (!jac || nblk == 0) return 0;
    Route r;
    if (int rc = calibrate_route(p, free_, con, jac, rg, &r)) return rc;
    if (p->routes.size() >= 16) p->routes.erase(p->routes.begin());
    p->routes.push_back(r);
    *out = &p->routes.back();
    return 0;
}

int eval_device(opty_hip_problem *p, int what, const double *free_,
                double *con, double *jac, const NodeRange &rg,
                bool with_inst) {
    const int S = p->d.jac_wgs_per_block, T = 64*p->d.jac_waves_per_wg;
    if (int rc = order_streams(p)) return rc;
    // Node-invariant sub-expressions: recomputed only when their inputs can
    // have changed (always, if they read unknown parameters / h from `free`).
    if (p->d.num_uniform > 0 && (p->uni_dirty || p->d.uniform_dynamic)) {
        if (int rc = launch(p, p->k_uni, -OPTY_UNI_WORKGROUPS, 64, free_, nullptr,
                                nullptr, rg)) return rc;
        p->uni_dirty = false;
    }
    // Small problems' modules carry the instance tails in the main kernels
    // (one more workgroup instead of one more launch: a launch costs such a
    // problem as much as its evaluation).
    const bool tails = with_inst && p->d.num_inst > 0;
    const bool folded = tails && p->d.inst_folded;
    // Which of the module's kernels serve this entry point.  The launch
    // plan's flags (measured on the tuner's box) are the starting point;
    // with OPTY_HIP_ROUTE_CALIBRATE the handle measures the candidates on
    // ITS device at the first launch of every size and keeps the faster
    // (calibrate_route); a kernel the build marked unusable (it spills
    // vector registers: OPTY_HIP_ROUTE_NO_*) is never launched.
    bool fused_loses = p->d.fused_loses != 0;
    bool jac_via_fused = p->d.jac_via_fused != 0;
    const int banned = p->d.routing & (OPTY_HIP_ROUTE_NO_JAC_KERNEL |
                                       OPTY_HIP_ROUTE_NO_FUSED_KERNEL);
    if ((p->d.routing & OPTY_HIP_ROUTE_CALIBRATE) && !banned &&
        routing_enabled() &&
        (what == OPTY_HIP_EVAL_FUSED || what == OPTY_HIP_EVAL_JAC)) {
        const Route *rt = nullptr;
        if (int rc = route_for(p, free_, con, jac, rg, &rt)) return rc;
        if (rt) {
            fused_loses = rt->fused_loses;
            jac_via_fused = rt->jac_via_fused;
        }
    }
    if (banned & OPTY_HIP_ROUTE_NO_JAC_KERNEL) {
        // opty_jac is out: the fused kernel serves EVAL_JAC (constraint
        // values to scratch) and the pair
        fused_loses = false;
        jac_via_fused = true;
        if (what == OPTY_HIP_EVAL_PAIR) what = OPTY_HIP_EVAL_FUSED;
    }
    if (banned & OPTY_HIP_ROUTE_NO_FUSED_KERNEL) {
        if (what == OPTY_HIP_EVAL_FUSED_KERNEL)
            return fail("the fused kernel of this module is marked unusable "
                        "(opty_hip_desc.routing)");
        fused_loses = true;
        jac_via_fused = false;
    }
    // the fused kernel was measured slower than the two it replaces: issue
    // those
    if (what == OPTY_HIP_EVAL_FUSED && fused_loses)
        what = OPTY_HIP_EVAL_PAIR;
    if (what == OPTY_HIP_EVAL_FUSED_KERNEL) what = OPTY_HIP_EVAL_FUSED;
    // ... or faster than the Jacobian-only one: its constraint values go to
    // scratch
    if (what == OPTY_HIP_EVAL_JAC && jac_via_fused && !fused_loses) {
        if (int rc = ensure(&p->d_con_scratch, (size_t)p->num_con()))
            return rc;
        NodeRange sr{rg.begin, rg.end, p->ncon_nodes()};
        if (int rc = launch(p, p->k_conjac, p->d.fused_wgs_per_block,
                            64*p->d.fused_waves_per_wg, free_,
                            p->d_con_scratch + rg.begin, jac, sr, folded,
                            p->d.fused_persist, &p->sched_fused,
                            p->d.fused_class_cost))
            return rc;
        if (tails && !folded)
            if (int rc = launch_instance(
                    p, free_, nullptr, jac + (rg.end - rg.begin)*p->P()))
                return rc;
        return 0;
    }
    if (what == OPTY_HIP_EVAL_CON || what == OPTY_HIP_EVAL_PAIR)
        if (int rc = launch(p, p->k_con, p->d.con_wgs_per_block,
                            64*p->d.con_waves_per_wg, free_, con, nullptr, rg,
                            folded))
            return rc;
    if (what == OPTY_HIP_EVAL_JAC || what == OPTY_HIP_EVAL_PAIR)
        if (int rc = launch(p, p->k_jac, S, T, free_, nullptr, jac, rg,
                            folded, p->d.jac_persist, &p->sched_jac,
                            p->d.jac_class_cost))
            return rc;
    if (what == OPTY_HIP_EVAL_FUSED)
        if (int rc = launch(p, p->k_conjac, p->d.fused_wgs_per_block,
                            64*p->d.fused_waves_per_wg, free_, con, jac, rg,
                            folded, p->d.fused_persist, &p->sched_fused,
                            p->d.fused_class_cost))
            return rc;
    if (tails && !folded) {
        double *c = (what == OPTY_HIP_EVAL_JAC) ? nullptr
            : con + (long long)p->d.M*rg.con_stride;
        double *j = (what == OPTY_HIP_EVAL_CON) ? nullptr
            : jac + (rg.end - rg.begin)*p->P();
        if (int rc = launch_instance(p, free_, c, j)) return rc;
    }
    return 0;
}

int check_shard(const opty_hip_problem *p, int what, const double *free_,
                const double *con, const double *jac, int64_t con_stride,
                int64_t node_begin, int64_t node_end) {
    if (!p) return fail("null handle");
    if (what != OPTY_HIP_EVAL_CON && what != OPTY_HIP_EVAL_JAC &&
        what != OPTY_HIP_EVAL_PAIR && what != OPTY_HIP_EVAL_FUSED &&
        what != OPTY_HIP_EVAL_FUSED_KERNEL)
        return fail("bad evaluation selector %d", what);
    if (p->d.layout != OPTY_HIP_LAYOUT_COO)
        return fail("only the node-major layout is node-sharded");
    if (node_begin < 0 || node_end < node_begin ||
        node_end > p->ncon_nodes())
        return fail("shard [%lld, %lld) outside the %lld constraint nodes",
                    (long long)node_begin, (long long)node_end,
                    (long long)p->ncon_nodes());
    const bool want_con = what != OPTY_HIP_EVAL_JAC;
    const bool want_jac = what != OPTY_HIP_EVAL_CON;
    if (!free_ || (want_con && !con) || (want_jac && !jac))
        return fail("null buffer");
    if (want_con && con_stride < node_end - node_begin)
        return fail("con_stride %lld is smaller than the shard's %lld nodes",
                    (long long)con_stride, (long long)(node_end - node_begin));
    return 0;
}


// total bytes of one host-side evaluation up to which the mapped-memory path
// is used
#define OPTY_LATENCY_PATH_BYTES (2u << 20)

// NUMA node of the current HIP device (its PCI function's numa_node in
// sysfs; -1: unknown / one node).
int device_numa_node() {
    int dev = 0;
    char bdf[64] = {0}, path[160];
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetPCIBusId(bdf, sizeof bdf, dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE *f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    return node;
}

// Page-locked host memory next to the current device.  It lands where the
// calling thread's memory policy puts it; a vector on the other socket than
// the GPU costs the DMA its rate (one box of r05: 4.6 ms per Jacobian next to
// the GPU, 7.7 ms across the socket link).  So the GPU's node is PREFERRED for
// the duration of the allocation (MPOL_PREFERRED; a container that refuses
// set_mempolicy keeps its default; OPTY_HIP_HOST_ALLOC_ANYWHERE=1 opts out).
hipError_t pinned_alloc(void **ptr, size_t bytes) {
    const int node = device_numa_node();
    bool bound = false;
    if (node >= 0 && node < 64 && !getenv("OPTY_HIP_HOST_ALLOC_ANYWHERE")) {
        unsigned long mask = 1UL << node;
        bound = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, &mask,
                        65UL) == 0;
    }
    hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
    if (bound)
        (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0UL);
    return e;
}


// Device-visible address of caller memory that is page-locked (hipHostMalloc
// / hipHostRegister), or null for pageable memory.
double *mapped_address(double *host) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (attr.type != hipMemoryTypeHost) return nullptr;
    return static_cast<double *>(attr.devicePointer);
}


int eval_mapped(opty_hip_problem *p, int what, const double *free_,
                double *con, double *jac) {
    const bool want_con = what != OPTY_HIP_EVAL_JAC;
    const bool want_jac = what != OPTY_HIP_EVAL_CON;
    const double t_in = std::chrono::duration<double, std::micro>(
        std::chrono::steady_clock::now().time_since_epoch()).count();
    if (int rc = ensure_pinned(&p->h_free, (size_t)p->num_free())) return rc;
    double *dcon = nullptr, *djac = nullptr;
    if (want_con) {
        dcon = mapped_address(con);
        if (!dcon) {
            if (int rc = ensure_pinned(&p->h_con, (size_t)p->num_con()))
                return rc;
            dcon = p->h_con;
        }
    }
    if (want_jac) {
        djac = mapped_address(jac);
        if (!djac) {
            if (int rc = ensure_pinned(&p->h_jac, (size_t)p->nnz())) return rc;
            djac = p->h_jac;
        }
    }
    if (int rc = order_streams(p)) return rc;
    // OPTY_HIP_TRACE=1: where the time of one call goes (stderr)
    static const bool trace = getenv("OPTY_HIP_TRACE") != nullptr;
    auto now = [] {
        return std::chrono::duration<double, std::micro>(
            std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t0 = trace ? now() : 0.0;
    memcpy(p->h_free, free_, p->num_free()*sizeof(double));
    const double t1 = trace ? now() : 0.0;
    if (int rc = eval_device(p, what, p->h_free, dcon, djac, whole(p), true))
        return rc;
    const double t2 = trace ? now() : 0.0;
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    const double t3 = trace ? now() : 0.0;
    if (want_con && dcon == p->h_con)
        memcpy(con, p->h_con, p->num_con()*sizeof(double));
    if (want_jac && djac == p->h_jac)
        memcpy(jac, p->h_jac, p->nnz()*sizeof(double));
    if (trace)
        fprintf(stderr, "opty_hip: mapped evaluation %d: pointer queries "
                "%.1f us, free in %.1f, launches %.1f, wait %.1f, results "
                "out %.1f\n", what, t0 - t_in, t1 - t0, t2 - t1, t3 - t2,
                now() - t3);
    return 0;
}


int eval_any(opty_hip_problem *p, int what, const double *free_, double *con,
             double *jac, int mem) {
    if (!p) return fail("null handle");
    if (int rc = use_device(p)) return rc;
    if (int rc = check_ready(p)) return rc;
    const bool want_con = what != OPTY_HIP_EVAL_JAC;
    const bool want_jac = what != OPTY_HIP_EVAL_CON;
    if (!free_ || (want_con && !con) || (want_jac && !jac))
        return fail("null buffer");
    if (mem != OPTY_HIP_HOST && mem != OPTY_HIP_DEVICE)
        return fail("bad memory kind %d", mem);
    if (p->d.layout == OPTY_HIP_LAYOUT_SEGMENTED && want_jac)
        return eval_segmented(p, what, free_, con, jac, mem, true);
    if (mem == OPTY_HIP_DEVICE)
        return eval_device(p, what, free_, con, jac, whole(p), true);
    // Small problems (BASELINE config 2: 240 KB in, 160 KB + 960 KB out) are
    // bound by the latency of the copies, not by their bytes: a pageable
    // hipMemcpyAsync costs 15-20 us whatever it moves.  Their kernels read
    // `free` from and write the results to page-locked, device-mapped host
    // memory directly -- no copy is enqueued; the caller's vectors are
    // reached by plain memcpy (or, when they are page-locked themselves,
    // like the persistent Jacobian array, written in place).
    const size_t moved = sizeof(double)*(size_t)(
        p->num_free() + (want_con ? p->num_con() : 0) +
        (want_jac ? p->nnz() : 0));
    if (moved <= OPTY_LATENCY_PATH_BYTES && !getenv("OPTY_HIP_NO_LATENCY_PATH"))
        return eval_mapped(p, what, free_, con, jac);
    // Host buffers (the cyipopt callback case): stage through device memory.
    if (int rc = ensure(&p->d_free, (size_t)p->num_free())) return rc;
    if (want_con)
        if (int rc = ensure(&p->d_con, (size_t)p->num_con())) return rc;
    if (want_jac)
        if (int rc = ensure(&p->d_jac, (size_t)p->nnz())) return rc;
    if (int rc = order_streams(p)) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_free, free_, p->num_free()*sizeof(double),
                           hipMemcpyHostToDevice, p->stream));
    if (int rc = eval_device(p, what, p->d_free, p->d_con, p->d_jac,
                             whole(p), true))
        return rc;
    if (want_con)
        HIP_TRY(hipMemcpyAsync(con, p->d_con, p->num_con()*sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
    if (want_jac)
        HIP_TRY(hipMemcpyAsync(jac, p->d_jac, p->nnz()*sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    return 0;
}

}  // namespace opty

using namespace opty;

extern "C" {

const char *opty_hip_last_error(void) { return g_error.c_str(); }

void *opty_hip_host_alloc(size_t bytes) {
    void *ptr = nullptr;
    if (bytes == 0) bytes = 8;
    hipError_t e = pinned_alloc(&ptr, bytes);
    if (e != hipSuccess) {
        fail("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return ptr;
}

int opty_hip_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return 0;
}

int opty_hip_abi_version(void) { return OPTY_HIP_ABI_VERSION; }

int opty_hip_list_schedule(int persist, int64_t node_blocks, int classes,
                           const float *class_cost, int32_t *table,
                           int64_t capacity, int64_t *count) {
    if (persist < 8 || persist % 8 || node_blocks < 0 || classes < 1 ||
        classes > OPTY_HIP_MAX_CLASSES || !class_cost || !count)
        return fail("bad list-schedule request (%d workgroups, %lld blocks, "
                    "%d classes)", persist, (long long)node_blocks, classes);
    std::vector<int> t = build_schedule(persist, node_blocks, classes,
                                        class_cost);
    *count = (int64_t)t.size();
    if (table) {
        if (capacity < (int64_t)t.size())
            return fail("schedule table needs %lld words",
                        (long long)t.size());
        memcpy(table, t.data(), t.size()*sizeof(int));
    }
    return 0;
}

int opty_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int opty_hip_create(const opty_hip_desc *desc, const char *code_object_path,
                    opty_hip_problem **out) {
    if (!desc || !code_object_path || !out) return fail("null argument");
    if (desc->N < 2) return fail("need at least 2 collocation nodes");
    if (desc->P < 0 || desc->P > desc->M*desc->C)
        return fail("P = %d stored entries per block, block is %d x %d",
                    desc->P, desc->M, desc->C);
    if (desc->layout != OPTY_HIP_LAYOUT_COO &&
        desc->layout != OPTY_HIP_LAYOUT_CSR &&
        desc->layout != OPTY_HIP_LAYOUT_SEGMENTED)
        return fail("bad layout %d", desc->layout);
    if (desc->layout == OPTY_HIP_LAYOUT_SEGMENTED &&
        desc->P != desc->M*desc->C)
        return fail("the segmented layout stores the whole %d x %d block",
                    desc->M, desc->C);
    if (desc->jac_wgs_per_block < 1 || desc->jac_waves_per_wg < 1 ||
        desc->jac_waves_per_wg > 16 || desc->fused_wgs_per_block < 1 ||
        desc->con_wgs_per_block < 1 || desc->fused_waves_per_wg < 1 ||
        desc->fused_waves_per_wg > 16 || desc->con_waves_per_wg < 1 ||
        desc->con_waves_per_wg > 16)
        return fail("bad Jacobian launch geometry (%d workgroups x %d waves)",
                    desc->jac_wgs_per_block, desc->jac_waves_per_wg);
    if (desc->jac_persist < 0 || desc->fused_persist < 0 ||
        desc->jac_persist % 8 || desc->fused_persist % 8 ||
        (desc->jac_persist && desc->jac_waves_per_wg != 1) ||
        (desc->fused_persist && desc->fused_waves_per_wg != 1))
        return fail("persistent kernels take a multiple of 8 one-wave "
                    "workgroups (jac_persist %d, fused_persist %d)",
                    desc->jac_persist, desc->fused_persist);
    if ((desc->jac_persist &&
         desc->jac_wgs_per_block > OPTY_HIP_MAX_CLASSES) ||
        (desc->fused_persist &&
         desc->fused_wgs_per_block > OPTY_HIP_MAX_CLASSES))
        return fail("a list schedule takes at most %d strip classes",
                    OPTY_HIP_MAX_CLASSES);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail("no HIP device is visible: the HIP backend has no CPU "
                    "fallback");
    if (desc->device < 0 || desc->device >= count)
        return fail("device %d out of range (have %d)", desc->device, count);
    HIP_TRY(hipSetDevice(desc->device));
    auto *p = new opty_hip_problem;
    p->d = *desc;
    hipError_t e = hipModuleLoad(&p->module, code_object_path);
    if (e != hipSuccess) {
        delete p;
        (void)hipGetLastError();
        return fail("hipModuleLoad(%s) failed: %s", code_object_path,
                    hipGetErrorString(e));
    }
    struct { const char *name; hipFunction_t *f; bool required; } ks[] = {
        {"opty_con", &p->k_con, true},
        {"opty_jac", &p->k_jac, true},
        {"opty_conjac", &p->k_conjac, true},
        {"opty_inst", &p->k_inst, desc->num_inst > 0},
        {"opty_uni", &p->k_uni, desc->num_uniform > 0},
    };
    for (auto &k : ks) {
        if (!k.required) continue;
        e = hipModuleGetFunction(k.f, p->module, k.name);
        if (e != hipSuccess) {
            (void)hipModuleUnload(p->module);
            delete p;
            return fail("kernel %s missing from %s: %s", k.name,
                        code_object_path, hipGetErrorString(e));
        }
    }
    auto allocate = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&p->own_stream,
                                         hipStreamNonBlocking));
        p->stream = p->own_stream;
        HIP_TRY(hipEventCreate(&p->ev0));
        HIP_TRY(hipEventCreate(&p->ev1));
        if (desc->p_known > 0)
            HIP_TRY(hipMalloc((void **)&p->d_params,
                              desc->p_known*sizeof(double)));
        if (desc->num_uniform > 0)
            HIP_TRY(hipMalloc((void **)&p->d_uni,
                              desc->num_uniform*sizeof(double)));
        if (desc->m_known > 0)
            HIP_TRY(hipMalloc((void **)&p->d_known,
                              (size_t)desc->m_known*desc->N*sizeof(double)));
        return 0;
    };
    if (int rc = allocate()) {
        (void)opty_hip_destroy(p);      // releases whatever was acquired
        return rc;
    }
    *out = p;
    return 0;
}

int opty_hip_destroy(opty_hip_problem *p) {
    if (!p) return 0;
    (void)hipSetDevice(p->d.device);
    (void)hipStreamSynchronize(sync_target(p->stream));
    void *bufs[] = {p->d_pattern, p->d_rowinfo, p->d_uni, p->d_params,
                    p->d_known, p->d_inst_idx, p->d_inst_rows,
                    p->d_inst_cols, p->d_free, p->d_con, p->d_jac, p->d_rows,
                    p->d_cols, p->d_var, p->d_packed, p->d_seg_order,
                    p->d_dense, p->d_seg, p->d_con_scratch};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (auto *v : {&p->sched_jac, &p->sched_fused})
        for (Schedule &sc : *v) (void)hipFree(sc.d_table);
    // (a straggling scatter worker of the sliced mode may still read
    // h_packed: ADVICE r05)
    opty::scatter_quiesce();
    void *pinned[] = {p->h_packed, p->h_free, p->h_con, p->h_jac};
    for (void *b : pinned)
        if (b) (void)hipHostFree(b);
    for (hipEvent_t e : p->chunk_events) (void)hipEventDestroy(e);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->ev_cal0) (void)hipEventDestroy(p->ev_cal0);
    if (p->ev_cal1) (void)hipEventDestroy(p->ev_cal1);
    if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    if (p->copy_stream2) (void)hipStreamDestroy(p->copy_stream2);
    if (p->module) (void)hipModuleUnload(p->module);
    delete p;
    return 0;
}

int opty_hip_set_stream(opty_hip_problem *p, void *hip_stream) {
    if (!p) return fail("null handle");
    p->stream = hip_stream ? (hipStream_t)hip_stream : p->own_stream;
    return 0;
}

int opty_hip_synchronize(opty_hip_problem *p) {
    if (!p) return fail("null handle");
    if (int rc = use_device(p)) return rc;
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    return 0;
}

int opty_hip_set_known_parameters(opty_hip_problem *p, const double *values,
                                  int32_t count) {
    if (!p) return fail("null handle");
    if (count != p->d.p_known)
        return fail("expected %d known parameters, got %d", p->d.p_known,
                    count);
    if (count == 0) return 0;
    if (!values) return fail("null values");
    if (int rc = use_device(p)) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_params, values, count*sizeof(double),
                           hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    p->uni_dirty = true;
    p->static_valid = p->shard_valid = false;   // invariant entries change
    p->have_params = true;
    return 0;
}

int opty_hip_set_interval(opty_hip_problem *p, double h) {
    if (!p) return fail("null handle");
    if (!(h > 0.0) || h > 1.79e308)
        return fail("the node time interval must be positive and finite, "
                    "got %g", h);
    p->h = h;
    p->have_h = true;
    p->uni_dirty = true;
    p->static_valid = p->shard_valid = false;
    return 0;
}

int opty_hip_set_known_trajectories(opty_hip_problem *p, const double *values,
                                    int32_t mem) {
    if (!p) return fail("null handle");
    if (p->d.m_known == 0) return 0;
    if (!values) return fail("null values");
    if (int rc = use_device(p)) return rc;
    const size_t bytes = (size_t)p->d.m_known*p->d.N*sizeof(double);
    HIP_TRY(hipMemcpyAsync(p->d_known, values, bytes,
                           mem == OPTY_HIP_DEVICE ? hipMemcpyDeviceToDevice
                                                  : hipMemcpyHostToDevice,
                           p->stream));
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    p->have_known = true;
    return 0;
}

int opty_hip_set_instance_indices(opty_hip_problem *p,
                                  const int64_t *atom_free_index,
                                  const int64_t *rows, const int64_t *cols) {
    if (!p) return fail("null handle");
    if (p->d.num_inst == 0) return 0;
    if (int rc = use_device(p)) return rc;
    const int na = p->d.num_inst_atoms, nz = p->d.nnz_inst;
    if (na > 0 && !atom_free_index) return fail("null atom index table");
    if (nz > 0 && (!rows || !cols)) return fail("null instance rows/cols");
    for (int a = 0; a < na; ++a)
        if (atom_free_index[a] < 0 || atom_free_index[a] >= p->num_free())
            return fail("instance atom %d: free index %lld out of range", a,
                        (long long)atom_free_index[a]);
    if (int rc = ensure(&p->d_inst_idx, (size_t)na)) return rc;
    if (int rc = ensure(&p->d_inst_rows, (size_t)nz)) return rc;
    if (int rc = ensure(&p->d_inst_cols, (size_t)nz)) return rc;
    if (na)
        HIP_TRY(hipMemcpy(p->d_inst_idx, atom_free_index, na*sizeof(int64_t),
                          hipMemcpyHostToDevice));
    if (nz) {
        HIP_TRY(hipMemcpy(p->d_inst_rows, rows, nz*sizeof(int64_t),
                          hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_inst_cols, cols, nz*sizeof(int64_t),
                          hipMemcpyHostToDevice));
    }
    p->have_inst = true;
    return 0;
}

int opty_hip_set_block_pattern(opty_hip_problem *p, const int32_t *jk) {
    if (!p || !jk) return fail("null argument");
    if (int rc = use_device(p)) return rc;
    for (int e = 0; e < p->d.P; ++e)
        if (jk[2*e] < 0 || jk[2*e] >= p->d.M || jk[2*e + 1] < 0 ||
            jk[2*e + 1] >= p->d.C)
            return fail("block pattern entry %d = (%d, %d) outside %d x %d",
                        e, jk[2*e], jk[2*e + 1], p->d.M, p->d.C);
    if (int rc = ensure(&p->d_pattern, (size_t)2*p->d.P)) return rc;
    HIP_TRY(hipMemcpy(p->d_pattern, jk, 2*p->d.P*sizeof(int32_t),
                      hipMemcpyHostToDevice));
    if (p->d.layout == OPTY_HIP_LAYOUT_CSR) {
        std::vector<int32_t> info(2*(size_t)p->d.P);
        for (int e = 0; e < p->d.P;) {
            int e1 = e;
            while (e1 < p->d.P && jk[2*e1] == jk[2*e]) ++e1;
            if (e > 0 && jk[2*e] <= jk[2*(e - 1)])
                return fail("CSR block pattern is not grouped by row");
            for (int t = e; t < e1; ++t) {
                info[2*t] = e;
                info[2*t + 1] = e1 - e;
            }
            e = e1;
        }
        if (int rc = ensure(&p->d_rowinfo, info.size())) return rc;
        HIP_TRY(hipMemcpy(p->d_rowinfo, info.data(),
                          info.size()*sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return 0;
}

int64_t opty_hip_num_free(const opty_hip_problem *p) {
    return p ? p->num_free() : -1;
}
int64_t opty_hip_num_constraints(const opty_hip_problem *p) {
    return p ? p->num_con() : -1;
}
int64_t opty_hip_nnz(const opty_hip_problem *p) { return p ? p->nnz() : -1; }

int opty_hip_eval_con(opty_hip_problem *p, const double *free_, double *con,
                      int32_t mem) {
    return eval_any(p, OPTY_HIP_EVAL_CON, free_, con, nullptr, mem);
}

int opty_hip_eval_jac(opty_hip_problem *p, const double *free_, double *jac,
                      int32_t mem) {
    return eval_any(p, OPTY_HIP_EVAL_JAC, free_, nullptr, jac, mem);
}

int opty_hip_eval_con_jac(opty_hip_problem *p, const double *free_,
                          double *con, double *jac, int32_t mem) {
    return eval_any(p, OPTY_HIP_EVAL_FUSED, free_, con, jac, mem);
}

// Indices of the constraint nodes [node_offset, node_offset + count) of a
// problem with N_global time nodes; `with_inst`: followed by the instance
// part (whole-problem calls only).
static int indices_impl(opty_hip_problem *p, int64_t N_global,
                        int64_t node_offset, int64_t count, bool with_inst,
                        int64_t *rows, int64_t *cols, int32_t mem) {
    if (!rows || !cols) return fail("null buffer");
    if (int rc = use_device(p)) return rc;
    if (p->d.num_inst > 0 && !p->have_inst)
        return fail("instance indices were never set");
    if (node_offset < 0 || count < 0 || node_offset + count > N_global - 1)
        return fail("shard [%lld, %lld) outside the %lld constraint nodes",
                    (long long)node_offset, (long long)(node_offset + count),
                    (long long)(N_global - 1));
    long long *dr = (long long *)rows, *dc = (long long *)cols;
    const int nnz_inst = with_inst ? p->d.nnz_inst : 0;
    const size_t nnz = (size_t)(p->P()*count + nnz_inst);
    if (nnz == 0) return 0;
    if (mem == OPTY_HIP_HOST) {
        if (int rc = ensure(&p->d_rows, nnz)) return rc;
        if (int rc = ensure(&p->d_cols, nnz)) return rc;
        dr = p->d_rows;
        dc = p->d_cols;
    } else if (mem != OPTY_HIP_DEVICE) {
        return fail("bad memory kind %d", mem);
    }
    IndexDims d;
    d.N = N_global;
    d.ncon = N_global - 1;
    d.offset = node_offset;
    d.count = count;
    d.n = p->d.n;
    d.q = p->d.q;
    d.M = p->d.M;
    d.C = p->d.C;
    d.tail = p->d.r + p->d.s;
    d.method = p->d.method;
    d.P = p->d.P;
    d.pattern = p->d_pattern;
    d.rowinfo = p->d.layout != OPTY_HIP_LAYOUT_COO ? p->d_rowinfo : nullptr;
    if (p->d.layout == OPTY_HIP_LAYOUT_CSR && !p->d_rowinfo)
        return fail("the CSR block pattern was never set "
                    "(opty_hip_set_block_pattern)");
    if (p->d.layout == OPTY_HIP_LAYOUT_SEGMENTED && !p->have_segments)
        return fail("the segments were never set (opty_hip_set_segments)");
    if (p->d.layout != OPTY_HIP_LAYOUT_COO &&
        (node_offset != 0 || count != N_global - 1))
        return fail("only the node-major layout is node-sharded");
    if (p->d.P != p->d.M*p->d.C && !p->d_pattern)
        return fail("the block pattern was never set "
                    "(opty_hip_set_block_pattern)");
    const int P = (int)p->P();
    // enough entries per block to keep 256 lanes busy
    int npb = P >= 1024 ? 1 : (1024 + P - 1)/P;
    const unsigned grid = (unsigned)((d.count + npb - 1)/npb);
    (void)hipGetLastError();    // drop whatever an earlier failed call left
    if (grid > 0) {
        hipLaunchKernelGGL(opty_indices_kernel, dim3(grid), dim3(256), 0,
                           p->stream, d, dr, dc, npb);
        HIP_TRY(hipGetLastError());
    }
    const size_t base = (size_t)(p->P()*count);
    if (nnz_inst > 0) {
        HIP_TRY(hipMemcpyAsync(dr + base, p->d_inst_rows,
                               p->d.nnz_inst*sizeof(int64_t),
                               hipMemcpyDeviceToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(dc + base, p->d_inst_cols,
                               p->d.nnz_inst*sizeof(int64_t),
                               hipMemcpyDeviceToDevice, p->stream));
    }
    if (mem == OPTY_HIP_HOST) {
        HIP_TRY(hipMemcpyAsync(rows, dr, nnz*sizeof(int64_t),
                               hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipMemcpyAsync(cols, dc, nnz*sizeof(int64_t),
                               hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
        // index arrays are setup-only: do not keep 16 bytes/entry resident
        (void)hipFree(p->d_rows);
        (void)hipFree(p->d_cols);
        p->d_rows = p->d_cols = nullptr;
    }
    return 0;
}

int opty_hip_jacobian_indices(opty_hip_problem *p, int64_t *rows,
                              int64_t *cols, int32_t mem) {
    if (!p) return fail("null handle");
    return indices_impl(p, p->d.N, 0, p->ncon_nodes(), true, rows, cols, mem);
}

int opty_hip_jacobian_indices_range(opty_hip_problem *p, int64_t node_begin,
                                    int64_t node_end, int64_t *rows,
                                    int64_t *cols, int32_t mem) {
    if (!p) return fail("null handle");
    if (node_end < node_begin) return fail("empty node range");
    return indices_impl(p, p->d.N, node_begin, node_end - node_begin, false,
                        rows, cols, mem);
}

int opty_hip_jacobian_indices_shard(opty_hip_problem *p, int64_t N_global,
                                    int64_t node_offset, int64_t *rows,
                                    int64_t *cols, int32_t mem) {
    if (!p) return fail("null handle");
    if (p->d.num_inst > 0)
        return fail("a slab handle cannot carry instance constraints (their "
                    "free indices are global): use one global handle and "
                    "opty_hip_jacobian_indices_range");
    if (p->d.layout != OPTY_HIP_LAYOUT_COO)
        return fail("only the node-major layout is node-sharded");
    return indices_impl(p, N_global, node_offset, p->ncon_nodes(), false,
                        rows, cols, mem);
}

static int time_impl(opty_hip_problem *p, int32_t what, const double *free_,
                     double *con, double *jac, const NodeRange &rg,
                     bool with_inst, int32_t iters, float *ms_per_iter) {
    if (!ms_per_iter) return fail("null argument");
    if (iters < 1) return fail("iters must be >= 1");
    if (int rc = use_device(p)) return rc;
    if (int rc = check_ready(p)) return rc;
    if (int rc = order_streams(p)) return rc;
    if (p->d.num_uniform > 0 && p->uni_dirty && !p->d.uniform_dynamic) {
        // keep the one-off table fill out of the timed region
        if (int rc = launch(p, p->k_uni, -OPTY_UNI_WORKGROUPS, 64, free_, nullptr,
                                nullptr, rg)) return rc;
        p->uni_dirty = false;
    }
    // ... and the one-off calibration of the routing of this launch size
    if ((p->d.routing & OPTY_HIP_ROUTE_CALIBRATE) && routing_enabled() &&
        !(p->d.routing & ~OPTY_HIP_ROUTE_CALIBRATE) &&
        (what == OPTY_HIP_EVAL_FUSED || what == OPTY_HIP_EVAL_JAC)) {
        const Route *rt = nullptr;
        if (int rc = route_for(p, free_, con, jac, rg, &rt)) return rc;
    }
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    for (int it = 0; it < iters; ++it)
        if (int rc = eval_device(p, what, free_, con, jac, rg, with_inst))
            return rc;
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    HIP_TRY(hipEventSynchronize(p->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *ms_per_iter = ms/iters;
    return 0;
}

int opty_hip_time_eval(opty_hip_problem *p, int32_t what, const double *free_,
                       double *con, double *jac, int32_t iters,
                       float *ms_per_iter) {
    if (!p) return fail("null argument");
    return time_impl(p, what, free_, con, jac, whole(p), true, iters,
                     ms_per_iter);
}

int opty_hip_eval_shard(opty_hip_problem *p, int32_t what, const double *free_,
                        double *con, int64_t con_stride, double *jac,
                        int64_t node_begin, int64_t node_end) {
    if (int rc = check_shard(p, what, free_, con, jac, con_stride, node_begin,
                             node_end)) return rc;
    if (int rc = use_device(p)) return rc;
    if (int rc = check_ready(p)) return rc;
    if (node_end == node_begin) return 0;
    return eval_device(p, what, free_, con, jac,
                       NodeRange{node_begin, node_end, con_stride}, false);
}

int opty_hip_eval_instance(opty_hip_problem *p, const double *free_,
                           double *con_tail, double *jac_tail) {
    if (!p) return fail("null handle");
    if (!free_) return fail("null buffer");
    if (p->d.num_inst == 0 || (!con_tail && !jac_tail)) return 0;
    if (int rc = use_device(p)) return rc;
    if (int rc = check_ready(p)) return rc;
    if (int rc = order_streams(p)) return rc;
    return launch_instance(p, free_, con_tail, jac_tail);
}

int opty_hip_time_eval_shard(opty_hip_problem *p, int32_t what,
                             const double *free_, double *con,
                             int64_t con_stride, double *jac,
                             int64_t node_begin, int64_t node_end,
                             int32_t iters, float *ms_per_iter) {
    if (int rc = check_shard(p, what, free_, con, jac, con_stride, node_begin,
                             node_end)) return rc;
    return time_impl(p, what, free_, con, jac,
                     NodeRange{node_begin, node_end, con_stride}, false, iters,
                     ms_per_iter);
}

int opty_hip_routing(opty_hip_problem *p, int64_t node_count,
                     int32_t *calibrated, int32_t *fused_loses,
                     int32_t *jac_via_fused, float *ms3) {
    if (!p) return fail("null handle");
    if (node_count < 0) return fail("negative node count");
    const long long nblk = (node_count + 63)/64;
    const Route *hit = nullptr;
    for (const Route &r : p->routes)
        if (r.nblk == nblk) hit = &r;
    const int banned = p->d.routing & (OPTY_HIP_ROUTE_NO_JAC_KERNEL |
                                       OPTY_HIP_ROUTE_NO_FUSED_KERNEL);
    bool fl = hit ? hit->fused_loses : p->d.fused_loses != 0;
    bool jv = hit ? hit->jac_via_fused : p->d.jac_via_fused != 0;
    if (banned & OPTY_HIP_ROUTE_NO_JAC_KERNEL) { fl = false; jv = true; }
    if (banned & OPTY_HIP_ROUTE_NO_FUSED_KERNEL) { fl = true; jv = false; }
    if (calibrated) *calibrated = hit ? 1 : 0;
    if (fused_loses) *fused_loses = fl;
    if (jac_via_fused) *jac_via_fused = jv && !fl;
    if (ms3) {
        ms3[0] = hit ? hit->ms_fused : 0.f;
        ms3[1] = hit ? hit->ms_con : 0.f;
        ms3[2] = hit ? hit->ms_jac : 0.f;
    }
    return 0;
}

}  // extern "C"
