// opty_hip.cpp -- runtime behind include/opty_hip.h (libopty_hip.so).
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
#include <hip/hip_runtime.h>

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/opty_hip.h"

namespace {

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

#define HIP_TRY(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            (void)hipGetLastError(); /* the runtime's error state is sticky */\
            return fail("%s failed: %s", #expr, hipGetErrorString(e_));       \
        }                                                                     \
    } while (0)

// opty_uni fills the node-invariant table with up to this many single-lane
// workgroups (the generated kernel switches on blockIdx.x; surplus ones exit).
#define OPTY_UNI_WORKGROUPS 16

// The packed kernarg buffer; must match the parameter list every generated
// kernel has (KERNEL_PARAMS in opty_amd/codegen/emit_hip.py).
struct KernelArgs {
    const double *free_;
    const double *known_traj;
    const double *params;
    const double *uni_c;
    double *uni_w;
    const long long *inst_idx;
    double *con;
    double *jac;
    double h;
    long long N;
    long long con_stride;
    long long node_begin;
    long long node_end;
    // kernels with a list schedule only (desc.jac_persist / fused_persist):
    // one more parameter, the schedule table (build_schedule below)
    const int *sched;
};

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

// ---------------------------------------------------------------------------
// Host-visible Jacobian: only what changed crosses PCIe.
//
// The reference hands IPOPT the DENSE per-node block (structural zeros and
// node-invariant entries included, opty/direct_collocation.py:2589-2593) in a
// persistent array (:2814).  For the 10-link pendulum 660 of the block's 990
// entries are the same at every node and every call; moving all 792 MB over
// PCIe Gen5 x16 takes 13.9 ms, the 264 MB that can change 4.6 ms.  The
// varying entries of every node are packed on the device (opty_pack_kernel),
// copied in chunks by the DMA engine into a page-locked staging vector, and
// scattered into the caller's dense vector by a small pool of host threads
// while the next chunk is in flight.  (Alternatives measured on MI355X,
// profiles/r03_host_scatter.txt: a kernel storing the runs straight into
// host-mapped memory 7.2-7.3 ms -- 64-byte PCIe writes, 36 GB/s; one
// hipMemcpy2D per run 59 ms.)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
opty_pack_kernel(const double *__restrict__ jac, double *__restrict__ packed,
                 const int *__restrict__ entries, int V, long long P,
                 long long total) {
    long long g = (long long)blockIdx.x*256 + threadIdx.x;
    const long long stride = (long long)gridDim.x*256;
    for (; g < total; g += stride) {
        const long long i = g/V;
        const int v = (int)(g - i*V);
        packed[g] = jac[i*P + entries[v]];
    }
}

// Persistent pool of host threads (one per library and process) that scatter
// packed node rows into a dense vector.  Between jobs the workers sleep on a
// condition variable; during a job they poll the number of chunks that have
// landed (a job lasts a few milliseconds).
class ScatterPool {
public:
    struct Job {
        const double *packed = nullptr;   // [node][V]
        double *dense = nullptr;          // [node][P]
        const int *run_start = nullptr, *run_len = nullptr;
        // entries that repeat another entry of their node's block
        const int *copy_dst = nullptr, *copy_src = nullptr;
        const double *copy_scale = nullptr;   // null: plain copies
        int nruns = 0, V = 0, chunks = 0, ncopies = 0;
        long long P = 0, nodes = 0;
        // segmented layout: seg_dst[i*L1 + k] = seg_src[i*L0 + seg_pos[k]]
        // (the repeated entries, filled from the varying entries that have
        // landed at the head of the same vector)
        const double *seg_src = nullptr;
        double *seg_dst = nullptr;
        const int *seg_pos = nullptr;
        int L0 = 0, L1 = 0;
        // staging: chunk c copies the columns [bounds[c], bounds[c + 1]) of
        // a (rows x pitch) matrix from rows_src to rows_dst (same layout) --
        // the caller's pageable vectors into / out of page-locked memory, in
        // the windows the device pipeline consumes / produces them
        const double *rows_src = nullptr;
        double *rows_dst = nullptr;
        const long long *bounds = nullptr;
        long long rows = 0, pitch = 0;
    };

    // chunk c of the running job has been finished by every worker
    bool chunk_finished(int c) const {
        return chunk_done_[c].load(std::memory_order_acquire) >= threads();
    }
    void wait_chunk(int c) const {
        for (unsigned spins = 0; !chunk_finished(c); ++spins) {
            if (spins < 4096) cpu_relax();
            else std::this_thread::yield();
        }
    }

    static ScatterPool &instance() {
        static ScatterPool *pool = nullptr;
        static std::mutex guard;
        std::lock_guard<std::mutex> lk(guard);
        // a forked child inherits the object but none of its threads
        if (!pool || pool->pid_ != getpid()) pool = new ScatterPool;
        return *pool;
    }

    static int default_threads() {
        if (const char *env = getenv("OPTY_HIP_HOST_THREADS")) {
            const int n = atoi(env);
            if (n > 0) return std::min(n, 256);
        }
        // one pool per process: the ranks of a node share its cores, and all
        // of them scatter into ONE vector, i.e. onto the cores of one NUMA
        // node (a quarter of the hardware threads on a two-socket SMT box)
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::max(2u, std::min(16u, hw/4/local_ranks().second));
    }

    int threads() const { return (int)workers_.size(); }
    int numa_node() const { return node_; }

    // (LOCAL_RANK, LOCAL_WORLD_SIZE) of this process (torch.distributed.run)
    static std::pair<unsigned, unsigned> local_ranks() {
        unsigned rank = 0, size = 1;
        if (const char *env = getenv("LOCAL_WORLD_SIZE"))
            size = (unsigned)std::max(1, atoi(env));
        if (const char *env = getenv("LOCAL_RANK"))
            rank = (unsigned)std::max(0, atoi(env)) % size;
        return {rank, size};
    }

    // CPUs of NUMA node `node` (the one that holds the vector being
    // assembled); restarts the workers there.  node < 0: unknown, nothing
    // changes.
    void set_numa_node(int node) {
        std::lock_guard<std::recursive_mutex> lk(busy_);
        if (node < 0 || node == node_) return;
        char path[96], buf[4096];
        snprintf(path, sizeof path,
                 "/sys/devices/system/node/node%d/cpulist", node);
        FILE *f = fopen(path, "r");
        if (!f) return;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (fgets(buf, sizeof buf, f)) {
            char *save = nullptr;
            for (char *tok = strtok_r(buf, ",\n", &save); tok;
                 tok = strtok_r(nullptr, ",\n", &save)) {
                int lo = 0, hi = 0;
                const int got = sscanf(tok, "%d-%d", &lo, &hi);
                if (got == 1) hi = lo;
                for (int c = lo; got >= 1 && c <= hi && c < CPU_SETSIZE; ++c)
                    CPU_SET(c, &set);
            }
        }
        fclose(f);
        if (CPU_COUNT(&set) == 0) return;
        // one CPU per physical core (the lowest of its hardware threads):
        // two workers on sibling hyperthreads share one core's load/store
        // bandwidth -- measured on the same box: scatter finished 0.2 ms
        // after the last DMA chunk in one process, 2.4 ms after it in the
        // next, depending on where the scheduler had put the 16 workers
        cores_.clear();
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET(c, &set)) continue;
            if (!CPU_ISSET(c, &allowed_)) {     // outside the process's mask
                CPU_CLR(c, &set);
                continue;
            }
            snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/"
                     "topology/thread_siblings_list", c);
            int first = c;
            if (FILE *g = fopen(path, "r")) {
                if (fscanf(g, "%d", &first) != 1) first = c;
                fclose(g);
            }
            if (first == c || !CPU_ISSET(first, &allowed_))
                cores_.push_back(c);
        }
        if (CPU_COUNT(&set) == 0) return;   // none of that node's CPUs is ours
        node_ = node;
        cpus_ = set;
        have_cpus_ = true;
        want_threads_ = std::max(want_threads_, threads());
        resize(want_threads_);
    }

    void resize(int n) {
        std::lock_guard<std::recursive_mutex> lk(busy_);
        stop();
        n = std::max(1, std::min(n, 256));
        if (!widen_) n = std::min(n, std::max(1, CPU_COUNT(&allowed_)));
        quit_ = false;
        // the epoch the new workers have seen is fixed HERE, by the thread
        // that also starts the jobs: a worker that read it on its own could
        // start late, after the first job was posted, and sleep through it
        // never more placed workers than the node has physical cores of
        // ours: two on one core share its load / store bandwidth, and a
        // handful of well-placed threads beat sixteen badly placed ones
        if (have_cpus_ && !cores_.empty()) {
            const long long share = std::max<long long>(
                1, (long long)cores_.size()/local_ranks().second);
            n = (int)std::min<long long>(n, share);
        }
        const unsigned long long seen = epoch_;
        worker_cpu_.assign((size_t)n, -1);
        for (int t = 0; t < n; ++t)
            workers_.emplace_back([this, t, n, seen] { work(t, n, seen); });
    }

    // -- placement that is verified, not assumed ------------------------------
    // The NUMA node of the caller's vector comes from get_mempolicy, which a
    // container's seccomp profile may refuse and which says nothing about how
    // the box's fabric treats that node: on one box of round 4 the default
    // placement finished the scatter 3.4 ms after the last DMA chunk (7.95 ms
    // per Jacobian, against 4.57 ms on another).  So the pool measures: a
    // call whose scatter ends late (`lag`: time after the last chunk landed)
    // twice in a row makes the pool try every NUMA node that has CPUs of ours
    // -- and the unplaced pool -- for one call each, and keep the best for
    // this vector.
    void target(const void *vector, int policy_node) {
        std::lock_guard<std::recursive_mutex> lk(busy_);
        if (vector != vector_) {
            vector_ = vector;
            cand_.clear();
            exploring_ = -1;
            settled_ = false;
            bad_streak_ = calls_ = sample_ = searches_ = 0;
            want_threads_ = std::max(want_threads_, threads());
        }
        if (exploring_ < 0 && !settled_) set_numa_node(policy_node);
    }

    void feedback(double dma_ms, double lag_ms) {
        std::lock_guard<std::recursive_mutex> lk(busy_);
        static const bool trace = getenv("OPTY_HIP_TRACE") != nullptr;
        static const bool off = [] {
            const char *e = getenv("OPTY_HIP_HOST_PLACEMENT");
            return e && strcmp(e, "fixed") == 0;
        }();
        if (off) return;
        ++calls_;
        // the first calls with a vector fault its pages in and fill the
        // staging buffers: not measurements
        if (calls_ <= 3) return;
        const bool bad = lag_ms > std::max(0.6, 0.15*dma_ms);
        if (exploring_ < 0) {
            if (settled_) {
                // keep watching: a placement that was right may stop being
                // so (another process took those cores); at most two more
                // searches per vector
                if (lag_ms <= std::max(0.6, 2.0*best_lag_)) {
                    bad_streak_ = 0;
                    return;
                }
                if (++bad_streak_ < 5 || searches_ >= 3) return;
                settled_ = false;
            } else {
                if (!bad) { bad_streak_ = 0; return; }
                if (++bad_streak_ < 2) return;
            }
            // every placement is measured afresh, the current one included:
            // each NUMA node that has CPUs of ours, and the unplaced pool
            ++searches_;
            bad_streak_ = 0;
            cand_.clear();
            cand_.push_back({node_, -1.0});
            for (int node = 0; node < 64; ++node) {
                char path[96];
                snprintf(path, sizeof path,
                         "/sys/devices/system/node/node%d/cpulist", node);
                if (node != node_ && access(path, R_OK) == 0)
                    cand_.push_back({node, -1.0});
            }
            if (node_ >= 0) cand_.push_back({-1, -1.0});   // unplaced
            exploring_ = 0;
            sample_ = 0;
            return;         // the next call measures candidate 0 as it is
        }
        // two calls per candidate, the better one counts (one late chunk or a
        // descheduled worker must not decide)
        auto &cur = cand_[(size_t)exploring_];
        cur.second = cur.second < 0 ? lag_ms : std::min(cur.second, lag_ms);
        if (++sample_ < 2) return;
        sample_ = 0;
        ++exploring_;
        while (exploring_ < (int)cand_.size()) {
            if (place(cand_[(size_t)exploring_].first)) return;
            cand_[(size_t)exploring_].second = 1e9;     // no CPUs of ours there
            ++exploring_;
        }
        size_t best = 0;
        for (size_t k = 1; k < cand_.size(); ++k)
            if (cand_[k].second >= 0 && cand_[k].second < cand_[best].second)
                best = k;
        place(cand_[best].first);
        best_lag_ = cand_[best].second;
        settled_ = true;
        exploring_ = -1;
        if (trace) {
            fprintf(stderr, "opty_hip: scatter placement settled on NUMA node "
                    "%d after measuring:", cand_[best].first);
            for (auto &c : cand_)
                fprintf(stderr, " node %d: +%.2f ms;", c.first, c.second);
            fprintf(stderr, "\n");
        }
    }

    bool settled() const { return settled_; }

    void request_threads(int n) {
        std::lock_guard<std::recursive_mutex> lk(busy_);
        want_threads_ = n;
        resize(n);
    }

    // "worker -> cpu" of the last job (OPTY_HIP_TRACE)
    void report(FILE *f) const {
        fprintf(f, "opty_hip: scatter workers (node %d):", node_);
        for (size_t t = 0; t < worker_cpu_.size(); ++t)
            fprintf(f, " %zu->cpu%d", t, worker_cpu_[t]);
        fprintf(f, "\n");
    }

private:
    // workers on `node` (-1: unplaced, the creating thread's mask); false
    // when that node has no CPUs of ours
    bool place(int node) {
        if (node < 0) {
            node_ = -1;
            have_cpus_ = false;
            cores_.clear();
            resize(std::max(want_threads_, 1));
            return true;
        }
        const int before = node_;
        node_ = -2;                 // force set_numa_node to act
        set_numa_node(node);
        if (node_ != node) { node_ = before; return false; }
        return true;
    }
public:

    // The caller publishes chunks [0, c) as landed with ready(c) and finally
    // waits for the workers.
    void start(const Job &job) {
        // one job at a time: handles used from different host threads share
        // the pool (released by wait())
        busy_.lock();
        quiesce();
        job_ = job;
        open_ = true;
        ready_.store(0, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        // scatter jobs hand their nodes out in slices (see work())
        slices_ = std::min(SLICES_PER_WORKER*threads(), MAX_SLICES);
        // opt-in (OPTY_HIP_SCATTER_SLICES=1): on the hosts this could be
        // measured on it bought nothing and cost the pruned layout 0.5-1 ms
        // (DESIGN.md 5.2); the default is one fixed share per worker
        static const bool slices = getenv("OPTY_HIP_SCATTER_SLICES") != nullptr;
        sliced_ = !job.rows_dst && job.chunks <= MAX_CHUNKS && slices;
        slices_done_.store(0, std::memory_order_relaxed);
        for (int c = 0; c < std::min(job.chunks, MAX_CHUNKS); ++c) {
            chunk_done_[c].store(0, std::memory_order_relaxed);
            slice_next_[c].store(0, std::memory_order_relaxed);
            if (sliced_)
                for (int k = 0; k < slices_; ++k)
                    slice_state_[c*MAX_SLICES + k].v.store(
                        0, std::memory_order_relaxed);
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            ++epoch_;
        }
        cv_.notify_all();
    }
    void ready(int chunks) { ready_.store(chunks, std::memory_order_release); }
    // A scatter job is over when every slice is in place -- whoever wrote it:
    // a worker that lost its core in the middle of a slice (the hosts are
    // shared: load averages of 40 were seen) is not waited for, the workers
    // that are done repeat slices that have been in flight for too long.  It
    // wakes up later and writes the same values once more; quiesce() keeps
    // the next job (and the next DMA into the staging buffer it reads) behind
    // it.
    void wait() {
        if (sliced_) {
            const int total = job_.chunks*slices_;
            while (slices_done_.load(std::memory_order_acquire) < total)
                std::this_thread::yield();
        } else {
            quiesce();
        }
        busy_.unlock();
    }
    // every worker has left the last job
    void quiesce() {
        std::lock_guard<std::recursive_mutex> lk(busy_);
        while (open_ && done_.load(std::memory_order_acquire) < threads())
            std::this_thread::yield();
        open_ = false;
    }

private:
    ScatterPool() : pid_(getpid()) {
        // The CPUs this pool may use: the affinity mask of the thread that
        // creates it (a taskset / cpuset / OpenMP binding the host
        // application chose is honoured: workers never run outside it and
        // there are never more workers than CPUs in it).
        // OPTY_HIP_HOST_AFFINITY=wide restores the round-3 behaviour for
        // applications whose binding pins only the calling thread
        // (OMP_PROC_BIND pins the thread that loads this library to one
        // core): all CPUs of the machine.
        CPU_ZERO(&allowed_);
        const char *mode = getenv("OPTY_HIP_HOST_AFFINITY");
        const bool wide = mode && strcmp(mode, "wide") == 0;
        if (wide || sched_getaffinity(0, sizeof allowed_, &allowed_) != 0 ||
            CPU_COUNT(&allowed_) == 0)
            for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &allowed_);
        widen_ = wide;
        resize(default_threads());
    }

    void stop() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
        workers_.clear();
        open_ = false;
    }

    static unsigned now_us() {
        return (unsigned)std::chrono::duration_cast<std::chrono::microseconds>(
            std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    // slice_state_[].v: 0 free, 1 done, else the time it was taken (odd)
    void finish_slice(int c, int k) {
        if (slice_state_[c*MAX_SLICES + k].v.exchange(
                1u, std::memory_order_acq_rel) != 1u)
            slices_done_.fetch_add(1, std::memory_order_release);
    }
    static void scatter_nodes(const Job &j, long long s0, long long s1) {
        if (j.seg_dst) {
            for (long long i = s0; i < s1; ++i) {
                const double *src = j.seg_src + i*j.L0;
                double *dst = j.seg_dst + i*j.L1;
                for (int q = 0; q < j.L1; ++q) dst[q] = src[j.seg_pos[q]];
            }
            return;
        }
        for (long long i = s0; i < s1; ++i) {
            const double *src = j.packed + i*j.V;
            double *dst = j.dense + i*j.P;
            for (int r = 0; r < j.nruns; ++r) {
                memcpy(dst + j.run_start[r], src,
                       (size_t)j.run_len[r]*sizeof(double));
                src += j.run_len[r];
            }
            if (j.copy_scale)
                for (int q = 0; q < j.ncopies; ++q)
                    dst[j.copy_dst[q]] = j.copy_scale[q]*dst[j.copy_src[q]];
            else
                for (int q = 0; q < j.ncopies; ++q)
                    dst[j.copy_dst[q]] = dst[j.copy_src[q]];
        }
    }

    void work(int t, int T, unsigned long long seen) {
        // The workers run on the cores of the NUMA node that holds the
        // caller's dense vector (set_numa_node).  Without that information
        // they inherit the creating thread's mask -- unless it is narrower
        // than the pool: a host application that binds its OpenMP team
        // (OMP_PROC_BIND) pins the thread that loads this library to ONE
        // core, and sixteen workers on one core turn 5.9 ms into 40.
        cpu_set_t mask;
        const auto lr = local_ranks();
        const long long all = (long long)T*lr.second;   // workers of the node
        if (have_cpus_ && (long long)cores_.size() >= all) {
            // a core of its own, the workers of all local ranks spread evenly
            // over the node (its CCDs / memory channels)
            const long long n = (long long)cores_.size();
            const long long g = (long long)lr.first*T + t;
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cores_[(size_t)(((2*g + 1)*n/(2*all)) % n)], &one);
            (void)sched_setaffinity(0, sizeof one, &one);
        } else if (have_cpus_) {
            (void)sched_setaffinity(0, sizeof cpus_, &cpus_);
        } else if (widen_ &&
                   sched_getaffinity(0, sizeof mask, &mask) == 0 &&
                   CPU_COUNT(&mask) < T) {
            (void)sched_setaffinity(0, sizeof allowed_, &allowed_);
        }
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || epoch_ != seen; });
                if (quit_) return;
                seen = epoch_;
            }
            const Job j = job_;
            if (t < (int)worker_cpu_.size()) worker_cpu_[(size_t)t] = sched_getcpu();
            for (int c = 0; c < j.chunks; ++c) {
                // a chunk lands every ~0.3 ms: spin briefly, then give the
                // core away between polls
                for (unsigned spins = 0;
                     ready_.load(std::memory_order_acquire) <= c; ++spins) {
                    if (spins < 4096) cpu_relax();
                    else if (spins < 4096 + 64) std::this_thread::yield();
                    else std::this_thread::sleep_for(
                        std::chrono::microseconds(20));
                }
                const long long a = j.nodes*c/j.chunks,
                                b = j.nodes*(c + 1)/j.chunks;
                if (j.rows_dst) {
                    const long long c0 = j.bounds[c], c1 = j.bounds[c + 1];
                    const long long s0 = c0 + (c1 - c0)*t/T,
                                    s1 = c0 + (c1 - c0)*(t + 1)/T;
                    if (s1 > s0)
                        for (long long r = 0; r < j.rows; ++r)
                            memcpy(j.rows_dst + r*j.pitch + s0,
                                   j.rows_src + r*j.pitch + s0,
                                   (size_t)(s1 - s0)*sizeof(double));
                    if (c < MAX_CHUNKS)
                        chunk_done_[c].fetch_add(1,
                                                 std::memory_order_release);
                    continue;
                }
                if (!sliced_) {
                    scatter_nodes(j, a + (b - a)*t/T, a + (b - a)*(t + 1)/T);
                    continue;
                }
                // The nodes of a chunk in slices that the workers take from
                // a counter, not one fixed share each: a worker that loses
                // its core holds back one slice, not a sixteenth of every
                // chunk.
                const int S = slices_;
                for (int k = slice_next_[c].fetch_add(
                         1, std::memory_order_relaxed); k < S;
                     k = slice_next_[c].fetch_add(
                         1, std::memory_order_relaxed)) {
                    slice_state_[c*MAX_SLICES + k].v.store(
                        std::max(2u, now_us()), std::memory_order_relaxed);
                    scatter_nodes(j, a + (b - a)*k/S, a + (b - a)*(k + 1)/S);
                    finish_slice(c, k);
                }
            }
            // ... and slices that have been in flight for longer than a few
            // of them take are written again by whoever is done (the same
            // values from the same staging buffer)
            while (sliced_) {
                const int total = j.chunks*slices_;
                if (slices_done_.load(std::memory_order_acquire) >= total)
                    break;
                bool helped = false;
                const unsigned now = now_us();
                for (int c = 0; c < j.chunks; ++c) {
                    const long long a = j.nodes*c/j.chunks,
                                    b = j.nodes*(c + 1)/j.chunks;
                    for (int k = 0; k < slices_; ++k) {
                        const unsigned at = slice_state_[c*MAX_SLICES + k].v
                            .load(std::memory_order_relaxed);
                        if (at == 0u) {
                            // taken (the counter is past it) but not stamped
                            // yet: its age counts from now
                            unsigned zero = 0u;
                            slice_state_[c*MAX_SLICES + k].v
                                .compare_exchange_strong(
                                    zero, std::max(2u, now),
                                    std::memory_order_relaxed);
                            continue;
                        }
                        if (at == 1u || now - at < STALE_US) continue;
                        scatter_nodes(j, a + (b - a)*k/slices_,
                                      a + (b - a)*(k + 1)/slices_);
                        finish_slice(c, k);
                        helped = true;
                    }
                }
                // (look again in a while: the scan reads every stamp, and
                // the workers still at it are writing theirs)
                if (!helped)
                    for (int spin = 0; spin < 256 &&
                         slices_done_.load(std::memory_order_acquire) < total;
                         ++spin)
                        cpu_relax();
            }
            done_.fetch_add(1, std::memory_order_release);
        }
    }

public:
    static inline void pause() { cpu_relax(); }
private:
    static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        __asm__ __volatile__("yield");
#else
        std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
    }

    pid_t pid_;
    cpu_set_t allowed_;          // CPUs the workers may run on
    bool widen_ = false;         // OPTY_HIP_HOST_AFFINITY=wide
    int node_ = -1;
    bool have_cpus_ = false;
    cpu_set_t cpus_;
    std::vector<int> cores_;     // one CPU per physical core of that node
    std::vector<int> worker_cpu_;    // where each worker ran its last job
    const void *vector_ = nullptr;   // the vector the placement was chosen for
    std::vector<std::pair<int, double>> cand_;  // (node, lag) while exploring
    int exploring_ = -1, bad_streak_ = 0, want_threads_ = 0;
    int calls_ = 0, sample_ = 0, searches_ = 0;
    double best_lag_ = 0.0;
    bool settled_ = false;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::recursive_mutex busy_;   // a job, or a restart of the workers
    std::condition_variable cv_;
    unsigned long long epoch_ = 0;
    bool quit_ = false;
    Job job_;
    std::atomic<int> ready_{0}, done_{0};
public:
    static constexpr int MAX_CHUNKS = 64;
private:
    static constexpr int SLICES_PER_WORKER = 4, MAX_SLICES = 256;
    static constexpr unsigned STALE_US = 250;   // a slice takes 20-70 us
    std::atomic<int> chunk_done_[MAX_CHUNKS];
    std::atomic<int> slice_next_[MAX_CHUNKS];
    // (a cache line each: sixteen workers stamp their slices all the time)
    struct alignas(64) SliceState { std::atomic<unsigned> v{0}; };
    SliceState slice_state_[MAX_CHUNKS*MAX_SLICES];
    std::atomic<int> slices_done_{0};
    int slices_ = 0;
    bool sliced_ = false, open_ = false;
};

}  // namespace

// one list schedule of a persistent kernel (build_schedule below), on the
// device: per launch size
struct Schedule {
    long long nblk = -1;
    int *d_table = nullptr;
    int npw = 0;
};

// Which kernels the entry points launch for one launch size, measured on
// the device the handle lives on (opty_hip_desc.routing, calibrate_route).
struct Route {
    long long nblk = -1;
    bool fused_loses = false, jac_via_fused = false;
    float ms_fused = 0.f, ms_con = 0.f, ms_jac = 0.f;   // per launch
};

struct opty_hip_problem {
    std::vector<Schedule> sched_jac, sched_fused;
    std::vector<Route> routes;
    hipEvent_t ev_cal0 = nullptr, ev_cal1 = nullptr;
    opty_hip_desc d{};
    hipModule_t module = nullptr;
    hipFunction_t k_con = nullptr, k_jac = nullptr, k_conjac = nullptr,
                  k_inst = nullptr, k_uni = nullptr;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t copy_stream = nullptr;  // device-to-host side of a pipeline
    double *d_params = nullptr, *d_known = nullptr, *d_uni = nullptr;
    bool uni_dirty = true;   // node-invariant table needs (re)computing
    long long *d_inst_idx = nullptr, *d_inst_rows = nullptr,
              *d_inst_cols = nullptr;
    int *d_pattern = nullptr;   // (j, k) per stored block entry when pruned
    int *d_rowinfo = nullptr;   // (S_j, L_j) per stored block entry (CSR)
    double *d_free = nullptr, *d_con = nullptr, *d_jac = nullptr;  // staging
    double *d_con_scratch = nullptr;    // jac_via_fused: discarded values
    long long *d_rows = nullptr, *d_cols = nullptr;                // staging
    double h = 0.0;
    bool have_params = false, have_known = false, have_inst = false,
         have_h = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the last enqueued work
    // host-visible Jacobian by varying entries (opty_hip_eval_jac_persistent)
    std::vector<int> var_entries, run_start, run_len;
    std::vector<int> copy_dst, copy_src;  // opty_hip_set_entry_copies
    std::vector<double> copy_scale;       // ..._scaled (empty: plain copies)
    int *d_var = nullptr;
    double *d_packed = nullptr, *h_packed = nullptr;
    // page-locked, device-mapped staging of the latency path (eval_mapped)
    double *h_free = nullptr, *h_con = nullptr, *h_jac = nullptr;
    // OPTY_HIP_LAYOUT_SEGMENTED (opty_hip_set_segments): block entries in
    // stored order, the lengths of the three segments, for every entry of
    // segment 1 the position in segment 0 it repeats
    std::vector<int> seg_order, seg_copy_src;
    int seg_len[3] = {0, 0, 0};
    bool have_segments = false;
    int *d_seg_order = nullptr;
    double *d_dense = nullptr;   // node-major blocks the kernels write
    double *d_seg = nullptr;     // the same values in segmented order
    std::vector<hipEvent_t> chunk_events;
    size_t packed_cap = 0;                // doubles in d_packed / h_packed
    const double *static_host = nullptr;  // vector whose invariant entries
    bool static_valid = false;            // ... are up to date
    const double *shard_host = nullptr;   // the same for a node shard copied
    long long shard_begin = 0, shard_end = 0;   // by opty_hip_shard_jac_to_host
    bool shard_valid = false;

    int64_t ncon_nodes() const { return d.N - 1; }
    int64_t P() const { return (int64_t)d.P; }
    int64_t num_free() const { return (int64_t)(d.n + d.q)*d.N + d.r + d.s; }
    int64_t num_con() const { return (int64_t)d.M*ncon_nodes() + d.num_inst; }
    int64_t nnz() const { return P()*ncon_nodes() + d.nnz_inst; }
};

namespace {

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

struct NodeRange {
    long long begin, end, con_stride;
};

NodeRange whole(const opty_hip_problem *p) {
    return NodeRange{0, p->ncon_nodes(), p->ncon_nodes()};
}

// A handle's device state (node-invariant table, staging buffers) belongs to
// one stream at a time.  When the caller moved the handle to another stream
// (opty_hip_set_stream), work issued there is ordered after everything the
// handle enqueued on the previous one: opty_uni may overwrite the table that
// kernels of the previous stream still read, and the first fill has to be
// visible to the new stream.
// hipStreamSynchronize target of a handle's stream: the legacy handle is
// synchronised through the null stream it stands for.
hipStream_t sync_target(hipStream_t s) {
    return s == (hipStream_t)OPTY_HIP_STREAM_LEGACY ? nullptr : s;
}

template <typename Handle>
int order_streams(Handle *p) {
    if (p->last_stream && p->last_stream != p->stream) {
        // A switch is rare (set-up code, tests): wait for the old stream on
        // the host.  (An event recorded on hipStreamLegacy and waited for
        // on another stream crashed inside the runtime, ROCm 7.0.2; the
        // legacy handle is synchronised through the null stream it stands
        // for.)
        HIP_TRY(hipStreamSynchronize(sync_target(p->last_stream)));
    }
    p->last_stream = p->stream;
    return 0;
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
template <typename T>
int ensure(T **ptr, size_t count);

bool routing_enabled() {
    // OPTY_HIP_ROUTING=plan: the launch plan's flags as they are (A/B runs)
    const char *e = getenv("OPTY_HIP_ROUTING");
    return !(e && !strcmp(e, "plan"));
}

// Average duration (ms) of one issue of `fn` on the handle's stream: one
// untimed issue, then the best of three timed batches (hipEvents; batches
// long enough for the event resolution).
template <typename Fn>
int time_issue(opty_hip_problem *p, Fn fn, float *ms_out) {
    if (int rc = fn()) return rc;
    int n = 2;
    float best = 1e30f;
    for (int round = 0; round < 3; ++round) {
        HIP_TRY(hipEventRecord(p->ev_cal0, p->stream));
        for (int i = 0; i < n; ++i)
            if (int rc = fn()) return rc;
        HIP_TRY(hipEventRecord(p->ev_cal1, p->stream));
        HIP_TRY(hipEventSynchronize(p->ev_cal1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev_cal0, p->ev_cal1));
        if (ms/n < best) best = ms/n;
        if (round == 0 && ms < 0.2f) {
            const float per = ms/n > 1e-4f ? ms/n : 1e-4f;
            const int want = (int)(0.25f/per) + 1;
            n = want > 64 ? 64 : (want < n ? n : want);
        }
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

template <typename T>
int ensure(T **ptr, size_t count) {
    if (*ptr == nullptr && count > 0)
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(ptr), count*sizeof(T)));
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

template <typename T>
int ensure_pinned(T **ptr, size_t count) {
    if (*ptr == nullptr && count > 0)
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(ptr),
                             count*sizeof(T)));
    return 0;
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

// Upload of the trajectory rows of a HOST `free` vector in node windows
// (window w of W covers the constraint nodes [ncn*w/W, ncn*(w+1)/W) and needs
// the time nodes up to its last node + 1): 2-D copies of the rows' columns,
// from the caller's vector when it is page-locked, else from a page-locked
// staging vector that the host threads fill window by window (a pageable
// vector would make every 2-D copy a blocking, slow staging inside the
// runtime).  With W == 1 one plain copy.  The parameters / node time interval
// at the end of `free` go first.
class FreeUploader {
public:
    int begin(opty_hip_problem *p, const double *free_, int W) {
        p_ = p;
        W_ = W;
        N_ = p->d.N;
        rows_ = (long long)p->d.n + p->d.q;
        const long long rest = p->num_free() - rows_*N_;
        src_ = free_;
        if (W <= 1) {
            HIP_TRY(hipMemcpyAsync(p->d_free, free_,
                                   (size_t)p->num_free()*sizeof(double),
                                   hipMemcpyHostToDevice, p->stream));
            return 0;
        }
        if (rest > 0)
            HIP_TRY(hipMemcpyAsync(p->d_free + rows_*N_, free_ + rows_*N_,
                                   (size_t)rest*sizeof(double),
                                   hipMemcpyHostToDevice, p->stream));
        if (mapped_address(const_cast<double *>(free_)) != nullptr) return 0;
        if (int rc = ensure_pinned(&p->h_free, (size_t)p->num_free()))
            return rc;
        const long long ncn = p->ncon_nodes();
        bounds_.assign((size_t)W + 1, 0);
        for (int w = 0; w < W; ++w) bounds_[(size_t)w + 1] = ncn*(w + 1)/W + 1;
        // (the workers stay where the last scatter put them: restarting
        // them on another NUMA node costs more than a remote memcpy)
        pool_ = &ScatterPool::instance();
        ScatterPool::Job job;
        job.rows_src = free_;
        job.rows_dst = p->h_free;
        job.bounds = bounds_.data();
        job.rows = rows_;
        job.pitch = N_;
        job.chunks = W;
        pool_->start(job);
        pool_->ready(W);
        src_ = p->h_free;
        return 0;
    }
    // columns [c0, c1) of every row, for window w
    int window(int w, long long c0, long long c1) {
        if (W_ <= 1 || c1 <= c0) return 0;
        if (pool_) pool_->wait_chunk(w);
        HIP_TRY(hipMemcpy2DAsync(p_->d_free + c0, (size_t)N_*sizeof(double),
                                 src_ + c0, (size_t)N_*sizeof(double),
                                 (size_t)(c1 - c0)*sizeof(double),
                                 (size_t)rows_, hipMemcpyHostToDevice,
                                 p_->stream));
        return 0;
    }
    // the staging job holds the pool: release it before another job starts
    void end() {
        if (pool_) pool_->wait();
        pool_ = nullptr;
    }
    ~FreeUploader() { end(); }

private:
    opty_hip_problem *p_ = nullptr;
    ScatterPool *pool_ = nullptr;
    const double *src_ = nullptr;
    std::vector<long long> bounds_;
    long long N_ = 0, rows_ = 0;
    int W_ = 1;
};

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

int eval_segmented(opty_hip_problem *p, int what, const double *free_,
                   double *con, double *jac, int mem, bool full);

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

}  // namespace

extern "C" {

const char *opty_hip_last_error(void) { return g_error.c_str(); }

// ---------------------------------------------------------------------------
// objective / objective gradient
// ---------------------------------------------------------------------------
}  // extern "C"

struct opty_hip_objective {
    opty_hip_objective_desc d{};
    hipModule_t module = nullptr;
    hipFunction_t k_grad = nullptr, k_fin = nullptr;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the last enqueued work
    double *d_partial = nullptr, *d_value = nullptr;
    double *d_free = nullptr, *d_grad = nullptr;   // staging for host callers
    long long nblk = 0;
    int64_t num_free() const { return (int64_t)(d.n + d.q)*d.N + d.r; }
};

extern "C" {

int opty_hip_objective_create(const opty_hip_objective_desc *desc,
                              const char *code_object_path,
                              opty_hip_objective **out) {
    if (!desc || !code_object_path || !out) return fail("null argument");
    if (desc->N < 2) return fail("need at least 2 collocation nodes");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail("no HIP device is visible: the HIP backend has no CPU "
                    "fallback");
    if (desc->device < 0 || desc->device >= count)
        return fail("device %d out of range (have %d)", desc->device, count);
    HIP_TRY(hipSetDevice(desc->device));
    auto *o = new opty_hip_objective;
    o->d = *desc;
    hipError_t e = hipModuleLoad(&o->module, code_object_path);
    if (e != hipSuccess) {
        delete o;
        (void)hipGetLastError();
        return fail("hipModuleLoad(%s) failed: %s", code_object_path,
                    hipGetErrorString(e));
    }
    if (hipModuleGetFunction(&o->k_grad, o->module, "opty_objgrad") !=
            hipSuccess ||
        hipModuleGetFunction(&o->k_fin, o->module, "opty_objfin") !=
            hipSuccess) {
        (void)hipModuleUnload(o->module);
        delete o;
        return fail("opty_objgrad/opty_objfin missing from %s",
                    code_object_path);
    }
    auto allocate = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&o->own_stream,
                                         hipStreamNonBlocking));
        o->stream = o->own_stream;
        o->nblk = (desc->N + 63)/64;
        HIP_TRY(hipMalloc((void **)&o->d_partial,
                          (size_t)o->nblk*(1 + desc->r)*sizeof(double)));
        HIP_TRY(hipMalloc((void **)&o->d_value, sizeof(double)));
        return 0;
    };
    if (int rc = allocate()) {
        (void)opty_hip_objective_destroy(o);
        return rc;
    }
    *out = o;
    return 0;
}

int opty_hip_objective_destroy(opty_hip_objective *o) {
    if (!o) return 0;
    (void)hipSetDevice(o->d.device);
    (void)hipStreamSynchronize(sync_target(o->stream));
    void *bufs[] = {o->d_partial, o->d_value, o->d_free, o->d_grad};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (o->own_stream) (void)hipStreamDestroy(o->own_stream);
    if (o->module) (void)hipModuleUnload(o->module);
    delete o;
    return 0;
}

int opty_hip_objective_set_stream(opty_hip_objective *o, void *hip_stream) {
    if (!o) return fail("null handle");
    o->stream = hip_stream ? (hipStream_t)hip_stream : o->own_stream;
    return 0;
}

int opty_hip_objective_eval(opty_hip_objective *o, const double *free_,
                            double *value, double *grad, int32_t mem) {
    if (!o || !free_ || !value) return fail("null argument");
    HIP_TRY(hipSetDevice(o->d.device));
    // d_partial / d_value (and the staging buffers) may still be in use on
    // the stream the handle was on before opty_hip_objective_set_stream
    if (int rc = order_streams(o)) return rc;
    const double *dfree = free_;
    double *dgrad = grad;
    if (mem == OPTY_HIP_HOST) {
        if (int rc = ensure(&o->d_free, (size_t)o->num_free())) return rc;
        HIP_TRY(hipMemcpyAsync(o->d_free, free_, o->num_free()*sizeof(double),
                               hipMemcpyHostToDevice, o->stream));
        dfree = o->d_free;
        if (grad) {
            if (int rc = ensure(&o->d_grad, (size_t)o->num_free())) return rc;
            dgrad = o->d_grad;
        }
    } else if (mem != OPTY_HIP_DEVICE) {
        return fail("bad memory kind %d", mem);
    }
    KernelArgs a{};
    a.free_ = dfree;
    a.uni_w = o->d_value;
    a.con = o->d_partial;
    a.jac = dgrad;
    a.h = o->d.h;
    a.N = o->d.N;
    a.con_stride = o->nblk;
    a.node_begin = 0;
    a.node_end = o->d.N;
    size_t size = sizeof a;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a,
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &size,
                      HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(o->k_grad, (unsigned)o->nblk, 1, 1, 64, 1, 1,
                                  0, o->stream, nullptr, config));
    HIP_TRY(hipModuleLaunchKernel(o->k_fin, 1, 1, 1, 64, 1, 1, 0, o->stream,
                                  nullptr, config));
    HIP_TRY(hipMemcpyAsync(value, o->d_value, sizeof(double),
                           hipMemcpyDeviceToHost, o->stream));
    if (mem == OPTY_HIP_HOST && grad)
        HIP_TRY(hipMemcpyAsync(grad, o->d_grad, o->num_free()*sizeof(double),
                               hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(sync_target(o->stream)));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// plain matrix functions (ufuncify_matrix call shape)
// ---------------------------------------------------------------------------
struct opty_hip_matrix {
    opty_hip_matrix_desc d{};
    hipModule_t module = nullptr;
    hipFunction_t k_mat = nullptr, k_uni = nullptr;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the last enqueued work
    double *d_args = nullptr, *d_result = nullptr, *d_const = nullptr,
           *d_uni = nullptr;
    size_t args_cap = 0, result_cap = 0;    // doubles
    std::vector<double> last_const;
    bool uni_valid = false;
};

namespace {

int grow(double **ptr, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(ptr), need*sizeof(double)));
    *cap = need;
    return 0;
}

}  // namespace

extern "C" {

int opty_hip_matrix_create(const opty_hip_matrix_desc *desc,
                           const char *code_object_path,
                           opty_hip_matrix **out) {
    if (!desc || !code_object_path || !out) return fail("null argument");
    if (desc->num_vec < 0 || desc->num_const < 0 || desc->rows < 1 ||
        desc->cols < 1)
        return fail("bad matrix shape %d x %d (%d vector, %d const args)",
                    desc->rows, desc->cols, desc->num_vec, desc->num_const);
    if (desc->wgs_per_block < 1 || desc->waves_per_wg < 1 ||
        desc->waves_per_wg > 16)
        return fail("bad launch geometry (%d workgroups x %d waves)",
                    desc->wgs_per_block, desc->waves_per_wg);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail("no HIP device is visible: the HIP backend has no CPU "
                    "fallback");
    if (desc->device < 0 || desc->device >= count)
        return fail("device %d out of range (have %d)", desc->device, count);
    HIP_TRY(hipSetDevice(desc->device));
    auto *m = new opty_hip_matrix;
    m->d = *desc;
    hipError_t e = hipModuleLoad(&m->module, code_object_path);
    if (e != hipSuccess) {
        delete m;
        (void)hipGetLastError();
        return fail("hipModuleLoad(%s) failed: %s", code_object_path,
                    hipGetErrorString(e));
    }
    if (hipModuleGetFunction(&m->k_mat, m->module, "opty_jac") != hipSuccess ||
        (desc->num_uniform > 0 &&
         hipModuleGetFunction(&m->k_uni, m->module, "opty_uni") !=
             hipSuccess)) {
        (void)hipModuleUnload(m->module);
        delete m;
        return fail("opty_jac/opty_uni missing from %s", code_object_path);
    }
    auto allocate = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&m->own_stream,
                                         hipStreamNonBlocking));
        m->stream = m->own_stream;
        if (desc->num_const > 0)
            HIP_TRY(hipMalloc((void **)&m->d_const,
                              desc->num_const*sizeof(double)));
        if (desc->num_uniform > 0)
            HIP_TRY(hipMalloc((void **)&m->d_uni,
                              desc->num_uniform*sizeof(double)));
        return 0;
    };
    if (int rc = allocate()) {
        (void)opty_hip_matrix_destroy(m);
        return rc;
    }
    *out = m;
    return 0;
}

int opty_hip_matrix_destroy(opty_hip_matrix *m) {
    if (!m) return 0;
    (void)hipSetDevice(m->d.device);
    if (m->stream) (void)hipStreamSynchronize(sync_target(m->stream));
    void *bufs[] = {m->d_args, m->d_result, m->d_const, m->d_uni};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    if (m->module) (void)hipModuleUnload(m->module);
    delete m;
    return 0;
}

int opty_hip_matrix_set_stream(opty_hip_matrix *m, void *hip_stream) {
    if (!m) return fail("null handle");
    m->stream = hip_stream ? (hipStream_t)hip_stream : m->own_stream;
    return 0;
}

int opty_hip_matrix_eval(opty_hip_matrix *m, double *result,
                         const double *const *vec_args,
                         const double *const_args, int64_t n, int32_t mem) {
    if (!m || !result) return fail("null argument");
    if (n < 1) return fail("need at least one evaluation row, got %lld",
                           (long long)n);
    if (m->d.num_vec > 0 && !vec_args) return fail("null vector arguments");
    if (m->d.num_const > 0 && !const_args)
        return fail("null const arguments");
    if (mem != OPTY_HIP_HOST && mem != OPTY_HIP_DEVICE)
        return fail("bad memory kind %d", mem);
    for (int k = 0; k < m->d.num_vec; ++k)
        if (!vec_args[k]) return fail("vector argument %d is null", k);
    HIP_TRY(hipSetDevice(m->d.device));
    // the packed arguments, the const table and the node-invariant table may
    // still be in use on the stream the handle was on before
    // opty_hip_matrix_set_stream
    if (int rc = order_streams(m)) return rc;
    const size_t size = (size_t)m->d.rows*m->d.cols;
    // the kernel reads the vector arguments as the rows of ONE (num_vec, n)
    // array (what `free` is to the collocation kernels): pack them
    if (int rc = grow(&m->d_args, &m->args_cap,
                      (size_t)std::max(1, m->d.num_vec)*n)) return rc;
    const hipMemcpyKind in = mem == OPTY_HIP_HOST ? hipMemcpyHostToDevice
                                                  : hipMemcpyDeviceToDevice;
    for (int k = 0; k < m->d.num_vec; ++k)
        HIP_TRY(hipMemcpyAsync(m->d_args + (size_t)k*n, vec_args[k],
                               n*sizeof(double), in, m->stream));
    // const arguments: by value in the reference; the table of
    // sub-expressions that depend on them alone is refilled when they change
    bool changed = !m->uni_valid ||
                   m->last_const.size() != (size_t)m->d.num_const;
    for (int k = 0; !changed && k < m->d.num_const; ++k)
        changed = std::memcmp(&m->last_const[k], &const_args[k],
                              sizeof(double)) != 0;
    double *out = result;
    if (mem == OPTY_HIP_HOST) {
        if (int rc = grow(&m->d_result, &m->result_cap, size*n)) return rc;
        out = m->d_result;
    }
    KernelArgs a{};
    a.free_ = m->d_args;
    a.params = m->d_const;
    a.uni_c = m->d_uni;
    a.uni_w = m->d_uni;
    a.jac = out;
    a.h = 1.0;
    a.N = n;
    a.con_stride = n;
    a.node_begin = 0;
    a.node_end = n;
    size_t asize = sizeof a;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a,
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &asize,
                      HIP_LAUNCH_PARAM_END};
    if (changed) {
        if (m->d.num_const > 0) {
            // pageable source: the copy has consumed it when the call returns
            HIP_TRY(hipMemcpyAsync(m->d_const, const_args,
                                   m->d.num_const*sizeof(double),
                                   hipMemcpyHostToDevice, m->stream));
            m->last_const.assign(const_args, const_args + m->d.num_const);
        }
        if (m->d.num_uniform > 0)
            HIP_TRY(hipModuleLaunchKernel(m->k_uni, OPTY_UNI_WORKGROUPS, 1, 1,
                                          64, 1, 1, 0, m->stream, nullptr,
                                          config));
        m->uni_valid = true;
    }
    const long long nblk = ((n + 63)/64 + 7)/8*8;
    HIP_TRY(hipModuleLaunchKernel(m->k_mat,
                                  (unsigned)(nblk*m->d.wgs_per_block), 1, 1,
                                  64*m->d.waves_per_wg, 1, 1, 0, m->stream,
                                  nullptr, config));
    if (mem == OPTY_HIP_HOST) {
        HIP_TRY(hipMemcpyAsync(result, out, size*n*sizeof(double),
                               hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(sync_target(m->stream)));
    }
    return 0;
}

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

#include "opty_poison.inc"

int opty_hip_poison_registers(unsigned pattern) {
    hipLaunchKernelGGL(opty_poison, dim3(4096), dim3(64), 0, nullptr, pattern,
                       (unsigned *)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
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

int opty_hip_set_host_threads(int32_t count) {
    if (count < 0) return fail("thread count must be >= 0");
    ScatterPool &pool = ScatterPool::instance();
    pool.request_threads(count == 0 ? ScatterPool::default_threads() : count);
    return 0;
}

int opty_hip_host_threads(void) { return ScatterPool::instance().threads(); }

int opty_hip_host_placement(int32_t *workers_node, int32_t *device_node,
                            int32_t *verified) {
    ScatterPool &pool = ScatterPool::instance();
    if (workers_node) *workers_node = pool.numa_node();
    if (device_node) *device_node = device_numa_node();
    if (verified) *verified = pool.settled() ? 1 : 0;
    return 0;
}

int opty_hip_set_varying_entries(opty_hip_problem *p, const int32_t *entries,
                                 int32_t count) {
    if (!p) return fail("null handle");
    if (p->d.layout != OPTY_HIP_LAYOUT_COO)
        return fail("varying entries apply to the node-major layout only");
    if (count < 0 || count > p->d.P) return fail("bad entry count %d", count);
    if (count > 0 && !entries) return fail("null entries");
    for (int v = 0; v < count; ++v)
        if (entries[v] < 0 || entries[v] >= p->d.P ||
            (v > 0 && entries[v] <= entries[v - 1]))
            return fail("varying entries must ascend within [0, %d)", p->d.P);
    if (int rc = use_device(p)) return rc;
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    p->var_entries.assign(entries, entries + count);
    p->copy_dst.clear();
    p->copy_src.clear();
    p->copy_scale.clear();
    p->run_start.clear();
    p->run_len.clear();
    for (int v = 0; v < count; ++v) {
        if (v > 0 && entries[v] == entries[v - 1] + 1) {
            ++p->run_len.back();
        } else {
            p->run_start.push_back(entries[v]);
            p->run_len.push_back(1);
        }
    }
    if (p->d_var) (void)hipFree(p->d_var);
    p->d_var = nullptr;
    if (count > 0) {
        HIP_TRY(hipMalloc((void **)&p->d_var, count*sizeof(int)));
        HIP_TRY(hipMemcpy(p->d_var, entries, count*sizeof(int),
                          hipMemcpyHostToDevice));
    }
    p->static_valid = p->shard_valid = false;
    return 0;
}

int opty_hip_set_entry_copies(opty_hip_problem *p, const int32_t *dst,
                              const int32_t *src, int32_t count) {
    return opty_hip_set_entry_copies_scaled(p, dst, src, nullptr, count);
}

int opty_hip_set_entry_copies_scaled(opty_hip_problem *p, const int32_t *dst,
                                     const int32_t *src, const double *scale,
                                     int32_t count) {
    if (!p) return fail("null handle");
    if (count < 0 || count > p->d.P) return fail("bad copy count %d", count);
    if (count > 0 && (!dst || !src)) return fail("null entries");
    const std::vector<int> &var = p->var_entries;
    for (int k = 0; k < count; ++k) {
        if (dst[k] < 0 || dst[k] >= p->d.P ||
            (k > 0 && dst[k] <= dst[k - 1]))
            return fail("copied entries must ascend within [0, %d)", p->d.P);
        if (std::binary_search(var.begin(), var.end(), dst[k]))
            return fail("entry %d is moved as a varying entry: it cannot be "
                        "a copy as well", dst[k]);
        if (!std::binary_search(var.begin(), var.end(), src[k]))
            return fail("entry %d is copied from entry %d, which is not a "
                        "varying entry (opty_hip_set_varying_entries)",
                        dst[k], src[k]);
    }
    if (int rc = use_device(p)) return rc;
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    if (scale)
        for (int k = 0; k < count; ++k)
            if (!(scale[k] == scale[k]) || scale[k] - scale[k] != 0.0)
                return fail("scale of copied entry %d is not finite", dst[k]);
    p->copy_dst.assign(dst, dst + count);
    p->copy_src.assign(src, src + count);
    p->copy_scale.clear();
    if (scale && count > 0) p->copy_scale.assign(scale, scale + count);
    p->static_valid = p->shard_valid = false;
    return 0;
}

// NUMA node that holds the page of `addr` (get_mempolicy; -1 when unknown).
// The scatter workers run where the caller's dense vector lives: page-locked
// memory sits on the node of the thread that allocated it, which need not be
// the GPU's (measured on a 2-socket box, 792 MB vector on node 1: workers on
// node 1 6.4 ms, wherever the scheduler puts them 8.0, on node 0 12.6).
// OPTY_HIP_HOST_NUMA=<node> overrides, =off leaves the workers unplaced.
static int host_numa_node(const void *addr) {
    const char *env = getenv("OPTY_HIP_HOST_NUMA");
    if (env && strcmp(env, "off") == 0) return -1;
    if (env && *env) return atoi(env);
    int node = -1;
    // MPOL_F_NODE | MPOL_F_ADDR
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0UL, addr, 1UL | 2UL) != 0)
        return -1;
    return node;
}

// Moves the dense blocks of `count` nodes from device memory (d_blocks) into
// host memory (h_blocks, page-locked): all of it (`full`), or only the varying
// entries -- packed on the device, copied in chunks, scattered by the host
// threads while the next chunk is in flight.  Synchronous.
static bool packing_pays(const opty_hip_problem *p);

//
// `produce(a, b)`, when given with !full, enqueues the evaluation of the nodes
// [a, b) of d_blocks on the handle's stream: the nodes are then evaluated and
// packed in windows while the previous window crosses PCIe on a stream of its
// own (the evaluation + packing of the whole problem, 0.2 of 5 ms for the
// 10-link pendulum at N = 10^5, is no longer serial).
typedef std::function<int(long long, long long)> Producer;

// node windows of a host pipeline that moves `bytes` per call over `count`
// nodes (OPTY_HIP_HOST_WINDOWS overrides); never on the legacy stream: an
// event recorded there and waited for on another stream crashed inside the
// runtime (ROCm 7.0.2)
static int host_windows(const opty_hip_problem *p, size_t bytes,
                        long long count) {
    if (p->stream == (hipStream_t)OPTY_HIP_STREAM_LEGACY || bytes == 0)
        return 1;
    const char *env_w = getenv("OPTY_HIP_HOST_WINDOWS");
    int W = env_w ? std::max(1, std::min(64, atoi(env_w)))
                  : (bytes >= (32u << 20) ? 8 : 1);
    return (int)std::min<long long>(W, std::max<long long>(1, count/64));
}

static int move_blocks_to_host(opty_hip_problem *p, const double *d_blocks,
                               double *h_blocks, long long count, bool full,
                               const Producer &produce = Producer(),
                               int windows = 0) {
    const int V = (int)p->var_entries.size();
    const long long P = p->P();
    if (count <= 0) return 0;
    if (full) {
        HIP_TRY(hipMemcpyAsync(h_blocks, d_blocks,
                               (size_t)count*P*sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
        if (!p->copy_dst.empty() && packing_pays(p)) {
            // later calls fill the repeated entries from their sources: the
            // vector holds the same values from the first call on (the
            // kernels evaluate both copies, possibly a rounding apart)
            ScatterPool &pool = ScatterPool::instance();
            pool.set_numa_node(host_numa_node(h_blocks));
            ScatterPool::Job job;
            job.dense = h_blocks;
            job.copy_dst = p->copy_dst.data();
            job.copy_src = p->copy_src.data();
    job.copy_scale = p->copy_scale.empty() ? nullptr : p->copy_scale.data();
            job.copy_scale = p->copy_scale.empty() ? nullptr
                                                   : p->copy_scale.data();
            job.ncopies = (int)p->copy_dst.size();
            job.chunks = 1;
            job.P = P;
            job.nodes = count;
            pool.start(job);
            pool.ready(1);
            pool.wait();
        }
        return 0;
    }
    if (V == 0)                 // a block of constants: nothing moves
        return produce ? produce(0, count) : 0;
    const size_t packed = (size_t)count*V;
    if (packed > p->packed_cap) {
        HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
        if (p->d_packed) (void)hipFree(p->d_packed);
        if (p->h_packed) (void)hipHostFree(p->h_packed);
        p->d_packed = p->h_packed = nullptr;
        p->packed_cap = 0;
        HIP_TRY(hipMalloc((void **)&p->d_packed, packed*sizeof(double)));
        HIP_TRY(pinned_alloc((void **)&p->h_packed, packed*sizeof(double)));
        p->packed_cap = packed;
    }
    // chunks of about 16 MB: long enough for the DMA engine's full rate,
    // short enough that the host threads start early and finish soon after
    // the last byte has landed
    ScatterPool::instance().quiesce();      // nobody reads h_packed any more
    int chunks = (int)std::max<size_t>(1, std::min<size_t>(
        32, packed*sizeof(double)/(16u << 20)));
    chunks = (int)std::min<long long>(chunks, count);
    // windows (see `produce`); never on the legacy stream: an event
    // recorded there and waited for on another stream crashed inside the
    // runtime (ROCm 7.0.2)
    int W = 1;
    if (produce) {
        W = windows > 0 ? windows
                        : host_windows(p, packed*sizeof(double), count);
        chunks = std::max(chunks, W);
    }
    while ((int)p->chunk_events.size() < chunks + W) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->chunk_events.push_back(e);
    }
    if (W > 1 && !p->copy_stream)
        HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream,
                                         hipStreamNonBlocking));
    hipStream_t out = W > 1 ? p->copy_stream : p->stream;
    int next_chunk = 0;
    for (int w = 0; w < W; ++w) {
        const long long wa = count*w/W, wb = count*(w + 1)/W;
        if (produce)
            if (int rc = produce(wa, wb)) return rc;
        const long long part = (wb - wa)*V;
        const unsigned grid = (unsigned)std::min<long long>(
            (part + 255)/256, 8192);
        (void)hipGetLastError();
        hipLaunchKernelGGL(opty_pack_kernel, dim3(grid), dim3(256), 0,
                           p->stream, d_blocks + wa*P, p->d_packed + wa*V,
                           p->d_var, V, P, part);
        HIP_TRY(hipGetLastError());
        if (W > 1) {
            hipEvent_t ready = p->chunk_events[(size_t)chunks + (size_t)w];
            HIP_TRY(hipEventRecord(ready, p->stream));
            HIP_TRY(hipStreamWaitEvent(out, ready, 0));
        }
        while (next_chunk < chunks &&
               count*(next_chunk + 1)/chunks <= wb) {
            const long long a = count*next_chunk/chunks,
                            b = count*(next_chunk + 1)/chunks;
            HIP_TRY(hipMemcpyAsync(p->h_packed + a*V, p->d_packed + a*V,
                                   (size_t)(b - a)*V*sizeof(double),
                                   hipMemcpyDeviceToHost, out));
            HIP_TRY(hipEventRecord(p->chunk_events[(size_t)next_chunk], out));
            ++next_chunk;
        }
    }
    ScatterPool &pool = ScatterPool::instance();
    pool.target(h_blocks, host_numa_node(h_blocks));
    ScatterPool::Job job;
    job.packed = p->h_packed;
    job.dense = h_blocks;
    job.run_start = p->run_start.data();
    job.run_len = p->run_len.data();
    job.nruns = (int)p->run_start.size();
    job.copy_dst = p->copy_dst.data();
    job.copy_src = p->copy_src.data();
    job.copy_scale = p->copy_scale.empty() ? nullptr : p->copy_scale.data();
    job.ncopies = (int)p->copy_dst.size();
    job.V = V;
    job.chunks = chunks;
    job.P = P;
    job.nodes = count;
    // OPTY_HIP_TRACE=1: where the time of one call goes (stderr)
    static const bool trace = getenv("OPTY_HIP_TRACE") != nullptr;
    auto now = [] {
        return std::chrono::duration<double, std::milli>(
            std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t0 = now();
    pool.start(job);
    int rc = 0;
    double t_first = 0.0;
    for (int c = 0; c < chunks; ++c) {
        // OPTY_HIP_EVENT_WAIT=poll: query in a loop instead of the runtime's
        // wait (experiment: slow modes of this pipeline on shared hosts)
        static const bool poll = [] {
            const char *v = getenv("OPTY_HIP_EVENT_WAIT");
            return v && strcmp(v, "poll") == 0;
        }();
        hipError_t e;
        if (poll) {
            while ((e = hipEventQuery(p->chunk_events[c])) ==
                   hipErrorNotReady)
                ScatterPool::pause();
        } else {
            e = hipEventSynchronize(p->chunk_events[c]);
        }
        if (e != hipSuccess && rc == 0) {
            (void)hipGetLastError();
            rc = fail("hipEventSynchronize failed: %s", hipGetErrorString(e));
        }
        if (trace && c == 0) t_first = now();
        pool.ready(c + 1);      // also after a failure: the workers must end
    }
    const double t_dma = now();
    pool.wait();
    const double t_end = now();
    if (trace) {
        fprintf(stderr, "opty_hip: %lld nodes x %d entries in %d chunks: "
                "first chunk landed +%.2f ms, last +%.2f ms, scatter done "
                "+%.2f ms; %d threads on NUMA node %d (vector on node %d), "
                "caller on cpu %d\n", count, V, chunks, t_first - t0,
                t_dma - t0, t_end - t0, pool.threads(), pool.numa_node(),
                host_numa_node(h_blocks), sched_getcpu());
        pool.report(stderr);
    }
    // late scatter -> the pool looks for a better placement (one candidate
    // per call, then keeps the best for this vector)
    if (rc == 0) pool.feedback(t_dma - t0, t_end - t_dma);
    return rc;
}

static double pack_ratio() {
    static const double ratio = [] {
        const char *env = getenv("OPTY_HIP_PACK_RATIO");
        const double r = env ? atof(env) : 0.0;
        // measured (pruned 10-link block, 275 of 390 stored entries:
        // 5.94 ms whole blocks, 4.85 ms packed): the pack kernel and the
        // host scatter cost less than the bytes they save well beyond half
        return r > 0.0 && r <= 1.0 ? r : 0.8;
    }();
    return ratio;
}

// nothing to gain from packing: no table, or most of the block varies
static bool packing_pays(const opty_hip_problem *p) {
    return p->d.layout == OPTY_HIP_LAYOUT_COO && p->d_var != nullptr &&
           (double)p->var_entries.size() <= pack_ratio()*(double)p->P();
}

double opty_hip_pack_ratio(void) { return pack_ratio(); }

int opty_hip_eval_jac_persistent(opty_hip_problem *p, const double *free_,
                                 double *jac, int32_t fresh) {
    if (!p) return fail("null handle");
    if (!free_ || !jac) return fail("null buffer");
    if (int rc = use_device(p)) return rc;
    if (int rc = check_ready(p)) return rc;
    if (p->d.layout == OPTY_HIP_LAYOUT_SEGMENTED) {
        // the invariant segment stays in `jac` between calls
        const bool all = fresh != 0 || !p->static_valid ||
                         p->static_host != jac;
        p->static_valid = false;
        if (int rc = eval_segmented(p, OPTY_HIP_EVAL_JAC, free_, nullptr, jac,
                                    OPTY_HIP_HOST, all))
            return rc;
        p->static_host = jac;
        p->static_valid = true;
        return 0;
    }
    const long long P = p->P(), ncn = p->ncon_nodes();
    if (int rc = ensure(&p->d_free, (size_t)p->num_free())) return rc;
    if (int rc = ensure(&p->d_jac, (size_t)p->nnz())) return rc;
    if (int rc = order_streams(p)) return rc;
    // `fresh`: the caller's word that `jac` does not hold this handle's
    // invariant entries.  The address alone proves nothing -- a freed block
    // can come back from the allocator at the same address.
    const bool full = fresh != 0 || !p->static_valid ||
                      p->static_host != jac || !packing_pays(p);
    Producer produce;
    FreeUploader up;
    int W = 1, w = 0;
    if (full) {
        if (int rc = up.begin(p, free_, 1)) return rc;
        if (int rc = eval_device(p, OPTY_HIP_EVAL_JAC, p->d_free, nullptr,
                                 p->d_jac, whole(p), true))
            return rc;
    } else {
        // uploaded, evaluated and packed window by window inside
        // move_blocks_to_host; the instance tails (they read all of `free`
        // and the node-invariant table the first window fills) behind the
        // last window
        W = host_windows(p, p->var_entries.size()*(size_t)ncn*sizeof(double),
                         ncn);
        if (int rc = up.begin(p, free_, W)) return rc;
        produce = [p, P, ncn, &up, &w](long long a, long long b) {
            if (int rc = up.window(w++, a == 0 ? a : a + 1, b + 1)) return rc;
            if (int rc = eval_device(p, OPTY_HIP_EVAL_JAC, p->d_free, nullptr,
                                     p->d_jac + a*P, NodeRange{a, b, ncn},
                                     false))
                return rc;
            if (b == ncn) {
                up.end();       // the scatter job needs the host threads
                if (p->d.num_inst > 0)
                    return launch_instance(p, p->d_free, nullptr,
                                           p->d_jac + P*ncn);
            }
            return 0;
        };
    }
    if (int rc = move_blocks_to_host(p, p->d_jac, jac, ncn, full, produce,
                                     W)) {
        p->static_valid = false;
        return rc;
    }
    if (p->d.nnz_inst > 0)
        HIP_TRY(hipMemcpyAsync(jac + P*ncn, p->d_jac + P*ncn,
                               p->d.nnz_inst*sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    if (p->copy_stream) HIP_TRY(hipStreamSynchronize(p->copy_stream));
    p->static_host = jac;
    p->static_valid = true;
    return 0;
}

int opty_hip_shard_jac_to_host(opty_hip_problem *p, const double *d_jac_shard,
                               double *host_jac, int64_t node_begin,
                               int64_t node_end, int32_t fresh) {
    if (!p) return fail("null handle");
    if (!d_jac_shard || !host_jac) return fail("null buffer");
    if (node_begin < 0 || node_end < node_begin ||
        node_end > p->ncon_nodes())
        return fail("shard [%lld, %lld) outside the %lld constraint nodes",
                    (long long)node_begin, (long long)node_end,
                    (long long)p->ncon_nodes());
    if (p->d.layout != OPTY_HIP_LAYOUT_COO)
        return fail("only the node-major layout is node-sharded");
    if (int rc = use_device(p)) return rc;
    if (int rc = order_streams(p)) return rc;
    const bool full = fresh != 0 || !p->shard_valid ||
                      p->shard_host != host_jac ||
                      p->shard_begin != node_begin ||
                      p->shard_end != node_end || !packing_pays(p);
    if (int rc = move_blocks_to_host(p, d_jac_shard,
                                     host_jac + node_begin*p->P(),
                                     node_end - node_begin, full)) {
        p->shard_valid = false;
        return rc;
    }
    p->shard_host = host_jac;
    p->shard_begin = node_begin;
    p->shard_end = node_end;
    p->shard_valid = true;
    return 0;
}

int opty_hip_set_segments(opty_hip_problem *p, const int32_t *order,
                          const int32_t *seg_len,
                          const int32_t *copy_source) {
    if (!p || !order || !seg_len) return fail("null argument");
    if (p->d.layout != OPTY_HIP_LAYOUT_SEGMENTED)
        return fail("segments apply to OPTY_HIP_LAYOUT_SEGMENTED only");
    const int P = p->d.P;
    const int L0 = seg_len[0], L1 = seg_len[1], L2 = seg_len[2];
    if (L0 < 0 || L1 < 0 || L2 < 0 || L0 + L1 + L2 != P)
        return fail("segment lengths %d + %d + %d do not add up to the %d "
                    "entries of a block", L0, L1, L2, P);
    if (L1 > 0 && !copy_source) return fail("null copy sources");
    std::vector<char> seen((size_t)P, 0);
    for (int e = 0; e < P; ++e) {
        if (order[e] < 0 || order[e] >= P || seen[(size_t)order[e]])
            return fail("the stored order is not a permutation of the "
                        "block's %d entries", P);
        seen[(size_t)order[e]] = 1;
    }
    for (int k = 0; k < L1; ++k)
        if (copy_source[k] < 0 || copy_source[k] >= L0)
            return fail("entry %d of segment 1 repeats position %d, outside "
                        "segment 0 (%d entries)", k, copy_source[k], L0);
    if (int rc = use_device(p)) return rc;
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    p->seg_order.assign(order, order + P);
    p->seg_copy_src.assign(copy_source, copy_source + L1);
    p->seg_len[0] = L0;
    p->seg_len[1] = L1;
    p->seg_len[2] = L2;
    // (j, k) of every stored entry and the segment it lies in: what the
    // index kernel needs (rows / cols in the order of the values)
    std::vector<int32_t> jk(2*(size_t)P), info(2*(size_t)P);
    const int start[4] = {0, L0, L0 + L1, P};
    for (int sgm = 0; sgm < 3; ++sgm)
        for (int e = start[sgm]; e < start[sgm + 1]; ++e) {
            jk[2*(size_t)e] = order[e]/p->d.C;
            jk[2*(size_t)e + 1] = order[e]%p->d.C;
            info[2*(size_t)e] = start[sgm];
            info[2*(size_t)e + 1] = start[sgm + 1] - start[sgm];
        }
    if (int rc = ensure(&p->d_pattern, jk.size())) return rc;
    if (int rc = ensure(&p->d_rowinfo, info.size())) return rc;
    if (int rc = ensure(&p->d_seg_order, (size_t)P)) return rc;
    HIP_TRY(hipMemcpy(p->d_pattern, jk.data(), jk.size()*sizeof(int32_t),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_rowinfo, info.data(), info.size()*sizeof(int32_t),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_seg_order, order, (size_t)P*sizeof(int32_t),
                      hipMemcpyHostToDevice));
    p->have_segments = true;
    p->static_valid = p->shard_valid = false;
    return 0;
}

}  // extern "C"

namespace {

// gathers segment `sgm` of every node's block from the node-major vector
int pack_segment(opty_hip_problem *p, const double *dense, double *out,
                 int sgm, long long count) {
    const int start[3] = {0, p->seg_len[0], p->seg_len[0] + p->seg_len[1]};
    const int L = p->seg_len[sgm];
    const long long total = count*L;
    if (total <= 0) return 0;
    const unsigned grid = (unsigned)std::min<long long>((total + 255)/256,
                                                        8192);
    (void)hipGetLastError();
    hipLaunchKernelGGL(opty_pack_kernel, dim3(grid), dim3(256), 0, p->stream,
                       dense, out, p->d_seg_order + start[sgm], L, p->P(),
                       total);
    HIP_TRY(hipGetLastError());
    return 0;
}

// OPTY_HIP_LAYOUT_SEGMENTED: the kernels write the node-major blocks into a
// staging vector; the caller gets [segment 0 of all nodes | segment 1 of all
// nodes | segment 2 of all nodes | instance tail].  HOST: segment 0 (the
// entries that vary) crosses PCIe in chunks straight into the head of `jac`,
// segment 1 (entries that repeat one of segment 0) is filled from that head
// by the host threads while the next chunk is in flight, segment 2 (the
// node-invariant entries) moves only when `full`.
int eval_segmented(opty_hip_problem *p, int what, const double *free_,
                   double *con, double *jac, int mem, bool full) {
    if (!p->have_segments)
        return fail("the segments were never set (opty_hip_set_segments)");
    static const bool trace = getenv("OPTY_HIP_TRACE") != nullptr;
    auto now = [] {
        return std::chrono::duration<double, std::milli>(
            std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t_in = trace ? now() : 0.0;
    const bool want_con = what != OPTY_HIP_EVAL_JAC;
    const long long P = p->P(), ncn = p->ncon_nodes();
    const long long L0 = p->seg_len[0], L1 = p->seg_len[1];
    const size_t tail = (size_t)p->d.nnz_inst;
    if (int rc = ensure(&p->d_dense, (size_t)p->nnz())) return rc;
    if (mem == OPTY_HIP_DEVICE) {
        if (int rc = eval_device(p, what, free_, con, p->d_dense, whole(p),
                                 true))
            return rc;
        for (int sgm = 0, at = 0; sgm < 3; at += p->seg_len[sgm], ++sgm)
            if (int rc = pack_segment(p, p->d_dense, jac + (long long)at*ncn,
                                      sgm, ncn))
                return rc;
        if (tail)
            HIP_TRY(hipMemcpyAsync(jac + P*ncn, p->d_dense + P*ncn,
                                   tail*sizeof(double),
                                   hipMemcpyDeviceToDevice, p->stream));
        return 0;
    }
    if (int rc = ensure(&p->d_free, (size_t)p->num_free())) return rc;
    if (want_con)
        if (int rc = ensure(&p->d_con, (size_t)p->num_con())) return rc;
    if (int rc = ensure(&p->d_seg, (size_t)p->nnz())) return rc;
    if (int rc = order_streams(p)) return rc;
    double *dcon = want_con ? p->d_con : nullptr;
    // Windows of nodes: the upload of `free`, the evaluation and the packing
    // of window w + 1 run while window w crosses PCIe the other way (its own
    // stream).  What is serial is one window's upload + evaluation, not the
    // whole problem's (0.5 of 4.8 ms for the 10-link pendulum at N = 10^5).
    // (An event recorded on hipStreamLegacy and waited for on another stream
    // crashed inside the runtime, ROCm 7.0.2: one window there.)
    const size_t head_bytes = (size_t)L0*ncn*sizeof(double);
    const int W = host_windows(p, head_bytes, ncn);
    if (W > 1 && !p->copy_stream)
        HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream,
                                         hipStreamNonBlocking));
    hipStream_t out = W > 1 ? p->copy_stream : p->stream;
    // DMA chunks of about 16 MB: long enough for the engine's full rate,
    // short enough that the host threads start early and finish soon after
    // the last byte has landed
    int chunks = 0;
    if (L0 > 0) {
        chunks = (int)std::max<size_t>(1, std::min<size_t>(
            32, head_bytes/(16u << 20)));
        chunks = (int)std::min<long long>(chunks, ncn);
        chunks = std::max(chunks, W);
    }
    while ((int)p->chunk_events.size() < std::max(chunks, 1) + W) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->chunk_events.push_back(e);
    }
    FreeUploader up;
    if (int rc = up.begin(p, free_, W)) return rc;
    int next_chunk = 0;
    for (int w = 0; w < W; ++w) {
        const long long a = ncn*w/W, b = ncn*(w + 1)/W;
        // time nodes [a, b] of every trajectory row (one-node halo; the
        // first column of a later window is there already)
        if (int rc = up.window(w, w == 0 ? a : a + 1, b + 1)) return rc;
        const NodeRange rg{a, b, ncn};
        if (int rc = eval_device(p, what, p->d_free, dcon ? dcon + a : nullptr,
                                 p->d_dense + a*P, rg, false))
            return rc;
        if (full && p->seg_len[2] > 0)
            if (int rc = pack_segment(p, p->d_dense + a*P,
                                      p->d_seg + (L0 + L1)*ncn +
                                          a*p->seg_len[2], 2, b - a))
                return rc;
        if (L0 > 0)
            if (int rc = pack_segment(p, p->d_dense + a*P, p->d_seg + a*L0, 0,
                                      b - a))
                return rc;
        if (W > 1) {
            hipEvent_t ready = p->chunk_events[(size_t)std::max(chunks, 1) +
                                               (size_t)w];
            HIP_TRY(hipEventRecord(ready, p->stream));
            HIP_TRY(hipStreamWaitEvent(out, ready, 0));
        }
        // the chunks that end inside this window
        while (next_chunk < chunks &&
               ncn*(next_chunk + 1)/chunks <= b) {
            const long long ca = ncn*next_chunk/chunks,
                            cb = ncn*(next_chunk + 1)/chunks;
            HIP_TRY(hipMemcpyAsync(jac + ca*L0, p->d_seg + ca*L0,
                                   (size_t)(cb - ca)*L0*sizeof(double),
                                   hipMemcpyDeviceToHost, out));
            HIP_TRY(hipEventRecord(p->chunk_events[(size_t)next_chunk], out));
            ++next_chunk;
        }
    }
    up.end();
    // instance tails (they read the whole free vector), constraints, and --
    // `full` -- the node-invariant segment, behind the entries that vary
    if (p->d.num_inst > 0)
        if (int rc = launch_instance(
                p, p->d_free, want_con ? dcon + (long long)p->d.M*ncn
                                       : nullptr,
                p->d_dense + P*ncn))
            return rc;
    if (want_con)
        HIP_TRY(hipMemcpyAsync(con, p->d_con, p->num_con()*sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
    if (tail)
        HIP_TRY(hipMemcpyAsync(jac + P*ncn, p->d_dense + P*ncn,
                               tail*sizeof(double), hipMemcpyDeviceToHost,
                               p->stream));
    if (full && p->seg_len[2] > 0)
        HIP_TRY(hipMemcpyAsync(jac + (L0 + L1)*ncn,
                               p->d_seg + (L0 + L1)*ncn,
                               (size_t)p->seg_len[2]*ncn*sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
    const double t0 = trace ? now() : 0.0;
    double t_first = 0.0, t_last = 0.0;
    int rc = 0;
    ScatterPool *pool = nullptr;
    if (L1 > 0 && chunks > 0) {
        pool = &ScatterPool::instance();
        pool->set_numa_node(host_numa_node(jac));
        ScatterPool::Job job;
        job.seg_src = jac;
        job.seg_dst = jac + L0*ncn;
        job.seg_pos = p->seg_copy_src.data();
        job.L0 = (int)L0;
        job.L1 = (int)L1;
        job.chunks = chunks;
        job.nodes = ncn;
        pool->start(job);
    }
    for (int c = 0; c < chunks; ++c) {
        const hipError_t e = hipEventSynchronize(p->chunk_events[(size_t)c]);
        if (e != hipSuccess && rc == 0)
            rc = fail("hipEventSynchronize: %s", hipGetErrorString(e));
        if (trace && c == 0) t_first = now();
        if (pool) pool->ready(c + 1);   // (also after an error: frees them)
    }
    if (trace) t_last = now();
    if (pool) pool->wait();
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    if (W > 1) HIP_TRY(hipStreamSynchronize(out));
    if (trace)
        fprintf(stderr, "opty_hip: segmented, %lld nodes x %lld entries, %d "
                "windows, %d chunks%s: enqueued +%.2f ms, first chunk landed "
                "+%.2f, last +%.2f, all done +%.2f\n", ncn, L0, W, chunks,
                full ? " (+ invariant segment)" : "", t0 - t_in,
                t_first - t_in, t_last - t_in, now() - t_in);
    return 0;
}

}  // namespace

extern "C" {

int opty_hip_host_numa_node(const void *ptr) {
    if (!ptr) return -1;
    int node = -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0UL, ptr, 1UL | 2UL) != 0)
        return -1;
    return node;
}

int opty_hip_host_register(void *ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail("null argument");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    return 0;
}

int opty_hip_host_unregister(void *ptr) {
    if (!ptr) return 0;
    HIP_TRY(hipHostUnregister(ptr));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// The instruction-tape kernel: the referee of generated code objects.
//
// hipcc 7.2 has miscompiled generated kernels that sit at the edge of the
// register file: builds cannot vouch for each other.  This
// kernel evaluates the expression DAG itself (opty_amd/codegen/tape.py), one
// lane per node, every value in HBM (vals[slot*nodes + node], coalesced), a
// handful of registers and the same device math library as the generated
// code.  ConstraintCollocator._verify_build holds every code object at the
// register limit to it on the verification nodes before a handle exists.  It
// never supplies values a caller sees.
// ---------------------------------------------------------------------------
namespace {

enum TapeOp { T_ADD = 0, T_SUB, T_MUL, T_DIV, T_NEG, T_POWI, T_POW, T_MAX,
              T_MIN, T_ATAN2, T_SELECT, T_UNARY0 = 16 };
// order of opty_amd/codegen/ir.py UNARY
enum TapeUnary { U_SQRT = 0, U_SIN, U_COS, U_TAN, U_EXP, U_LOG, U_ABS, U_SIGN,
                 U_ASIN, U_ACOS, U_ATAN, U_SINH, U_COSH, U_TANH, U_STEP, U_ERF,
                 U_ERFC, U_FLOOR, U_CEIL, U_ASINH, U_ACOSH, U_ATANH, U_LOG1P,
                 U_EXPM1, U_LOG2, U_LOG10, U_EXP2, U_CBRT, U_TGAMMA, U_LGAMMA,
                 U_COUNT };

__device__ double tape_unary(int u, double x) {
    switch (u) {
    case U_SQRT: return sqrt(x);
    case U_SIN: return sin(x);
    case U_COS: return cos(x);
    case U_TAN: return tan(x);
    case U_EXP: return exp(x);
    case U_LOG: return log(x);
    case U_ABS: return fabs(x);
    case U_SIGN: return (double)((x > 0.0) - (x < 0.0));
    case U_ASIN: return asin(x);
    case U_ACOS: return acos(x);
    case U_ATAN: return atan(x);
    case U_SINH: return sinh(x);
    case U_COSH: return cosh(x);
    case U_TANH: return tanh(x);
    case U_STEP: return x > 0.0 ? 1.0 : 0.0;
    case U_ERF: return erf(x);
    case U_ERFC: return erfc(x);
    case U_FLOOR: return floor(x);
    case U_CEIL: return ceil(x);
    case U_ASINH: return asinh(x);
    case U_ACOSH: return acosh(x);
    case U_ATANH: return atanh(x);
    case U_LOG1P: return log1p(x);
    case U_EXPM1: return expm1(x);
    case U_LOG2: return log2(x);
    case U_LOG10: return log10(x);
    case U_EXP2: return exp2(x);
    case U_CBRT: return cbrt(x);
    case U_TGAMMA: return tgamma(x);
    default: return lgamma(x);
    }
}

// x^n the way the generated code multiplies it out (opty_device.h opty_powi)
// (even n: the square of x^(n/2); odd n: x times x^(n-1) -- which is the
// left-to-right binary method)
__device__ double tape_powi(double x, int n) {
    int bit = 31 - __builtin_clz((unsigned)n);
    double r = x;
    while (bit-- > 0) {
        r = r*r;
        if ((n >> bit) & 1) r = x*r;
    }
    return r;
}

__global__ void __launch_bounds__(64)
opty_tape_kernel(const int *__restrict__ code, long long ninstr,
                 double *vals, long long nodes) {
    const long long t = (long long)blockIdx.x*64 + threadIdx.x;
    if (t >= nodes) return;
    for (long long k = 0; k < ninstr; ++k) {
        const int *w = code + 8*k;        // wave-uniform: scalar loads
        const int op = w[0];
        const double a = vals[(long long)w[2]*nodes + t];
        double r;
        if (op >= T_UNARY0) {
            r = tape_unary(op - T_UNARY0, a);
        } else if (op == T_NEG) {
            r = -a;
        } else if (op == T_POWI) {
            r = tape_powi(a, w[6]);
        } else {
            const double b = vals[(long long)w[3]*nodes + t];
            switch (op) {
            case T_ADD: r = a + b; break;
            case T_SUB: r = a - b; break;
            case T_MUL: r = a*b; break;
            case T_DIV: r = a/b; break;
            case T_POW: r = pow(a, b); break;
            case T_MAX: r = fmax(a, b); break;
            case T_MIN: r = fmin(a, b); break;
            case T_ATAN2: r = atan2(a, b); break;
            default: {      // T_SELECT: (a rel b) ? c : d
                const int rel = w[6];
                const bool take = rel == 0 ? a < b : rel == 1 ? a <= b
                                : rel == 2 ? a == b : a != b;
                r = vals[(long long)(take ? w[4] : w[5])*nodes + t];
            }
            }
        }
        vals[(long long)w[1]*nodes + t] = r;
    }
}

}  // namespace

extern "C" {

int opty_hip_tape_run(int32_t device, const int32_t *code, int64_t ninstr,
                      double *vals, int64_t nslots, int64_t nodes) {
    if (!code || !vals || ninstr < 0 || nslots <= 0 || nodes <= 0)
        return fail("opty_hip_tape_run: bad argument");
    for (int64_t k = 0; k < ninstr; ++k) {
        const int32_t *w = code + 8*k;
        const int op = w[0];
        const bool unary = op >= T_UNARY0 && op < T_UNARY0 + U_COUNT;
        if (!unary && (op < 0 || op > T_SELECT))
            return fail("opty_hip_tape_run: instruction %lld has unknown "
                        "opcode %d", (long long)k, op);
        const int nsrc = unary || op == T_NEG || op == T_POWI ? 1
                       : op == T_SELECT ? 4 : 2;
        for (int s = 1; s <= 1 + nsrc; ++s)
            if (w[s] < 0 || w[s] >= nslots)
                return fail("opty_hip_tape_run: instruction %lld refers to "
                            "slot %d of %lld", (long long)k, w[s],
                            (long long)nslots);
        if (op == T_POWI && (w[6] < 1 || w[6] > 4096))
            return fail("opty_hip_tape_run: instruction %lld: exponent %d",
                        (long long)k, w[6]);
        if (op == T_SELECT && (w[6] < 0 || w[6] > 3))
            return fail("opty_hip_tape_run: instruction %lld: relation %d",
                        (long long)k, w[6]);
    }
    HIP_TRY(hipSetDevice(device));
    int *d_code = nullptr;
    double *d_vals = nullptr;
    const size_t cbytes = (size_t)ninstr*8*sizeof(int32_t);
    const size_t vbytes = (size_t)nslots*(size_t)nodes*sizeof(double);
    if (ninstr) HIP_TRY(hipMalloc(&d_code, cbytes));
    hipError_t e = hipMalloc(&d_vals, vbytes);
    if (e == hipSuccess && ninstr)
        e = hipMemcpy(d_code, code, cbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_vals, vals, vbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && ninstr) {
        hipLaunchKernelGGL(opty_tape_kernel, dim3((unsigned)((nodes + 63)/64)),
                           dim3(64), 0, 0, d_code, (long long)ninstr, d_vals,
                           (long long)nodes);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(vals, d_vals, vbytes, hipMemcpyDeviceToHost);
    if (d_code) (void)hipFree(d_code);
    if (d_vals) (void)hipFree(d_vals);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail("opty_hip_tape_run failed: %s", hipGetErrorString(e));
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Node-sharded problems over several GPUs: the RCCL side.
//
// The evaluation itself never communicates (every rank reads the global free
// vector in its own HBM and writes its node range); what does is (i) getting
// that vector to every rank and (ii) re-assembling the shards for a consumer
// that wants whole vectors on one GPU.  Both are RCCL calls on the problem
// handle's stream, issued by this library -- no PyTorch process group
// involved.  librccl is loaded on first use (dlopen): a single-GPU process
// never needs it.
// ---------------------------------------------------------------------------
#include <dlfcn.h>

namespace {

typedef struct ncclComm *rccl_comm_t;
struct RcclId { char internal[OPTY_HIP_COMM_ID_BYTES]; };
enum { RCCL_FLOAT64 = 8 };          // ncclDouble (rccl.h)

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(rccl_comm_t *, int, RcclId, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, rccl_comm_t,
                     hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, rccl_comm_t,
                hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    const char *names[] = {getenv("OPTY_HIP_RCCL_LIBRARY"), "librccl.so.1",
                           "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    // a copy that the process holds already (PyTorch bundles its own next
    // to its HIP runtime) comes first: one RCCL, one HIP runtime
    for (const char *n : {"librccl.so", "librccl.so.1"})
        if (!getenv("OPTY_HIP_RCCL_LIBRARY") &&
            (lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD))) break;
    for (const char *n : names)
        if (!lib && n && *n && (lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!lib)
        return fail("librccl.so could not be loaded (%s): node-sharded "
                    "problems need RCCL", dlerror());
    Rccl r;
    r.lib = lib;
#define OPTY_RCCL_SYM(field, name)                                            \
    *reinterpret_cast<void **>(&r.field) = dlsym(lib, name);                  \
    if (!r.field) return fail("librccl.so has no symbol %s", name)
    OPTY_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    OPTY_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    OPTY_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    OPTY_RCCL_SYM(Broadcast, "ncclBroadcast");
    OPTY_RCCL_SYM(Send, "ncclSend");
    OPTY_RCCL_SYM(Recv, "ncclRecv");
    OPTY_RCCL_SYM(GroupStart, "ncclGroupStart");
    OPTY_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    OPTY_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef OPTY_RCCL_SYM
    g_rccl = r;
    return 0;
}

#define RCCL_TRY(expr)                                                        \
    do {                                                                      \
        int r_ = (expr);                                                      \
        if (r_ != 0)                                                          \
            return fail("%s failed: %s", #expr, g_rccl.GetErrorString(r_));   \
    } while (0)

}  // namespace

struct opty_hip_comm {
    rccl_comm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    double *d_stage = nullptr;      // constraint blocks of the peers (root)
    size_t stage_cap = 0;           // doubles
};

extern "C" {

void *opty_hip_device_alloc(int32_t device, size_t bytes) {
    void *ptr = nullptr;
    if (hipSetDevice(device) != hipSuccess ||
        hipMalloc(&ptr, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        fail("hipMalloc of %zu bytes on device %d failed", bytes, device);
        return nullptr;
    }
    return ptr;
}

int opty_hip_device_free(void *ptr) {
    if (ptr) HIP_TRY(hipFree(ptr));
    return 0;
}

int opty_hip_memcpy(void *dst, const void *src, size_t bytes, int32_t kind) {
    if (kind < 0 || kind > 2) return fail("bad copy kind %d", kind);
    if (bytes && (!dst || !src)) return fail("null buffer");
    const hipMemcpyKind kinds[] = {hipMemcpyHostToDevice,
                                   hipMemcpyDeviceToHost,
                                   hipMemcpyDeviceToDevice};
    HIP_TRY(hipMemcpy(dst, src, bytes, kinds[kind]));
    return 0;
}

int opty_hip_comm_unique_id(void *id_out) {
    if (!id_out) return fail("null buffer");
    if (int rc = load_rccl()) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(static_cast<RcclId *>(id_out)));
    return 0;
}

int opty_hip_comm_create(const void *unique_id, int32_t rank, int32_t world,
                         int32_t device, opty_hip_comm **out) {
    if (!unique_id || !out) return fail("null argument");
    if (world < 1 || rank < 0 || rank >= world)
        return fail("rank %d outside a world of %d", rank, world);
    if (int rc = load_rccl()) return rc;
    HIP_TRY(hipSetDevice(device));
    RcclId id;
    memcpy(&id, unique_id, sizeof id);
    auto *c = new opty_hip_comm;
    c->rank = rank;
    c->world = world;
    c->device = device;
    int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) {
        delete c;
        return fail("ncclCommInitRank(rank %d of %d, device %d) failed: %s",
                    rank, world, device, g_rccl.GetErrorString(r));
    }
    *out = c;
    return 0;
}

int opty_hip_comm_destroy(opty_hip_comm *c) {
    if (!c) return 0;
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

int opty_hip_comm_rank(const opty_hip_comm *c) { return c ? c->rank : -1; }
int opty_hip_comm_world(const opty_hip_comm *c) { return c ? c->world : 0; }

int opty_hip_bcast_free(opty_hip_comm *c, opty_hip_problem *p,
                        double *free_dev, int32_t root) {
    if (!c || !p || !free_dev) return fail("null argument");
    if (root < 0 || root >= c->world)
        return fail("root %d outside a world of %d", root, c->world);
    if (int rc = use_device(p)) return rc;
    if (int rc = order_streams(p)) return rc;
    if (c->world == 1) return 0;
    RCCL_TRY(g_rccl.Broadcast(free_dev, free_dev, (size_t)p->num_free(),
                              RCCL_FLOAT64, root, c->comm, p->stream));
    return 0;
}

int opty_hip_gather_v(opty_hip_comm *c, opty_hip_problem *p,
                      const int64_t *bounds, const double *con_shard,
                      const double *jac_shard, double *con_global,
                      double *jac_global, int32_t root, int32_t what) {
    if (!c || !p || !bounds) return fail("null argument");
    if (root < 0 || root >= c->world)
        return fail("root %d outside a world of %d", root, c->world);
    if (what != OPTY_HIP_EVAL_CON && what != OPTY_HIP_EVAL_JAC &&
        what != OPTY_HIP_EVAL_PAIR)
        return fail("bad selector %d (OPTY_HIP_EVAL_CON / _JAC / _PAIR)",
                    what);
    if (p->d.layout != OPTY_HIP_LAYOUT_COO)
        return fail("only the node-major layout is node-sharded");
    const bool want_con = what != OPTY_HIP_EVAL_JAC;
    const bool want_jac = what != OPTY_HIP_EVAL_CON;
    const int64_t ncn = p->ncon_nodes(), M = p->d.M, P = p->P();
    if (bounds[0] != 0 || bounds[c->world] != ncn)
        return fail("bounds must run from 0 to the %lld constraint nodes",
                    (long long)ncn);
    for (int g = 0; g < c->world; ++g)
        if (bounds[g + 1] < bounds[g])
            return fail("bounds are not ascending at rank %d", g);
    const bool is_root = c->rank == root;
    if (is_root && ((want_con && !con_global) || (want_jac && !jac_global)))
        return fail("the root needs the global vectors");
    if (!is_root && ((want_con && !con_shard) || (want_jac && !jac_shard)))
        return fail("a sending rank needs its shard buffers");
    if (int rc = use_device(p)) return rc;
    if (int rc = order_streams(p)) return rc;
    const int64_t a = bounds[c->rank], b = bounds[c->rank + 1];
    hipStream_t st = p->stream;
    if (is_root) {
        // the root's own shard: copied unless it was evaluated in place
        // (null shard pointers, or pointers into the global vectors)
        if (want_jac && jac_shard && jac_shard != jac_global + a*P && b > a)
            HIP_TRY(hipMemcpyAsync(jac_global + a*P, jac_shard,
                                   (size_t)(b - a)*P*sizeof(double),
                                   hipMemcpyDeviceToDevice, st));
        if (want_con && con_shard && con_shard != con_global + a && b > a)
            HIP_TRY(hipMemcpy2DAsync(
                con_global + a, (size_t)ncn*sizeof(double), con_shard,
                (size_t)(b - a)*sizeof(double),
                (size_t)(b - a)*sizeof(double), (size_t)M,
                hipMemcpyDeviceToDevice, st));
        if (c->world == 1) return 0;
        // peers' constraint blocks land densely in staging (a message is
        // contiguous), then ONE strided copy per peer puts the M rows of a
        // block at j*(N-1) + a (equation-major, opty/direct_collocation.py:
        // 2446); their Jacobian slices land in place (node-major: contiguous)
        size_t need = 0;
        if (want_con)
            for (int g = 0; g < c->world; ++g)
                if (g != root) need += (size_t)M*(bounds[g + 1] - bounds[g]);
        if (need > c->stage_cap) {
            if (c->d_stage) HIP_TRY(hipFree(c->d_stage));
            c->d_stage = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_stage),
                              need*sizeof(double)));
            c->stage_cap = need;
        }
        RCCL_TRY(g_rccl.GroupStart());
        size_t off = 0;
        for (int g = 0; g < c->world; ++g) {
            if (g == root) continue;
            const int64_t ga = bounds[g], gb = bounds[g + 1];
            if (gb == ga) continue;
            if (want_jac)
                RCCL_TRY(g_rccl.Recv(jac_global + ga*P, (size_t)(gb - ga)*P,
                                     RCCL_FLOAT64, g, c->comm, st));
            if (want_con) {
                RCCL_TRY(g_rccl.Recv(c->d_stage + off, (size_t)M*(gb - ga),
                                     RCCL_FLOAT64, g, c->comm, st));
                off += (size_t)M*(gb - ga);
            }
        }
        RCCL_TRY(g_rccl.GroupEnd());
        off = 0;
        for (int g = 0; want_con && g < c->world; ++g) {
            if (g == root) continue;
            const int64_t ga = bounds[g], gb = bounds[g + 1];
            if (gb == ga) continue;
            HIP_TRY(hipMemcpy2DAsync(
                con_global + ga, (size_t)ncn*sizeof(double),
                c->d_stage + off, (size_t)(gb - ga)*sizeof(double),
                (size_t)(gb - ga)*sizeof(double), (size_t)M,
                hipMemcpyDeviceToDevice, st));
            off += (size_t)M*(gb - ga);
        }
        return 0;
    }
    if (b == a) return 0;
    RCCL_TRY(g_rccl.GroupStart());
    if (want_jac)
        RCCL_TRY(g_rccl.Send(jac_shard, (size_t)(b - a)*P, RCCL_FLOAT64, root,
                             c->comm, st));
    if (want_con)
        RCCL_TRY(g_rccl.Send(con_shard, (size_t)M*(b - a), RCCL_FLOAT64, root,
                             c->comm, st));
    RCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

}  // extern "C"
