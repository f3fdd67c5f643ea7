// host_scatter.cpp -- the host-buffer (cyipopt callback) path of
// libopty_hip.so: what the device computed reaches the caller's dense host
// vector -- packing of the varying entries on the device, chunked DMA into a
// page-locked staging vector, a pool of host threads (placed on the NUMA node
// of the vector / the device) that scatter the chunks while the next one is
// in flight; the scatter-free segmented layout; uploads of `free` in node
// windows.  The reference returns a persistent dense array
// (opty/direct_collocation.py:2814); this is how it is kept current.
#include "opty_internal.h"

using namespace opty;

namespace {

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
        // chunk c covers the nodes [chunk_bounds[c], chunk_bounds[c + 1]);
        // null: `chunks` equal parts of `nodes`
        const long long *chunk_bounds = nullptr;
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
    // r06: ... and then for every worker to have left the job
    // all the same.  A straggler that still wrote into the caller's vector
    // after the API call had returned was a use-after-free in waiting (the
    // reference-shaped callbacks hand back FRESH arrays, which the caller
    // may free at once); what the slices still buy is that the straggler's
    // remaining slice is at most one, not its whole share.
    void wait() {
        if (sliced_) {
            const int total = job_.chunks*slices_;
            while (slices_done_.load(std::memory_order_acquire) < total)
                std::this_thread::yield();
        }
        quiesce();
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
                const long long a = j.chunk_bounds ? j.chunk_bounds[c]
                                                   : j.nodes*c/j.chunks,
                                b = j.chunk_bounds ? j.chunk_bounds[c + 1]
                                                   : j.nodes*(c + 1)/j.chunks;
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
                    const long long a = j.chunk_bounds
                        ? j.chunk_bounds[c] : j.nodes*c/j.chunks,
                                    b = j.chunk_bounds
                        ? j.chunk_bounds[c + 1] : j.nodes*(c + 1)/j.chunks;
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
    // wb: W + 1 constraint-node bounds of the windows (null: equal parts)
    int begin(opty_hip_problem *p, const double *free_, int W,
              const long long *wb = nullptr) {
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
        for (int w = 0; w < W; ++w)
            bounds_[(size_t)w + 1] = (wb ? wb[w + 1] : ncn*(w + 1)/W) + 1;
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

}  // namespace

namespace opty {
void scatter_quiesce() { ScatterPool::instance().quiesce(); }
}  // namespace opty

extern "C" {

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

// OPTY_HIP_HOST_SCHEDULE=2 (r06 experiment -> default if it pays): windows and
// DMA chunks that are NOT equal parts.  A copy runs at 54.5 GB/s inside and
// costs 11-18 us between two of them whatever its size, the first byte can
// only leave once the first window is uploaded, evaluated and packed, and the
// call ends one chunk's scatter after the last byte: so a tiny first window
// (1 % of the nodes) with a chunk of its own, growing chunks (4 %, 15 %), 32 MB
// copies through the bulk, and a tail that halves down to 2 MB.
struct HostSchedule {
    std::vector<long long> windows, chunks;     // node bounds, [0] = 0
};

static bool host_schedule(const opty_hip_problem *p, long long count, int V,
                          HostSchedule *out) {
    static const int mode = [] {
        const char *v = getenv("OPTY_HIP_HOST_SCHEDULE");
        return v ? atoi(v) : 1;
    }();
    const double bytes = (double)count*V*sizeof(double);
    if (mode != 2 || bytes < (96u << 20) || count < 4096 ||
        p->stream == (hipStream_t)OPTY_HIP_STREAM_LEGACY)
        return false;
    const long long per_mb = (long long)((1 << 20)/(V*sizeof(double))) + 1;
    const double frac[] = {0.0, 0.01, 0.05, 0.20, 1.0};
    out->windows.clear();
    out->chunks.clear();
    for (double f : frac) out->windows.push_back((long long)(count*f));
    out->windows.back() = count;
    for (int w = 0; w <= 3; ++w) out->chunks.push_back(out->windows[(size_t)w]);
    // the bulk window: 32 MB copies, then 16, 8, 4, 2, 2
    long long at = out->windows[3];
    const long long tail_mb[] = {16, 8, 4, 2, 2};
    long long tail_nodes = 0;
    for (long long t : tail_mb) tail_nodes += t*per_mb;
    const long long big = 32*per_mb;
    while (count - at > tail_nodes + big/2) {
        const long long left = count - at - tail_nodes;
        at += left < big + big/2 ? left : big;
        out->chunks.push_back(at);
    }
    for (long long t : tail_mb) {
        at = std::min(count, at + t*per_mb);
        if (at > out->chunks.back()) out->chunks.push_back(at);
    }
    if (out->chunks.back() != count) out->chunks.push_back(count);
    return (int)out->chunks.size() - 1 <= ScatterPool::MAX_CHUNKS;
}

static int move_blocks_to_host(opty_hip_problem *p, const double *d_blocks,
                               double *h_blocks, long long count, bool full,
                               const Producer &produce = Producer(),
                               int windows = 0,
                               const HostSchedule *plan = nullptr) {
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
    // r06 A/B (profiles/r06_host_path.txt), both OFF by default because
    // neither paid: (i) tapering the LAST chunks (.. 1, 1, 1/2, 1/4, 1/4 of
    // a regular one: OPTY_HIP_HOST_TAPER=1) does not shorten the 0.26 ms
    // between the last byte and the last scattered value -- that lag is the
    // host threads running a constant distance behind the DMA engine, not
    // the last chunk's own scatter; (ii) two copy streams
    // (OPTY_HIP_COPY_STREAMS=2, below) share the link instead of filling
    // each other's gaps: the download takes 4.5 ms instead of 3.7.
    static const bool taper = [] {
        const char *v = getenv("OPTY_HIP_HOST_TAPER");
        return v && v[0] == '1';
    }();
    // windows (see `produce`); never on the legacy stream: an event
    // recorded there and waited for on another stream crashed inside the
    // runtime (ROCm 7.0.2)
    int W = 1;
    if (produce) {
        W = windows > 0 ? windows
                        : host_windows(p, packed*sizeof(double), count);
        chunks = std::max(chunks, W);
    }
    if (plan) {
        W = (int)plan->windows.size() - 1;
        chunks = (int)plan->chunks.size() - 1;
    }
    // chunk c covers the nodes [bound(c), bound(c + 1)); kept in the handle:
    // the scatter workers read it (nobody does right now: quiesce() above)
    std::vector<long long> &bound = p->chunk_bounds;
    bound.clear();
    if (plan) {
        bound = plan->chunks;
    } else {
        const bool tp = taper && chunks >= 4 &&
                        chunks + 2 <= ScatterPool::MAX_CHUNKS && produce;
        for (int c = 0; c <= chunks; ++c)
            bound.push_back(count*c/chunks);
        if (tp) {
            const long long a = bound[(size_t)chunks - 1], b = count;
            bound.pop_back();
            bound.push_back(a + (b - a)/2);
            bound.push_back(a + (b - a)*3/4);
            bound.push_back(b);
            chunks += 2;
        }
    }
    while ((int)p->chunk_events.size() < chunks + W) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->chunk_events.push_back(e);
    }
    if (W > 1 && !p->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream,
                                         hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream2,
                                         hipStreamNonBlocking));
    }
    // OPTY_HIP_COPY_STREAMS=2: the chunks alternate between two copy
    // streams (measured slower, see above)
    static const bool two = [] {
        const char *v = getenv("OPTY_HIP_COPY_STREAMS");
        return v && v[0] == '2';
    }();
    hipStream_t outs[2] = {W > 1 ? p->copy_stream : p->stream,
                           W > 1 && two ? p->copy_stream2
                                        : (W > 1 ? p->copy_stream : p->stream)};
    int next_chunk = 0;
    for (int w = 0; w < W; ++w) {
        const long long wa = plan ? plan->windows[(size_t)w] : count*w/W,
                        wb = plan ? plan->windows[(size_t)w + 1]
                                  : count*(w + 1)/W;
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
            HIP_TRY(hipStreamWaitEvent(outs[0], ready, 0));
            if (outs[1] != outs[0])
                HIP_TRY(hipStreamWaitEvent(outs[1], ready, 0));
        }
        while (next_chunk < chunks && bound[(size_t)next_chunk + 1] <= wb) {
            const long long a = bound[(size_t)next_chunk],
                            b = bound[(size_t)next_chunk + 1];
            hipStream_t out = outs[next_chunk & 1];
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
    job.chunk_bounds = bound.data();
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

static double trace_now_ms() {
    return std::chrono::duration<double, std::milli>(
        std::chrono::steady_clock::now().time_since_epoch()).count();
}

int opty_hip_eval_jac_persistent(opty_hip_problem *p, const double *free_,
                                 double *jac, int32_t fresh) {
    if (!p) return fail("null handle");
    if (!free_ || !jac) return fail("null buffer");
    static const bool trace_call = getenv("OPTY_HIP_TRACE") != nullptr;
    const double t_call = trace_call ? trace_now_ms() : 0.0;
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
    HostSchedule sched;
    const HostSchedule *plan = nullptr;
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
        if (host_schedule(p, ncn, (int)p->var_entries.size(), &sched)) {
            plan = &sched;
            W = (int)sched.windows.size() - 1;
        }
        if (int rc = up.begin(p, free_, W,
                              plan ? plan->windows.data() : nullptr))
            return rc;
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
    const double t_move = trace_call ? trace_now_ms() : 0.0;
    if (int rc = move_blocks_to_host(p, p->d_jac, jac, ncn, full, produce,
                                     W, plan)) {
        p->static_valid = false;
        return rc;
    }
    const double t_moved = trace_call ? trace_now_ms() : 0.0;
    if (p->d.nnz_inst > 0)
        HIP_TRY(hipMemcpyAsync(jac + P*ncn, p->d_jac + P*ncn,
                               p->d.nnz_inst*sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(sync_target(p->stream)));
    if (p->copy_stream) HIP_TRY(hipStreamSynchronize(p->copy_stream));
    p->static_host = jac;
    p->static_valid = true;
    if (trace_call)
        fprintf(stderr, "opty_hip: jacobian call: set-up %.3f ms (buffers, "
                "upload begin), pipeline %.3f ms, final synchronisation "
                "%.3f ms\n", t_move - t_call, t_moved - t_move,
                trace_now_ms() - t_moved);
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

namespace opty {

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

}  // namespace opty

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
