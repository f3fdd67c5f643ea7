// Developer micro-benchmark (GPU box): how fast can the node-VARYING entries
// of a dense-block Jacobian reach a dense page-locked HOST vector whose
// node-invariant entries are already in place?
//
//   hipcc --offload-arch=gfx950 -O3 tools/host_scatter_bench.hip -lpthread \
//       -o tools/host_scatter_bench.bin
//   tools/host_scatter_bench.bin tools/patterns/config3_varying.txt 99999
//
// Variants: (a) plain D2H of the dense vector and of a packed vector of the
// varying entries, (b) one hipMemcpy2DAsync per run of varying entries
// (pitch = row), (c) a kernel that stores the varying entries straight into
// the host-mapped dense vector (8-byte lanes following the entry table),
// from the dense device vector or from a packed one, (d) packed D2H in
// chunks overlapped with a threaded CPU scatter.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static double now() {
    return std::chrono::duration<double>(
        std::chrono::steady_clock::now().time_since_epoch()).count();
}

// one workgroup of 256 lanes per `npb` nodes; lane sweeps the V varying
// entries of a node: consecutive lanes -> consecutive table entries, i.e.
// consecutive addresses inside a run
template <bool PACKED>
__global__ void __launch_bounds__(256)
scatter_to_host(const double *__restrict__ src, double *__restrict__ dst,
                const int *__restrict__ entries, int V, long long P,
                long long nnodes, int npb) {
    const long long n0 = (long long)blockIdx.x*npb;
    for (int s = 0; s < npb; ++s) {
        const long long i = n0 + s;
        if (i >= nnodes) return;
        for (int v = threadIdx.x; v < V; v += 256) {
            const int e = entries[v];
            const double x = PACKED ? src[i*V + v] : src[i*P + e];
            __builtin_nontemporal_store(x, dst + i*P + e);
        }
    }
}

// the same flattened: every lane owns one (node, v) pair of a long list, so
// that a wave's 64 stores are 64 consecutive varying entries
__global__ void __launch_bounds__(256)
scatter_flat(const double *__restrict__ src, double *__restrict__ dst,
             const int *__restrict__ entries, int V, long long P,
             long long total) {
    long long g = (long long)blockIdx.x*256 + threadIdx.x;
    const long long stride = (long long)gridDim.x*256;
    for (; g < total; g += stride) {
        const long long i = g/V;
        const int v = (int)(g - i*V);
        const int e = entries[v];
        dst[i*P + e] = src[i*P + e];
    }
}

// 16-byte pieces where a run allows it: piece table (entry, width 1|2)
__global__ void __launch_bounds__(256)
scatter_pieces(const double *__restrict__ src, double *__restrict__ dst,
               const int *__restrict__ piece_e, const int *__restrict__ piece_w,
               int NP, long long P, long long nnodes, int npb) {
    const long long n0 = (long long)blockIdx.x*npb;
    for (int s = 0; s < npb; ++s) {
        const long long i = n0 + s;
        if (i >= nnodes) return;
        for (int v = threadIdx.x; v < NP; v += 256) {
            const int e = piece_e[v];
            if (piece_w[v] == 2) {
                const double2 x = *reinterpret_cast<const double2 *>(src + i*P + e);
                *reinterpret_cast<double2 *>(dst + i*P + e) = x;
            } else {
                dst[i*P + e] = src[i*P + e];
            }
        }
    }
}

__global__ void pack_kernel(const double *__restrict__ src,
                            double *__restrict__ packed,
                            const int *__restrict__ entries, int V,
                            long long P, long long total) {
    long long g = (long long)blockIdx.x*256 + threadIdx.x;
    const long long stride = (long long)gridDim.x*256;
    for (; g < total; g += stride) {
        const long long i = g/V;
        const int v = (int)(g - i*V);
        packed[g] = src[i*P + entries[v]];
    }
}

#include <atomic>
#include <thread>
struct Pool {
    std::vector<std::thread> th;
    std::atomic<int> ready{0}, done{0}, epoch{0}, quit{0};
    const double *packed; double *dense; const int *run_s; const int *run_l;
    int nruns, V, T, chunks; long long P, n;
    void work(int t) {
        int seen = 0;
        while (true) {
            while (epoch.load(std::memory_order_acquire) == seen) {
                if (quit.load()) return;
                __builtin_ia32_pause();
            }
            seen = epoch.load();
            for (int c = 0; c < chunks; ++c) {
                while (ready.load(std::memory_order_acquire) <= c) __builtin_ia32_pause();
                long long a = n*c/chunks, b = n*(c + 1)/chunks;
                long long i0 = a + (b - a)*t/T, i1 = a + (b - a)*(t + 1)/T;
                for (long long i = i0; i < i1; ++i) {
                    const double *sp = packed + i*V;
                    double *d = dense + i*P;
                    for (int r = 0; r < nruns; ++r) {
                        memcpy(d + run_s[r], sp, run_l[r]*8);
                        sp += run_l[r];
                    }
                }
            }
            done.fetch_add(1, std::memory_order_release);
        }
    }
};

struct Job {
    const double *packed; double *dense; const int *entries;
    int V; long long P, n0, n1;
};
static void *scatter_thread(void *arg) {
    Job *j = (Job *)arg;
    for (long long i = j->n0; i < j->n1; ++i) {
        const double *s = j->packed + i*j->V;
        double *d = j->dense + i*j->P;
        for (int v = 0; v < j->V; ++v) d[j->entries[v]] = s[v];
    }
    return nullptr;
}

int main(int argc, char **argv) {
    if (argc < 3) { printf("usage: pattern.txt nodes\n"); return 1; }
    FILE *f = fopen(argv[1], "r");
    int P, V;
    if (!f || fscanf(f, "%d %d", &P, &V) != 2) { printf("bad pattern\n"); return 1; }
    std::vector<int> ent(V);
    for (int k = 0; k < V; ++k) if (fscanf(f, "%d", &ent[k]) != 1) return 1;
    fclose(f);
    const long long n = atoll(argv[2]);
    const size_t dense_b = (size_t)n*P*8, packed_b = (size_t)n*V*8;
    int nruns = 0;
    for (int k = 0; k < V; ++k) if (k == 0 || ent[k] != ent[k - 1] + 1) ++nruns;
    printf("P %d varying %d (%d runs) nodes %lld: dense %.1f MB packed %.1f MB\n",
           P, V, nruns, n, dense_b/1e6, packed_b/1e6);
    double *d_dense, *d_packed, *h_dense, *h_packed;
    int *d_ent;
    CHECK(hipMalloc(&d_dense, dense_b));
    CHECK(hipMalloc(&d_packed, packed_b));
    CHECK(hipMalloc(&d_ent, V*sizeof(int)));
    CHECK(hipHostMalloc(&h_dense, dense_b, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_packed, packed_b, hipHostMallocDefault));
    CHECK(hipMemcpy(d_ent, ent.data(), V*sizeof(int), hipMemcpyHostToDevice));
    std::vector<double> init((size_t)n*P);
    for (size_t k = 0; k < init.size(); ++k) init[k] = (double)(k % 1000003);
    CHECK(hipMemcpy(d_dense, init.data(), dense_b, hipMemcpyHostToDevice));
    memset(h_dense, 0, dense_b);
    memset(h_packed, 0, packed_b);
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const long long total = n*V;
    pack_kernel<<<4096, 256, 0, st>>>(d_dense, d_packed, d_ent, V, P, total);
    CHECK(hipStreamSynchronize(st));

    auto timeit = [&](const char *label, double bytes, auto fn) {
        fn(); CHECK(hipStreamSynchronize(st));
        double best = 1e9, sum = 0; const int reps = 5;
        for (int r = 0; r < reps; ++r) {
            double t0 = now(); fn(); CHECK(hipStreamSynchronize(st));
            double t = now() - t0; best = t < best ? t : best; sum += t;
        }
        printf("%-52s min %7.3f ms  mean %7.3f ms  (%.1f GB/s of moved bytes)\n",
               label, best*1e3, sum/reps*1e3, bytes/best/1e9);
    };
    auto check = [&](const char *label) {
        long long bad = 0;
        for (long long i = 0; i < n; i += 997)
            for (int v = 0; v < V; ++v)
                if (h_dense[i*P + ent[v]] != init[i*P + ent[v]]) ++bad;
        if (bad) printf("  !! %s: %lld wrong entries\n", label, bad);
        for (long long i = 0; i < n; ++i)
            for (int v = 0; v < V; v += 7) h_dense[i*P + ent[v]] = -1.0;
    };

    timeit("D2H dense (today)", dense_b, [&] {
        CHECK(hipMemcpyAsync(h_dense, d_dense, dense_b, hipMemcpyDeviceToHost, st)); });
    timeit("D2H packed, contiguous", packed_b, [&] {
        CHECK(hipMemcpyAsync(h_packed, d_packed, packed_b, hipMemcpyDeviceToHost, st)); });
    check("warm");
    timeit("memcpy2D per run, dense -> dense", packed_b, [&] {
        for (int k = 0; k < V;) {
            int k1 = k + 1;
            while (k1 < V && ent[k1] == ent[k1 - 1] + 1) ++k1;
            CHECK(hipMemcpy2DAsync(h_dense + ent[k], (size_t)P*8, d_dense + ent[k],
                                   (size_t)P*8, (size_t)(k1 - k)*8, (size_t)n,
                                   hipMemcpyDeviceToHost, st));
            k = k1;
        }});
    check("memcpy2D");
    for (int npb : {1, 4, 16}) {
        char label[96];
        snprintf(label, sizeof label, "kernel -> mapped host, from dense, %d nodes/wg", npb);
        timeit(label, packed_b, [&] {
            scatter_to_host<false><<<(unsigned)((n + npb - 1)/npb), 256, 0, st>>>(
                d_dense, h_dense, d_ent, V, P, n, npb); });
        check(label);
        snprintf(label, sizeof label, "kernel -> mapped host, from packed, %d nodes/wg", npb);
        timeit(label, packed_b, [&] {
            scatter_to_host<true><<<(unsigned)((n + npb - 1)/npb), 256, 0, st>>>(
                d_packed, h_dense, d_ent, V, P, n, npb); });
        check(label);
    }
    for (int grid : {256, 1024, 4096}) {
        char label[96];
        snprintf(label, sizeof label, "kernel flat -> mapped host, grid %d", grid);
        timeit(label, packed_b, [&] {
            scatter_flat<<<grid, 256, 0, st>>>(d_dense, h_dense, d_ent, V, P, total); });
        check(label);
    }
    {   // 16-byte pieces
        std::vector<int> pe, pw;
        for (int k = 0; k < V;) {
            int k1 = k + 1;
            while (k1 < V && ent[k1] == ent[k1 - 1] + 1) ++k1;
            int e = ent[k], len = k1 - k;
            if (e & 1) { pe.push_back(e); pw.push_back(1); ++e; --len; }
            for (; len >= 2; len -= 2, e += 2) { pe.push_back(e); pw.push_back(2); }
            if (len) { pe.push_back(e); pw.push_back(1); }
            k = k1;
        }
        int *d_pe, *d_pw;
        CHECK(hipMalloc(&d_pe, pe.size()*4)); CHECK(hipMalloc(&d_pw, pw.size()*4));
        CHECK(hipMemcpy(d_pe, pe.data(), pe.size()*4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_pw, pw.data(), pw.size()*4, hipMemcpyHostToDevice));
        char label[96];
        snprintf(label, sizeof label, "kernel 16-byte pieces (%zu per node) -> mapped host", pe.size());
        timeit(label, packed_b, [&] {
            scatter_pieces<<<(unsigned)((n + 3)/4), 256, 0, st>>>(
                d_dense, h_dense, d_pe, d_pw, (int)pe.size(), P, n, 4); });
        check(label);
    }
    {   // (e) persistent pool, runs copied with memcpy, chunked D2H
        std::vector<int> rs, rl;
        for (int k = 0; k < V;) {
            int k1 = k + 1;
            while (k1 < V && ent[k1] == ent[k1 - 1] + 1) ++k1;
            rs.push_back(ent[k]); rl.push_back(k1 - k); k = k1;
        }
        for (int T : {8, 16, 24, 32, 48}) {
            for (int chunks : {8, 16, 32}) {
                Pool pool;
                pool.packed = h_packed; pool.dense = h_dense;
                pool.run_s = rs.data(); pool.run_l = rl.data();
                pool.nruns = (int)rs.size(); pool.V = V; pool.T = T;
                pool.chunks = chunks; pool.P = P; pool.n = n;
                for (int t = 0; t < T; ++t)
                    pool.th.emplace_back([&pool, t] { pool.work(t); });
                std::vector<hipEvent_t> ev(chunks);
                for (auto &evt : ev) CHECK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
                char label[96];
                snprintf(label, sizeof label, "pool: %d threads, %d chunks, SDMA packed + host scatter", T, chunks);
                timeit(label, packed_b, [&] {
                    pool.ready.store(0); pool.done.store(0);
                    for (int c = 0; c < chunks; ++c) {
                        long long a = n*c/chunks, b = n*(c + 1)/chunks;
                        CHECK(hipMemcpyAsync(h_packed + a*V, d_packed + a*V, (size_t)(b - a)*V*8,
                                             hipMemcpyDeviceToHost, st));
                        CHECK(hipEventRecord(ev[c], st));
                    }
                    pool.epoch.fetch_add(1, std::memory_order_release);
                    for (int c = 0; c < chunks; ++c) {
                        CHECK(hipEventSynchronize(ev[c]));
                        pool.ready.store(c + 1, std::memory_order_release);
                    }
                    while (pool.done.load(std::memory_order_acquire) < T) __builtin_ia32_pause();
                });
                check(label);
                pool.quit.store(1);
                for (auto &t : pool.th) t.join();
            }
        }
    }
    // (d) packed D2H in chunks + CPU scatter threads
    for (int threads : {16}) {
        for (int chunks : {1, 8}) {
            char label[96];
            snprintf(label, sizeof label, "packed D2H in %d chunk(s) + %d scatter threads", chunks, threads);
            std::vector<pthread_t> th(threads);
            std::vector<Job> jobs(threads);
            std::vector<hipEvent_t> ev(chunks);
            for (auto &evt : ev) CHECK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
            timeit(label, packed_b, [&] {
                for (int c = 0; c < chunks; ++c) {
                    long long a = n*c/chunks, b = n*(c + 1)/chunks;
                    CHECK(hipMemcpyAsync(h_packed + a*V, d_packed + a*V, (size_t)(b - a)*V*8,
                                         hipMemcpyDeviceToHost, st));
                    CHECK(hipEventRecord(ev[c], st));
                }
                for (int c = 0; c < chunks; ++c) {
                    long long a = n*c/chunks, b = n*(c + 1)/chunks;
                    CHECK(hipEventSynchronize(ev[c]));
                    for (int t = 0; t < threads; ++t) {
                        jobs[t] = Job{h_packed, h_dense, ent.data(), V, P,
                                      a + (b - a)*t/threads, a + (b - a)*(t + 1)/threads};
                        pthread_create(&th[t], nullptr, scatter_thread, &jobs[t]);
                    }
                    for (int t = 0; t < threads; ++t) pthread_join(th[t], nullptr);
                }});
            check(label);
        }
    }
    return 0;
}
