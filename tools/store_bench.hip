// Developer micro-benchmark (GPU box): what does the MI355X memory system
// sustain for the Jacobian's store pattern?  Each 64-lane workgroup owns 64
// consecutive "nodes" of ROW bytes each and writes them in chunks of SEG bytes
// per node (what opty_flush16 does), or fully contiguously.
//
//   hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o /tmp/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double vec2d __attribute__((ext_vector_type(2)));
// one 16-byte non-temporal store (global_store_dwordx4 ... nt), what the
// generated kernels' flush issues (there as buffer stores with nt | sc1)
__device__ inline void store16_nt(void *p, double a, double b) {
    vec2d v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<vec2d *>(p));
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// pattern 0: per chunk, lanes cover (64*16/SEG) nodes x SEG bytes per store
// pattern 1: contiguous 1 KB per store instruction over the block's region
template <int SEG, bool ALIGN>
__global__ void __launch_bounds__(64)
seg_store(double *out, long long row_doubles, long long nnodes, int lds_pad) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long long node0 = (long long)blockIdx.x*64;
    if (node0 >= nnodes) return;
    if (lds_pad < 0) lds[lane] = 1.0;           // keep the allocation alive
    constexpr int LPN = SEG/16;                 // lanes per node
    constexpr int NPS = 64/LPN;                 // nodes per store
    const int sub = lane % LPN, nsel = lane/LPN;
    const long long row_bytes = row_doubles*8;
    const long long nchunk = row_bytes/SEG;     // whole chunks only
    char *base = (char *)out + node0*row_bytes;
    for (long long c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int p = 0; p < 64/NPS; ++p) {
            const int nd = p*NPS + nsel;
            if (node0 + nd < nnodes) {
                long long off = nd*row_bytes + c*SEG + sub*16;
                if (ALIGN) {
                    // shift every node's window so that it starts on a
                    // 128-byte line (what a line-aligned flush would do)
                    long long start = nd*row_bytes + c*SEG;
                    long long shift = (128 - ((node0*row_bytes + start) & 127)) & 127;
                    off += shift;
                    if (off + 16 > 64*row_bytes) continue;
                }
                double2 v = make_double2((double)lane, (double)c);
                *reinterpret_cast<double2 *>(base + off) = v;
            }
        }
    }
}

// pattern 2: like the real kernel -- G waves per 64-node block, each writing
// a strip of every row in SEG-byte pieces whose boundaries are aligned to
// ALIGNB bytes (flat address), XCD-aware block placement.
template <int SEG, int ALIGNB>
__global__ void __launch_bounds__(64)
strip_store(double *out, long long row_doubles, long long nnodes, int G) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    if (G < 0) lds[lane] = 1.0;
    const long long xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long blk = (slot/G)*8 + xcd;
    const int g = (int)(slot % G);
    const long long node0 = blk*64;
    if (node0 >= nnodes) return;
    constexpr int LPN = SEG/16, NPS = 64/LPN;
    const int sub = lane % LPN, nsel = lane/LPN;
    const long long row_bytes = row_doubles*8;
    const long long strip = (row_bytes/G)/SEG*SEG;       // bytes per strip
    const long long s0 = g*strip, s1 = (g == G - 1) ? row_bytes : s0 + strip;
    char *base = (char *)out + node0*row_bytes;
    const long long region = 64*row_bytes;
    for (long long c = s0; c < s1; c += SEG) {
#pragma unroll
        for (int p = 0; p < 64/NPS; ++p) {
            const int nd = p*NPS + nsel;
            long long start = nd*row_bytes + c;
            // first ALIGNB boundary at or after the nominal start
            long long addr = (long long)base + start;
            long long al = (addr + ALIGNB - 1)/ALIGNB*ALIGNB - (long long)base;
            long long off = al + sub*16;
            if (node0 + nd < nnodes && off + 16 <= region) {
                double2 v = make_double2((double)lane, (double)c);
                *reinterpret_cast<double2 *>(base + off) = v;
            }
        }
    }
}

// pattern 2 with non-temporal stores: like the real kernel -- G waves per 64-node block, each writing
// a strip of every row in SEG-byte pieces whose boundaries are aligned to
// ALIGNB bytes (flat address), XCD-aware block placement.
template <int SEG, int ALIGNB>
__global__ void __launch_bounds__(64)
strip_store_nt(double *out, long long row_doubles, long long nnodes, int G) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    if (G < 0) lds[lane] = 1.0;
    const long long xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long blk = (slot/G)*8 + xcd;
    const int g = (int)(slot % G);
    const long long node0 = blk*64;
    if (node0 >= nnodes) return;
    constexpr int LPN = SEG/16, NPS = 64/LPN;
    const int sub = lane % LPN, nsel = lane/LPN;
    const long long row_bytes = row_doubles*8;
    const long long strip = (row_bytes/G)/SEG*SEG;       // bytes per strip
    const long long s0 = g*strip, s1 = (g == G - 1) ? row_bytes : s0 + strip;
    char *base = (char *)out + node0*row_bytes;
    const long long region = 64*row_bytes;
    for (long long c = s0; c < s1; c += SEG) {
#pragma unroll
        for (int p = 0; p < 64/NPS; ++p) {
            const int nd = p*NPS + nsel;
            long long start = nd*row_bytes + c;
            // first ALIGNB boundary at or after the nominal start
            long long addr = (long long)base + start;
            long long al = (addr + ALIGNB - 1)/ALIGNB*ALIGNB - (long long)base;
            long long off = al + sub*16;
            if (node0 + nd < nnodes && off + 16 <= region) {
                store16_nt(base + off, (double)lane, (double)c);
            }
        }
    }
}

__global__ void __launch_bounds__(64)
contig_store(double *out, long long row_doubles, long long nnodes, int lds_pad) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long long node0 = (long long)blockIdx.x*64;
    if (node0 >= nnodes) return;
    if (lds_pad < 0) lds[lane] = 1.0;
    long long nn = nnodes - node0 < 64 ? nnodes - node0 : 64;
    const long long bytes = nn*row_doubles*8;
    char *base = (char *)out + node0*row_doubles*8;
    for (long long off = lane*16; off + 16 <= bytes; off += 1024) {
        double2 v = make_double2((double)lane, (double)off);
        *reinterpret_cast<double2 *>(base + off) = v;
    }
}

// pattern 3: ONE chunk per short-lived workgroup: block b writes the SEG-byte
// pieces (line-aligned) of chunk (b % nchunk) of node block (b / nchunk).
template <int SEG>
__global__ void __launch_bounds__(64)
chunk_store(double *out, long long row_doubles, long long nnodes, int nchunk) {
    const int lane = threadIdx.x;
    const long long blk = blockIdx.x/nchunk;
    const int c = blockIdx.x % nchunk;
    const long long node0 = blk*64;
    if (node0 >= nnodes) return;
    constexpr int LPN = SEG/16, NPS = 64/LPN;
    const int sub = lane % LPN, nsel = lane/LPN;
    const long long row_bytes = row_doubles*8;
    char *base = (char *)out + node0*row_bytes;
    const long long region = 64*row_bytes;
#pragma unroll
    for (int p = 0; p < 64/NPS; ++p) {
        const int nd = p*NPS + nsel;
        long long start = nd*row_bytes + (long long)c*SEG;
        long long addr = (long long)base + start;
        long long al = (addr + 127)/128*128 - (long long)base;
        long long off = al + sub*16;
        if (node0 + nd < nnodes && off + 16 <= region) {
            double2 v = make_double2((double)lane, (double)c);
            *reinterpret_cast<double2 *>(base + off) = v;
        }
    }
}

// pattern 4 (r04): PERSISTENT grid, work taken from an atomic ticket in
// address order.  Ticket t -> node block t / G, strip t % G: the resident
// waves always cover the lowest unfinished window of the output (R resident
// workgroups -> a window of about R / G node blocks).  One counter for the
// chip, or one per XCD (xcd = blockIdx.x % 8 takes the blocks 8k + xcd).
template <int SEG, bool PER_XCD>
__global__ void __launch_bounds__(64)
ticket_store(double *out, long long row_doubles, long long nnodes, int G,
             unsigned *tickets) {
    const int lane = threadIdx.x;
    const long long nblk = (nnodes + 63)/64;
    const long long row_bytes = row_doubles*8;
    const long long strip = (row_bytes/G)/SEG*SEG;
    constexpr int LPN = SEG/16, NPS = 64/LPN;
    const int sub = lane % LPN, nsel = lane/LPN;
    const int xcd = blockIdx.x & 7;
    __shared__ unsigned t_sh;
    for (;;) {
        if (lane == 0) t_sh = atomicAdd(tickets + (PER_XCD ? xcd*32 : 0), 1u);
        __syncthreads();
        const unsigned t = t_sh;
        __syncthreads();
        long long blk = t/G;
        const int g = (int)(t % G);
        if (PER_XCD) blk = blk*8 + xcd;
        if (blk >= nblk) return;
        const long long node0 = blk*64;
        const long long s0 = g*strip,
                        s1 = (g == G - 1) ? row_bytes : s0 + strip;
        char *base = (char *)out + node0*row_bytes;
        const long long region = 64*row_bytes;
        for (long long c = s0; c < s1; c += SEG) {
#pragma unroll
            for (int p = 0; p < 64/NPS; ++p) {
                const int nd = p*NPS + nsel;
                long long addr = (long long)base + nd*row_bytes + c;
                long long al = (addr + 127)/128*128 - (long long)base;
                long long off = al + sub*16;
                if (node0 + nd < nnodes && off + 16 <= region) {
                    store16_nt(base + off, (double)lane, (double)c);
                }
            }
        }
    }
}

// 256-thread workgroups, each streaming its own contiguous region (4 KB per
// workgroup-instruction)
__global__ void __launch_bounds__(256)
contig256(double2 *out, long long n2, long long per_block) {
    const long long b0 = (long long)blockIdx.x*per_block;
    long long e = b0 + per_block < n2 ? b0 + per_block : n2;
    for (long long i = b0 + threadIdx.x; i < e; i += 256)
        out[i] = make_double2(1.0, 2.0);
}

// short-lived 256-thread blocks, but block b writes the (b / S)-th 4 KB piece
// of stream (b % S): S interleaved long streams instead of one global sweep
__global__ void __launch_bounds__(256)
strided_fill(double2 *out, long long n2, int S) {
    const long long pieces = (n2 + 255)/256;          // 4 KB pieces
    const long long per_stream = (pieces + S - 1)/S;
    const long long piece = (blockIdx.x % S)*per_stream + blockIdx.x/S;
    const long long i = piece*256 + threadIdx.x;
    if (blockIdx.x/S < per_stream && i < n2) out[i] = make_double2(1.0, 2.0);
}

__global__ void __launch_bounds__(256)
strided_fill_nt(double2 *out, long long n2, int S) {
    const long long pieces = (n2 + 255)/256;
    const long long per_stream = (pieces + S - 1)/S;
    const long long piece = (blockIdx.x % S)*per_stream + blockIdx.x/S;
    const long long i = piece*256 + threadIdx.x;
    if (blockIdx.x/S < per_stream && i < n2) store16_nt(out + i, 1.0, 2.0);
}

// plain streaming fill with 64-thread blocks, one 1 KB store per block
__global__ void __launch_bounds__(64)
stream_fill64(double2 *out, long long n2) {
    long long i = (long long)blockIdx.x*64 + threadIdx.x;
    if (i < n2) out[i] = make_double2(1.0, 2.0);
}

// plain grid-stride fill, 256-thread blocks (the classic streaming write)
__global__ void __launch_bounds__(256)
stream_fill(double2 *out, long long n2) {
    long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x*blockDim.x;
    for (; i < n2; i += stride) out[i] = make_double2(1.0, 2.0);
}

template <typename F>
float time_ms(F launch, int iters = 20) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms/iters;
}

// usage: store_bench.bin [row doubles (990)] [nodes (99999)]: the strip
// pattern of a P = row block (the 24-link stand-in: 5100 49999)
int main(int argc, char **argv) {
    const long long arg_row = argc > 1 ? atoll(argv[1]) : 990;
    const long long nnodes = argc > 2 ? atoll(argv[2]) : 99999;
    double *out;
    const long long max_row = arg_row > 1024 ? arg_row + 8 : 1024;
    CHECK(hipMalloc(&out, nnodes*max_row*8 + 4096));
    const int grid = (int)((nnodes + 63)/64);
    for (long long row : {arg_row, arg_row + 2}) {
        const double gb = nnodes*row*8/1e9;
        for (int lds_kb : {36}) {
            const size_t lds = lds_kb*1024;
#define RUN(NAME, KERN)                                                       \
            { float ms = time_ms([&] { hipLaunchKernelGGL(KERN, dim3(grid),   \
                  dim3(64), lds, 0, out, row, nnodes, 0); });                 \
              printf("row=%lld lds=%2dKB %-22s %.4f ms  %7.0f GB/s\n", row,   \
                     lds_kb, NAME, ms, gb/ms*1e3); }
            RUN("contig", contig_store)
            RUN("seg128", (seg_store<128, false>))
            RUN("seg256", (seg_store<256, false>))
            RUN("seg512", (seg_store<512, false>))
            RUN("seg1024", (seg_store<1024, false>))
            RUN("seg256-linealigned", (seg_store<256, true>))
            RUN("seg512-linealigned", (seg_store<512, true>))
            fflush(stdout);
        }
    }
    {
        const long long row = arg_row;
        const double gb = nnodes*row*8/1e9;
        for (int G : {1, 4, 8, 20, 32}) {
            if (G > 8 && row < 2000) continue;
            const int nblk = (int)(((nnodes + 63)/64 + 7)/8*8);
            for (int lds_kb : {16, 36}) {
                const size_t lds = lds_kb*1024;
#define RUNS(NAME, KERN)                                                      \
                { float ms = time_ms([&] { hipLaunchKernelGGL(KERN,           \
                      dim3(nblk*G), dim3(64), lds, 0, out, row, nnodes, G); });\
                  printf("strips G=%d lds=%2dKB %-14s %.4f ms  %7.0f GB/s\n", \
                         G, lds_kb, NAME, ms, gb/ms*1e3); }
                RUNS("seg256/a128", (strip_store<256, 128>))
                RUNS("seg256/a256", (strip_store<256, 256>))
                RUNS("seg512/a128", (strip_store<512, 128>))
                RUNS("seg512/a512", (strip_store<512, 512>))
                RUNS("seg1024/a128", (strip_store<1024, 128>))
                RUNS("seg1024/a1024", (strip_store<1024, 1024>))
                fflush(stdout);
            }
        }
    }
    {
        const long long row = arg_row;
        const double gb = nnodes*row*8/1e9;
        const int nblk = (int)((nnodes + 63)/64);
        for (int seg : {256, 512, 1024}) {
            const int nchunk = (int)(row*8/seg);
            float ms;
            if (seg == 256) ms = time_ms([&] { hipLaunchKernelGGL((chunk_store<256>), dim3(nblk*nchunk), dim3(64), 0, 0, out, row, nnodes, nchunk); });
            else if (seg == 512) ms = time_ms([&] { hipLaunchKernelGGL((chunk_store<512>), dim3(nblk*nchunk), dim3(64), 0, 0, out, row, nnodes, nchunk); });
            else ms = time_ms([&] { hipLaunchKernelGGL((chunk_store<1024>), dim3(nblk*nchunk), dim3(64), 0, 0, out, row, nnodes, nchunk); });
            printf("chunk-per-block seg%d  %.4f ms %7.0f GB/s (%d blocks)\n", seg, ms, gb/ms*1e3*(nchunk*seg)/(row*8.0), nblk*nchunk);
        }
    }
    {
        // r04: ticket-ordered persistent grids (see ticket_store): resident
        // workgroups R x strips per block G -> window of ~R/G node blocks
        const long long row = arg_row;
        const double gb = nnodes*row*8/1e9;
        unsigned *tickets;
        CHECK(hipMalloc(&tickets, 8*32*sizeof(unsigned)));
        for (int per_xcd = 0; per_xcd < 2; ++per_xcd)
            for (int G : {8, 20, 32, 64, 128})
                for (int R : {1024, 2048, 4096}) {
                    if (G > 32 && row < 2000) continue;
                    auto run = [&] {
                        CHECK(hipMemsetAsync(tickets, 0,
                                             8*32*sizeof(unsigned), 0));
                        if (per_xcd)
                            hipLaunchKernelGGL((ticket_store<256, true>),
                                               dim3(R), dim3(64), 0, 0, out,
                                               row, nnodes, G, tickets);
                        else
                            hipLaunchKernelGGL((ticket_store<256, false>),
                                               dim3(R), dim3(64), 0, 0, out,
                                               row, nnodes, G, tickets);
                    };
                    float ms = time_ms(run);
                    printf("ticket %s G=%-3d resident=%-4d seg256 nt  %.4f ms"
                           "  %7.0f GB/s\n", per_xcd ? "per-xcd" : "global ",
                           G, R, ms, gb/ms*1e3);
                    fflush(stdout);
                }
        CHECK(hipFree(tickets));
        // the static XCD-aware dispatch with the same non-temporal stores
        for (int G : {8, 20, 32}) {
            if (G > 8 && row < 2000) continue;
            const int nblk = (int)(((nnodes + 63)/64 + 7)/8*8);
            float ms = time_ms([&] { hipLaunchKernelGGL(
                (strip_store_nt<256, 128>), dim3(nblk*G), dim3(64), 36*1024, 0,
                out, row, nnodes, G); });
            printf("strips G=%d static dispatch seg256/a128 nt  %.4f ms  "
                   "%7.0f GB/s\n", G, ms, gb/ms*1e3);
        }
    }
    const long long n2 = nnodes*arg_row/2;
    for (int nb : {512, 2048, 8192}) {
        const long long per_block = (n2 + nb - 1)/nb;
        float ms = time_ms([&] { hipLaunchKernelGGL(contig256, dim3(nb), dim3(256), 0, 0, (double2 *)out, n2, per_block); });
        printf("contig256 blocks=%d (%.0f KB each) %.4f ms %7.0f GB/s\n", nb, per_block*16/1024.0, ms, n2*16/1e9/ms*1e3);
    }
    // bandwidth against the number of concurrently advancing address streams
    for (int S : {1, 4, 16, 64, 256, 512, 1024, 2048, 3072, 4096, 8192}) {
        const long long pieces = (n2 + 255)/256;
        const long long per_stream = (pieces + S - 1)/S;
        float ms = time_ms([&] { hipLaunchKernelGGL(strided_fill, dim3((unsigned)(per_stream*S)), dim3(256), 0, 0, (double2 *)out, n2, S); });
        printf("strided_fill streams=%d %.4f ms %7.0f GB/s\n", S, ms, n2*16/1e9/ms*1e3);
        ms = time_ms([&] { hipLaunchKernelGGL(strided_fill_nt, dim3((unsigned)(per_stream*S)), dim3(256), 0, 0, (double2 *)out, n2, S); });
        printf("strided_fill streams=%d non-temporal %.4f ms %7.0f GB/s\n", S, ms, n2*16/1e9/ms*1e3);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(stream_fill64, dim3((unsigned)((n2 + 63)/64)), dim3(64), 0, 0, (double2 *)out, n2); });
        printf("stream_fill64 (1 KB per 64-thread block) %.4f ms %7.0f GB/s\n", ms, n2*16/1e9/ms*1e3);
    }
    for (int g : {2048, 8192, 65536}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(stream_fill, dim3(g),
                                                    dim3(256), 0, 0,
                                                    (double2 *)out, n2); });
        printf("stream_fill grid=%d %.4f ms %7.0f GB/s\n", g, ms,
               n2*16/1e9/ms*1e3);
    }
    float ms = time_ms([&] { CHECK(hipMemsetAsync(out, 0, n2*16, 0)); });
    printf("hipMemsetAsync %.4f ms %7.0f GB/s\n", ms, n2*16/1e9/ms*1e3);
    return 0;
}
