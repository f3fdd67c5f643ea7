// opty_device.h -- hand-written CDNA4 (gfx950) device runtime that every
// generated collocation kernel is built on.
//
// Work decomposition (the replacement for the reference's
// `for i in prange(n)` node loop, opty/utils.py:524-526):
//   * one 64-lane wavefront == 64 consecutive constraint nodes, lane l owns
//     node  node0 + l; a workgroup is one wave, or W waves that evaluate
//     different entry ranges of the same 64 nodes and share the input slab;
//   * the trajectory rows those nodes need (65 time nodes per row: the one-node
//     halo of opty/direct_collocation.py:2411-2413) are pulled from HBM with
//     one coalesced 512-byte load per row into an LDS slab (the generated
//     prologue issues every load before the first LDS write), and every lane
//     then picks its "current" and "adjacent" value from LDS;
//   * the per-node expressions (generated straight-line float64 code) are
//     evaluated in registers;
//   * constraints are stored equation-major straight from registers (lane ==
//     node == consecutive addresses, opty/direct_collocation.py:2446);
//   * the Jacobian block of a node is node-major in memory
//     (jac[i*P + j*C + k], opty/direct_collocation.py:2885-2887), i.e. lanes
//     are 8*P bytes apart.  A wave therefore stages entries of all its 64
//     nodes in an LDS ring tile (entry-major, conflict-free 8-byte writes) and
//     flushes whole 128-byte lines of the flat output with branch-free
//     16-byte buffer stores (opty_flush_lines below); tiny blocks (P < 64)
//     use the simple per-chunk flushes opty_flush16 / opty_flush8.
//
// The waves of a workgroup exchange data only through their own ring tile:
// LDS operations of one wave execute in order, so a compiler-level fence
// (opty_wave_sync) is all the tile hand-off needs; the shared slab is
// published with one __syncthreads() when W > 1.
#pragma once
#include <hip/hip_runtime.h>

// Every generated kernel takes the same argument list (mirrored by
// `struct KernelArgs` in opty_hip.cpp, which is passed as the packed kernarg
// buffer):
//   const double *free_       (n+q)*N + r + s free variables
//   const double *known_traj  m_known x N known input trajectories
//   const double *params      known parameter values
//   const double *uni_c       node-invariant table (read side)
//   double       *uni_w       the same table (written by opty_uni only)
//   const long long *inst_idx free index of every instance-function atom
//   double *con, *jac         outputs
//   double h                  node time interval when it is not free
//   long long N               number of collocation (time) nodes
//   long long con_stride      distance between two equations in `con`
//   long long node_begin/end  constraint-node range this launch evaluates

// Cache-policy bits of the flush stores (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).
#ifndef OPTY_STORE_AUX
#define OPTY_STORE_AUX 0
#endif

#define OPTY_WAVE 64
// LDS row stride (in doubles) of both the input slab and the output tile.
// 65 = 1 (mod 16) makes the transposing reads of the flush hit 32 distinct
// bank pairs (ds_read_b64 sees 64 banks; 2*65*2 dwords = 4 (mod 64)).
#define OPTY_TS 65

// Compiler-only ordering point between the lanes of the (single) wave of a
// workgroup that exchange data through LDS.
__device__ __forceinline__ void opty_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Flushes an entry-major LDS tile (KC entries x 64 nodes, row stride OPTY_TS)
// to the node-major Jacobian: node nd's KC values go to out[nd*P + 0..KC).
// 16-byte stores; requires KC even and (P even, entry offset even) so that
// every store is 16-byte aligned.
template <int KC>
__device__ __forceinline__ void opty_flush16(const double *tile, double *out,
                                             long long P, int nvalid,
                                             int lane) {
    constexpr int PAIRS = KC/2;                 // 16-byte pieces per node
    constexpr int PASSES = (OPTY_WAVE*PAIRS + OPTY_WAVE - 1)/OPTY_WAVE;
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const int g = j*OPTY_WAVE + lane;
        const int nd = g/PAIRS;
        const int pr = g - nd*PAIRS;
        if (nd < nvalid && nd < OPTY_WAVE) {
            double2 v;
            v.x = tile[(2*pr)*OPTY_TS + nd];
            v.y = tile[(2*pr + 1)*OPTY_TS + nd];
            *reinterpret_cast<double2 *>(out + nd*P + 2*pr) = v;
        }
    }
}

// 8-byte variant for odd P / odd offsets (small problems only).
template <int KC>
__device__ __forceinline__ void opty_flush8(const double *tile, double *out,
                                            long long P, int nvalid,
                                            int lane) {
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        const int g = j*OPTY_WAVE + lane;
        const int nd = g/KC;
        const int k = g - nd*KC;
        if (nd < nvalid)
            out[nd*P + k] = tile[k*OPTY_TS + nd];
    }
}

// ---------------------------------------------------------------------------
// Row-sorted ("csr") layout: the L entries of equation j of 64 consecutive
// nodes are ONE contiguous span of 64*L doubles, dst[nd*L + k].  The tile holds
// them entry-major ([k][nd]); the wave sweeps the span front to back, every
// lane storing one 16-byte aligned pair per step, i.e. 1 KB of consecutive,
// line-aligned bytes per store instruction: whole 128-byte lines everywhere
// but at the two ends of the span.
// ---------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void opty_flush_flat(const double *tile,
                                                double *dst, int nvalid,
                                                int lane) {
    const int total = nvalid*L;
    // Pieces are counted from the 128-byte line that holds dst, so that every
    // store instruction of the wave covers eight whole lines (a sweep whose
    // instructions start at arbitrary 16-byte offsets runs at 4.2 TB/s, this
    // one at 6.7 TB/s, MI355X, 10-link pendulum).  `a` doubles of that line
    // precede dst.
    const int a = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15);
    typedef double opty_d2 __attribute__((ext_vector_type(2)));
    typedef unsigned opty_u4 __attribute__((ext_vector_type(4)));
    // buffer over the aligned span; out-of-range offsets drop the store
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        dst - a, (short)0, (total + a)*8, 0x00020000);
    constexpr int STEPS = (OPTY_WAVE*L + 15 + 2*OPTY_WAVE - 1)/(2*OPTY_WAVE);
#pragma unroll 4
    for (int it = 0; it < STEPS; ++it) {
        const int g = it*OPTY_WAVE + lane;          // 16-byte piece
        const int f = 2*g - a, f1 = f + 1;
        const bool ok = f >= 0 && f1 < total;
        const int c0 = f < 0 ? 0 : (f < total ? f : total - 1);
        const int c1 = f1 < 0 ? 0 : (f1 < total ? f1 : total - 1);
        const int n0 = c0/L, n1 = c1/L;
        opty_d2 v;
        v.x = tile[(c0 - n0*L)*OPTY_TS + n0];
        v.y = tile[(c1 - n1*L)*OPTY_TS + n1];
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(opty_u4, v), rsrc, ok ? 16*g : 0x7ffffff0, 0,
            OPTY_STORE_AUX);
    }
    // the two half pieces at the ends of the span (dst 8 mod 16 / odd end)
    if ((a & 1) && lane == 0)
        __builtin_nontemporal_store(tile[0], dst);
    if (((total + a) & 1) && lane == 1)
        __builtin_nontemporal_store(
            tile[((total - 1) % L)*OPTY_TS + (total - 1)/L], dst + total - 1);
}

// ---------------------------------------------------------------------------
// Line-aligned flush (blocks with P >= 64).
//
// Measured on MI355X (tools/store_bench.hip): streaming stores that cover whole
// 128-byte lines reach ~6.1 TB/s, the same bytes written as 256-byte segments
// that start at arbitrary 16-byte offsets only ~3.3 TB/s -- every partially
// written line costs about as much as two full ones.  Node rows are 8*P bytes
// (7920 for the 10-link pendulum), so row starts drift through all sixteen
// 8-byte offsets of a line.  The flush therefore works on *lines of the flat
// output*, not on entries of a node:
//
//   * the wave's 64 node rows are one contiguous region; position
//     L = nd*P + v (node nd, "virtual entry" v); v >= P simply continues into
//     node nd+1's row;
//   * the tile is a ring of R >= K+16 entry rows; after a chunk of K entries
//     every node flushes the (K/16) lines that have just become complete,
//     wherever their boundaries fall for that node;
//   * a line is owned by the wave whose entry range holds the line's FIRST
//     entry; waves evaluate up to 15 entries past their range (the last range
//     wraps to entries 0..14, which lane nd+1 holds for node nd's last line),
//     so every line is written exactly once, whole;
//   * only the two ends of the 64-node block are partial (the line shared with
//     the neighbouring block): written element-wise by opty_head_piece and by
//     the straddling-piece branch below.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int opty_line_phase(const double *p) {
    return (int)((reinterpret_cast<unsigned long long>(p) >> 3) & 15ULL);
}

typedef unsigned int opty_u32x4 __attribute__((ext_vector_type(4)));

// Buffer resource over the wave's output block [jrow, jrow + bytes): raw
// (stride 0) addressing with hardware range checking -- a store whose byte
// offset is out of range is dropped, which is how the flush predicates its
// stores without branching (gfx950 data-format word: 0x00020000).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t
opty_block_rsrc(double *jrow, int nvalid, int P) {
    return __builtin_amdgcn_make_buffer_rsrc(
        jrow, (short)0, nvalid*P*8, 0x00020000);
}

// NLP: line slots per node and flush (a power of two >= the lines that can
// have completed); R: ring rows; UNR: pieces kept in flight.
// lo = first candidate line start (chunk start - 15), lor = lo mod R (>= 0).
template <int NLP, int R, int UNR>
__device__ __forceinline__ void opty_flush_lines(const double *tile,
                                                 double *jrow, int P, int b0,
                                                 int lo, int lor, int own_lo,
                                                 int own_hi, int avail,
                                                 int nvalid, int lane) {
    constexpr int PPN = NLP*8;          // 16-byte pieces per node and flush
    constexpr int NPP = OPTY_WAVE/PPN;  // nodes covered by one pass
    static_assert((PPN & (PPN - 1)) == 0 && PPN <= OPTY_WAVE, "NLP: 1,2,4,8");
    static_assert(16*NLP + 16 <= R, "ring too small for the line window");
    const __amdgpu_buffer_rsrc_t rsrc = opty_block_rsrc(jrow, nvalid, P);
    // Lane roles are fixed for the whole flush: which line slot and which
    // 16-byte piece of it; passes walk over the nodes, so everything that
    // depends on the node advances by constants.
    const int w = lane & (PPN - 1);
    const int k16 = (w >> 3) << 4;
    const int p2 = (w & 7) << 1;
    int nd = lane/PPN;
    int o = (b0 + nd*P) & 15;           // line phase of the node's row start
    const int dphase = (NPP*P) & 15;
    int pos0 = nd*P;
    // Branch-free: all LDS reads (ring slots always in range) and all stores
    // are issued unconditionally; pieces that must not be written get an
    // out-of-range buffer offset and are dropped by the hardware.  Partially
    // unrolled so that UNR pieces are in flight while the code stays small.
#pragma unroll UNR
    for (int j = 0; j < PPN; ++j) {
        const int d = ((-(o + lo)) & 15) + k16;     // line start - lo
        const int v0 = lo + d;
        const int v = v0 + p2;
        const bool ok = nd < nvalid && v0 >= own_lo && v0 < own_hi &&
                        v0 + 16 <= avail;
        int r0 = lor + d + p2;                      // < 2R by construction
        r0 = r0 >= R ? r0 - R : r0;
        const int r1 = r0 + 1 == R ? 0 : r0 + 1;
        const int c0 = nd + (v >= P ? 1 : 0);
        const int c1 = nd + (v + 1 >= P ? 1 : 0);
        double2 x;
        x.x = tile[r0*OPTY_TS + c0];
        x.y = tile[r1*OPTY_TS + c1];
        const int pos = pos0 + v;
        const bool full = ok && c1 < nvalid;
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(opty_u32x4, x), rsrc,
            full ? pos*8 : 0x7ffffff0, 0, OPTY_STORE_AUX);
        // piece straddling the end of the block's last node (odd line parity
        // only): its first element alone
        if (ok && !full && c0 < nvalid) jrow[pos] = x.x;
        nd += NPP;
        o = (o + dphase) & 15;
        pos0 += NPP*P;
    }
}

// Entries [0, s) of the block's first node share their line with the previous
// block's last node; nobody else holds them, so the first wave stores them
// element-wise while entries 0..15 are still in the ring.
template <int R>
__device__ __forceinline__ void opty_head_piece(const double *tile,
                                                double *jrow, int P, int b0,
                                                int lane) {
    const int s = (-b0) & 15;
    if (lane < s && lane < P) jrow[lane] = tile[(lane % R)*OPTY_TS];
}

// ---------------------------------------------------------------------------
// sin / cos of a trajectory value.
//
// Multibody equations reach their angles only through sin and cos, and every
// wave that evaluates a strip of a block needs (nearly) all of them: the
// 24-link stand-in of BASELINE config 5 spends ~2.9 k of its ~3.5 k vector
// instructions per wave in 24 library sincos calls (profiles/r02_pmc_24link:
// 86.6 k VALU instructions per 64-node block = 30 waves x 24 sincos).  The
// library routine carries the Payne-Hanek reduction for arbitrarily large
// arguments inline at every call site; collocation angles are small.  So:
//
//   |x| <= 2^20 : k = rint(x*2/pi); r = x - k*pi/2 by three FMAs against a
//     three-part pi/2 (each product exact inside the FMA: the reduced argument
//     is good to an ulp of r itself, also next to multiples of pi/2), then the
//     classic minimax kernels on [-pi/4, pi/4] (coefficients of the public
//     fdlibm k_sin.c / k_cos.c, < 1 ulp each) and a branch-free quadrant fix;
//     ~50 vector instructions for the pair (the compiler's two-address
//     v_fmac_f64 needs every Horner constant copied into a VGPR pair first;
//     a three-address v_fma_f64 with the constant in SGPRs, forced through
//     inline assembly, saved those copies but returned non-repeatable values
//     in one equation of the 24-link row-sorted module, next to hundreds of
//     SGPR spills -- not pursued: the kernels do not wait on the vector ALU);
//   otherwise (and Inf) : the library sincos, in a branch marked unlikely so
//     that its code sits behind the kernel's hot path (a real call would
//     need a stack, i.e. scratch memory, in every kernel).
// NaN falls through the fast path and stays NaN.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void opty_sincos(double x, double *s, double *c) {
    if (__builtin_expect(!(__builtin_fabs(x) <= 1048576.0) && x == x, 0)) {
        sincos(x, s, c);
        return;
    }
    const double k = __builtin_rint(x*6.36619772367581382433e-01);  // 2/pi
    // pi/2 = hi + mid + lo, each the nearest double of what is left
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    r = __builtin_fma(-k, -1.49738490485916983294e-33, r);
    const double z = r*r;
    // sin(r) = r + r^3*(S1 + z*(S2 + ... z*S6))
    // (innermost step as multiply + add: one scalar constant per instruction)
    double ps = z*1.58969099521155010221e-10 + -2.50507602534068634195e-08;
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(z*r, ps, r);
    // cos(r) = w + (((1 - w) - z/2) + z*z*(C1 + z*(C2 + ... z*C6))), w = 1 - z/2
    double pc = z*-1.13596475577881948265e-11 + 2.08757232129817482790e-09;
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5*z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + (z*z)*pc);
    // quadrant n = k mod 4: (s, c) = (sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr)
    const int n = (int)k;
    const bool swap = n & 1;
    const double a = swap ? cr : sr;
    const double b = swap ? sr : cr;
    *s = (n & 2) ? -a : a;
    *c = ((n + 1) & 2) ? -b : b;
}

__device__ __forceinline__ double opty_sin(double x) {
    double s, c;
    opty_sincos(x, &s, &c);
    return s;
}

__device__ __forceinline__ double opty_cos(double x) {
    double s, c;
    opty_sincos(x, &s, &c);
    return c;
}

template <int N>
__device__ __forceinline__ double opty_powi(double x) {
    if constexpr (N == 1) return x;
    else if constexpr (N % 2 == 0) { double y = opty_powi<N/2>(x); return y*y; }
    else { return x*opty_powi<N - 1>(x); }
}
