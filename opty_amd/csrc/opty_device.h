// opty_device.h -- hand-written CDNA4 (gfx950) device runtime that every
// generated collocation kernel is built on.
//
// Work decomposition (the replacement for the reference's
// `for i in prange(n)` node loop, opty/utils.py:524-526):
//   * one 64-lane wavefront == one workgroup == 64 consecutive constraint
//     nodes, lane l owns node  node0 + l;
//   * the trajectory rows those nodes need (65 time nodes per row: the one-node
//     halo of opty/direct_collocation.py:2411-2413) are pulled from HBM with
//     one coalesced 512-byte load per row into an LDS slab, and every lane then
//     picks its "current" and "adjacent" value from LDS;
//   * the per-node expressions (generated straight-line float64 code) are
//     evaluated in registers;
//   * constraints are stored equation-major straight from registers (lane ==
//     node == consecutive addresses, opty/direct_collocation.py:2446);
//   * the Jacobian block of a node is node-major in memory
//     (jac[i*P + j*C + k], opty/direct_collocation.py:2885-2887), i.e. lanes
//     are 8*P bytes apart.  The wave therefore stages KC consecutive entries of
//     all its 64 nodes in an LDS tile (entry-major, conflict-free 8-byte
//     writes), and flushes the tile with 16-byte-per-lane stores in which 2*KC
//     / 16 ... lanes cover one node's KC*8 contiguous bytes.
//
// A single-wave workgroup needs no s_barrier: LDS operations of one wave
// execute in order, so a compiler-level fence is all the tile hand-off needs.
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

// Loads time nodes [t0, t0+64] of one trajectory row into slab row `r`.
// `tmax` is the last valid time node (N-1); loads are clamped, the clamped
// lanes belong to nodes that are never stored.
__device__ __forceinline__ void opty_slab_load(double *slab, int r,
                                               const double *row,
                                               long long t0, long long tmax,
                                               int lane) {
    long long t = t0 + lane;
    slab[r*OPTY_TS + lane] = row[t < tmax ? t : tmax];
    if (lane == 0) {
        long long te = t0 + OPTY_WAVE;
        slab[r*OPTY_TS + OPTY_WAVE] = row[te < tmax ? te : tmax];
    }
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

template <int N>
__device__ __forceinline__ double opty_powi(double x) {
    if constexpr (N == 1) return x;
    else if constexpr (N % 2 == 0) { double y = opty_powi<N/2>(x); return y*y; }
    else { return x*opty_powi<N - 1>(x); }
}
