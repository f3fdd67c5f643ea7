// opty_internal.h -- what the translation units of libopty_hip.so share:
// error reporting, the problem handle, the launch / evaluation core's
// prototypes.  Not installed; the public interface is include/opty_hip.h.
//
//   runtime.cpp       handles, kernel launches, routing, the entry points of
//                     the evaluation itself, index kernels, objective and
//                     matrix programs
//   host_scatter.cpp  the host-buffer (cyipopt callback) path: packing on the
//                     device, chunked DMA, the scatter pool of host threads
//                     and its NUMA placement, the segmented layout
//   comm.cpp          RCCL communicator, broadcast of `free`, gather-v
//   referee.cpp       (libopty_hip_referee.so) instruction-tape kernel and
//                     register poisoner of the build verification
#pragma once
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

namespace opty {

extern thread_local std::string g_error;
int fail(const char *fmt, ...);

}  // namespace opty

using opty::fail;

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
    hipStream_t copy_stream2 = nullptr; // ... its chunks alternate between two
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
    std::vector<long long> chunk_bounds;  // node ranges of the DMA chunks
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


namespace opty {

struct NodeRange {
    long long begin, end, con_stride;
};

int use_device(const opty_hip_problem *p);
int check_ready(const opty_hip_problem *p);
NodeRange whole(const opty_hip_problem *p);
hipStream_t sync_target(hipStream_t s);
std::vector<int> build_schedule(int persist, long long nblk, int sets,
                                const float *cost);
int launch_instance(opty_hip_problem *p, const double *free_, double *con_tail,
                    double *jac_tail);
int eval_device(opty_hip_problem *p, int what, const double *free_,
                double *con, double *jac, const NodeRange &rg,
                bool with_inst);
int device_numa_node();
hipError_t pinned_alloc(void **ptr, size_t bytes);
double *mapped_address(double *host);
// host_scatter.cpp
int eval_segmented(opty_hip_problem *p, int what, const double *free_,
                   double *con, double *jac, int mem, bool full);
void scatter_quiesce();

// A handle's device state (node-invariant table, staging buffers) belongs to
// one stream at a time.  When the caller moved the handle to another stream
// (opty_hip_set_stream), work issued there is ordered after everything the
// handle enqueued on the previous one: opty_uni may overwrite the table that
// kernels of the previous stream still read, and the first fill has to be
// visible to the new stream.
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

template <typename T>
int ensure(T **ptr, size_t count) {
    if (*ptr == nullptr && count > 0)
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(ptr), count*sizeof(T)));
    return 0;
}

template <typename T>
int ensure_pinned(T **ptr, size_t count) {
    if (*ptr == nullptr && count > 0)
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(ptr),
                             count*sizeof(T)));
    return 0;
}

}  // namespace opty
