// comm.cpp -- the RCCL side of libopty_hip.so.
#include "opty_internal.h"

using namespace opty;

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
    if (c->device != p->d.device)
        return fail("the communicator lives on device %d, the problem handle "
                    "on device %d", c->device, p->d.device);
    if (int rc = use_device(p)) return rc;
    if (int rc = order_streams(p)) return rc;
    const int64_t a = bounds[c->rank], b = bounds[c->rank + 1];
    hipStream_t st = p->stream;
    // Inside a group every call is issued whatever its neighbours return and
    // the group is ALWAYS closed: a return between ncclGroupStart and
    // ncclGroupEnd would leave the group open and the peers blocked (ADVICE
    // r05).  The first error is reported after the group has ended.
    int gerr = 0;
    auto in_group = [&](int r) { if (r != 0 && gerr == 0) gerr = r; };
    auto group_result = [&](const char *what_) -> int {
        return gerr == 0 ? 0 : fail("%s failed: %s", what_,
                                    g_rccl.GetErrorString(gerr));
    };
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
            // strided copies of an earlier call may still read the old
            // block: waited for explicitly, not through hipFree's implicit
            // synchronisation
            HIP_TRY(hipStreamSynchronize(sync_target(st)));
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
                in_group(g_rccl.Recv(jac_global + ga*P, (size_t)(gb - ga)*P,
                                     RCCL_FLOAT64, g, c->comm, st));
            if (want_con) {
                in_group(g_rccl.Recv(c->d_stage + off, (size_t)M*(gb - ga),
                                     RCCL_FLOAT64, g, c->comm, st));
                off += (size_t)M*(gb - ga);
            }
        }
        in_group(g_rccl.GroupEnd());
        if (int rc = group_result("grouped ncclRecv")) return rc;
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
        in_group(g_rccl.Send(jac_shard, (size_t)(b - a)*P, RCCL_FLOAT64, root,
                             c->comm, st));
    if (want_con)
        in_group(g_rccl.Send(con_shard, (size_t)M*(b - a), RCCL_FLOAT64, root,
                             c->comm, st));
    in_group(g_rccl.GroupEnd());
    return group_result("grouped ncclSend");
}

}  // extern "C"

