// programs.cpp -- the two other generated programs libopty_hip.so drives:
// the objective / objective gradient (SURVEY.md 8(f) rank 1,
// opty/utils.py:329-470) and plain matrix functions in the reference's
// ufuncify_matrix call shape (opty/utils.py:639-640).
#include "opty_internal.h"

using namespace opty;

// ---------------------------------------------------------------------------
// objective / objective gradient
// ---------------------------------------------------------------------------
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


}  // extern "C"
