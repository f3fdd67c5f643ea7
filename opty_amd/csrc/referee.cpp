// referee.cpp -- libopty_hip_referee.so: the build verification's device side
// (include/opty_hip.h, "build verification").  Kept OUT of libopty_hip.so:
// a process that only evaluates never loads it; the Python host loads it when
// it verifies a code object (ConstraintCollocator._verify_build).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/opty_hip_referee.h"

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
            (void)hipGetLastError();                                          \
            return fail("%s failed: %s", #expr, hipGetErrorString(e_));       \
        }                                                                     \
    } while (0)

}  // namespace

extern "C" {

const char *opty_hip_referee_last_error(void) { return g_error.c_str(); }

#include "opty_poison.inc"

int opty_hip_poison_registers(unsigned pattern) {
    hipLaunchKernelGGL(opty_poison, dim3(4096), dim3(64), 0, nullptr, pattern,
                       (unsigned *)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
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
