// Developer tool (GPU box): do the workgroups of a small-register kernel run
// BESIDE a 512-VGPR kernel launched on another stream (on the SIMDs the big
// kernel leaves free), or only after it?  Decides whether the store-only
// strips of an arithmetic-bound block can be a kernel of their own.
//   hipcc --offload-arch=gfx950 -O2 tools/concurrent_probe.hip -o tools/concurrent_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// big: one wave, all 512 registers (the attribute sets the allocation)
extern "C" __global__ void __launch_bounds__(64)
__attribute__((amdgpu_num_vgpr(512)))
heavy(long long *rec, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { rec[2*blockIdx.x] = t0; rec[2*blockIdx.x + 1] = wall_clock64(); }
}

extern "C" __global__ void __launch_bounds__(64)
__attribute__((amdgpu_num_vgpr(64)))
light(long long *rec, long long ticks, double *out) {
    __shared__ double lds[32*65];
    const long long t0 = wall_clock64();
    lds[threadIdx.x] = (double)blockIdx.x;
    // 36 KB of 16-byte stores, then wait out the rest of the time
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v; v.x = lds[threadIdx.x]; v.y = 1.0;
    d2 *dst = reinterpret_cast<d2 *>(out) + (long long)blockIdx.x*2304 + threadIdx.x;
    for (int i = 0; i < 36; ++i) __builtin_nontemporal_store(v, dst + 64*i);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { rec[2*blockIdx.x] = t0; rec[2*blockIdx.x + 1] = wall_clock64(); }
}

int main() {
    const int NH = 782, NL = 2346;
    long long *rh, *rl; double *out;
    CK(hipMalloc(&rh, NH*16)); CK(hipMalloc(&rl, NL*16));
    CK(hipMalloc(&out, (size_t)NL*2304*16));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join, e0, e1;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long TH = 2000, TL = 300;    // 20 us, 3 us (100 MHz ticks)
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipEventRecord(e0, s0));
            if (mode == 0) {            // serial on one stream
                hipLaunchKernelGGL(heavy, NH, 64, 0, s0, rh, TH);
                hipLaunchKernelGGL(light, NL, 64, 0, s0, rl, TL, out);
            } else if (mode == 1) {     // light first on the aux stream
                CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
                hipLaunchKernelGGL(light, NL, 64, 0, s1, rl, TL, out);
                CK(hipEventRecord(join, s1));
                hipLaunchKernelGGL(heavy, NH, 64, 0, s0, rh, TH);
                CK(hipStreamWaitEvent(s0, join, 0));
            } else if (mode == 3 || mode == 4) {     // one stream, second kernel without the barrier bit
                void *ah[] = {&rh, (void *)&TH};
                double *o = out; void *al[] = {&rl, (void *)&TL, &o};
                if (mode == 3) {
                    CK(hipExtLaunchKernel((const void *)heavy, dim3(NH), dim3(64), ah, 0, s0, nullptr, nullptr, 0));
                    CK(hipExtLaunchKernel((const void *)light, dim3(NL), dim3(64), al, 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch));
                } else {
                    CK(hipExtLaunchKernel((const void *)light, dim3(NL), dim3(64), al, 0, s0, nullptr, nullptr, 0));
                    CK(hipExtLaunchKernel((const void *)heavy, dim3(NH), dim3(64), ah, 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch));
                }
            } else {                    // heavy first
                CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
                hipLaunchKernelGGL(heavy, NH, 64, 0, s0, rh, TH);
                hipLaunchKernelGGL(light, NL, 64, 0, s1, rl, TL, out);
                CK(hipEventRecord(join, s1));
                CK(hipStreamWaitEvent(s0, join, 0));
            }
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        std::vector<long long> h(2*NH), l(2*NL);
        CK(hipMemcpy(h.data(), rh, NH*16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(l.data(), rl, NL*16, hipMemcpyDeviceToHost));
        long long t0 = h[0], hend = 0, lfirst = l[0], llast = 0, lend = 0;
        for (int i = 0; i < NH; ++i) { t0 = std::min(t0, h[2*i]); hend = std::max(hend, h[2*i+1]); }
        for (int i = 0; i < NL; ++i) { lfirst = std::min(lfirst, l[2*i]); llast = std::max(llast, l[2*i]); lend = std::max(lend, l[2*i+1]); }
        int during = 0;
        for (int i = 0; i < NL; ++i) if (l[2*i+1] <= hend && l[2*i] >= t0) ++during;
        printf("mode %d (%s): %.4f ms best of 20; heavy ends +%.1f us; light starts +%.1f .. +%.1f us, ends +%.1f us; %d of %d light waves ran inside the heavy kernel's window\n",
               mode, mode == 0 ? "serial" : mode == 1 ? "two streams, light enqueued first" : mode == 2 ? "two streams, heavy enqueued first" : mode == 3 ? "one stream, heavy then light(any order)" : "one stream, light then heavy(any order)",
               best, (hend - t0)*0.01, (lfirst - t0)*0.01, (llast - t0)*0.01, (lend - t0)*0.01, during, NL);
    }
    return 0;
}
