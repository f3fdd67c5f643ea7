// Developer micro-benchmark (GPU box): can the free vector be uploaded by node
// ranges?  `free` is (rows x N) row-major; a node range is a column window of
// every row, i.e. a 2-D copy.  Times one 1-D hipMemcpyAsync of the whole
// vector against S column windows through hipMemcpy2DAsync, from pageable and
// from page-locked host memory.
//   hipcc --offload-arch=gfx950 -O2 tools/h2d_2d_bench.hip -o tools/h2d_2d_bench.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(
        std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 23;
    const long long N = argc > 2 ? atoll(argv[2]) : 100000;
    const size_t bytes = (size_t)rows*N*sizeof(double);
    double *pageable = (double *)malloc(bytes), *pinned = nullptr, *dev = nullptr;
    CHECK(hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault));
    CHECK(hipMalloc((void **)&dev, bytes));
    for (long long i = 0; i < rows*N; ++i) pageable[i] = pinned[i] = (double)i;
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int pin = 0; pin < 2; ++pin) {
        const double *src = pin ? pinned : pageable;
        for (int S : {1, 2, 4, 8}) {
            double best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                const double t0 = now_ms();
                if (S == 1) {
                    CHECK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
                } else {
                    for (int k = 0; k < S; ++k) {
                        const long long a = N*k/S, b = N*(k + 1)/S;
                        CHECK(hipMemcpy2DAsync(dev + a, N*sizeof(double), src + a,
                                               N*sizeof(double), (b - a)*sizeof(double),
                                               rows, hipMemcpyHostToDevice, s));
                    }
                }
                CHECK(hipStreamSynchronize(s));
                best = std::min(best, now_ms() - t0);
            }
            printf("%s host memory, %d window(s): %.3f ms (%.1f GB/s)\n",
                   pin ? "page-locked" : "pageable   ", S, best, bytes/best/1e6);
        }
    }
    return 0;
}
