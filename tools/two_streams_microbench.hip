// Which call of a short piece of work in stream B waits while a long kernel of few workgroups runs in stream A?
// (r05: the word scan of the drop-in's main thread sat behind the SDP passes of the flight thread.)
// hipcc --offload-arch=gfx950 -O2 two_streams.hip -o two_streams -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long cycles, int *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (out) out[blockIdx.x] = 1;
}
__global__ void touch(const unsigned char *in, int n, unsigned long long *sum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && in[i] == 255) atomicAdd(sum, 1ull);
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 303, lds = argc > 2 ? atoi(argv[2]) : 0;
    hipStream_t a, b;
    OK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    OK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int n = 10 << 20;
    unsigned char *h = nullptr, *d = nullptr; unsigned long long *dsum = nullptr, *hsum = nullptr; int *dout = nullptr;
    OK(hipHostMalloc((void **)&h, n, hipHostMallocDefault)); OK(hipHostMalloc((void **)&hsum, 8, hipHostMallocDefault));
    OK(hipMalloc((void **)&d, n)); OK(hipMalloc((void **)&dsum, 8)); OK(hipMalloc((void **)&dout, 4 * 4096));
    memset(h, 1, n);
    OK(hipMemsetAsync(dsum, 0, 8, b));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 1000, nullptr);      // code objects loaded
    hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, b, d, n, dsum);
    OK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        const double t0 = now_ms();
        std::thread ta([&] {
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), lds, a, 30000000ll /* 100 MHz clock: 300 ms */, dout);
            (void)hipStreamSynchronize(a);
            printf("  A: long kernel done at %.1f ms\n", now_ms() - t0);
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        double t = now_ms();
        auto lap = [&](const char *what) { const double u = now_ms(); printf("  B: %-34s %8.2f ms (at %.1f)\n", what, u - t, u - t0); t = u; };
        (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, b); lap("hipMemcpyAsync H2D 10 MB (pinned)");
        (void)hipStreamSynchronize(b); lap("sync");
        hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, b, d, n, dsum); lap("launch");
        (void)hipStreamSynchronize(b); lap("sync");
        (void)hipMemcpyAsync(hsum, dsum, 8, hipMemcpyDeviceToHost, b); lap("hipMemcpyAsync D2H 8 B (pinned)");
        (void)hipStreamSynchronize(b); lap("sync");
        void *m = nullptr; (void)hipMalloc(&m, 64 << 20); lap("hipMalloc 64 MB");
        (void)hipFree(m); lap("hipFree");
        ta.join();
    }
    return 0;
}
