// fetch_calibration.hip — known-size reads in the access widths the Viterbi kernels use, to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: the counter reports one half
// of a 16 B/lane streaming read; other widths are "uncalibrated: calibrate on a known byte count in your own
// access pattern").  Every kernel touches each byte of a BYTES-sized buffer exactly once per launch (the
// sliding kernel: 64 times, through the caches), so the true HBM bytes per launch are BYTES.
//
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- build/fetch_calibration
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- build/fetch_calibration
// tools/summarise_calibration.py divides the counters by the known bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr long long BYTES = 2LL << 30;        // 2 GiB: eight times the 256 MiB Infinity Cache

template <class T>
__global__ void calib_stream(const T *__restrict__ p, long long n, int *sink) {      // sizeof(T) bytes per lane
    long long acc = 0;
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        if constexpr (sizeof(T) == 16) { const int4 v = ((const int4 *)p)[x]; acc += v.x + v.y + v.z + v.w; }
        else acc += (long long)p[x];
    }
    if (acc == 0x7fffffffffffLL) sink[0] = 1;
}

// the Viterbi kernels' pattern: at step s lane l reads element s - l (adjacent lanes = adjacent columns,
// each step shifts the window by one), one wave per contiguous slab
template <class T>
__global__ void calib_sliding(const T *__restrict__ p, long long per_wave, int *sink) {
    const T *base = p + (long long)blockIdx.x * per_wave;
    long long acc = 0;
    const int lane = threadIdx.x;
    for (long long s = 0; s < per_wave + 63; s++) {
        long long j = s - lane;
        j = j < 0 ? 0 : (j >= per_wave ? per_wave - 1 : j);
        acc += (long long)base[j];
    }
    if (acc == 0x7fffffffffffLL) sink[0] = 1;
}

__global__ void calib_write4(int *__restrict__ p, long long n) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) p[x] = (int)x;
}
__global__ void calib_write16(int4 *__restrict__ p, long long n) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x)
        p[x] = make_int4((int)x, 1, 2, 3);
}

int main() {
    void *buf; int *sink;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, BYTES));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16, block = 256;
    hipLaunchKernelGGL(calib_stream<uint8_t>, dim3(grid), dim3(block), 0, 0, (const uint8_t *)buf, BYTES, sink);
    hipLaunchKernelGGL(calib_stream<uint16_t>, dim3(grid), dim3(block), 0, 0, (const uint16_t *)buf, BYTES / 2, sink);
    hipLaunchKernelGGL(calib_stream<int>, dim3(grid), dim3(block), 0, 0, (const int *)buf, BYTES / 4, sink);
    hipLaunchKernelGGL(calib_stream<int4>, dim3(grid), dim3(block), 0, 0, (const int4 *)buf, BYTES / 16, sink);
    // sliding windows: 8192 waves, each over its own contiguous slab
    hipLaunchKernelGGL(calib_sliding<uint8_t>, dim3(8192), dim3(64), 0, 0, (const uint8_t *)buf, BYTES / 8192, sink);
    hipLaunchKernelGGL(calib_sliding<int>, dim3(8192), dim3(64), 0, 0, (const int *)buf, BYTES / 4 / 8192, sink);
    hipLaunchKernelGGL(calib_write4, dim3(grid), dim3(block), 0, 0, (int *)buf, BYTES / 4);
    hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(block), 0, 0, (int4 *)buf, BYTES / 16);
    CHECK(hipDeviceSynchronize());
    printf("{\"bytes_per_launch\": %lld}\n", BYTES);
    return 0;
}
