// Calibration of rocprofv3's FETCH_SIZE for the access patterns of the gather kernels (MI355X_MICROARCH.md: the x2
// correction is established for wide coalesced reads only).  hipcc --offload-arch=gfx950 -O3 -o fc tools/fetch_calib.hip
//   ./fc <mode> : 0 = coalesced 16 B/lane stream of 1 GiB; 1 = random 8-byte gathers over a 59 MiB table (the hash
//   grid: fits the 256 MiB Infinity Cache, not the 4 MiB L2s); 2 = random 8-byte gathers over 1 GiB (HBM misses)
// Run each mode under `rocprofv3 --pmc FETCH_SIZE` and compare with the printed byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void stream_k(const f4 *p, size_t n, float *out) {
    f4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1;
}
__global__ void gather_k(const float2 *p, unsigned mask, int iters, float *out) {
    unsigned idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 1;
    float s = 0;
    for (int i = 0; i < iters; i++) {
        idx = idx * 1664525u + 1013904223u;
        const float2 v = p[(idx >> 5) & mask];
        s += v.x + v.y;
    }
    if (s == 12345.f) out[0] = 1;
}
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t bytes = mode == 1 ? (64u << 20) : (1u << 30);
    void *d; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    float *out; hipMalloc(&out, 4);
    hipDeviceSynchronize();
    if (mode == 0) {
        hipLaunchKernelGGL(stream_k, dim3(4096), dim3(256), 0, 0, (const f4 *)d, bytes / 16, out);
        printf("mode 0: coalesced stream, %zu bytes read\n", bytes);
    } else {
        const unsigned entries = (unsigned)(bytes / 8);          // power of two
        const int iters = 64, blocks = 8192;
        hipLaunchKernelGGL(gather_k, dim3(blocks), dim3(256), 0, 0, (const float2 *)d, entries - 1, iters, out);
        const double n = (double)blocks * 256 * iters;
        printf("mode %d: %.0f random 8-byte gathers over %zu MiB = %.0f useful bytes; x64 B lines = %.0f, x128 B = %.0f\n", mode, n,
               bytes >> 20, n * 8, n * 64, n * 128);
    }
    hipDeviceSynchronize();
    return 0;
}
