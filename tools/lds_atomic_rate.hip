#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    __shared__ float s[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned idx = (wave * 977 + lane) & 32767;
    for (int it = 0; it < iters; it++) {
        idx = (idx * 1664525u + 1013904223u);
        unsigned a = ((idx >> 8) & 0x7FC0) | lane;      // 64 consecutive floats per wave, random row
        if (MODE == 0) atomicAdd(&s[a], 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&s[a]), 1u);
        else if (MODE == 3) atomicAdd(reinterpret_cast<double*>(&s[a & ~1u]), 1.0);
        else if (MODE == 4) __hip_atomic_fetch_add(&s[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 5) unsafeAtomicAdd(&s[a], 1.0f);
        else s[a] += 1.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[0] + s[100];
}
int main() {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, d, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, d, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, d, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, d, iters);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, d, iters);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.3f ms -> %.1f cycles per wave-instruction per CU (16 waves)\n", mode, ms, ms * 1e-3 * 2.4e9 / (iters * 16.0));
        }
    }
    return 0;
}
