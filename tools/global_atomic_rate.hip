// Global fp32 atomic-add rate on MI355X by memory scope, and whether XCD-local (workgroup-scope = L2) atomics
// are exact when every address is only ever touched from ONE XCD.   hipcc --offload-arch=gfx950 -O3 -o gar tools/global_atomic_rate.hip
//
// mode 0: atomicAdd (agent scope)               random addresses over the whole table, any XCD
// mode 1: workgroup-scope fetch_add             the same traffic (NOT exact across XCDs: timing only)
// mode 2: workgroup-scope, XCD-partitioned      a workgroup reads its XCC id and only touches slice [xcc] of the table
// mode 3: agent scope, XCD-partitioned          same partition, device-scope atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *table, unsigned entries_per_slice, int iters, unsigned *xcc_hist) {
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[xcc], 1u);
    unsigned idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const unsigned total = entries_per_slice * 8;
    for (int it = 0; it < iters; it++) {
        idx = idx * 1664525u + 1013904223u;
        unsigned e = (idx >> 7);
        float *p;
        if (MODE >= 2) p = table + (size_t)xcc * entries_per_slice + (e % entries_per_slice);
        else p = table + (e % total);
        if (MODE == 0 || MODE == 3) atomicAdd(p, 1.0f);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

int main() {
    const unsigned entries_per_slice = 2u << 20;            // 8 MiB of floats per XCD slice (two 2^19 x 2 levels)
    const size_t total = (size_t)entries_per_slice * 8;
    float *d; hipMalloc(&d, total * 4);
    unsigned *hist; hipMalloc(&hist, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 256, blocks = 8192;
    std::vector<float> h(total);
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(d, 0, total * 4); hipMemset(hist, 0, 64);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, entries_per_slice, iters, hist);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, entries_per_slice, iters, hist);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, entries_per_slice, iters, hist);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, entries_per_slice, iters, hist);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
                hipMemcpy(h.data(), d, total * 4, hipMemcpyDeviceToHost);
                double sum = 0; for (size_t i = 0; i < total; i++) sum += h[i];
                unsigned hh[16]; hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost);
                const double n = (double)blocks * 256 * iters;
                printf("mode %d: %.3f ms  %.2f G atomics/s  sum %.0f of %.0f (%s)  blocks per xcc:", mode, ms, n / ms * 1e-6, sum, n,
                       sum == n ? "exact" : "LOST UPDATES");
                for (int x = 0; x < 8; x++) printf(" %u", hh[x]);
                printf("\n");
            }
        }
    }
    return 0;
}
