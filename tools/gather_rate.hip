// What a gather instruction costs on MI355X as a function of how many distinct 128-byte lines its 64 lanes touch, the
// bytes per lane and the residency of the table (L1 / L2 / Infinity Cache).  Answers the question behind
// sample_features8_kernel's lane layout: is the texture path's cost per instruction, per lane, or per distinct line?
//   hipcc --offload-arch=gfx950 -O3 -o gr tools/gather_rate.hip && ./gr
// Output: cycles per wave-instruction per CU (2.4 GHz assumed) for each (table size, bytes/lane, lanes sharing a line).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int BYTES>
__global__ __launch_bounds__(256, 4) void gather_k(const char *__restrict__ p, unsigned line_mask, int share, int iters,
                                                   float *out, int pattern) {
    const unsigned lane = threadIdx.x & 63;
    // pattern 0: `share` ADJACENT lanes read one 128-byte line; 1: the sharing lanes are 64/share apart (interleaved);
    // 2: adjacent, but the groups are shifted by one lane (misaligned pairs / quads)
    unsigned grp = lane / share, sub = lane % share;
    if (pattern == 1) grp = lane % (64 / share), sub = lane / (64 / share);
    if (pattern == 2) grp = ((lane + 1) & 63) / share, sub = ((lane + 1) & 63) % share;
    unsigned idx = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + grp) * 2654435761u + 1;
    float s = 0;
#pragma unroll 4
    for (int i = 0; i < iters; i++) {
        idx = idx * 1664525u + 1013904223u;
        const unsigned off = ((idx >> 7) & line_mask) * 128u + (sub * BYTES) % 128u;
        if (BYTES == 4) s += *reinterpret_cast<const float *>(p + off);
        if (BYTES == 8) {
            const float2 v = *reinterpret_cast<const float2 *>(p + off);
            s += v.x + v.y;
        }
        if (BYTES == 16) {
            const float4 v = *reinterpret_cast<const float4 *>(p + off);
            s += v.x + v.y + v.z + v.w;
        }
    }
    if (s == 12345.f) out[0] = 1;
}

int main() {
    void *d;
    const size_t cap = 256u << 20;
    hipMalloc(&d, cap);
    hipMemset(d, 0, cap);
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 4 * 4, iters = 512;
    const size_t sizes[] = {8u << 10, 2u << 20, 64u << 20};
    const char *names[] = {"8 KiB (L1)", "2 MiB (L2)", "64 MiB (MALL)"};
    printf("%-14s %6s %6s %10s %12s\n", "table", "B/lane", "share", "lines/inst", "cyc/inst/CU");
    for (int pattern = 0; pattern < 3; pattern++)
    for (int t = 0; t < (pattern ? 2 : 3); t++)
        for (int bytes = (pattern ? 8 : 4); bytes <= (pattern ? 8 : 16); bytes *= 2)
            for (int share = 1; share <= 64; share *= 2) {
                if (share * bytes > 128 && share != 64) continue;   // more lanes than fit a line only for the broadcast case
                const unsigned line_mask = (unsigned)(sizes[t] / 128 - 1);
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (bytes == 4) hipLaunchKernelGGL(gather_k<4>, dim3(blocks), dim3(256), 0, 0, (const char *)d, line_mask, share, iters, out, pattern);
                    if (bytes == 8) hipLaunchKernelGGL(gather_k<8>, dim3(blocks), dim3(256), 0, 0, (const char *)d, line_mask, share, iters, out, pattern);
                    if (bytes == 16) hipLaunchKernelGGL(gather_k<16>, dim3(blocks), dim3(256), 0, 0, (const char *)d, line_mask, share, iters, out, pattern);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double insts_per_cu = (double)blocks * 4 * iters / 256.0;
                printf("p%d %-14s %6d %6d %10d %12.1f\n", pattern, names[t], bytes, share, 64 / share, ms * 1e-3 * 2.4e9 / insts_per_cu);
            }
    return 0;
}
