// Do fp32 MFMA waves and VALU / gather waves of ANOTHER kernel fill different units of a CDNA4 SIMD, or the same one?
// One workgroup of 8 waves per CU (100 KiB of unused dynamic LDS keeps a second one away: two waves per SIMD): waves 0-3 issue v_mfma_f32_16x16x4_f32 back to back (16
// independent accumulators, the canonical MLP's inner loop without its memory traffic), waves 4-7 play the co-resident kernel:
//   role V   v_fma_f32 on 16 independent chains (a VALU-bound kernel: the kNN's distance / k-best arithmetic)
//   role P   v_pk_fma_f32 (packed fp32 pairs, what the kNN's distance evaluations use)
//   role I   v_add_u32 / v_xor / v_mul_lo chains (integer VALU: hash index arithmetic)
//   role G   dependent random 16-byte gathers from a 64 MiB table (a texture-path / latency-bound kernel: the feature kernel)
// Each role runs alone and beside the MFMA waves; per role the kernel reports its own cycles (s_memtime deltas, max over the
// role's waves of block 0..) so that "beside" / "alone" is the slowdown each side suffers.
//   hipcc --offload-arch=gfx950 -O3 -o cohab tools/mfma_valu_cohab.hip && ./cohab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int INWAVE>
__global__ __launch_bounds__(512) void k(int mfma_iters, int role, int role_iters, int prio, const uint4 *__restrict__ tab, unsigned tab_mask,
                                            unsigned long long *cycles /*[blocks][8]*/, float *sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = wall_clock64();
    float keep = 0.f;
    if (wave < 4) {
        if (mfma_iters > 0) {
            f32x4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = f32x4{0, 0, 0, 0};
            const float a = lane * 1e-3f, b = 1.f + lane * 1e-4f;
            float y[16];
#pragma unroll
            for (int i = 0; i < 16; i++) y[i] = lane + i;
            const float m = 1.0000001f, c = 1e-7f;
            for (int it = 0; it < mfma_iters; it++) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    // INWAVE independent v_fma_f32 of the SAME wave behind every MFMA
#pragma unroll
                    for (int j = 0; j < INWAVE; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[(i + 4 * j) & 15]) : "v"(m), "v"(c));
                }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) keep += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + y[i];
        }
    } else if (role_iters > 0) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        if (role == 0) {                       // V: 16 independent v_fma_f32 chains
            float x[16];
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = lane * 1e-3f + i;
            const float m = 1.0000001f, c = 1e-7f;
            for (int it = 0; it < role_iters; it++) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
            }
#pragma unroll
            for (int i = 0; i < 16; i++) keep += x[i];
        } else if (role == 1) {                // P: v_pk_fma_f32
            f32x2 x[16];
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = f32x2{lane * 1e-3f + i, 1.f};
            const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, 2e-7f};
            for (int it = 0; it < role_iters; it++) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
            }
#pragma unroll
            for (int i = 0; i < 16; i++) keep += x[i][0] + x[i][1];
        } else if (role == 2) {                // I: integer VALU
            unsigned x[16];
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = lane * 2654435761u + i;
            for (int it = 0; it < role_iters; it++) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(805459861u), "v"(it));
            }
#pragma unroll
            for (int i = 0; i < 16; i++) keep += (float)(x[i] & 0xFF);
        } else {                               // G: 4 dependent chains of random 16-byte gathers
            unsigned idx[4];
#pragma unroll
            for (int i = 0; i < 4; i++) idx[i] = (lane * 2654435761u + i * 40503u + blockIdx.x * 9781u + wave * 77u) & tab_mask;
            for (int it = 0; it < role_iters; it++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint4 v = tab[idx[i]];
                    idx[i] = (v.x + v.y + it) & tab_mask;
                }
            }
            keep += (float)(idx[0] ^ idx[1] ^ idx[2] ^ idx[3]);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    if (keep == 123.456f) sink[0] = keep;
}

template <int INWAVE = 0>
static void run(const char *name, int mfma_iters, int role, int role_iters, int prio, const uint4 *tab, unsigned mask, unsigned long long *cyc,
                float *sink, double *mfma_us, double *role_us) {
    const int blocks = 256 * 2 * 4;        // 8 rounds of 1 workgroup per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<INWAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<INWAVE>, dim3(blocks), dim3(512), 100 * 1024, 0, mfma_iters, role, role_iters, prio, tab, mask, cyc, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, r = 0;
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < 8; w++) (w < 4 ? m : r) += (double)h[b * 8 + w] / (blocks * 4.0);
    *mfma_us = m / 100.0;                   // wall_clock64: 100 MHz
    *role_us = r / 100.0;
    const double tf = mfma_iters ? (double)blocks * 4 * mfma_iters * 16 * 2048.0 / (m / 100.0 * 1e-6) / 1e12 / (blocks / 512.0) : 0;
    printf("%-44s kernel %7.3f ms | MFMA waves: mean %8.1f us each%s | other waves: mean %8.1f us each\n", name, ms, *mfma_us,
           mfma_iters ? "" : " (idle)", *role_us);
    (void)tf;
}

int main() {
    const unsigned n = 1u << 22;           // 4 M x 16 B = 64 MiB
    uint4 *tab;
    (void)hipMalloc(&tab, (size_t)n * 16);
    std::vector<uint4> h(n);
    unsigned s = 12345;
    for (unsigned i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        h[i] = uint4{s, s >> 7, 0, 0};
    }
    (void)hipMemcpy(tab, h.data(), (size_t)n * 16, hipMemcpyHostToDevice);
    unsigned long long *cyc;
    float *sink;
    (void)hipMalloc(&cyc, 256 * 2 * 4 * 8 * 8);
    (void)hipMalloc(&sink, 4);
    const int MI = 4096;
    double m0, r0, m, r;
    run("MFMA waves alone (1 per SIMD)", MI, 0, 0, 0, tab, n - 1, cyc, sink, &m0, &r0);
    // the same wave issuing independent VALU work behind each of its MFMAs: does it fit into the MFMA's 32 cycles?
    run<1>("MFMA + 1 v_fma_f32 of the SAME wave per MFMA", MI, 0, 0, 0, tab, n - 1, cyc, sink, &m, &r);
    printf("    -> %.2fx the MFMA-only time\n", m / m0);
    run<2>("MFMA + 2 v_fma_f32 of the SAME wave per MFMA", MI, 0, 0, 0, tab, n - 1, cyc, sink, &m, &r);
    printf("    -> %.2fx the MFMA-only time\n", m / m0);
    run<4>("MFMA + 4 v_fma_f32 of the SAME wave per MFMA", MI, 0, 0, 0, tab, n - 1, cyc, sink, &m, &r);
    printf("    -> %.2fx the MFMA-only time\n", m / m0);
    run<6>("MFMA + 6 v_fma_f32 of the SAME wave per MFMA", MI, 0, 0, 0, tab, n - 1, cyc, sink, &m, &r);
    printf("    -> %.2fx the MFMA-only time\n", m / m0);
    const char *names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_xad_u32 (integer)", "dependent 16-byte gathers"};
    const int iters[4] = {4096 * 4, 4096 * 2, 4096 * 4, 600};
    for (int role = 0; role < 4; role++) {
        char buf[128];
        double ra;
        snprintf(buf, sizeof buf, "%s waves alone (1 per SIMD)", names[role]);
        run(buf, 0, role, iters[role], 0, tab, n - 1, cyc, sink, &m, &ra);
        for (int prio = 0; prio < 2; prio++) {
            snprintf(buf, sizeof buf, "MFMA + %s%s", names[role], prio ? ", s_setprio 3" : "");
            run(buf, MI, role, iters[role], prio, tab, n - 1, cyc, sink, &m, &r);
            printf("    -> MFMA waves %.2fx their time alone, %s waves %.2fx theirs; serial = %.1f us, co-resident = %.1f us\n",
                   m / m0, names[role], r / ra, m0 + ra, m > r ? m : r);
        }
    }
    return 0;
}
