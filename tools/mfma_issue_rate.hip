// Bottom-up attribution of the fp32 matrix pipe's issue rate on MI355X: what v_mfma_f32_16x16x4_f32 sustains per SIMD with
// the ingredients of csrc/mlp16.hip added one at a time (2 waves per SIMD, 16 independent accumulators per wave, groups of
// 64 MFMAs = one 16 KiB weight chunk):
//   mode 0  MFMAs only, operands in registers
//   mode 1  + 16 ds_read_b128 per group (the chunk's A operands, read one half-group ahead)
//   mode 2  + one s_barrier per group (4-wave workgroups)
//   mode 3  + 4 LDS-DMA fragments per wave per group (global_load_lds_dwordx4, SGPR base + lane offset) and the counted vmcnt
//   mode 4  mode 3 + a 64-instruction VALU epilogue (bias + ReLU) every 16 groups
//   hipcc --offload-arch=gfx950 -O3 -o mir tools/mfma_issue_rate.hip && ./mir
// Prints TFLOP/s and the fraction of the 157.3 TFLOP/s fp32-matrix peak (256 CUs x 4 SIMDs x 2.4 GHz x 512 FLOP / 8 cycles
// per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
// shader clock held during a kernel: block 0 / lane 0 reads the shader-cycle counter and the constant 100 MHz counter at its
// start and end
__device__ unsigned long long g_clk[4];
#define CLK_BEGIN() if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = clock64(); g_clk[1] = wall_clock64(); }
#define CLK_END() if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[2] = clock64(); g_clk[3] = wall_clock64(); }
static double held_ghz() {
    unsigned long long h[4];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
    return (double)(h[2] - h[0]) / ((double)(h[3] - h[1]) * 10.0);      // cycles per 10 ns tick -> GHz
}
constexpr int kWaves = 4, kSlots = 4, kChunkF4 = 1024;      // 16 KiB chunks, 4-slot ring

template <int MODE, int WAVES_PER_SIMD>
__global__ __launch_bounds__(kWaves * 64, WAVES_PER_SIMD) void k(const float *__restrict__ pk, int groups, float *out) {
    // (one wave per SIMD is forced by doubling the LDS image: 128 KiB leaves room for one workgroup per CU)
    __shared__ __attribute__((aligned(16))) f32x4 ring[kSlots * kChunkF4 * (WAVES_PER_SIMD == 1 ? 2 : 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kSlots * kChunkF4; i += kWaves * 64) ring[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    CLK_BEGIN()
    f32x4 acc[16], act[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        acc[i] = f32x4{0, 0, 0, 0};
        act[i] = f32x4{lane * 1e-3f, 1.f, 2.f, 3.f};
    }
    const f32x4 *stream = reinterpret_cast<const f32x4 *>(pk);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) f32x4 *)ring;
    auto issue1 = [&](int c, int f) {
        const int frag = wave * 4 + f;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane * 16), "s"(stream + (size_t)(c & 63) * kChunkF4 + frag * 64),
                       "s"(ring_lds + (unsigned)(((c & (kSlots - 1)) * kChunkF4 + frag * 64) * 16))
                     : "memory");
    };
    int c = 0;
    if (MODE >= 3) {
        for (int j = 0; j < 3; j++)
            for (int f = 0; f < 4; f++) issue1(j, f);
    }
    f32x4 wA[8], wB[8];
#pragma unroll
    for (int i = 0; i < 8; i++) wA[i] = wB[i] = f32x4{1e-3f, 2e-3f, 1e-3f, 2e-3f};
    const f32x4 *slot = ring;
#pragma unroll 1
    for (int gi = 0; gi < groups; gi++) {
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) wB[i] = slot[(8 + i) * 64 + lane];
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
#pragma unroll
            for (int ob = 0; ob < 8; ob++)
                acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ob][rr], act[gi & 15][rr], acc[ob], 0, 0, 0);
        }
        if (MODE >= 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        if (MODE >= 2) __builtin_amdgcn_s_barrier();
        slot = ring + (c & (kSlots - 1)) * kChunkF4;
        c++;
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) wA[i] = slot[i * 64 + lane];
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            if (MODE >= 3) issue1(c + 2, rr);
#pragma unroll
            for (int ob = 0; ob < 8; ob++)
                acc[8 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[ob][rr], act[gi & 15][rr], acc[8 + ob], 0, 0, 0);
        }
        if (MODE >= 4 && (gi & 15) == 15) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
#pragma unroll
                for (int r = 0; r < 4; r++) act[i][r] = fmaxf(acc[i][r] * 1e-3f, 0.0f);
                acc[i] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
            }
        }
    }
    if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLK_END()
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + act[i][0];
    if (s == 12345.678f) out[0] = s;
}

// mode 5: as mode 3, but the four MFMA waves never issue a DMA: a FIFTH wave of the workgroup (64 lanes, a handful of
// registers) waits for the chunk to land, joins the barrier and requests the refill -- 16 fragments per chunk.
__global__ __launch_bounds__(5 * 64, 2) void k_loader(const float *__restrict__ pk, int groups, float *out) {
    __shared__ __attribute__((aligned(16))) f32x4 ring[kSlots * kChunkF4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kSlots * kChunkF4; i += 5 * 64) ring[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    CLK_BEGIN()
    const f32x4 *stream = reinterpret_cast<const f32x4 *>(pk);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) f32x4 *)ring;
    if (wave == 4) {        // ---- loader ----
        auto issue1 = [&](int c, int frag) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane * 16), "s"(stream + (size_t)(c & 63) * kChunkF4 + frag * 64),
                           "s"(ring_lds + (unsigned)(((c & (kSlots - 1)) * kChunkF4 + frag * 64) * 16))
                         : "memory");
        };
        for (int j = 0; j < 3; j++)
            for (int f = 0; f < 16; f++) issue1(j, f);
        int c = 0;
#pragma unroll 1
        for (int gi = 0; gi < groups; gi++) {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // the oldest of three chunks has landed
            __builtin_amdgcn_s_barrier();
            c++;
#pragma unroll
            for (int f = 0; f < 16; f++) issue1(c + 2, f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    f32x4 acc[16], act[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        acc[i] = f32x4{0, 0, 0, 0};
        act[i] = f32x4{lane * 1e-3f, 1.f, 2.f, 3.f};
    }
    int c = 0;
    f32x4 wA[8], wB[8];
#pragma unroll
    for (int i = 0; i < 8; i++) wA[i] = wB[i] = f32x4{1e-3f, 2e-3f, 1e-3f, 2e-3f};
    const f32x4 *slot = ring;
#pragma unroll 1
    for (int gi = 0; gi < groups; gi++) {
#pragma unroll
        for (int i = 0; i < 8; i++) wB[i] = slot[(8 + i) * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
#pragma unroll
            for (int ob = 0; ob < 8; ob++)
                acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ob][rr], act[gi & 15][rr], acc[ob], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = ring + (c & (kSlots - 1)) * kChunkF4;
        c++;
#pragma unroll
        for (int i = 0; i < 8; i++) wA[i] = slot[i * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
#pragma unroll
            for (int ob = 0; ob < 8; ob++)
                acc[8 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[ob][rr], act[gi & 15][rr], acc[8 + ob], 0, 0, 0);
        }
    }
    CLK_END()
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + act[i][0];
    if (s == 12345.678f) out[0] = s;
}

// mode 6: the shape of csrc/nonrigid16.hip -- two 16-sample tiles per wave, every weight register feeds one MFMA of each tile:
// 8 KiB chunks (8 ds_read_b128 and 2 DMA fragments per wave per 64 MFMAs), one barrier per chunk.
__global__ __launch_bounds__(kWaves * 64, 2) void k_two_tiles(const float *__restrict__ pk, int groups, float *out) {
    constexpr int kChunk8 = 512;
    __shared__ __attribute__((aligned(16))) f32x4 ring[kSlots * kChunk8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kSlots * kChunk8; i += kWaves * 64) ring[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    CLK_BEGIN()
    f32x4 acc[2][8], act[2][8];
#pragma unroll
    for (int T = 0; T < 2; T++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc[T][i] = f32x4{0, 0, 0, 0};
            act[T][i] = f32x4{lane * 1e-3f, 1.f, 2.f, 3.f};
        }
    const f32x4 *stream = reinterpret_cast<const f32x4 *>(pk);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) f32x4 *)ring;
    auto issue1 = [&](int c, int f) {
        const int frag = wave * 2 + f;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane * 16), "s"(stream + (size_t)(c & 63) * kChunk8 + frag * 64),
                       "s"(ring_lds + (unsigned)(((c & (kSlots - 1)) * kChunk8 + frag * 64) * 16))
                     : "memory");
    };
    int c = 0;
    for (int j = 0; j < 3; j++)
        for (int f = 0; f < 2; f++) issue1(j, f);
    f32x4 wA[4], wB[4];
#pragma unroll
    for (int i = 0; i < 4; i++) wA[i] = wB[i] = f32x4{1e-3f, 2e-3f, 1e-3f, 2e-3f};
    const f32x4 *slot = ring;
#pragma unroll 1
    for (int gi = 0; gi < groups; gi++) {
#pragma unroll
        for (int i = 0; i < 4; i++) wB[i] = slot[(4 + i) * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int ob = 0; ob < 4; ob++)
#pragma unroll
                for (int T = 0; T < 2; T++)
                    acc[T][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ob][rr], act[T][gi & 7][rr], acc[T][ob], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = ring + (c & (kSlots - 1)) * kChunk8;
        c++;
#pragma unroll
        for (int i = 0; i < 4; i++) wA[i] = slot[i * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            if (rr < 2) issue1(c + 2, rr);
#pragma unroll
            for (int ob = 0; ob < 4; ob++)
#pragma unroll
                for (int T = 0; T < 2; T++)
                    acc[T][4 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[ob][rr], act[T][gi & 7][rr], acc[T][4 + ob], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLK_END()
    float s = 0;
#pragma unroll
    for (int T = 0; T < 2; T++)
#pragma unroll
        for (int i = 0; i < 8; i++) s += acc[T][i][0] + acc[T][i][1] + acc[T][i][2] + acc[T][i][3] + act[T][i][0];
    if (s == 12345.678f) out[0] = s;
}

// mode 7: one 16-sample tile per wave at width 128 (8 accumulators), 8 KiB chunks = 32 MFMAs per wave (8 ds_read_b128, 2 DMA
// fragments, 1 barrier per 32 MFMAs), FOUR waves per SIMD (4 workgroups of 4 waves per CU).
__global__ __launch_bounds__(kWaves * 64, 4) void k_one_tile_w128(const float *__restrict__ pk, int groups, float *out) {
    constexpr int kChunk8 = 512;
    __shared__ __attribute__((aligned(16))) f32x4 ring[kSlots * kChunk8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kSlots * kChunk8; i += kWaves * 64) ring[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    CLK_BEGIN()
    f32x4 acc[8], act[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc[i] = f32x4{0, 0, 0, 0};
        act[i] = f32x4{lane * 1e-3f, 1.f, 2.f, 3.f};
    }
    const f32x4 *stream = reinterpret_cast<const f32x4 *>(pk);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) f32x4 *)ring;
    auto issue1 = [&](int c, int f) {
        const int frag = wave * 2 + f;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane * 16), "s"(stream + (size_t)(c & 63) * kChunk8 + frag * 64),
                       "s"(ring_lds + (unsigned)(((c & (kSlots - 1)) * kChunk8 + frag * 64) * 16))
                     : "memory");
    };
    int c = 0;
    for (int j = 0; j < 3; j++)
        for (int f = 0; f < 2; f++) issue1(j, f);
    f32x4 wA[4], wB[4];
#pragma unroll
    for (int i = 0; i < 4; i++) wA[i] = wB[i] = f32x4{1e-3f, 2e-3f, 1e-3f, 2e-3f};
    const f32x4 *slot = ring;
#pragma unroll 1
    for (int gi = 0; gi < groups; gi++) {
#pragma unroll
        for (int i = 0; i < 4; i++) wB[i] = slot[(4 + i) * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int ob = 0; ob < 4; ob++)
                acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ob][rr], act[gi & 7][rr], acc[ob], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = ring + (c & (kSlots - 1)) * kChunk8;
        c++;
#pragma unroll
        for (int i = 0; i < 4; i++) wA[i] = slot[i * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            if (rr < 2) issue1(c + 2, rr);
#pragma unroll
            for (int ob = 0; ob < 4; ob++)
                acc[4 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[ob][rr], act[gi & 7][rr], acc[4 + ob], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLK_END()
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + act[i][0];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int W>
static void run(const float *pk, float *out, const char *what) {
    const int groups = 2048, blocks = 256 * W * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(kWaves * 64), 0, 0, pk, groups, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = (double)blocks * kWaves * groups * 64 * 2.0 * 16 * 16 * 4;
    const double tf = flop / (ms * 1e-3) / 1e12;
    printf("mode %d, %d wave(s) per SIMD: %-58s %8.2f ms  %7.1f TFLOP/s  %.3f of 157.3  (block 0 ran at %.3f GHz)\n", MODE, W, what, ms, tf, tf / 157.3, held_ghz());
}

int main() {
    float *pk, *out;
    (void)hipMalloc(&pk, 64 * kChunkF4 * 16);
    (void)hipMemset(pk, 0, 64 * kChunkF4 * 16);
    (void)hipMalloc(&out, 4);
    run<0, 1>(pk, out, "MFMAs only");
    run<0, 2>(pk, out, "MFMAs only");
    run<1, 2>(pk, out, "+ 16 ds_read_b128 per 64 MFMAs");
    run<2, 2>(pk, out, "+ s_barrier per 64 MFMAs (4-wave workgroups)");
    run<3, 2>(pk, out, "+ 4 LDS-DMA fragments per wave per 64 MFMAs, vmcnt(8)");
    run<4, 2>(pk, out, "+ bias/ReLU epilogue (128 VALU) every 16 groups");
    run<3, 1>(pk, out, "mode 3 with one wave per SIMD");
    {
        const int groups = 2048, blocks = 256 * 2 * 8;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_loader, dim3(blocks), dim3(5 * 64), 0, 0, pk, groups, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double tf = (double)blocks * kWaves * groups * 64 * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
        printf("mode 5, 2 wave(s) per SIMD: %-58s %8.2f ms  %7.1f TFLOP/s  %.3f of 157.3  (block 0 ran at %.3f GHz)\n",
               "mode 3 with a fifth wave per workgroup as the only DMA issuer", ms, tf, tf / 157.3, held_ghz());
    }
    {
        const int groups = 2048, blocks = 256 * 2 * 8;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_two_tiles, dim3(blocks), dim3(kWaves * 64), 0, 0, pk, groups, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double tf = (double)blocks * kWaves * groups * 64 * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
        printf("mode 6, 2 wave(s) per SIMD: %-58s %8.2f ms  %7.1f TFLOP/s  %.3f of 157.3  (block 0 ran at %.3f GHz)\n",
               "two tiles per wave: 8 ds_read, 2 DMA, 1 barrier per 64 MFMAs", ms, tf, tf / 157.3, held_ghz());
    }
    {
        const int groups = 4096, blocks = 256 * 4 * 8;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_one_tile_w128, dim3(blocks), dim3(kWaves * 64), 0, 0, pk, groups, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double tf = (double)blocks * kWaves * groups * 32 * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
        printf("mode 7, 4 wave(s) per SIMD: %-58s %8.2f ms  %7.1f TFLOP/s  %.3f of 157.3  (block 0 ran at %.3f GHz)\n",
               "one tile, width 128: 8 ds_read, 2 DMA, 1 barrier per 32 MFMAs", ms, tf, tf / 157.3, held_ghz());
    }
    return 0;
}
