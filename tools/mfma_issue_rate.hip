// Bottom-up attribution of the fp32 matrix pipe's issue rate on MI355X: what v_mfma_f32_16x16x4_f32 sustains per SIMD with
// the ingredients of csrc/mlp16.hip added one at a time (2 waves per SIMD, 16 independent accumulators per wave, groups of
// 64 MFMAs = one 16 KiB weight chunk):
//   mode 0  MFMAs only, operands in registers
//   mode 1  + 16 ds_read_b128 per group (the chunk's A operands, read one half-group ahead)
//   mode 2  + one s_barrier per group (4-wave workgroups)
//   mode 3  + 4 LDS-DMA fragments per wave per group (global_load_lds_dwordx4, SGPR base + lane offset) and the counted vmcnt
//   mode 4  mode 3 + a 64-instruction VALU epilogue (bias + ReLU) every 16 groups
//   mode 5-7  a dedicated loader wave; two sample tiles per wave (csrc/nonrigid16.hip's steady state); one tile at four waves per SIMD
//   mode 8  the TILES of csrc/nonrigid16.hip (46 chunks, two of them quarter-filled) with its MFMA-free phases added one at a time:
//           bias/ReLU per layer, the output layer as VALU dots, inputs through the row list + the sincos embedding
//   mode 9  mode 8 with the next tile's inputs and embedding moved inside the current tile's last layer
//   mode 10 mode 9 with the two dependent loads issued layers ahead of their use
//   mode 13 (MIR_MODE13=1) mode 3 with the DMA fragments issued as `buffer_load_dwordx4 off, V#, soffset lds` through a
//           descriptor with ADD_TID_ENABLE: no address VGPR
//   mode 11 mode 10 with every request and first use placed right after a chunk's rendezvous (before that chunk's DMA issue)
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
    // mode 13: the same fragments by `buffer_load_dwordx4 off, V#, soffset lds` through a descriptor with ADD_TID_ENABLE
    // (address = base + soffset + 16 x lane formed by the address unit: no address VGPR is read at all)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long sbase = (unsigned long long)(size_t)stream;
    const u32x4 vdesc = {(unsigned)sbase, (unsigned)((sbase >> 32) & 0xFFFFu) | (16u << 16), 1u << 28, (1u << 23) | 0x7000u};
    auto issue1b = [&](int c, int f) {
        const int frag = wave * 4 + f;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %1, %2 lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(vdesc), "s"((unsigned)(((c & 63) * kChunkF4 + frag * 64) * 16)),
                       "s"(ring_lds + (unsigned)(((c & (kSlots - 1)) * kChunkF4 + frag * 64) * 16))
                     : "memory");
    };
#define ISSUE1(C, F) do { if (MODE == 13) issue1b((C), (F)); else issue1((C), (F)); } while (0)
    int c = 0;
    if (MODE >= 3) {
        for (int j = 0; j < 3; j++)
            for (int f = 0; f < 4; f++) ISSUE1(j, f);
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
            if (MODE >= 3) ISSUE1(c + 2, rr);
#pragma unroll
            for (int ob = 0; ob < 8; ob++)
                acc[8 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[ob][rr], act[gi & 15][rr], acc[8 + ob], 0, 0, 0);
        }
        if (MODE >= 4 && MODE < 13 && (gi & 15) == 15) {
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


// modes 8-10: mode 6's steady state cut into the TILES of csrc/nonrigid16.hip -- 46 chunks per 2 x 16 samples per wave, the third
// chunk of layer 0 and the eleventh of the skip layer carrying 1 k-step of 4 -- with the kernel's MFMA-free phases added:
//   PH & 1  per layer (6 per tile): bias from LDS (8 ds_read_b128) + ReLU (64 v_max)
//   PH & 2  per tile: the 128 -> 3 output layer as VALU dot products (24 ds_read_b128, 192 fma, 12 cross-lane adds) + the store
//   PH & 4  per tile: inputs through a row list (two dependent global loads) and 2 x 10 sincosf for the embedding
//   PH & 8  (with PH & 4) the NEXT tile's inputs are requested and embedded inside the current tile's last layer instead
// The persistent workgroup walks `tiles` tiles.  FLOP counts executed MFMAs only (40 full chunks + 6 quarter... see run_tiles).
template <int PH>
__global__ __launch_bounds__(kWaves * 64, 2) void k_nr_tiles(const float *__restrict__ pk, int tiles, const int *__restrict__ rows,
                                                             const float *__restrict__ xyz, float *out) {
    constexpr int kChunk8 = 512;
    __shared__ __attribute__((aligned(16))) f32x4 smem[kSlots * kChunk8 + 512];
    f32x4 *ring = smem;
    const f32x4 *aux = smem + kSlots * kChunk8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    for (int i = threadIdx.x; i < kSlots * kChunk8 + 512; i += kWaves * 64) smem[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    CLK_BEGIN()
    f32x4 acc[2][8], act[2][8];
    float e[2][12], en[2][12], p[2][3];
#pragma unroll
    for (int T = 0; T < 2; T++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc[T][i] = f32x4{0, 0, 0, 0};
            act[T][i] = f32x4{lane * 1e-3f, 1.f, 2.f, 3.f};
        }
#pragma unroll
        for (int i = 0; i < 12; i++) e[T][i] = en[T][i] = 1e-3f * (i + lane);
#pragma unroll
        for (int i = 0; i < 3; i++) p[T][i] = 0.f;
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
    int rq[2];
    float xq[2][3];
    auto request_rows = [&](int64_t n) {
#pragma unroll
        for (int T = 0; T < 2; T++) rq[T] = rows[(n + T * 16 + (lane & 15)) & 0xFFFFF];
    };
    auto request_xyz = [&]() {
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) xq[T][c2] = xyz[(int64_t)rq[T] * 3 + c2];
    };
    auto embed_from = [&](float (&ee)[2][12]) {
#pragma unroll
        for (int T = 0; T < 2; T++) {
#pragma unroll
            for (int m = 0; m < 5; m++) {
                const int A = m < 4 ? m * 4 + g : 16 + (g >> 1);
                const int oct = A / 3, c2 = A - 3 * oct;
                const float a = (c2 == 0 ? xq[T][0] : c2 == 1 ? xq[T][1] : xq[T][2]) * (float)(1 << oct);
                float sa, ca;
                sincosf(a, &sa, &ca);
                if (m < 4) {
                    ee[T][2 * m] = sa;
                    ee[T][2 * m + 1] = ca;
                } else {
                    ee[T][8] = (g & 1) ? ca : sa;
                }
            }
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) p[T][c2] = xq[T][c2];
        }
    };
    auto embed = [&](int64_t n, float (&ee)[2][12]) {          // inputs of one tile pair + their embedding
#pragma unroll
        for (int T = 0; T < 2; T++) {
            const int r = rows[(n + T * 16 + (lane & 15)) & 0xFFFFF];
            float q[3];
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) q[c2] = xyz[(int64_t)r * 3 + c2];
#pragma unroll
            for (int m = 0; m < 5; m++) {
                const int A = m < 4 ? m * 4 + g : 16 + (g >> 1);
                const int oct = A / 3, c2 = A - 3 * oct;
                const float a = (c2 == 0 ? q[0] : c2 == 1 ? q[1] : q[2]) * (float)(1 << oct);
                float sa, ca;
                sincosf(a, &sa, &ca);
                if (m < 4) {
                    ee[T][2 * m] = sa;
                    ee[T][2 * m + 1] = ca;
                } else {
                    ee[T][8] = (g & 1) ? ca : sa;
                }
            }
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) p[T][c2] = q[c2];
        }
    };
    int c = 0;
    for (int j = 0; j < 3; j++)
        for (int f = 0; f < 2; f++) issue1(j, f);
    f32x4 wA[4], wB[4];
#pragma unroll
    for (int i = 0; i < 4; i++) wA[i] = wB[i] = f32x4{1e-3f, 2e-3f, 1e-3f, 2e-3f};
    const f32x4 *slot = ring;
    if ((PH & 4) && (PH & 8)) embed((int64_t)(blockIdx.x * kWaves + wave) * 32, en);
    rq[0] = rq[1] = 0;
    xq[0][0] = xq[0][1] = xq[0][2] = xq[1][0] = xq[1][1] = xq[1][2] = 0.f;
#pragma unroll 1
    for (int tile = 0; tile < tiles; tile++) {
        const int64_t n0 = ((int64_t)(tile * gridDim.x + blockIdx.x) * kWaves + wave) * 32;
        if (PH & 4) {
            if (PH & 8) {
#pragma unroll
                for (int T = 0; T < 2; T++)
#pragma unroll
                    for (int i = 0; i < 12; i++) e[T][i] = en[T][i];
            } else {
                embed(n0, e);
            }
        }
        // one chunk: CI = chunk index inside its layer (compile time), KS = k-steps of the layer, BOP(T, t) = B operand of k-step t
#define NRT_CHUNK_H(CI, KS, BOP, HOOK)                                                                                    \
        {                                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) wB[i] = slot[(4 + i) * 64 + lane];                        \
            _Pragma("unroll") for (int rr = 0; rr < 4; rr++) {                                                      \
                if ((CI) * 4 + rr < (KS)) {                                                                         \
                    _Pragma("unroll") for (int ob = 0; ob < 4; ob++)                                                \
                        _Pragma("unroll") for (int T = 0; T < 2; T++)                                               \
                            acc[T][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ob][rr], BOP(T, (CI) * 4 + rr), acc[T][ob], 0, 0, 0); \
                }                                                                                                   \
            }                                                                                                       \
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                             \
            __builtin_amdgcn_s_barrier();                                                                           \
            slot = ring + (c & (kSlots - 1)) * kChunk8;                                                             \
            c++;                                                                                                    \
            HOOK                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; i++) wA[i] = slot[i * 64 + lane];                              \
            _Pragma("unroll") for (int rr = 0; rr < 4; rr++) {                                                      \
                if (rr < 2) issue1(c + 2, rr);                                                                      \
                if ((CI) * 4 + rr < (KS)) {                                                                         \
                    _Pragma("unroll") for (int ob = 0; ob < 4; ob++)                                                \
                        _Pragma("unroll") for (int T = 0; T < 2; T++)                                               \
                            acc[T][4 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[ob][rr], BOP(T, (CI) * 4 + rr), acc[T][4 + ob], 0, 0, 0); \
                }                                                                                                   \
            }                                                                                                       \
        }
#define NRT_CHUNK(CI, KS, BOP) NRT_CHUNK_H(CI, KS, BOP, )
#define NRT_LAYER_END(BOFF)                                                                                         \
        if (PH & 1) {                                                                                               \
            _Pragma("unroll") for (int T = 0; T < 2; T++)                                                           \
                _Pragma("unroll") for (int ob = 0; ob < 8; ob++)                                                    \
                    _Pragma("unroll") for (int r = 0; r < 4; r++) act[T][ob][r] = fmaxf(acc[T][ob][r], 0.0f);       \
            _Pragma("unroll") for (int ob = 0; ob < 8; ob++) {                                                      \
                const f32x4 b = aux[(BOFF) * 32 + ob * 4 + g];                                                      \
                acc[0][ob] = b;                                                                                     \
                acc[1][ob] = b;                                                                                     \
            }                                                                                                       \
        }
#define BOP_E(T, t) e[T][(t) < 12 ? (t) : 0]
#define BOP_A(T, t) act[T][((t) >> 2) & 7][(t) & 3]
#define BOP_S(T, t) ((t) < 32 ? act[T][((t) >> 2) & 7][(t) & 3] : e[T][(t) - 32 < 12 && (t) >= 32 ? (t) - 32 : 0])
        NRT_CHUNK(0, 9, BOP_E) NRT_CHUNK(1, 9, BOP_E) NRT_CHUNK(2, 9, BOP_E)
        NRT_LAYER_END(0)
#pragma unroll 1
        for (int l = 0; l < 4; l++) {
            if (l == 3) {
                NRT_CHUNK(0, 41, BOP_S) NRT_CHUNK(1, 41, BOP_S) NRT_CHUNK(2, 41, BOP_S) NRT_CHUNK(3, 41, BOP_S)
                NRT_CHUNK(4, 41, BOP_S) NRT_CHUNK(5, 41, BOP_S) NRT_CHUNK(6, 41, BOP_S) NRT_CHUNK(7, 41, BOP_S)
                NRT_CHUNK(8, 41, BOP_S) NRT_CHUNK(9, 41, BOP_S) NRT_CHUNK(10, 41, BOP_S)
                NRT_LAYER_END(5)
            }
            // mode 10/11: the dependent loads are issued, and their results first touched, RIGHT AFTER a chunk's rendezvous --
            // where the compiler's `s_waitcnt vmcnt(0)` for them finds the ring's DMAs landed (they were issued a chunk ago)
            NRT_CHUNK_H(0, 32, BOP_A, if ((PH & 32) && l == 0) request_rows(n0 + (int64_t)gridDim.x * kWaves * 32);)
            NRT_CHUNK(1, 32, BOP_A)
            NRT_CHUNK_H(2, 32, BOP_A, if ((PH & 32) && l == 0) request_xyz();)
            NRT_CHUNK_H(3, 32, BOP_A, if ((PH & 32) && l == 1) embed_from(en);)
            if ((PH & 4) && (PH & 8) && !(PH & 48) && l == 3)    // next tile's inputs + embedding inside the last layer
                embed(n0 + (int64_t)gridDim.x * kWaves * 32, en);
            if (PH & 16) {                                       // ... requested two layers ahead of their use (mode 10)
                if (l == 0) request_rows(n0 + (int64_t)gridDim.x * kWaves * 32);
                if (l == 1) request_xyz();
                if (l == 3) embed_from(en);
            }
            NRT_CHUNK(4, 32, BOP_A) NRT_CHUNK(5, 32, BOP_A) NRT_CHUNK(6, 32, BOP_A) NRT_CHUNK(7, 32, BOP_A)
            NRT_LAYER_END(l + 1)
        }
#undef NRT_CHUNK
#undef NRT_CHUNK_H
#undef NRT_LAYER_END
#undef BOP_E
#undef BOP_A
#undef BOP_S
        if (PH & 2) {
#pragma unroll
            for (int T = 0; T < 2; T++) {
                float off[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float sum = 0.0f;
#pragma unroll
                    for (int ob = 0; ob < 8; ob++) {
                        const f32x4 w = aux[256 + ch * 32 + ob * 4 + g];
#pragma unroll
                        for (int r = 0; r < 4; r++) sum = fmaf(w[r], act[T][ob][r], sum);
                    }
                    sum += __shfl_xor(sum, 16);
                    off[ch] = sum + __shfl_xor(sum, 32);
                }
                if (g == 0) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) out[16 + ((n0 + T * 16 + (lane & 15)) & 0xFFFFF) * 3 + ch] = p[T][ch] + off[ch];
                }
            }
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

template <int PH>
static void run_tiles(const float *pk, const int *rows, const float *xyz, float *out, const char *what, int mode) {
    const int tiles = 48, blocks = 256 * 2;                  // persistent: 2 workgroups per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_nr_tiles<PH>), dim3(blocks), dim3(kWaves * 64), 0, 0, pk, tiles, rows, xyz, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfma_per_tile = (44 * 4 + 2 * 1) * 8 * 2.0;              // k-steps x output blocks x two sample tiles
    const double tf = (double)blocks * kWaves * tiles * mfma_per_tile * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
    printf("mode %d, 2 wave(s) per SIMD: %-58s %8.2f ms  %7.1f TFLOP/s  %.3f of 157.3  (block 0 ran at %.3f GHz)\n", mode, what, ms,
           tf, tf / 157.3, held_ghz());
}


// mode 12: mode 3 with HALF the rendezvous and DMA-issue count per MFMA: one workgroup of EIGHT waves per CU (still two waves
// per SIMD) shares a 4-slot ring of 32 KiB chunks = 128 MFMAs per wave per barrier (32 ds_read_b128, 4 DMA fragments, 1
// eight-wave barrier per 128 MFMAs).  The experiment VERDICT r02 item 9 asks for before touching csrc/mlp16.hip.
__global__ __launch_bounds__(8 * 64, 1) void k_big_chunks(const float *__restrict__ pk, int groups, float *out) {
    constexpr int kChunk32 = 2048, kW8 = 8;
    __shared__ __attribute__((aligned(16))) f32x4 ring[kSlots * kChunk32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kSlots * kChunk32; i += kW8 * 64) ring[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
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
                     : "v"(lane * 16), "s"(stream + (size_t)(c & 31) * kChunk32 + frag * 64),
                       "s"(ring_lds + (unsigned)(((c & (kSlots - 1)) * kChunk32 + frag * 64) * 16))
                     : "memory");
    };
    int c = 0;
    for (int j = 0; j < 3; j++)
        for (int f = 0; f < 4; f++) issue1(j, f);
    f32x4 wA[8], wB[8];
#pragma unroll
    for (int i = 0; i < 8; i++) wA[i] = wB[i] = f32x4{1e-3f, 2e-3f, 1e-3f, 2e-3f};
    const f32x4 *slot = ring;
#pragma unroll 1
    for (int gi = 0; gi < groups; gi++) {
        // quarters 0..3 of the current slot: (ob 0-7, k 0-3), (ob 8-15, k 0-3), (ob 0-7, k 4-7), (ob 8-15, k 4-7)
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            f32x4 *cur = (qd & 1) ? wB : wA, *nxt = (qd & 1) ? wA : wB;
            if (qd < 3) {
#pragma unroll
                for (int i = 0; i < 8; i++) nxt[i] = slot[((qd + 1) * 8 + i) * 64 + lane];
            } else {
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                slot = ring + (c & (kSlots - 1)) * kChunk32;
                c++;
#pragma unroll
                for (int i = 0; i < 8; i++) nxt[i] = slot[i * 64 + lane];
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                if (qd == 3) issue1(c + 2, rr);
#pragma unroll
                for (int ob = 0; ob < 8; ob++)
                    acc[(qd & 1) * 8 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[ob][rr], act[(gi * 2 + (qd >> 1)) & 15][rr],
                                                                                  acc[(qd & 1) * 8 + ob], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLK_END()
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + act[i][0];
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
    if (getenv("MIR_MODE13")) run<13, 2>(pk, out, "mode 3, DMA by buffer_load ... lds with ADD_TID (no address VGPR)");
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
    {
        const int groups = 1024, blocks = 256 * 8;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_big_chunks, dim3(blocks), dim3(8 * 64), 0, 0, pk, groups, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double tf = (double)blocks * 8 * groups * 128 * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
        printf("mode 12, 2 wave(s) per SIMD: %-57s %8.2f ms  %7.1f TFLOP/s  %.3f of 157.3  (block 0 ran at %.3f GHz)\n",
               "mode 3 with 8-wave workgroups, 32 KiB chunks: 1 barrier, 4 DMA per 128 MFMAs", ms, tf, tf / 157.3, held_ghz());
    }
    {
        int *rows;
        float *xyz, *big;
        (void)hipMalloc(&rows, (1 << 20) * 4);
        (void)hipMalloc(&xyz, (size_t)(1 << 20) * 12);
        (void)hipMalloc(&big, (size_t)(16 + 3 * (1 << 20)) * 4);
        (void)hipMemset(xyz, 0, (size_t)(1 << 20) * 12);
        int *h = (int *)malloc((1 << 20) * 4);
        for (int i = 0; i < (1 << 20); i++) h[i] = (int)(((long long)i * 2654435761ll) & 0xFFFFF);      // a scattered row list
        (void)hipMemcpy(rows, h, (1 << 20) * 4, hipMemcpyHostToDevice);
        run_tiles<0>(pk, rows, xyz, big, "nr16 tiles (46 chunks, two quarter chunks), MFMA + ring only", 8);
        run_tiles<1>(pk, rows, xyz, big, "+ bias/ReLU per layer", 8);
        run_tiles<3>(pk, rows, xyz, big, "+ output layer as VALU dots + store per tile", 8);
        run_tiles<7>(pk, rows, xyz, big, "+ inputs through the row list + sincos embedding per tile", 8);
        run_tiles<15>(pk, rows, xyz, big, "same, next tile's inputs + embedding inside the last layer", 9);
        run_tiles<13>(pk, rows, xyz, big, "mode 9 without the output dots", 9);
        run_tiles<31>(pk, rows, xyz, big, "row ids requested 3 layers, positions 2 layers before the embedding", 10);
        run_tiles<47>(pk, rows, xyz, big, "same, every request / first use right after a chunk rendezvous", 11);
    }
    return 0;
}
