// Training step, forward of the two MLP trunks in ONE kernel (occnerf_mlp.py:183-199 as
// the reference trains through it, trainer.py:239-249).
//
// Round 5's forward ran the ten layers as ten streaming passes of csrc/linear.hip: every 256-wide activation is written by one
// pass (512 B per row in bf16) and read back by the next (8 976 B per row in all, 2.0 ms per step for 786 432 rows, ~3.8 TB/s).
// The reads are pure waste -- the renderer's kernels (mlp.hip, mlp16.hip) already keep a sample's activations in registers
// through all ten layers.  This kernel is that scheme on v_mfma_f32_32x32x16_bf16 with ONE product per operand pair (plain
// bf16, BASELINE configs[4]), and it WRITES what the backward needs as it goes: the post-ReLU activation of every hidden layer,
// the geometry head and the bf16 input row, row-major, exactly the tensors the staged forward produced (the backward is
// unchanged).  Per row: 68 fp32 values in, 8 x 512 + 2 x 192 B of saved activations and 16 B of results out -- 4.7 KB
// instead of 9.0 KB, no read of any activation.
//
//   * one wave = 32 samples; a layer is computed transposed, D'[feature][sample] = W x act, so its accumulators -- after bias,
//     ReLU and the rounding to bf16 that the staged forward applied when it stored them -- ARE the next layer's B operands
//     (mlp_layout.h: register <-> feature map); same rounding points as the staged forward, different summation order.
//   * weights: 2-byte stream [k-step][output block][lane] x 16 B, fetched once per 4-wave workgroup by LDS-DMA into a 4-slot
//     ring of 16 KiB chunks (two 16-wide k-steps of 8 output blocks), counted vmcnt + one raw s_barrier per chunk (mlp.hip).
//   * saved activations leave through LDS: a wave writes its 32 x 128 tile in the accumulator layout (8 B per lane), reads it
//     back row by row and stores 16 B per lane -- 4 rows x 256 contiguous bytes per instruction (linear.hip measured what the
//     direct 8-byte scatter costs).  Stores and LDS-DMA share vmcnt and complete out of order with respect to each other:
//     see OCC_ENTER for how the ring's depth keeps the stores off the critical path.
//   * sigma and the three colour logits are VALU dot products over the fp32 accumulators with fp32 weights (as in the
//     renderer's kernels) -- the staged forward took them from bf16 MFMA columns; both are bf16-grade evaluations of the same row.
// One wave per SIMD (128 accumulators + 64 activation + 2 x 64 operand registers).  What bounds it (profiles/r06_trunks_forward.md,
// diagnostic builds OCC_TRUNKS_EXP_*): 1.31 ms for 786 432 rows, of which 1.09 ms remain when nothing is saved -- 1 500 cycles
// per 16-MFMA chunk against 512 of matrix time.  Halving the rendezvous count (two k-steps per chunk) and reading the operands
// one chunk ahead changed nothing: the cost is the ISSUE of the LDS-DMA pieces, 4 per wave per chunk at 100-185 cycles each
// (MI355X_MICROARCH.md) with one wave per SIMD and nothing to cover them -- a cost per KiB of weights, i.e. per 128 samples;
// only more samples per workgroup (64 per wave: twice the accumulators) would amortise it further.
#include "common.h"
#include "mlp_layout.h"

namespace occ {
namespace trunks {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kRing = 4;                                                   // chunks resident / in flight (see OCC_ENTER)
constexpr int kChunkUnits = 1024;                                          // 16-byte units per chunk: 2 k-steps x 8 blocks x 64 lanes
constexpr int kT_L0Geo = (kS_L0Geo + 1) / 2 * 2, kT_L0Rgb = (kS_L0Rgb + 1) / 2 * 2;      // k-steps padded to whole chunks: 6, 10
constexpr int kChunks = kT_L0Geo / 2 + 3 * kS_Hidden / 2 + kS_Hidden / 8 + kT_L0Rgb / 2 + 3 * kS_Hidden / 2;      // 58
constexpr int kTailChunks = kRing;                                         // zero chunks the prefetch may touch (one chunk is read ahead)
constexpr int kStagePitch = 272;                                           // bytes per staged row: 128 features + 16 B pad
constexpr int kStageBytes = 32 * kStagePitch;                              // per wave

// stream offsets in chunks
struct Stream {
    static constexpr int kGeoL0 = 0;
    static constexpr int kGeoH = kGeoL0 + kT_L0Geo / 2;
    static constexpr int kGeoHead = kGeoH + 3 * kS_Hidden / 2;
    static constexpr int kRgbL0 = kGeoHead + kS_Hidden / 8;
    static constexpr int kRgbH = kRgbL0 + kT_L0Rgb / 2;
    static constexpr int kTotal = kRgbH + 3 * kS_Hidden / 2;
};
static_assert(Stream::kTotal == kChunks, "stream layout");

// fp32 side data in LDS (floats): biases in accumulator order + the dot-row weights (as mlp.hip's Aux)
struct Aux {
    static constexpr int kGeoL0B = 0;
    static constexpr int kGeoHB = 256;          // 3 x 256
    static constexpr int kGeoHeadB = 1024;      // 64
    static constexpr int kSigma = 1088;         // 256 weights + bias (+3 pad)
    static constexpr int kRgbL0B = 1348;
    static constexpr int kRgbHB = 1604;         // 3 x 256
    static constexpr int kOut = 2372;           // 3 x 256 weights + 3 biases (+1 pad)
    static constexpr int kTotal = 3144;
};

// [step][ob][lane][8] bf16: element e -> (step, ob, lane, i)
__global__ void pack_layer_kernel(const float *__restrict__ W, int kind, int in_dim, int out_dim, int steps, int ob_count,
                                  __bf16 *__restrict__ Wp) {
    const int total = steps * ob_count * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int i = e & 7, lane = (e >> 3) & 63;
        const int rest = e >> 9;
        const int ob = rest % ob_count, step = rest / ob_count;
        const int col = slot_feature(kind, step * 8 + i, lane >> 5);
        const int row = out_row(kind, ob * 32 + (lane & 31), out_dim);
        Wp[e] = (__bf16)((col >= 0 && row >= 0) ? W[(size_t)row * in_dim + col] : 0.0f);
    }
}

struct FwdArgs {
    const float *agg, *var, *enc;      // [M,35], [M,1], [M,32] fp32
    int64_t M;
    const float *pk;                   // fp32 blob of occnerf_canonical_mlp_pack (biases, sigma row, colour rows)
    const bf16x8 *pkh;                 // the 2-byte weight stream of occnerf_trunks_pack_bf16
    __bf16 *X0;                        // [M,96]   = [agg 35 | var | enc 32 | 0]
    __bf16 *A[4];                      // [M,256]  pts_linears.{0,2,4,6} after ReLU
    __bf16 *GEO;                       // [M,96]   geometry features in columns 0..63, sigma in column 64, zeros
    __bf16 *B[4];                      // [M,256]  rgb_linears.{0,2,4,6} after ReLU
    float *raw4;                       // [M,4]    colour logits, sigma
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

__global__ __launch_bounds__(256, 1) void trunks_forward_kernel(const FwdArgs a) {
    // ONE __shared__ object (a second one makes hipcc drain vmcnt before every ds_read): [ring | stage x 4 waves | aux]
    __shared__ __attribute__((aligned(16))) bf16x8 smem[kRing * kChunkUnits + 4 * kStageBytes / 16 + Aux::kTotal / 4];
    bf16x8 *ring = smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *stage = reinterpret_cast<char *>(smem + kRing * kChunkUnits) + wave * kStageBytes;
    float *aux = reinterpret_cast<float *>(smem + kRing * kChunkUnits + 4 * kStageBytes / 16);
    const int j = lane & 31, h = lane >> 5;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 32;            // first row of this wave's tile
    const int64_t n = m0 + j;
    const int64_t ns = n < a.M ? n : a.M - 1;                            // the whole workgroup stays alive for the barriers

    auto copy = [&](int dst, int64_t src, int count) {
        for (int i = threadIdx.x; i < count; i += 256) aux[dst + i] = a.pk[src + i];
    };
    copy(Aux::kGeoL0B, Blob::kGeoL0B, 256);
    for (int l = 0; l < 3; l++) copy(Aux::kGeoHB + l * 256, Blob::kGeoHW + l * Blob::kHiddenStride + wsz(kG_Hidden, kOB), 256);
    copy(Aux::kGeoHeadB, Blob::kGeoHeadB, 64);
    copy(Aux::kSigma, Blob::kSigmaW, 260);
    copy(Aux::kRgbL0B, Blob::kRgbL0B, 256);
    for (int l = 0; l < 3; l++) copy(Aux::kRgbHB + l * 256, Blob::kRgbHW + l * Blob::kHiddenStride + wsz(kG_Hidden, kOB), 256);
    copy(Aux::kOut, Blob::kOutW, 772);

    // ---- inputs: features h * 34 + t of [agg 35 | var | enc 32] -> the 5 k-steps of the first layer, and the bf16 row X0 ----
    bf16x8 bx[kT_L0Geo];
    {
        float x[8 * kT_L0Geo];
        if (h == 0) {
#pragma unroll
            for (int t = 0; t < 34; t++) x[t] = a.agg[ns * 35 + t];
        } else {
            x[0] = a.agg[ns * 35 + 34];
            x[1] = a.var[ns];
#pragma unroll
            for (int t = 2; t < 34; t++) x[t] = a.enc[ns * 32 + (t - 2)];
        }
#pragma unroll
        for (int t = 34; t < 8 * kT_L0Geo; t++) x[t] = 0.0f;
#pragma unroll
        for (int s = 0; s < kT_L0Geo; s++) {
#pragma unroll
            for (int i = 0; i < 8; i++) bx[s][i] = (__bf16)x[s * 8 + i];
        }
        if (n < a.M) {
            uint32_t *row = reinterpret_cast<uint32_t *>(a.X0 + n * 96 + h * 34);        // 68-byte offset: dword aligned
#pragma unroll
            for (int t = 0; t < 17; t++) row[t] = pack2(x[2 * t], x[2 * t + 1]);
            uint32_t *pad = reinterpret_cast<uint32_t *>(a.X0 + n * 96 + 68 + h * 14);   // columns 68..95, 14 per half
#pragma unroll
            for (int t = 0; t < 7; t++) pad[t] = 0u;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the input loads and the X0 stores, before any DMA is counted
    __syncthreads();

    // ---- weight stream: chunk g lives in ring slot g & 3; a wave fetches 2 of a chunk's 8 KiB ----
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) bf16x8 *)ring;
    auto issue_piece = [&](int g, int f) {
        const int frag = wave * 4 + f;
        glds16(a.pkh + (size_t)g * kChunkUnits + frag * 64, lane * 16,
               ring_lds + (unsigned)(((g & (kRing - 1)) * kChunkUnits + frag * 64) * 16));
    };
    int g = 0;                     // next chunk to consume
#pragma unroll
    for (int c = 0; c < kRing; c++) {      // (one chunk is read ahead into registers: its slot refills a chunk later)
#pragma unroll
        for (int f = 0; f < 4; f++) issue_piece(c, f);
    }
    // Stores and LDS-DMA share vmcnt and retire out of order with respect to each other, so a counted wait is only meaningful
    // while DMAs alone are in flight.  A store phase therefore BEGINS with vmcnt(0) -- the 3 chunks ahead are in LDS (they were
    // issued 2-6 k-steps ago) -- and the next 3 chunk entries need no wait at all; by the 4th, whose DMA was issued after the
    // stores, 6 k-steps (~3 K cycles) have passed and the stores have retired: the counted wait resumes without a stall.
    int landed = 0;                // chunks ahead known to be in LDS (set by a store phase)

    // enter chunk g (to READ it ahead, while chunk g - 1 is multiplied): this wave's pieces of it have landed, rendezvous --
    // every wave holds chunk g - 1 in registers by then, so its slot is free for chunk g + 3
#define OCC_ENTER()                                                                    \
    if (landed > 0) {                                                                  \
        landed--;                                                                      \
    } else {                                                                           \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                               \
    }                                                                                  \
    __builtin_amdgcn_s_barrier();                                                      \
    const bf16x8 *slot_ = ring + (g & (kRing - 1)) * kChunkUnits;                      \
    g++;
#define OCC_REFILL(F)                                   \
    __builtin_amdgcn_sched_barrier(0);                  \
    issue_piece(g + (kRing - 2), F);                    \
    __builtin_amdgcn_sched_barrier(0);
#define OCC_STORE_PHASE()                               \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
    landed = kRing - 1;
#define OCC_MMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A_), (B_), (C_), 0, 0, 0)

    // Two 16-wide k-steps per chunk, 8 output blocks.  The 16 operand fragments of the NEXT chunk are read while the current
    // chunk's 16 MFMAs run: right behind the chunk barrier all four waves would otherwise read (4 x 16 KiB through a 128 B/clk
    // LDS = 512 cycles) and only then multiply (512 cycles) -- measured 1 500 cycles per chunk, profiles/r06_trunks_forward.md.
    // A chunk is 16 fragments x 64 lanes x 16 B whatever the layer: [k-step][block] here, [k-step][2 blocks] x 8 in the head.
    bf16x8 W[16];
#define OCC_NEXT(DST)                                                                      \
    OCC_ENTER()                                                                            \
    _Pragma("unroll") for (int f_ = 0; f_ < 16; f_++) DST[f_] = slot_[f_ * 64 + lane];
#define OCC_LAYER8(STEPS, ACC, BOPS)                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < (STEPS); s_ += 2) {                            \
        bf16x8 nw_[16];                                                                    \
        OCC_NEXT(nw_)                                                                      \
        const bf16x8 b0_ = BOPS(s_), b1_ = BOPS(s_ + 1);                                   \
        _Pragma("unroll") for (int ob_ = 0; ob_ < 4; ob_++) ACC[ob_] = OCC_MMA(W[ob_], b0_, ACC[ob_]); \
        OCC_REFILL(0)                                                                      \
        _Pragma("unroll") for (int ob_ = 4; ob_ < kOB; ob_++) ACC[ob_] = OCC_MMA(W[ob_], b0_, ACC[ob_]); \
        OCC_REFILL(1)                                                                      \
        _Pragma("unroll") for (int ob_ = 0; ob_ < 4; ob_++) ACC[ob_] = OCC_MMA(W[kOB + ob_], b1_, ACC[ob_]); \
        OCC_REFILL(2)                                                                      \
        _Pragma("unroll") for (int ob_ = 4; ob_ < kOB; ob_++) ACC[ob_] = OCC_MMA(W[kOB + ob_], b1_, ACC[ob_]); \
        OCC_REFILL(3)                                                                      \
        _Pragma("unroll") for (int f_ = 0; f_ < 16; f_++) W[f_] = nw_[f_];                 \
    }

    auto bias = [&](f32x16 (&acc_)[kOB], const float *src) {
        const f32x4 *B4 = reinterpret_cast<const f32x4 *>(src);
#pragma unroll
        for (int ob = 0; ob < kOB; ob++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = B4[(ob * 4 + q) * 2 + h];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) acc_[ob][q * 4 + rr] = v[rr];
            }
        }
    };
    // relu(acc) -> bf16: the next layer's 16 B operands AND the saved activation (row-major, through the wave's LDS tile)
    auto relu_keep = [&](bf16x8 (&b)[2 * kOB], const f32x16 (&acc_)[kOB], __bf16 *__restrict__ dst) {
#pragma unroll
        for (int ob = 0; ob < kOB; ob++) {
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
#pragma unroll
                for (int i = 0; i < 8; i++) b[ob * 2 + sub][i] = (__bf16)fmaxf(acc_[ob][sub * 8 + i], 0.0f);
            }
        }
#ifdef OCC_TRUNKS_EXP_NO_SAVE      // (tools/trunks_phases.py: what the kernel costs without writing the activations)
        return;
#endif
#pragma unroll
        for (int half = 0; half < 2; half++) {            // 128 features at a time
#pragma unroll
            for (int ob = 0; ob < 4; ob++) {
#pragma unroll
                for (int sub = 0; sub < 2; sub++) {
                    // b[..][0..3]: features 32 ob + 8 (2 sub) + 4 h + 0..3;  b[..][4..7]: the same + 8
                    const u32x4 v = __builtin_bit_cast(u32x4, b[(half * 4 + ob) * 2 + sub]);
                    char *p = stage + j * kStagePitch + (ob * 32 + 16 * sub + 4 * h) * 2;
                    *reinterpret_cast<u32x2 *>(p) = u32x2{v[0], v[1]};
                    *reinterpret_cast<u32x2 *>(p + 16) = u32x2{v[2], v[3]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's own tile: no barrier needed
            if (half == 0) { OCC_STORE_PHASE() }                        // (as late as possible: the newest DMA is one k-step old)
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int r = it * 4 + (lane >> 4), c = lane & 15;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + r * kStagePitch + c * 16);
#ifdef OCC_TRUNKS_EXP_NO_STORE     // (tools/trunks_phases.py: staging through LDS, no global store)
                if (v[0] == 0x12345678u && m0 + r < a.M) *reinterpret_cast<u32x4 *>(dst + (m0 + r) * 256 + half * 128 + c * 8) = v;
#else
                if (m0 + r < a.M) *reinterpret_cast<u32x4 *>(dst + (m0 + r) * 256 + half * 128 + c * 8) = v;
#endif
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // tile read before the next half overwrites it
        }
    };

    f32x16 acc[kOB];
    bf16x8 bact[2 * kOB];
    OCC_NEXT(W)                    // chunk 0

    // ---------------- geometry trunk ----------------
    bias(acc, aux + Aux::kGeoL0B);
#define BOPS_X(s) bx[s]
    OCC_LAYER8(kT_L0Geo, acc, BOPS_X)
    relu_keep(bact, acc, a.A[0]);
#define BOPS_ACT(s) bact[s]
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        bias(acc, aux + Aux::kGeoHB + l * 256);
        OCC_LAYER8(kS_Hidden, acc, BOPS_ACT)
        relu_keep(bact, acc, a.A[l + 1]);
    }
    // (acc still holds the last hidden layer before its ReLU: sigma from the fp32 values)
    float sigma;
    {
        const f32x4 *W4 = reinterpret_cast<const f32x4 *>(aux + Aux::kSigma);
        float sacc = 0.0f;
#pragma unroll
        for (int kb = 0; kb < kOB; kb++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 w = W4[(kb * 4 + q) * 2 + h];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) sacc = __fmaf_rn(w[rr], fmaxf(acc[kb][q * 4 + rr], 0.0f), sacc);
            }
        }
        sigma = (sacc + __shfl_xor(sacc, 32)) + aux[Aux::kSigma + 256];
    }
    // geometry head: 2 output blocks; a chunk carries 8 k-steps [step][ob][lane]
    f32x16 geo[2];
    {
        const f32x4 *B4 = reinterpret_cast<const f32x4 *>(aux + Aux::kGeoHeadB);
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = B4[(ob * 4 + q) * 2 + h];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) geo[ob][q * 4 + rr] = v[rr];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < kS_Hidden / 8; c++) {
        bf16x8 nw[16];
        OCC_NEXT(nw)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bf16x8 b = bact[c * 8 + q];
#pragma unroll
            for (int ob = 0; ob < 2; ob++) geo[ob] = OCC_MMA(W[q * 2 + ob], b, geo[ob]);
            if (q == 1) { OCC_REFILL(0) }
            if (q == 3) { OCC_REFILL(1) }
            if (q == 5) { OCC_REFILL(2) }
            if (q == 7) { OCC_REFILL(3) }
        }
#pragma unroll
        for (int f = 0; f < 16; f++) W[f] = nw[f];
    }
    bf16x8 bgeo[4];          // 64 geometry features (no activation) as 4 k-steps
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
#pragma unroll
            for (int i = 0; i < 8; i++) bgeo[b * 2 + sub][i] = (__bf16)geo[b][sub * 8 + i];
        }
    }
    OCC_STORE_PHASE()
    if (n < a.M) {           // GEO row: 64 features (8-byte pieces), sigma + zeros; raw4 later with the colour logits
        __bf16 *grow = a.GEO + n * 96;
#pragma unroll
        for (int b = 0; b < 2; b++) {
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const u32x4 v = __builtin_bit_cast(u32x4, bgeo[b * 2 + sub]);
                *reinterpret_cast<u32x2 *>(grow + b * 32 + 16 * sub + 4 * h) = u32x2{v[0], v[1]};
                *reinterpret_cast<u32x2 *>(grow + b * 32 + 16 * sub + 4 * h + 8) = u32x2{v[2], v[3]};
            }
        }
        u32x4 *tail = reinterpret_cast<u32x4 *>(grow + 64 + h * 16);      // columns 64..79 (h = 0: sigma first), 80..95
        u32x4 t0 = u32x4{0u, 0u, 0u, 0u};
        if (h == 0) t0[0] = pack2(sigma, 0.0f);
        tail[0] = t0;
        tail[1] = u32x4{0u, 0u, 0u, 0u};
    }

    // ---------------- colour trunk ----------------
    // (the input row again, from the bf16 copy this lane stored at the top: 20 registers not carried through the geometry trunk)
    {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(a.X0 + ns * 96 + h * 34);
        uint32_t xw[4 * kT_L0Geo];
#pragma unroll
        for (int t = 0; t < 17; t++) xw[t] = n < a.M ? row[t] : 0u;
#pragma unroll
        for (int t = 17; t < 4 * kT_L0Geo; t++) xw[t] = 0u;
#pragma unroll
        for (int s = 0; s < kT_L0Geo; s++) bx[s] = __builtin_bit_cast(bf16x8, u32x4{xw[4 * s], xw[4 * s + 1], xw[4 * s + 2], xw[4 * s + 3]});
    }
    bias(acc, aux + Aux::kRgbL0B);
#define BOPS_RGB0(s) ((s) < 4 ? bgeo[(s) & 3] : bx[((s) - 4) < 0 ? 0 : ((s) - 4) >= kT_L0Geo ? kT_L0Geo - 1 : ((s) - 4)])
    OCC_LAYER8(kT_L0Rgb, acc, BOPS_RGB0)
    relu_keep(bact, acc, a.B[0]);
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        bias(acc, aux + Aux::kRgbHB + l * 256);
        OCC_LAYER8(kS_Hidden, acc, BOPS_ACT)
        relu_keep(bact, acc, a.B[l + 1]);
    }
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const f32x4 *W4 = reinterpret_cast<const f32x4 *>(aux + Aux::kOut + c * kWidth);
        float sacc = 0.0f;
#pragma unroll
        for (int kb = 0; kb < kOB; kb++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 w = W4[(kb * 4 + q) * 2 + h];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) sacc = __fmaf_rn(w[rr], fmaxf(acc[kb][q * 4 + rr], 0.0f), sacc);
            }
        }
        rgb[c] = (sacc + __shfl_xor(sacc, 32)) + aux[Aux::kOut + 3 * kWidth + c];
    }
    if (h == 0 && n < a.M) *reinterpret_cast<f32x4 *>(a.raw4 + n * 4) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the 3 tail chunks: nobody reads them
#undef BOPS_X
#undef BOPS_ACT
#undef BOPS_RGB0
#undef OCC_LAYER8
#undef OCC_NEXT
#undef OCC_MMA
#undef OCC_REFILL
#undef OCC_STORE_PHASE
#undef OCC_ENTER
}

}  // namespace trunks
}  // namespace occ

OCC_API int64_t occnerf_trunks_packed_bytes(void) {
    return (int64_t)(occ::trunks::kChunks + occ::trunks::kTailChunks) * occ::trunks::kChunkUnits * 16;
}

/* h_W: host array of the ten device pointers pts_linears.{0,2,4,6}, geo_linear.0, rgb_linears.{0,2,4,6}, output_linear.0
 * (torch layout [out, in], fp32; the last one is not streamed: the colour rows come from occnerf_canonical_mlp_pack's blob).
 * packed: occnerf_trunks_packed_bytes() bytes, zero-initialised by the caller once (the tail chunks stay zero). */
OCC_API int occnerf_trunks_pack_bf16(const float *const *h_W, void *packed, void *stream) {
    using namespace occ;
    using namespace occ::trunks;
    OCC_REQUIRE(h_W && packed, "trunks_pack_bf16: null argument");
    for (int i = 0; i < 9; i++) OCC_REQUIRE(h_W[i], "trunks_pack_bf16: layer %d missing", i);
    hipStream_t st = as_stream(stream);
    __bf16 *base = reinterpret_cast<__bf16 *>(packed);
    auto layer = [&](int li, int kind, int in_dim, int out_dim, int steps, int ob, int chunk) {
        hipLaunchKernelGGL(trunks::pack_layer_kernel, dim3(128), dim3(256), 0, st, h_W[li], kind, in_dim, out_dim, steps, ob,
                           base + (size_t)chunk * kChunkUnits * 8);
    };
    layer(0, kL0Geo, kInGeo, kWidth, kT_L0Geo, kOB, Stream::kGeoL0);
    for (int l = 0; l < 3; l++) layer(1 + l, kHidden, kWidth, kWidth, kS_Hidden, kOB, Stream::kGeoH + l * kS_Hidden / 2);
    layer(4, kGeoHead, kWidth, 65, kS_Hidden, 2, Stream::kGeoHead);
    layer(5, kL0Rgb, kInRgb, kWidth, kT_L0Rgb, kOB, Stream::kRgbL0);
    for (int l = 0; l < 3; l++) layer(6 + l, kHidden, kWidth, kWidth, kS_Hidden, kOB, Stream::kRgbH + l * kS_Hidden / 2);
    return check_launch("trunks_pack_bf16");
}

/* The forward of both trunks for M rows in one launch (bf16 arithmetic, fp32 accumulation).  agg[M,35], var[M,1], enc[M,32]
 * fp32; packed_f32 = the blob of occnerf_canonical_mlp_pack (biases, sigma row, colour rows), packed_bf16 = the stream of
 * occnerf_trunks_pack_bf16.  Outputs, all row-major: X0[M,96], A1..A4[M,256], GEO[M,96], B1..B4[M,256] in bf16 (what
 * occnerf_linear_forward wrote layer by layer) and raw4[M,4] fp32 = (colour logits, sigma). */
OCC_API int occnerf_trunks_forward_bf16(const float *agg, const float *var, const float *enc, int64_t M, const float *packed_f32,
                                        const void *packed_bf16, void *X0, void *const *h_A /*4*/, void *GEO,
                                        void *const *h_B /*4*/, float *raw4, void *stream) {
    using namespace occ;
    if (M <= 0) return 0;
    OCC_REQUIRE(agg && var && enc && packed_f32 && packed_bf16 && X0 && h_A && GEO && h_B && raw4, "trunks_forward_bf16: null argument");
    trunks::FwdArgs a;
    a.agg = agg, a.var = var, a.enc = enc, a.M = M, a.pk = packed_f32;
    a.pkh = reinterpret_cast<const trunks::bf16x8 *>(packed_bf16);
    a.X0 = reinterpret_cast<__bf16 *>(X0);
    a.GEO = reinterpret_cast<__bf16 *>(GEO);
    for (int i = 0; i < 4; i++) {
        OCC_REQUIRE(h_A[i] && h_B[i], "trunks_forward_bf16: activation buffer %d missing", i);
        a.A[i] = reinterpret_cast<__bf16 *>(h_A[i]);
        a.B[i] = reinterpret_cast<__bf16 *>(h_B[i]);
    }
    a.raw4 = raw4;
    const int64_t blocks = (M + 127) / 128;
    OCC_REQUIRE(blocks < (1ll << 31), "trunks_forward_bf16: M too large");
    hipLaunchKernelGGL(trunks::trunks_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    return check_launch("trunks_forward_bf16");
}
