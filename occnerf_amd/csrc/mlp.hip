// Canonical density/colour MLP on fp32 MFMA -- the 32-sample-per-wave,
// direct-load kernel (occnerf_canonical_mlp_direct; the default is the LDS-staged kernel in mlp16.hip)
// and the split-bf16 variants:
//   geometry trunk  68 -> 256 -> 256 -> 256 -> 256 -> 65   (sigma = row 0)
//   colour trunk   131 -> 256 -> 256 -> 256 -> 256 -> 3
// occnerf_mlp.py:183-199, 461 568 MAC = 923 136 FLOP per sample -- the dominant
// arithmetic of the whole path and the kernel BASELINE.json's roofline is quoted on.
//
// Design (CDNA4, v_mfma_f32_32x32x2_f32, exact fp32 = fmaf chain):
//  * one wave = 32 samples, all ten layers, activations never leave registers.
//    The layer is computed transposed, D[feature][sample] = W[feature][k] * act[k][sample]:
//    in the 32x32 C/D layout lane l holds sample l&31 and, in register r, feature
//    (r&3) + 8*(r>>2) + 4*(l>>5) of a 32-feature block.  For the NEXT layer the B operand
//    of k-step r wants "k=0 from lanes 0-31, k=1 from lanes 32-63" -- which is exactly
//    register r of that block (features f and f+4).  So a layer's output registers ARE the
//    next layer's B operands; only the weights (A operand) have to be stored in the matching
//    k order, which occnerf_canonical_mlp_pack does once per checkpoint.
//  * A operand (weights): one float4 per lane per 4 k-steps, laid out [group][out-block]
//    [lane] so a wave reads 1 KiB contiguous; streamed from L2 with one group (8 KiB) of
//    register prefetch -- 2048 MFMA cycles of cover for an L2 hit.  All four waves of the
//    workgroup read the same stream, so L1 absorbs 3 of 4 reads.
//  * sigma (1 row) and the 3-row colour head are 256-long dot products: VALU fma over the
//    lane's 128 features + one cross-half shuffle, instead of a 97 %-empty MFMA block.
//  * 1 wave per SIMD (acc 128 + act 128 + inputs 36 + weight buffers 64 registers).
//
// MFMA count per wave: 288 + 3*1024 + 256 + 544 + 3*1024 = 7232 (ideal 7212): 99.7 % of
// the issued matrix work is algorithmic.  Roofline: fp32 MFMA, 157.3 TFLOP/s.
#include "common.h"
#include "mlp_layout.h"
#include "split.h"

namespace occ {

__global__ void pack_layer_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                  int kind, int in_dim, int out_dim, int groups, int ob_count,
                                  float *__restrict__ Wp, float *__restrict__ Bp) {
    const int total = groups * ob_count * 64 * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int rr = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
        const int ob = rest % ob_count, g = rest / ob_count;
        const int col = slot_feature(kind, g * 4 + rr, lane >> 5);
        const int row = out_row(kind, ob * 32 + (lane & 31), out_dim);
        Wp[e] = (col >= 0 && row >= 0) ? W[(size_t)row * in_dim + col] : 0.0f;
    }
    // bias in accumulator order: [ob][q][half][4]
    const int nb = ob_count * 32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nb; e += gridDim.x * blockDim.x) {
        const int rr = e & 3, h = (e >> 2) & 1, q = (e >> 3) & 3, ob = e >> 5;
        const int row = out_row(kind, ob * 32 + rr + 8 * q + 4 * h, out_dim);
        Bp[e] = row >= 0 ? b[row] : 0.0f;
    }
}

// rows evaluated as VALU dot products: [row][kb][q][half][4], bias appended
__global__ void pack_rows_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                 int nrows, float *__restrict__ Wp, float *__restrict__ Bp) {
    const int total = nrows * kWidth;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int rr = e & 3, h = (e >> 2) & 1, q = (e >> 3) & 3, kb = (e >> 5) & 7, row = e >> 8;
        Wp[e] = W[(size_t)row * kWidth + kb * 32 + rr + 8 * q + 4 * h];
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) Bp[threadIdx.x] = (int)threadIdx.x < nrows ? b[threadIdx.x] : 0.0f;
}

// ---------------------------------------------------------------------------------------
// the MLP kernel
// ---------------------------------------------------------------------------------------
#define OCC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int OB>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[OB], const float *__restrict__ Bp, int h) {
    const f32x4 *B4 = reinterpret_cast<const f32x4 *>(Bp);
#pragma unroll
    for (int ob = 0; ob < OB; ob++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v = B4[(ob * 4 + q) * 2 + h];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) acc[ob][q * 4 + rr] = v[rr];
        }
    }
}

template <int OB>
__device__ __forceinline__ void load_group(f32x4 (&w)[OB], const f32x4 *__restrict__ W4, int g,
                                           int lane) {
#pragma unroll
    for (int ob = 0; ob < OB; ob++) w[ob] = W4[(g * OB + ob) * 64 + lane];
}

// One dense layer: acc[ob] += W(group g) x B-operand(k-step t) for all groups.
// BOP(t) must be an expression with compile-time register indices after unrolling.
#define OCC_LAYER(GROUPS, OB, W4PTR, ACC, BOP)                                      \
    {                                                                               \
        f32x4 wc_[OB], wn_[OB];                                                     \
        load_group<OB>(wc_, (W4PTR), 0, lane);                                      \
        _Pragma("unroll") for (int g_ = 0; g_ < (GROUPS); g_++) {                   \
            if (g_ + 1 < (GROUPS)) load_group<OB>(wn_, (W4PTR), g_ + 1, lane);      \
            _Pragma("unroll") for (int rr_ = 0; rr_ < 4; rr_++) {                   \
                const int t_ = g_ * 4 + rr_;                                        \
                _Pragma("unroll") for (int ob_ = 0; ob_ < (OB); ob_++)              \
                    ACC[ob_] = OCC_MFMA(wc_[ob_][rr_], BOP(t_), ACC[ob_]);          \
            }                                                                       \
            _Pragma("unroll") for (int ob_ = 0; ob_ < (OB); ob_++) wc_[ob_] = wn_[ob_]; \
        }                                                                           \
    }

__device__ __forceinline__ void relu_into(f32x16 (&act)[kOB], const f32x16 (&acc)[kOB]) {
#pragma unroll
    for (int ob = 0; ob < kOB; ob++) {
#pragma unroll
        for (int r = 0; r < 16; r++) act[ob][r] = fmaxf(acc[ob][r], 0.0f);
    }
}

// dot product of one packed weight row with the lane's 128 activations; the other 128 live
// in the partner half-wave (lane ^ 32)
__device__ __forceinline__ float dot_row(const f32x16 (&act)[kOB], const float *__restrict__ Wrow,
                                         int h) {
    const f32x4 *W4 = reinterpret_cast<const f32x4 *>(Wrow);
    float s = 0.0f;
#pragma unroll
    for (int kb = 0; kb < kOB; kb++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 w = W4[(kb * 4 + q) * 2 + h];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) s = __fmaf_rn(w[rr], act[kb][q * 4 + rr], s);
        }
    }
    return s + __shfl_xor(s, 32);
}

__global__ __launch_bounds__(256, 1) void canonical_mlp_kernel(const float *__restrict__ mlp_in,
                                                               int64_t N,
                                                               const float *__restrict__ pk,
                                                               float *__restrict__ raw) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= N) return;
    const int64_t n = tile * 32 + j;
    const int64_t nsrc = n < N ? n : N - 1;

    // layer-0 B operands: k-step t carries feature t (lanes 0-31) / 34 + t (lanes 32-63)
    float x[kXRegs];
    {
        const float *src = mlp_in + nsrc * kInGeo + h * 34;
#pragma unroll
        for (int t = 0; t < 34; t += 2) {
            const float2 v = *reinterpret_cast<const float2 *>(src + t);
            x[t] = v.x;
            x[t + 1] = v.y;
        }
        x[34] = 0.0f;
        x[35] = 0.0f;
    }

    f32x16 acc[kOB], act[kOB];

    // ---------------- geometry trunk ----------------
    load_bias<kOB>(acc, pk + Blob::kGeoL0B, h);
#define BOP_X(t) x[t]
    OCC_LAYER(kG_L0Geo, kOB, reinterpret_cast<const f32x4 *>(pk + Blob::kGeoL0W), acc, BOP_X)
    relu_into(act, acc);
#define BOP_ACT(t) act[(t) >> 4][(t) & 15]
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        const float *base = pk + Blob::kGeoHW + l * Blob::kHiddenStride;
        load_bias<kOB>(acc, base + wsz(kG_Hidden, kOB), h);
        OCC_LAYER(kG_Hidden, kOB, reinterpret_cast<const f32x4 *>(base), acc, BOP_ACT)
        relu_into(act, acc);
    }
    // geometry head: 64 features on MFMA (two blocks, no activation), sigma as a dot row
    f32x16 geo[2];
    load_bias<2>(geo, pk + Blob::kGeoHeadB, h);
    OCC_LAYER(kG_Hidden, 2, reinterpret_cast<const f32x4 *>(pk + Blob::kGeoHeadW), geo, BOP_ACT)
    const float sigma = dot_row(act, pk + Blob::kSigmaW, h) + pk[Blob::kSigmaB];

    // ---------------- colour trunk ----------------
    load_bias<kOB>(acc, pk + Blob::kRgbL0B, h);
#define BOP_RGB0(t) ((t) < 32 ? geo[((t) >> 4) & 1][(t) & 15] : x[((t) - 32) < 0 ? 0 : ((t) - 32)])
    OCC_LAYER(kG_L0Rgb, kOB, reinterpret_cast<const f32x4 *>(pk + Blob::kRgbL0W), acc, BOP_RGB0)
    relu_into(act, acc);
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        const float *base = pk + Blob::kRgbHW + l * Blob::kHiddenStride;
        load_bias<kOB>(acc, base + wsz(kG_Hidden, kOB), h);
        OCC_LAYER(kG_Hidden, kOB, reinterpret_cast<const f32x4 *>(base), acc, BOP_ACT)
        relu_into(act, acc);
    }
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; c++)
        rgb[c] = dot_row(act, pk + Blob::kOutW + c * kWidth, h) + pk[Blob::kOutB + c];

    if (h == 0 && n < N) {
        float *o = raw + n * 5;
        o[0] = rgb[0];
        o[1] = rgb[1];
        o[2] = rgb[2];
        o[3] = sigma;
    }
#undef BOP_X
#undef BOP_ACT
#undef BOP_RGB0
}


// =======================================================================================
// bf16x3 variant: the same network on the bf16 matrix pipe with split operands.
//
// Every fp32 operand v is written v = hi + lo, hi = bf16(v), lo = bf16(v - hi) (16 significand
// bits together), and each product is formed as Wh*xh + Wh*xl + Wl*xh with fp32 accumulation
// (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate -> 5.3x after the three products).  The
// dropped Wl*xl term and the representation error are ~2^-17 relative per product; measured
// on the reference's own inputs the raw outputs move by <= 1.2e-5
// and pixels by <= 5.4e-7 (rgb) / 1.3e-6 (depth), two orders inside the 1e-4 gate, where
// plain bf16 (2.2e-4 / 7.6e-4) fails it.
//
// Structure is the fp32 kernel's: one wave = 32 samples, accumulators of a layer become the
// next layer's B operands after bias/ReLU and the hi/lo split (done in registers, 3 VALU per
// value).  A k-step is now 16 wide: lane half h supplies 8 consecutive registers of a block.
// =======================================================================================
// bf16 blob layout in units of bf16x8 (16 bytes): [step][hi|lo][ob][lane]
constexpr int64_t bsz(int steps, int ob) { return (int64_t)steps * 2 * ob * 64; }
struct BlobH {
    static constexpr int64_t kGeoL0 = 0;
    static constexpr int64_t kGeoH = kGeoL0 + bsz(kS_L0Geo, kOB);         // 3 layers
    static constexpr int64_t kHiddenStride = bsz(kS_Hidden, kOB);
    static constexpr int64_t kGeoHead = kGeoH + 3 * kHiddenStride;
    static constexpr int64_t kRgbL0 = kGeoHead + bsz(kS_Hidden, 2);
    static constexpr int64_t kRgbH = kRgbL0 + bsz(kS_L0Rgb, kOB);
    static constexpr int64_t kTotal = kRgbH + 3 * kHiddenStride;            // in bf16x8 units
};

template <typename P>
__global__ void pack_layer_split_kernel(const float *__restrict__ W, int kind, int in_dim, int out_dim,
                                        int steps, int ob_count, typename P::E *__restrict__ Wp) {
    // element e -> (step, which, ob, lane, i)
    const int total = steps * 2 * ob_count * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int i = e & 7, lane = (e >> 3) & 63;
        int rest = e >> 9;
        const int ob = rest % ob_count;
        rest /= ob_count;
        const int which = rest & 1, step = rest >> 1;
        const int col = slot_feature(kind, step * 8 + i, lane >> 5);
        const int row = out_row(kind, ob * 32 + (lane & 31), out_dim);
        const float w = (col >= 0 && row >= 0) ? W[(size_t)row * in_dim + col] : 0.0f;
        const typename P::E hi = P::w_hi(w);
        Wp[e] = which == 0 ? hi : P::w_lo(w, hi);
    }
}

template <typename P>
struct SplitT {      // B operand of one 16-wide k-step: 8 values per lane, as hi and lo pieces
    typename P::V8 hi, lo;
};
typedef SplitT<Bf16x3> SplitB;

template <typename P>
__device__ __forceinline__ SplitT<P> split8t(const float (&v)[8]) {
    SplitT<P> o;
    P::split8(v, o.hi, o.lo);
    return o;
}
__device__ __forceinline__ SplitB split8(const float (&v)[8]) { return split8t<Bf16x3>(v); }

#define OCC_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <int OB>
__device__ __forceinline__ void load_step(bf16x8 (&ah)[OB], bf16x8 (&al)[OB], const bf16x8 *__restrict__ Wp,
                                          int s, int lane) {
#pragma unroll
    for (int ob = 0; ob < OB; ob++) {
        ah[ob] = Wp[((s * 2 + 0) * OB + ob) * 64 + lane];
        al[ob] = Wp[((s * 2 + 1) * OB + ob) * 64 + lane];
    }
}

// acc[ob] += Wh*xh + Wh*xl + Wl*xh over STEPS k-steps; BOPS(s) yields the SplitB of step s
#define OCC_LAYER_BF16(STEPS, OB, WPTR, ACC, BOPS)                                          \
    {                                                                                       \
        bf16x8 ah_[OB], al_[OB], nh_[OB], nl_[OB];                                          \
        load_step<OB>(ah_, al_, (WPTR), 0, lane);                                           \
        _Pragma("unroll") for (int s_ = 0; s_ < (STEPS); s_++) {                            \
            if (s_ + 1 < (STEPS)) load_step<OB>(nh_, nl_, (WPTR), s_ + 1, lane);            \
            const SplitB &b_ = BOPS(s_);                                                    \
            _Pragma("unroll") for (int ob_ = 0; ob_ < (OB); ob_++)                          \
                ACC[ob_] = OCC_MFMA_BF16(ah_[ob_], b_.hi, ACC[ob_]);                        \
            _Pragma("unroll") for (int ob_ = 0; ob_ < (OB); ob_++)                          \
                ACC[ob_] = OCC_MFMA_BF16(ah_[ob_], b_.lo, ACC[ob_]);                        \
            _Pragma("unroll") for (int ob_ = 0; ob_ < (OB); ob_++)                          \
                ACC[ob_] = OCC_MFMA_BF16(al_[ob_], b_.hi, ACC[ob_]);                        \
            _Pragma("unroll") for (int ob_ = 0; ob_ < (OB); ob_++) {                        \
                ah_[ob_] = nh_[ob_];                                                        \
                al_[ob_] = nl_[ob_];                                                        \
            }                                                                               \
        }                                                                                   \
    }

// relu(acc) -> the 16 split B operands of the next layer (2 per 32-feature block)
template <typename P, bool WATCH = true>
__device__ __forceinline__ void relu_split(SplitT<P> (&b)[2 * kOB], const f32x16 (&acc)[kOB], float &amax) {
#pragma unroll
    for (int ob = 0; ob < kOB; ob++) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = P::relu(acc[ob][sub * 8 + i]);
            b[ob * 2 + sub] = split8t<P>(v);
            if constexpr (WATCH) P::watch_hi(amax, b[ob * 2 + sub].hi);
        }
    }
}
template <typename P>
__device__ __forceinline__ void relu_split(SplitT<P> (&b)[2 * kOB], const f32x16 (&acc)[kOB]) {
    float unused = 0.0f;
    relu_split<P, false>(b, acc, unused);
}

__device__ __forceinline__ float dot_row_relu(const f32x16 (&acc)[kOB], const float *__restrict__ Wrow, int h) {
    const f32x4 *W4 = reinterpret_cast<const f32x4 *>(Wrow);
    float s = 0.0f;
#pragma unroll
    for (int kb = 0; kb < kOB; kb++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 w = W4[(kb * 4 + q) * 2 + h];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) s = __fmaf_rn(w[rr], fmaxf(acc[kb][q * 4 + rr], 0.0f), s);
        }
    }
    return s + __shfl_xor(s, 32);
}

__global__ __launch_bounds__(256, 1) void canonical_mlp_bf16x3_kernel(
    const float *__restrict__ mlp_in, const int32_t *__restrict__ in_rows /*nullable: input row of entry n*/, int64_t N_max,
    const int32_t *__restrict__ n_dev /*nullable: device-side entry count*/, const float *__restrict__ pk,
    const bf16x8 *__restrict__ pkh, float *__restrict__ raw) {
    const int64_t N = n_dev ? (int64_t)*n_dev : N_max;
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= N) return;
    const int64_t n = tile * 32 + j;
    const int64_t nsrc0 = n < N ? n : N - 1;
    const int64_t nsrc = in_rows ? (int64_t)in_rows[nsrc0] : nsrc0;

    // layer-0 operands: slot t of half h carries input feature h*34 + t (t < 34), 5 k-steps
    SplitB bx[kS_L0Geo];
    {
        const float *src = mlp_in + nsrc * kInGeo + h * 34;
#pragma unroll
        for (int s = 0; s < kS_L0Geo; s++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (s * 8 + i) < 34 ? src[s * 8 + i] : 0.0f;
            bx[s] = split8(v);
        }
    }

    f32x16 acc[kOB];
    SplitB bact[2 * kOB];

    // ---------------- geometry trunk ----------------
    load_bias<kOB>(acc, pk + Blob::kGeoL0B, h);
#define BOPS_X(s) bx[s]
    OCC_LAYER_BF16(kS_L0Geo, kOB, pkh + BlobH::kGeoL0, acc, BOPS_X)
    relu_split(bact, acc);
#define BOPS_ACT(s) bact[s]
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        load_bias<kOB>(acc, pk + Blob::kGeoHW + l * Blob::kHiddenStride + wsz(kG_Hidden, kOB), h);
        OCC_LAYER_BF16(kS_Hidden, kOB, pkh + BlobH::kGeoH + l * BlobH::kHiddenStride, acc, BOPS_ACT)
        if (l < 2) relu_split(bact, acc);
    }
    // acc = pre-activation of the last geometry hidden layer: sigma from the fp32 values
    const float sigma = dot_row_relu(acc, pk + Blob::kSigmaW, h) + pk[Blob::kSigmaB];
    relu_split(bact, acc);
    f32x16 geo[2];
    load_bias<2>(geo, pk + Blob::kGeoHeadB, h);
    OCC_LAYER_BF16(kS_Hidden, 2, pkh + BlobH::kGeoHead, geo, BOPS_ACT)
    SplitB bgeo[4];          // 64 geometry features (no activation) as 4 k-steps
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = geo[b][sub * 8 + i];
            bgeo[b * 2 + sub] = split8(v);
        }
    }

    // ---------------- colour trunk ----------------
    load_bias<kOB>(acc, pk + Blob::kRgbL0B, h);
#define BOPS_RGB0(s) ((s) < 4 ? bgeo[(s) & 3] : bx[((s) - 4) < 0 ? 0 : ((s) - 4)])
    OCC_LAYER_BF16(kS_L0Rgb, kOB, pkh + BlobH::kRgbL0, acc, BOPS_RGB0)
    relu_split(bact, acc);
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        load_bias<kOB>(acc, pk + Blob::kRgbHW + l * Blob::kHiddenStride + wsz(kG_Hidden, kOB), h);
        OCC_LAYER_BF16(kS_Hidden, kOB, pkh + BlobH::kRgbH + l * BlobH::kHiddenStride, acc, BOPS_ACT)
        if (l < 2) relu_split(bact, acc);
    }
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; c++)
        rgb[c] = dot_row_relu(acc, pk + Blob::kOutW + c * kWidth, h) + pk[Blob::kOutB + c];

    if (h == 0 && n < N) {
        float *o = raw + n * 5;
        o[0] = rgb[0];
        o[1] = rgb[1];
        o[2] = rgb[2];
        o[3] = sigma;
    }
#undef BOPS_X
#undef BOPS_ACT
#undef BOPS_RGB0
}


// ---------------------------------------------------------------------------------------
// bf16x3, LDS-staged weights.  The direct-load kernel above makes each of the 4 waves of a
// workgroup pull the whole 1.8 MB weight stream through L1 itself (85 B/clk/CU against a
// 64 B/clk L1): it runs at ~40 % of the bf16 pipe.  Here the stream is cut into 16 KiB chunks
// (one 16-wide k-step x {hi,lo} x 8 output blocks) that the workgroup fetches ONCE with
// LDS-DMA (global_load_lds_dwordx4, 4 x 1 KiB per wave per chunk) into a 4-slot ring and all
// four waves read with ds_read_b128.  One raw s_barrier per chunk; the DMA of chunks g+1..g+3
// stays in flight across it behind a counted s_waitcnt vmcnt(8) (hipcc would drain to
// vmcnt(0) before the first ds_read if the DMA were a builtin it can see, so the DMA is inline
// asm and the wait is placed by hand).  Biases and the VALU head rows are copied to LDS once.
// ---------------------------------------------------------------------------------------
constexpr int kRingSlots = 4;
constexpr int kChunkUnits = 1024;                                  // 16-byte units per chunk
constexpr int kChunksTotal = kS_L0Geo + 3 * kS_Hidden + kS_Hidden / 4 + kS_L0Rgb + 3 * kS_Hidden;
static_assert(BlobH::kTotal == (int64_t)kChunksTotal * kChunkUnits, "bf16 blob is the chunk stream");
constexpr int kTailChunks = kRingSlots - 1;                        // zero chunks the prefetch may touch

// fp32 side data in LDS (floats): biases in accumulator order + the dot-row weights
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

template <int OB>
__device__ __forceinline__ void lds_bias(f32x16 (&acc)[OB], const float *aux, int h) {
    const f32x4 *B4 = reinterpret_cast<const f32x4 *>(aux);
#pragma unroll
    for (int ob = 0; ob < OB; ob++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v = B4[(ob * 4 + q) * 2 + h];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) acc[ob][q * 4 + rr] = v[rr];
        }
    }
}

template <typename P, bool WATCH /* P::kBounded policies: track the clamped values and report leaving the domain (split.h) */>
__global__ __launch_bounds__(256, 1) void canonical_mlp_split_lds_kernel(
    const float *__restrict__ mlp_in, const int32_t *__restrict__ in_rows /*nullable: input row of entry n*/, int64_t N_max,
    const int32_t *__restrict__ n_dev /*nullable: device-side entry count*/, const float *__restrict__ pk,
    const typename P::V8 *__restrict__ pkh, float *__restrict__ raw, uint32_t *__restrict__ domain_flag /*nullable*/) {
    typedef typename P::V8 V8;
    typedef SplitT<P> Split;
    float amax = 0.0f;      // P::kBounded policies: running packed-half maximum of the clamped values' hi pieces (split.h)
    constexpr float kSx = P::kSx, kInvSx = 1.0f / P::kSx;
    const int64_t N = n_dev ? (int64_t)*n_dev : N_max;
    if ((int64_t)blockIdx.x * 128 >= N) return;      // launches are sized for the worst case; uniform per workgroup
    // ONE __shared__ object (a second one makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) V8 smem[kRingSlots * kChunkUnits + Aux::kTotal / 4];
    V8 *ring = smem;
    float *aux = reinterpret_cast<float *>(smem + kRingSlots * kChunkUnits);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n = tile * 32 + j;
    const int64_t nsrc0 = n < N ? n : N - 1;      // whole workgroup stays alive for the barriers
    const int64_t nsrc = in_rows ? (int64_t)in_rows[nsrc0] : nsrc0;

    // ---- side data -> LDS, inputs -> registers (ordinary loads, before any DMA is in flight) ----
    // (the biases of the MFMA layers travel in the activations' scale; the two VALU head rows and their biases do not)
    auto copy = [&](int dst, int64_t src, int count, float scale) {
        for (int i = threadIdx.x; i < count; i += 256) aux[dst + i] = pk[src + i] * scale;
    };
    copy(Aux::kGeoL0B, Blob::kGeoL0B, 256, kSx);
    for (int l = 0; l < 3; l++) copy(Aux::kGeoHB + l * 256, Blob::kGeoHW + l * Blob::kHiddenStride + wsz(kG_Hidden, kOB), 256, kSx);
    copy(Aux::kGeoHeadB, Blob::kGeoHeadB, 64, kSx);
    copy(Aux::kSigma, Blob::kSigmaW, 260, 1.0f);
    copy(Aux::kRgbL0B, Blob::kRgbL0B, 256, kSx);
    for (int l = 0; l < 3; l++) copy(Aux::kRgbHB + l * 256, Blob::kRgbHW + l * Blob::kHiddenStride + wsz(kG_Hidden, kOB), 256, kSx);
    copy(Aux::kOut, Blob::kOutW, 772, 1.0f);

    Split bx[kS_L0Geo];
    {
        const float *src = mlp_in + nsrc * kInGeo + h * 34;
#pragma unroll
        for (int s = 0; s < kS_L0Geo; s++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (s * 8 + i) < 34 ? P::sym(src[s * 8 + i] * kSx) : 0.0f;
            if constexpr (WATCH) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if ((s * 8 + i) < 34) P::watch_abs(amax, src[s * 8 + i] * kSx);
            }
            bx[s] = split8t<P>(v);
        }
    }
    __syncthreads();

    // ---- weight stream: chunk g lives in ring slot g & 3 ----
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) V8 *)ring;
    auto issue = [&](int g) {      // this wave's quarter of chunk g: 4 x 1 KiB
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int frag = wave * 4 + f;
            glds16(pkh + (size_t)g * kChunkUnits + frag * 64, lane * 16,
                   ring_lds + (unsigned)(((g & (kRingSlots - 1)) * kChunkUnits + frag * 64) * 16));
        }
    };
    int g = 0;                     // next chunk to consume
    issue(0);
    issue(1);
    issue(2);

    // wait for chunk g (own quarter), rendezvous, refill the slot freed by chunk g-1
#define OCC_CHUNK_ENTER()                                              \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   \
    __builtin_amdgcn_s_barrier();                                      \
    issue(g + 3);                                                      \
    const V8 *slot_ = ring + (g & (kRingSlots - 1)) * kChunkUnits; \
    g++;

    // The four LDS-DMA pieces of the refill are not issued
    // right behind the barrier, amid the 16 ds_read_b128 of the k-step (where one piece costs the issuing wave 100-185 cycles:
    // MI355X_MICROARCH.md), but one by one in the second half of the k-step's MFMAs.  Measured with s_memtime stamps
    // (profiles/r05_split_kernel_phases.md): a hidden layer 17.5 K instead of 19.0 K cycles, a tile 147.7 K instead of 153.5 K,
    // the launch 39.8 instead of 40.3 ms -- two thirds of the cycles saved come back as a lower clock
    auto issue_piece = [&](int g, int f) {
        const int frag = wave * 4 + f;
        glds16(pkh + (size_t)g * kChunkUnits + frag * 64, lane * 16,
               ring_lds + (unsigned)(((g & (kRingSlots - 1)) * kChunkUnits + frag * 64) * 16));
    };
#define OCC_PIECE(F)                                    \
    __builtin_amdgcn_sched_barrier(0);                  \
    issue_piece(g + 2, F);                              \
    __builtin_amdgcn_sched_barrier(0);

    // one 16-wide k-step per chunk, 8 output blocks
#define OCC_LAYER_LDS8(STEPS, ACC, BOPS)                                                   \
    _Pragma("unroll") for (int s_ = 0; s_ < (STEPS); s_++) {                               \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                               \
            __builtin_amdgcn_s_barrier();                                                  \
            const V8 *slot_ = ring + (g & (kRingSlots - 1)) * kChunkUnits;                 \
            g++;                                                                           \
            V8 ah_[kOB], al_[kOB];                                                         \
            _Pragma("unroll") for (int ob_ = 0; ob_ < kOB; ob_++) ah_[ob_] = slot_[ob_ * 64 + lane];         \
            const Split &b_ = BOPS(s_);                                                    \
            const V8 b3_ = P::third(b_.hi);                                                \
            _Pragma("unroll") for (int ob_ = 0; ob_ < kOB; ob_++) ACC[ob_] = P::mfma(ah_[ob_], b_.hi, ACC[ob_]); \
            _Pragma("unroll") for (int ob_ = 0; ob_ < kOB; ob_++) al_[ob_] = slot_[(kOB + ob_) * 64 + lane]; \
            _Pragma("unroll") for (int ob_ = 0; ob_ < 4; ob_++) ACC[ob_] = P::mfma(ah_[ob_], b_.lo, ACC[ob_]); \
            OCC_PIECE(0)                                                                   \
            _Pragma("unroll") for (int ob_ = 4; ob_ < kOB; ob_++) ACC[ob_] = P::mfma(ah_[ob_], b_.lo, ACC[ob_]); \
            OCC_PIECE(1)                                                                   \
            _Pragma("unroll") for (int ob_ = 0; ob_ < 4; ob_++) ACC[ob_] = P::mfma(al_[ob_], b3_, ACC[ob_]);   \
            OCC_PIECE(2)                                                                   \
            _Pragma("unroll") for (int ob_ = 4; ob_ < kOB; ob_++) ACC[ob_] = P::mfma(al_[ob_], b3_, ACC[ob_]); \
            OCC_PIECE(3)                                                                   \
    }

    f32x16 acc[kOB];
    Split bact[2 * kOB];

    // ---------------- geometry trunk ----------------
    lds_bias<kOB>(acc, aux + Aux::kGeoL0B, h);
#define BOPS_X(s) bx[s]
    OCC_LAYER_LDS8(kS_L0Geo, acc, BOPS_X)
    relu_split<P, WATCH>(bact, acc, amax);
#define BOPS_ACT(s) bact[s]
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        lds_bias<kOB>(acc, aux + Aux::kGeoHB + l * 256, h);
        OCC_LAYER_LDS8(kS_Hidden, acc, BOPS_ACT)
        if (l < 2) relu_split<P, WATCH>(bact, acc, amax);
    }
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
        sigma = (sacc + __shfl_xor(sacc, 32)) * kInvSx + aux[Aux::kSigma + 256];
    }
    relu_split<P, WATCH>(bact, acc, amax);
    // geometry head: 2 output blocks; a chunk carries 4 k-steps [step][hi|lo][ob][lane]
    f32x16 geo[2];
    lds_bias<2>(geo, aux + Aux::kGeoHeadB, h);
#pragma unroll
    for (int c = 0; c < kS_Hidden / 4; c++) {
        OCC_CHUNK_ENTER()
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const Split &b = bact[c * 4 + q];
            const V8 b3 = P::third(b.hi);
            V8 ah[2], al[2];
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                ah[ob] = slot_[((q * 2 + 0) * 2 + ob) * 64 + lane];
                al[ob] = slot_[((q * 2 + 1) * 2 + ob) * 64 + lane];
            }
#pragma unroll
            for (int ob = 0; ob < 2; ob++) geo[ob] = P::mfma(ah[ob], b.hi, geo[ob]);
#pragma unroll
            for (int ob = 0; ob < 2; ob++) geo[ob] = P::mfma(ah[ob], b.lo, geo[ob]);
#pragma unroll
            for (int ob = 0; ob < 2; ob++) geo[ob] = P::mfma(al[ob], b3, geo[ob]);
        }
    }
    Split bgeo[4];
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = P::sym(geo[b][sub * 8 + i]);
            if constexpr (WATCH) {
#pragma unroll
                for (int i = 0; i < 8; i++) P::watch_abs(amax, geo[b][sub * 8 + i]);
            }
            bgeo[b * 2 + sub] = split8t<P>(v);
        }
    }

    // ---------------- colour trunk ----------------
    lds_bias<kOB>(acc, aux + Aux::kRgbL0B, h);
#define BOPS_RGB0(s) ((s) < 4 ? bgeo[(s) & 3] : bx[((s) - 4) < 0 ? 0 : ((s) - 4)])
    OCC_LAYER_LDS8(kS_L0Rgb, acc, BOPS_RGB0)
    relu_split<P, WATCH>(bact, acc, amax);
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        lds_bias<kOB>(acc, aux + Aux::kRgbHB + l * 256, h);
        OCC_LAYER_LDS8(kS_Hidden, acc, BOPS_ACT)
        if (l < 2) relu_split<P, WATCH>(bact, acc, amax);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the 3 tail chunks: nobody reads them
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
        rgb[c] = (sacc + __shfl_xor(sacc, 32)) * kInvSx + aux[Aux::kOut + 3 * kWidth + c];
    }
    if (h == 0 && n < N) {
        float *o = raw + n * 5;
        o[0] = rgb[0];
        o[1] = rgb[1];
        o[2] = rgb[2];
        o[3] = sigma;
    }
    if constexpr (WATCH) split_report<P>(amax, domain_flag);
#undef BOPS_X
#undef BOPS_ACT
#undef BOPS_RGB0
#undef OCC_LAYER_LDS8
#undef OCC_PIECE
#undef OCC_CHUNK_ENTER
}


}  // namespace occ

OCC_API int64_t occnerf_canonical_mlp_packed_floats(void) { return occ::Blob::kTotal + occ::mlp_lds_packed_floats(); }

OCC_API int occnerf_canonical_mlp_pack(const float *const *h_W, const float *const *h_b,
                                       float *packed, void *stream) {
    using namespace occ;
    OCC_REQUIRE(h_W && h_b && packed, "canonical_mlp_pack: null argument");
    for (int i = 0; i < 10; i++) OCC_REQUIRE(h_W[i] && h_b[i], "canonical_mlp_pack: layer %d missing", i);
    hipStream_t st = as_stream(stream);
    auto layer = [&](int li, int kind, int in_dim, int out_dim, int groups, int ob, int64_t woff, int64_t boff) {
        hipLaunchKernelGGL(pack_layer_kernel, dim3(256), dim3(256), 0, st, h_W[li], h_b[li], kind, in_dim,
                           out_dim, groups, ob, packed + woff, packed + boff);
    };
    layer(0, kL0Geo, kInGeo, kWidth, kG_L0Geo, kOB, Blob::kGeoL0W, Blob::kGeoL0B);
    for (int l = 0; l < 3; l++) {
        const int64_t base = Blob::kGeoHW + l * Blob::kHiddenStride;
        layer(1 + l, kHidden, kWidth, kWidth, kG_Hidden, kOB, base, base + wsz(kG_Hidden, kOB));
    }
    layer(4, kGeoHead, kWidth, 65, kG_Hidden, 2, Blob::kGeoHeadW, Blob::kGeoHeadB);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(4), dim3(256), 0, st, h_W[4], h_b[4], 1,
                       packed + Blob::kSigmaW, packed + Blob::kSigmaB);
    layer(5, kL0Rgb, kInRgb, kWidth, kG_L0Rgb, kOB, Blob::kRgbL0W, Blob::kRgbL0B);
    for (int l = 0; l < 3; l++) {
        const int64_t base = Blob::kRgbHW + l * Blob::kHiddenStride;
        layer(6 + l, kHidden, kWidth, kWidth, kG_Hidden, kOB, base, base + wsz(kG_Hidden, kOB));
    }
    hipLaunchKernelGGL(pack_rows_kernel, dim3(4), dim3(256), 0, st, h_W[9], h_b[9], 3,
                       packed + Blob::kOutW, packed + Blob::kOutB);
    if (int rc = check_launch("canonical_mlp_pack")) return rc;
    return mlp_lds_pack(h_W, h_b, packed + Blob::kTotal, st);
}

OCC_API int64_t occnerf_canonical_mlp_packed_bf16_bytes(void) {
    return (occ::BlobH::kTotal + (int64_t)occ::kTailChunks * occ::kChunkUnits) * 16;   // + zero tail chunks
}

template <typename P>
static int mlp_pack_split(const float *const *h_W, void *packed_split, void *stream) {
    using namespace occ;
    OCC_REQUIRE(h_W && packed_split, "canonical_mlp_pack (split): null argument");
    for (int i = 0; i < 10; i++) OCC_REQUIRE(h_W[i], "canonical_mlp_pack (split): layer %d missing", i);
    hipStream_t st = as_stream(stream);
    typename P::E *base = reinterpret_cast<typename P::E *>(packed_split);
    auto layer = [&](int li, int kind, int in_dim, int out_dim, int steps, int ob, int64_t off) {
        hipLaunchKernelGGL(pack_layer_split_kernel<P>, dim3(256), dim3(256), 0, st, h_W[li], kind, in_dim, out_dim,
                           steps, ob, base + off * 8);
    };
    layer(0, kL0Geo, kInGeo, kWidth, kS_L0Geo, kOB, BlobH::kGeoL0);
    for (int l = 0; l < 3; l++) layer(1 + l, kHidden, kWidth, kWidth, kS_Hidden, kOB, BlobH::kGeoH + l * BlobH::kHiddenStride);
    layer(4, kGeoHead, kWidth, 65, kS_Hidden, 2, BlobH::kGeoHead);
    layer(5, kL0Rgb, kInRgb, kWidth, kS_L0Rgb, kOB, BlobH::kRgbL0);
    for (int l = 0; l < 3; l++) layer(6 + l, kHidden, kWidth, kWidth, kS_Hidden, kOB, BlobH::kRgbH + l * BlobH::kHiddenStride);
    return check_launch("canonical_mlp_pack (split)");
}

OCC_API int occnerf_canonical_mlp_pack_bf16(const float *const *h_W, void *packed_bf16, void *stream) {
    return mlp_pack_split<occ::Bf16x3>(h_W, packed_bf16, stream);
}

OCC_API int occnerf_canonical_mlp_pack_f16(const float *const *h_W, void *packed_f16, void *stream) {
    return mlp_pack_split<occ::F16x3>(h_W, packed_f16, stream);
}

static int mlp_bf16x3_launch(const float *mlp_in, const int32_t *in_rows, int64_t N_max, const int32_t *n_dev,
                             const float *packed, const void *packed_bf16, float *raw, int32_t variant, void *stream) {
    using namespace occ;
    const int64_t blocks = (N_max + 127) / 128;
    OCC_REQUIRE(blocks < (1ll << 31), "canonical_mlp_bf16x3: N too large");
    const bf16x8 *pkh = reinterpret_cast<const bf16x8 *>(packed_bf16);
    if (variant == 0)
        hipLaunchKernelGGL((canonical_mlp_split_lds_kernel<Bf16x3, false>), dim3((unsigned)blocks), dim3(256), 0,
                           as_stream(stream), mlp_in, in_rows, N_max, n_dev, packed, pkh, raw, (uint32_t *)nullptr);
    else
        hipLaunchKernelGGL(canonical_mlp_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0,
                           as_stream(stream), mlp_in, in_rows, N_max, n_dev, packed, pkh, raw);
    return check_launch("canonical_mlp_bf16x3");
}

OCC_API int occnerf_canonical_mlp_bf16x3(const float *mlp_in, int64_t N, const float *packed,
                                         const void *packed_bf16, float *raw, int32_t variant,
                                         void *stream) {
    if (N <= 0) return 0;
    OCC_REQUIRE(mlp_in && packed && packed_bf16 && raw, "canonical_mlp_bf16x3: null argument");
    return mlp_bf16x3_launch(mlp_in, nullptr, N, nullptr, packed, packed_bf16, raw, variant, stream);
}

/* The fp32-grade split (F16x3 above): same packed fp32 blob for biases / head rows, weights from occnerf_canonical_mlp_pack_f16.
 * in_rows / n_dev nullable (all N_max rows, identity). */
OCC_API int occnerf_canonical_mlp_f16x3(const float *mlp_in, const int32_t *in_rows, int64_t N_max, const int32_t *n_dev,
                                        const float *packed, const void *packed_f16, float *raw, uint32_t *domain_flag,
                                        void *stream) {
    using namespace occ;
    if (N_max <= 0) return 0;
    OCC_REQUIRE(mlp_in && packed && packed_f16 && raw, "canonical_mlp_f16x3: null argument");
    OCC_REQUIRE(!in_rows || n_dev, "canonical_mlp_f16x3: a row list needs its device-side count");
    const int64_t blocks = (N_max + 127) / 128;
    OCC_REQUIRE(blocks < (1ll << 31), "canonical_mlp_f16x3: N too large");
    const f16x8 *pkh = reinterpret_cast<const f16x8 *>(packed_f16);
    const dim3 grid((unsigned)blocks), wg(256);
    // (the watched form costs ~6 % of the launch: callers that pass no flag word get the round-5 kernel)
    if (domain_flag)
        hipLaunchKernelGGL((canonical_mlp_split_lds_kernel<F16x3, true>), grid, wg, 0, as_stream(stream), mlp_in, in_rows, N_max, n_dev,
                           packed, pkh, raw, domain_flag);
    else
        hipLaunchKernelGGL((canonical_mlp_split_lds_kernel<F16x3, false>), grid, wg, 0, as_stream(stream), mlp_in, in_rows, N_max, n_dev,
                           packed, pkh, raw, domain_flag);
    return check_launch("canonical_mlp_f16x3");
}

OCC_API int occnerf_canonical_mlp_bf16x3_rows(const float *mlp_in, const int32_t *in_rows, int64_t N_max,
                                              const int32_t *n_dev, const float *packed, const void *packed_bf16,
                                              float *raw, int32_t variant, void *stream) {
    if (N_max <= 0) return 0;
    OCC_REQUIRE(mlp_in && n_dev && packed && packed_bf16 && raw, "canonical_mlp_bf16x3_rows: null argument");
    return mlp_bf16x3_launch(mlp_in, in_rows, N_max, n_dev, packed, packed_bf16, raw, variant, stream);
}

OCC_API int occnerf_canonical_mlp(const float *mlp_in, int64_t N, const float *packed, float *raw,
                                  void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(mlp_in && packed && raw, "canonical_mlp: null argument");
    return mlp_lds_launch(mlp_in, nullptr, N, nullptr, packed + Blob::kTotal, raw, as_stream(stream));
}

OCC_API int occnerf_canonical_mlp_counted(const float *mlp_in, int64_t N_max, const int32_t *n_dev,
                                          const float *packed, float *raw, void *stream) {
    using namespace occ;
    if (N_max <= 0) return 0;
    OCC_REQUIRE(mlp_in && n_dev && packed && raw, "canonical_mlp_counted: null argument");
    return mlp_lds_launch(mlp_in, nullptr, N_max, n_dev, packed + Blob::kTotal, raw, as_stream(stream));
}

OCC_API int occnerf_canonical_mlp_rows(const float *mlp_in, const int32_t *in_rows, int64_t N_max, const int32_t *n_dev,
                                       const float *packed, float *raw, void *stream) {
    using namespace occ;
    if (N_max <= 0) return 0;
    OCC_REQUIRE(mlp_in && in_rows && n_dev && packed && raw, "canonical_mlp_rows: null argument");
    return mlp_lds_launch(mlp_in, in_rows, N_max, n_dev, packed + Blob::kTotal, raw, as_stream(stream));
}

OCC_API int occnerf_canonical_mlp_direct(const float *mlp_in, int64_t N, const float *packed, float *raw,
                                         void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(mlp_in && packed && raw, "canonical_mlp_direct: null argument");
    const int64_t blocks = (N + 127) / 128;
    OCC_REQUIRE(blocks < (1ll << 31), "canonical_mlp_direct: N too large");
    hipLaunchKernelGGL(canonical_mlp_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       mlp_in, N, packed, raw);
    return check_launch("canonical_mlp_direct");
}
