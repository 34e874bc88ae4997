// Pose-conditioned non-rigid offset MLP on fp32 MFMA: the 32-sample-wave
// direct-load kernel (occnerf_nonrigid_direct; the default is the LDS-staged kernel in nonrigid16.hip) and
// the split-bf16 variant.
//
//   emb = Hann-windowed Fourier embedding of xyz, 6 octaves x (sin, cos) x 3 = 36
//         (embedders/hannw_fourier.py:9-63; the window is all ones at render time)
//   h   = [cond(69), emb(36)] -> 128 -> 128 -> 128 -> 128 -> [h, emb](164) -> 128 -> 128 -> 3
//         (non_rigid_motion_mlps/mlp_offset.py:7-62, skip at layer index 4)
//   xyz += offset                                                    (network.py:225-232)
//
// Same register-resident scheme as mlp.hip: one wave = 32 samples, layer outputs in the
// 32x32 MFMA C/D layout are the next layer's B operands, weights pre-packed in matching k
// order.  The 69 condition inputs are identical for every sample of a frame, so their
// contribution to layer 0 is folded into its bias once per call (same fma order as the
// dense evaluation: bias first, then k = 0..68), which removes 8 832 of the 100 352 MAC.
//
// MFMA per wave: 80 + 3*256 + 336 + 256 = 1440 (algorithmic 1424).  Bound: fp32 MFMA.
#include "common.h"
#include "split.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNrW = 128, kNrOB = 4;
constexpr int kCond = 69, kEmb = 36, kEmbHalf = 18;
constexpr int kERegs = 20;                       // 18 embedding k-steps + 2 pad
constexpr int kG_E = kERegs / 4;                 // 5
constexpr int kG_H = kNrW / 2 / 4;               // 16

constexpr int64_t nr_wsz(int groups) { return (int64_t)groups * kNrOB * 64 * 4; }
struct NrBlob {
    static constexpr int64_t kL0W = 0;                                  // embedding part only
    static constexpr int64_t kL0B = kL0W + nr_wsz(kG_E);               // folded bias (per call)
    static constexpr int64_t kHW = kL0B + kNrW;                        // layers 1..3
    static constexpr int64_t kHStride = nr_wsz(kG_H) + kNrW;
    static constexpr int64_t kSkipW = kHW + 3 * kHStride;              // layer 4: 16 + 5 groups
    static constexpr int64_t kSkipB = kSkipW + nr_wsz(kG_H + kG_E);
    static constexpr int64_t kL5W = kSkipB + kNrW;
    static constexpr int64_t kL5B = kL5W + nr_wsz(kG_H);
    static constexpr int64_t kOutW = kL5B + kNrW;                      // 3 dot rows
    static constexpr int64_t kOutB = kOutW + 3 * kNrW;
    static constexpr int64_t kTotal = kOutB + 4;
};

enum NrKind { kNrL0 = 0, kNrHidden = 1, kNrSkip = 2 };

__host__ __device__ inline int nr_slot_feature(int kind, int t, int h) {
    const int blk = t >> 4, r = t & 15;
    const int cd = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    switch (kind) {
        case kNrL0: return t < kEmbHalf ? kCond + h * kEmbHalf + t : -1;
        case kNrHidden: return cd;
        case kNrSkip:
            if (t < 64) return cd;
            return (t - 64) < kEmbHalf ? kNrW + h * kEmbHalf + (t - 64) : -1;
    }
    return -1;
}

__global__ void nr_pack_layer_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                     int kind, int in_dim, int groups, float *__restrict__ Wp,
                                     float *__restrict__ Bp) {
    const int total = groups * kNrOB * 64 * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int rr = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
        const int ob = rest % kNrOB, g = rest / kNrOB;
        const int col = nr_slot_feature(kind, g * 4 + rr, lane >> 5);
        Wp[e] = col >= 0 ? W[(size_t)(ob * 32 + (lane & 31)) * in_dim + col] : 0.0f;
    }
    if (Bp) {
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kNrW; e += gridDim.x * blockDim.x) {
            const int rr = e & 3, h = (e >> 2) & 1, q = (e >> 3) & 3, ob = e >> 5;
            Bp[e] = b[ob * 32 + rr + 8 * q + 4 * h];
        }
    }
}

__global__ void nr_pack_rows_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                    float *__restrict__ Wp, float *__restrict__ Bp) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 3 * kNrW; e += gridDim.x * blockDim.x) {
        const int rr = e & 3, h = (e >> 2) & 1, q = (e >> 3) & 3, kb = (e >> 5) & 3, row = e >> 7;
        Wp[e] = W[(size_t)row * kNrW + kb * 32 + rr + 8 * q + 4 * h];
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) Bp[threadIdx.x] = threadIdx.x < 3 ? b[threadIdx.x] : 0.0f;
}

// layer-0 bias with the frame's condition code folded in: b + W[:, :69] cond, fma chain
// in k order starting from the bias (the order a dense k = 0..104 evaluation would use)
__global__ void nr_fold_bias_kernel(const float *__restrict__ W0, const float *__restrict__ b0,
                                    const float *__restrict__ cond, float *__restrict__ Bp) {
    const int e = threadIdx.x;
    if (e >= kNrW) return;
    const int rr = e & 3, h = (e >> 2) & 1, q = (e >> 3) & 3, ob = e >> 5;
    const int row = ob * 32 + rr + 8 * q + 4 * h;
    float acc = b0[row];
    for (int k = 0; k < kCond; k++) acc = __fmaf_rn(W0[(size_t)row * (kCond + kEmb) + k], cond[k], acc);
    Bp[e] = acc;
}

#define OCC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void nr_load_bias(f32x16 (&acc)[kNrOB], const float *__restrict__ Bp, int h) {
    const f32x4 *B4 = reinterpret_cast<const f32x4 *>(Bp);
#pragma unroll
    for (int ob = 0; ob < kNrOB; ob++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v = ld32(B4, (uint32_t)(((ob * 4 + q) * 2 + h) * 16));
#pragma unroll
            for (int rr = 0; rr < 4; rr++) acc[ob][q * 4 + rr] = v[rr];
        }
    }
}

// Weight loads are buffer loads: SGPR descriptor of the packed blob + 32-bit lane offset + scalar layer
// offset.  A global_load whose address is a 64-bit VGPR pair per lane costs the issuing SIMD ~40 cycles of
// matrix-pipe time per instruction on gfx950 (measured on the canonical MLP).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 nr_bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// W_OFF: byte offset of the layer's weights inside the packed blob (wave-uniform)
#define NR_LAYER(GROUPS, W_OFF, ACC, BOP)                                                     \
    {                                                                                         \
        const unsigned w_off_ = (unsigned)(W_OFF);                                            \
        f32x4 wc_[kNrOB], wn_[kNrOB];                                                         \
        _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++)                               \
            wc_[ob_] = nr_bload(rsrc, lane16 + ob_ * 1024, w_off_);                           \
        _Pragma("unroll") for (int g_ = 0; g_ < (GROUPS); g_++) {                             \
            if (g_ + 1 < (GROUPS)) {                                                          \
                _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++)                       \
                    wn_[ob_] = nr_bload(rsrc, lane16 + ((g_ + 1) * kNrOB + ob_) * 1024, w_off_); \
            }                                                                                 \
            _Pragma("unroll") for (int rr_ = 0; rr_ < 4; rr_++) {                             \
                const int t_ = g_ * 4 + rr_;                                                  \
                _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++)                       \
                    ACC[ob_] = OCC_MFMA(wc_[ob_][rr_], BOP(t_), ACC[ob_]);                    \
            }                                                                                 \
            _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++) wc_[ob_] = wn_[ob_];      \
        }                                                                                     \
    }

struct NrParams {
    float hann[6];
};

__global__ __launch_bounds__(256, 2) void nonrigid_kernel(const float *xyz_in /* may alias xyz_out (in-place): no __restrict__ */, int64_t N,
                                                          const float *__restrict__ pk, NrParams prm,
                                                          float *xyz_out) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= N) return;
    const int64_t n = tile * 32 + j;
    const int64_t nsrc = n < N ? n : N - 1;
    const float p[3] = {xyz_in[nsrc * 3], xyz_in[nsrc * 3 + 1], xyz_in[nsrc * 3 + 2]};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pk), 0, (int)(NrBlob::kTotal * sizeof(float)), 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;

    // embedding feature f = octave*6 + {sin: 0..2, cos: 3..5}; this half-wave holds features
    // h*18 .. h*18+17, i.e. octaves 3h .. 3h+2
    float e[kERegs];
#pragma unroll
    for (int o = 0; o < 3; o++) {
        const int oct = 3 * h + o;
        const float freq = (float)(1 << oct);
        const float wgt = oct == 0 ? prm.hann[0] : oct == 1 ? prm.hann[1] : oct == 2 ? prm.hann[2]
                        : oct == 3 ? prm.hann[3] : oct == 4 ? prm.hann[4] : prm.hann[5];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float a = __fmul_rn(p[c], freq);
            e[o * 6 + c] = __fmul_rn(wgt, sinf(a));
            e[o * 6 + 3 + c] = __fmul_rn(wgt, cosf(a));
        }
    }
    e[18] = 0.0f;
    e[19] = 0.0f;

    f32x16 acc[kNrOB], act[kNrOB];
#define BOP_E(t) e[t]
#define BOP_A(t) act[(t) >> 4][(t) & 15]
#define BOP_SKIP(t) ((t) < 64 ? act[((t) >> 4) & 3][(t) & 15] : e[((t) - 64) < 0 ? 0 : ((t) - 64)])
#define NR_RELU()                                                                     \
    _Pragma("unroll") for (int ob = 0; ob < kNrOB; ob++) {                            \
        _Pragma("unroll") for (int r = 0; r < 16; r++) act[ob][r] = fmaxf(acc[ob][r], 0.0f); \
    }
    nr_load_bias(acc, pk + NrBlob::kL0B, h);
    NR_LAYER(kG_E, NrBlob::kL0W * 4, acc, BOP_E)
    NR_RELU()
#pragma unroll 1
    for (int l = 0; l < 3; l++) {
        const float *base = pk + NrBlob::kHW + l * NrBlob::kHStride;
        nr_load_bias(acc, base + nr_wsz(kG_H), h);
        NR_LAYER(kG_H, (NrBlob::kHW + l * NrBlob::kHStride) * 4, acc, BOP_A)
        NR_RELU()
    }
    nr_load_bias(acc, pk + NrBlob::kSkipB, h);
    NR_LAYER(kG_H + kG_E, NrBlob::kSkipW * 4, acc, BOP_SKIP)
    NR_RELU()
    nr_load_bias(acc, pk + NrBlob::kL5B, h);
    NR_LAYER(kG_H, NrBlob::kL5W * 4, acc, BOP_A)
    NR_RELU()

    float off[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const f32x4 *W4 = reinterpret_cast<const f32x4 *>(pk + NrBlob::kOutW + c * kNrW);
        float s = 0.0f;
#pragma unroll
        for (int kb = 0; kb < kNrOB; kb++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 w = W4[(kb * 4 + q) * 2 + h];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) s = __fmaf_rn(w[rr], act[kb][q * 4 + rr], s);
            }
        }
        off[c] = s + __shfl_xor(s, 32) + pk[NrBlob::kOutB + c];
    }
    if (h == 0 && n < N) {
#pragma unroll
        for (int c = 0; c < 3; c++) xyz_out[n * 3 + c] = __fadd_rn(p[c], off[c]);
    }
}


// =======================================================================================
// bf16x3 variant (see mlp.hip): hi/lo bf16 operands, Wh*xh + Wh*xl + Wl*xh, fp32 accumulate,
// weights streamed once per workgroup through an LDS ring by LDS-DMA.  A 16 KiB chunk holds two
// 16-wide k-steps x {hi,lo} x 4 output blocks; 23 chunks per tile.
// =======================================================================================
constexpr int kNrS_E = 3;                       // embedding: 18 slots per half -> 3 k-steps (6 pad)
constexpr int kNrS_H = kNrW / 16;               // 8
constexpr int kNrSteps = kNrS_E + 3 * kNrS_H + (kNrS_H + kNrS_E) + kNrS_H;   // 46
static_assert(kNrSteps % 2 == 0, "two k-steps per chunk");
constexpr int kNrChunks = kNrSteps / 2;
constexpr int kNrChunkUnits = 1024;             // 16-byte units per chunk
constexpr int kNrRing = 4;
constexpr int kNrStepUnits = 2 * kNrOB * 64;    // [hi|lo][ob][lane]

struct NrAux {      // floats in LDS
    static constexpr int kL0B = 0, kHB = 128, kSkipB = 512, kL5B = 640, kOut = 768, kTotal = 1156;
};

template <typename P>
__global__ void nr_pack_split_kernel(const float *__restrict__ W, int kind, int in_dim, int steps,
                                     typename P::E *__restrict__ Wp) {
    const int total = steps * 2 * kNrOB * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int i = e & 7, lane = (e >> 3) & 63;
        int rest = e >> 9;
        const int ob = rest % kNrOB;
        rest /= kNrOB;
        const int which = rest & 1, step = rest >> 1;
        const int col = nr_slot_feature(kind, step * 8 + i, lane >> 5);
        const float w = col >= 0 ? W[(size_t)(ob * 32 + (lane & 31)) * in_dim + col] : 0.0f;
        const typename P::E hi = P::w_hi(w);
        Wp[e] = which == 0 ? hi : P::w_lo(w, hi);
    }
}

template <typename P>
struct NrSplitT {
    typename P::V8 hi, lo;
};

template <typename P>
__device__ __forceinline__ NrSplitT<P> nr_split8(const float (&v)[8]) {
    NrSplitT<P> o;
    P::split8(v, o.hi, o.lo);
    return o;
}

// LDS-DMA of 64 x 16 B: wave-uniform source base (SGPR pair) + 32-bit lane offset ("saddr" form -- a
// 64-bit VGPR address per lane costs the issuing SIMD ~40 cycles of matrix-pipe time per instruction on
// gfx950), wave-uniform LDS destination in M0 (lane i lands at +16 i).
__device__ __forceinline__ void nr_glds16(const void *gbase, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

template <typename P, bool WATCH /* track the ReLU outputs and report leaving the policy's domain (split.h) */>
__global__ __launch_bounds__(256, 2) void nonrigid_split_kernel(const float *xyz_in /* may alias xyz_out (in-place): no __restrict__ */, int64_t N_max,
                                                                 const int32_t *__restrict__ rows /*nullable: sample of entry n*/,
                                                                 const int32_t *__restrict__ n_dev /*nullable: device-side entry count*/,
                                                                 const float *__restrict__ pk,
                                                                 const typename P::V8 *__restrict__ pkh, NrParams prm,
                                                                 float *xyz_out, uint32_t *__restrict__ domain_flag /*nullable*/) {
    typedef typename P::V8 V8;
    typedef NrSplitT<P> NrSplit;
    float amax = 0.0f;      // P::kBounded policies: running packed-half maximum of the ReLU outputs' hi pieces (split.h)
    constexpr float kSx = P::kSx, kInvSx = 1.0f / P::kSx;
    __shared__ __attribute__((aligned(16))) V8 smem[kNrRing * kNrChunkUnits + NrAux::kTotal / 4];
    V8 *ring = smem;
    float *aux = reinterpret_cast<float *>(smem + kNrRing * kNrChunkUnits);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t N = n_dev ? (int64_t)*n_dev : N_max;
    if ((int64_t)blockIdx.x * 128 >= N) return;      // launches are sized for the worst case; uniform per workgroup
    const int64_t n = ((int64_t)blockIdx.x * 4 + wave) * 32 + j;
    const int64_t nsrc0 = n < N ? n : N - 1;
    const int64_t nsrc = rows ? (int64_t)rows[nsrc0] : nsrc0;      // (with a row list the offsets are written back to the listed rows)

    // (the biases of the MFMA layers travel in the activations' scale; the output rows and their bias do not)
    auto copy = [&](int dst, int64_t src, int count, float scale) {
        for (int i = threadIdx.x; i < count; i += 256) aux[dst + i] = pk[src + i] * scale;
    };
    copy(NrAux::kL0B, NrBlob::kL0B, 128, kSx);
    for (int l = 0; l < 3; l++) copy(NrAux::kHB + l * 128, NrBlob::kHW + l * NrBlob::kHStride + nr_wsz(kG_H), 128, kSx);
    copy(NrAux::kSkipB, NrBlob::kSkipB, 128, kSx);
    copy(NrAux::kL5B, NrBlob::kL5B, 128, kSx);
    copy(NrAux::kOut, NrBlob::kOutW, 388, 1.0f);

    const float p[3] = {xyz_in[nsrc * 3], xyz_in[nsrc * 3 + 1], xyz_in[nsrc * 3 + 2]};
    NrSplit be[kNrS_E];      // embedding operands: slots 0..17 of this half, 6 zero pads
    {
        float e[24];
#pragma unroll
        for (int o = 0; o < 3; o++) {
            const int oct = 3 * h + o;
            const float freq = (float)(1 << oct);
            const float wgt = oct == 0 ? prm.hann[0] : oct == 1 ? prm.hann[1] : oct == 2 ? prm.hann[2]
                            : oct == 3 ? prm.hann[3] : oct == 4 ? prm.hann[4] : prm.hann[5];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float a = __fmul_rn(p[c], freq);
                e[o * 6 + c] = __fmul_rn(wgt, sinf(a));
                e[o * 6 + 3 + c] = __fmul_rn(wgt, cosf(a));
            }
        }
#pragma unroll
        for (int t = 18; t < 24; t++) e[t] = 0.0f;
#pragma unroll
        for (int s = 0; s < kNrS_E; s++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = e[s * 8 + i] * kSx;
            be[s] = nr_split8<P>(v);
        }
    }
    __syncthreads();

    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) V8 *)ring;
    auto issue = [&](int g) {
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int frag = wave * 4 + f;
            nr_glds16(pkh + (size_t)g * kNrChunkUnits + frag * 64, lane * 16,
                      ring_lds + (unsigned)(((g & (kNrRing - 1)) * kNrChunkUnits + frag * 64) * 16));
        }
    };
    issue(0);
    issue(1);
    issue(2);

    // The k-steps of the whole network form one stream: step index `st` (compile-time after
    // unrolling) -> chunk st/2, half st%2.  Entering a new chunk = counted wait + barrier + refill.
    const V8 *slot = ring;
#define NR_STEP(ST, ACC, BSPLIT)                                                                   \
    {                                                                                              \
        if (((ST) & 1) == 0) {                                                                     \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                       \
            __builtin_amdgcn_s_barrier();                                                          \
            issue((ST) / 2 + 3);                                                                   \
            slot = ring + (((ST) / 2) & (kNrRing - 1)) * kNrChunkUnits;                            \
        }                                                                                          \
        const V8 *st_ = slot + ((ST) & 1) * kNrStepUnits;                                          \
        V8 ah_[kNrOB], al_[kNrOB];                                                                 \
        _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++) ah_[ob_] = st_[ob_ * 64 + lane];   \
        _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++) al_[ob_] = st_[(kNrOB + ob_) * 64 + lane]; \
        const NrSplit &b_ = (BSPLIT);                                                              \
        const V8 b3_ = P::third(b_.hi);                                                            \
        _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++) ACC[ob_] = P::mfma(ah_[ob_], b_.hi, ACC[ob_]); \
        _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++) ACC[ob_] = P::mfma(ah_[ob_], b_.lo, ACC[ob_]); \
        _Pragma("unroll") for (int ob_ = 0; ob_ < kNrOB; ob_++) ACC[ob_] = P::mfma(al_[ob_], b3_, ACC[ob_]);   \
    }
#define NR_BIAS(OFF)                                                                               \
    {                                                                                              \
        const f32x4 *B4 = reinterpret_cast<const f32x4 *>(aux + (OFF));                            \
        _Pragma("unroll") for (int ob = 0; ob < kNrOB; ob++) {                                     \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                        \
                const f32x4 v = B4[(ob * 4 + q) * 2 + h];                                          \
                _Pragma("unroll") for (int rr = 0; rr < 4; rr++) acc[ob][q * 4 + rr] = v[rr];      \
            }                                                                                      \
        }                                                                                          \
    }
#define NR_RELU_SPLIT()                                                                            \
    _Pragma("unroll") for (int ob = 0; ob < kNrOB; ob++) {                                         \
        _Pragma("unroll") for (int sub = 0; sub < 2; sub++) {                                      \
            float v[8];                                                                            \
            _Pragma("unroll") for (int i = 0; i < 8; i++) v[i] = P::relu(acc[ob][sub * 8 + i]);    \
            bact[ob * 2 + sub] = nr_split8<P>(v);                                                  \
            if constexpr (WATCH) P::watch_hi(amax, bact[ob * 2 + sub].hi);                         \
        }                                                                                          \
    }

    f32x16 acc[kNrOB];
    NrSplit bact[2 * kNrOB];
    // layer 0: embedding only (condition code folded into the bias): steps 0..2
    NR_BIAS(NrAux::kL0B)
#pragma unroll
    for (int s = 0; s < kNrS_E; s++) NR_STEP(s, acc, be[s])
    NR_RELU_SPLIT()
    // layers 1..3: steps 3..26
#pragma unroll
    for (int l = 0; l < 3; l++) {
        NR_BIAS(NrAux::kHB + l * 128)
#pragma unroll
        for (int s = 0; s < kNrS_H; s++) NR_STEP(kNrS_E + l * kNrS_H + s, acc, bact[s])
        NR_RELU_SPLIT()
    }
    // layer 4 (skip): [h(128), emb(36)]: steps 27..37
    NR_BIAS(NrAux::kSkipB)
#pragma unroll
    for (int s = 0; s < kNrS_H; s++) NR_STEP(kNrS_E + 3 * kNrS_H + s, acc, bact[s])
#pragma unroll
    for (int s = 0; s < kNrS_E; s++) NR_STEP(kNrS_E + 4 * kNrS_H + s, acc, be[s])
    NR_RELU_SPLIT()
    // layer 5: steps 38..45
    NR_BIAS(NrAux::kL5B)
#pragma unroll
    for (int s = 0; s < kNrS_H; s++) NR_STEP(2 * kNrS_E + 4 * kNrS_H + s, acc, bact[s])
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float off[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const f32x4 *W4 = reinterpret_cast<const f32x4 *>(aux + NrAux::kOut + c * kNrW);
        float sacc = 0.0f;
#pragma unroll
        for (int kb = 0; kb < kNrOB; kb++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 w = W4[(kb * 4 + q) * 2 + h];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) sacc = __fmaf_rn(w[rr], fmaxf(acc[kb][q * 4 + rr], 0.0f), sacc);
            }
        }
        off[c] = (sacc + __shfl_xor(sacc, 32)) * kInvSx + aux[NrAux::kOut + 3 * kNrW + c];
    }
    if (h == 0 && n < N) {
#pragma unroll
        for (int c = 0; c < 3; c++) xyz_out[nsrc * 3 + c] = __fadd_rn(p[c], off[c]);
    }
    if constexpr (WATCH) split_report<P>(amax, domain_flag);
#undef NR_STEP
#undef NR_BIAS
#undef NR_RELU_SPLIT
}

}  // namespace occ

OCC_API int64_t occnerf_nonrigid_packed_floats(void) { return occ::NrBlob::kTotal + occ::nr_lds_packed_floats(); }

OCC_API int occnerf_nonrigid_pack(const float *const *h_W, const float *const *h_b, float *packed,
                                  void *stream) {
    using namespace occ;
    OCC_REQUIRE(h_W && h_b && packed, "nonrigid_pack: null argument");
    for (int i = 0; i < 7; i++) OCC_REQUIRE(h_W[i] && h_b[i], "nonrigid_pack: layer %d missing", i);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(nr_pack_layer_kernel, dim3(64), dim3(256), 0, st, h_W[0], h_b[0], (int)kNrL0,
                       kCond + kEmb, kG_E, packed + NrBlob::kL0W, (float *)nullptr);
    for (int l = 0; l < 3; l++) {
        const int64_t base = NrBlob::kHW + l * NrBlob::kHStride;
        hipLaunchKernelGGL(nr_pack_layer_kernel, dim3(64), dim3(256), 0, st, h_W[1 + l], h_b[1 + l],
                           (int)kNrHidden, kNrW, kG_H, packed + base, packed + base + nr_wsz(kG_H));
    }
    hipLaunchKernelGGL(nr_pack_layer_kernel, dim3(64), dim3(256), 0, st, h_W[4], h_b[4], (int)kNrSkip,
                       kNrW + kEmb, kG_H + kG_E, packed + NrBlob::kSkipW, packed + NrBlob::kSkipB);
    hipLaunchKernelGGL(nr_pack_layer_kernel, dim3(64), dim3(256), 0, st, h_W[5], h_b[5], (int)kNrHidden,
                       kNrW, kG_H, packed + NrBlob::kL5W, packed + NrBlob::kL5B);
    hipLaunchKernelGGL(nr_pack_rows_kernel, dim3(2), dim3(256), 0, st, h_W[6], h_b[6],
                       packed + NrBlob::kOutW, packed + NrBlob::kOutB);
    if (int rc = check_launch("nonrigid_pack")) return rc;
    return nr_lds_pack(h_W, h_b, packed + NrBlob::kTotal, st);
}

OCC_API int64_t occnerf_nonrigid_packed_bf16_bytes(void) {
    return ((int64_t)occ::kNrChunks + occ::kNrRing - 1) * occ::kNrChunkUnits * 16;     // + read-ahead tail
}

template <typename P>
static int nr_pack_split(const float *const *h_W, void *packed_split, void *stream) {
    using namespace occ;
    OCC_REQUIRE(h_W && packed_split, "nonrigid_pack (split): null argument");
    for (int i = 0; i < 6; i++) OCC_REQUIRE(h_W[i], "nonrigid_pack (split): layer %d missing", i);
    hipStream_t st = as_stream(stream);
    typename P::E *base = reinterpret_cast<typename P::E *>(packed_split);
    int step = 0;
    auto layer = [&](int li, int kind, int in_dim, int steps) {
        hipLaunchKernelGGL(nr_pack_split_kernel<P>, dim3(64), dim3(256), 0, st, h_W[li], kind, in_dim, steps,
                           base + (size_t)step * kNrStepUnits * 8);
        step += steps;
    };
    layer(0, kNrL0, kCond + kEmb, kNrS_E);
    for (int l = 0; l < 3; l++) layer(1 + l, kNrHidden, kNrW, kNrS_H);
    layer(4, kNrSkip, kNrW + kEmb, kNrS_H + kNrS_E);
    layer(5, kNrHidden, kNrW, kNrS_H);
    return check_launch("nonrigid_pack (split)");
}

// xyz_in may alias xyz_out; rows / n_dev nullable (all N_max samples)
template <typename P>
static int nr_split_launch(const float *xyz_in, int64_t N_max, const int32_t *rows, const int32_t *n_dev, const float *cond,
                           const float *h_hann, const float *W0, const float *b0, float *packed, const void *packed_split,
                           float *xyz_out, uint32_t *domain_flag, void *stream, const char *what) {
    using namespace occ;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(nr_fold_bias_kernel, dim3(1), dim3(128), 0, st, W0, b0, cond, packed + NrBlob::kL0B);
    NrParams prm;
    for (int i = 0; i < 6; i++) prm.hann[i] = h_hann[i];
    const int64_t blocks = (N_max + 127) / 128;
    OCC_REQUIRE(blocks < (1ll << 31), "%s: N too large", what);
    if (P::kBounded && domain_flag)
        hipLaunchKernelGGL((nonrigid_split_kernel<P, true>), dim3((unsigned)blocks), dim3(256), 0, st, xyz_in, N_max, rows, n_dev, packed,
                           reinterpret_cast<const typename P::V8 *>(packed_split), prm, xyz_out, domain_flag);
    else
        hipLaunchKernelGGL((nonrigid_split_kernel<P, false>), dim3((unsigned)blocks), dim3(256), 0, st, xyz_in, N_max, rows, n_dev, packed,
                           reinterpret_cast<const typename P::V8 *>(packed_split), prm, xyz_out, domain_flag);
    return check_launch(what);
}

OCC_API int occnerf_nonrigid_pack_bf16(const float *const *h_W, void *packed_bf16, void *stream) {
    return nr_pack_split<occ::Bf16x3>(h_W, packed_bf16, stream);
}

OCC_API int occnerf_nonrigid_pack_f16(const float *const *h_W, void *packed_f16, void *stream) {
    return nr_pack_split<occ::F16x3>(h_W, packed_f16, stream);
}

OCC_API int occnerf_nonrigid_bf16x3(const float *xyz_in, int64_t N, const float *cond, const float *h_hann,
                                    const float *W0, const float *b0, float *packed, const void *packed_bf16,
                                    float *xyz_out, void *stream) {
    if (N <= 0) return 0;
    OCC_REQUIRE(xyz_in && cond && h_hann && W0 && b0 && packed && packed_bf16 && xyz_out,
                "nonrigid_bf16x3: null argument");
    return nr_split_launch<occ::Bf16x3>(xyz_in, N, nullptr, nullptr, cond, h_hann, W0, b0, packed, packed_bf16, xyz_out, nullptr, stream,
                                        "nonrigid_bf16x3");
}

OCC_API int occnerf_nonrigid_bf16x3_rows(float *xyz, int64_t N_max, const int32_t *rows, const int32_t *n_dev,
                                         const float *cond, const float *h_hann, const float *W0, const float *b0,
                                         float *packed, const void *packed_bf16, void *stream) {
    if (N_max <= 0) return 0;
    OCC_REQUIRE(xyz && rows && n_dev && cond && h_hann && W0 && b0 && packed && packed_bf16,
                "nonrigid_bf16x3_rows: null argument");
    return nr_split_launch<occ::Bf16x3>(xyz, N_max, rows, n_dev, cond, h_hann, W0, b0, packed, packed_bf16, xyz, nullptr, stream,
                                        "nonrigid_bf16x3_rows");
}

/* The fp32-grade split (split.h F16x3).  rows / n_dev nullable: all N_max samples; xyz_in may alias xyz_out. */
OCC_API int occnerf_nonrigid_f16x3(const float *xyz_in, int64_t N_max, const int32_t *rows, const int32_t *n_dev,
                                   const float *cond, const float *h_hann, const float *W0, const float *b0, float *packed,
                                   const void *packed_f16, float *xyz_out, uint32_t *domain_flag, void *stream) {
    if (N_max <= 0) return 0;
    OCC_REQUIRE(xyz_in && cond && h_hann && W0 && b0 && packed && packed_f16 && xyz_out, "nonrigid_f16x3: null argument");
    OCC_REQUIRE(!rows || n_dev, "nonrigid_f16x3: a row list needs its device-side count");
    return nr_split_launch<occ::F16x3>(xyz_in, N_max, rows, n_dev, cond, h_hann, W0, b0, packed, packed_f16, xyz_out, domain_flag, stream,
                                       "nonrigid_f16x3");
}

OCC_API int occnerf_nonrigid(const float *xyz_in, int64_t N, const float *cond, const float *h_hann,
                             const float *W0, const float *b0, float *packed, float *xyz_out,
                             void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(xyz_in && cond && h_hann && W0 && b0 && packed && xyz_out, "nonrigid: null argument");
    return nr_lds_launch(xyz_in, N, nullptr, nullptr, cond, h_hann, W0, b0, packed + NrBlob::kTotal, xyz_out,
                         as_stream(stream));
}

OCC_API int occnerf_nonrigid_rows(float *xyz, int64_t N_max, const int32_t *rows, const int32_t *n_dev,
                                  const float *cond, const float *h_hann, const float *W0, const float *b0,
                                  float *packed, void *stream) {
    using namespace occ;
    if (N_max <= 0) return 0;
    OCC_REQUIRE(xyz && rows && n_dev && cond && h_hann && W0 && b0 && packed, "nonrigid_rows: null argument");
    return nr_lds_launch(xyz, N_max, rows, n_dev, cond, h_hann, W0, b0, packed + NrBlob::kTotal, xyz,
                         as_stream(stream));
}

OCC_API int occnerf_nonrigid_direct(const float *xyz_in, int64_t N, const float *cond, const float *h_hann,
                                    const float *W0, const float *b0, float *packed, float *xyz_out,
                                    void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(xyz_in && cond && h_hann && W0 && b0 && packed && xyz_out, "nonrigid_direct: null argument");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(nr_fold_bias_kernel, dim3(1), dim3(128), 0, st, W0, b0, cond, packed + NrBlob::kL0B);
    NrParams prm;
    for (int i = 0; i < 6; i++) prm.hann[i] = h_hann[i];
    const int64_t blocks = (N + 127) / 128;
    OCC_REQUIRE(blocks < (1ll << 31), "nonrigid_direct: N too large");
    hipLaunchKernelGGL(nonrigid_kernel, dim3((unsigned)blocks), dim3(256), 0, st, xyz_in, N, packed, prm,
                       xyz_out);
    return check_launch("nonrigid_direct");
}
