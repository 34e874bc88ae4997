// Pose-conditioned non-rigid offset MLP -- the default fp32 kernel (row a9; hannw_fourier.py:9-63,
// mlp_offset.py:7-62, network.py:225-232): the LDS-staged 16x16x4 scheme of mlp16.hip at width 128.
//
//   emb = Hann-windowed Fourier embedding of xyz, 6 octaves x (sin, cos) x 3 = 36
//   h   = [cond(69), emb(36)] -> 128 -> 128 -> 128 -> 128 -> [h, emb](164) -> 128 -> 128 -> 3;  xyz += offset
//
// A wave carries TWO tiles of 16 samples (2 x (32 + 32) registers of activations + accumulators) and every
// weight register feeds one MFMA of each tile, so LDS reads and DMA per sample are half of what one tile
// per wave would need; 2 waves per SIMD.  The weight stream (46 chunks of 8 KiB = one group of 4 k-steps x
// 8 output blocks) goes through a 4-slot LDS ring exactly as in mlp16.hip (LDS-DMA with SGPR base + 32-bit
// lane offset, one raw barrier + counted vmcnt per chunk, weight registers pipelined across chunks and
// layers).  The 69 condition inputs are the same for every sample of a frame: their share of layer 0 is
// folded into its bias once per call (bias first, then k = 0..68 -- the order a dense evaluation uses).
//
// Embedding k order (free to choose, the weights are packed to match): k-steps 2m, 2m+1 (m < 4) carry
// sin and cos of angle A = 4m + g in lane group g, k-step 8 carries angles 16, 17 (g>>1) as sin/cos
// (g&1); angle A = (octave A/3, axis A%3), torch feature index octave*6 + cos*3 + axis.  So a lane
// evaluates 5 sin + 5 cos per sample.
#include "common.h"

namespace occ {
namespace nr16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kW = 128, kOB = kW / 16;            // 8 output blocks of 16 features
constexpr int kCond = 69, kEmb = 36;
constexpr int kKS_E = 9, kKS_H = kW / 4, kKS_Skip = kKS_H + kKS_E;
constexpr int groups_of(int ks) { return (ks + 3) / 4; }
constexpr int kC_E = groups_of(kKS_E);            // 3
constexpr int kC_H = groups_of(kKS_H);            // 8
constexpr int kC_Skip = groups_of(kKS_Skip);      // 11 (8 activation groups + 3 embedding groups)
constexpr int kChunks = kC_E + 3 * kC_H + kC_Skip + kC_H;      // 46
constexpr int kWaves = 4, kFrags = 2;             // 8 KiB chunk = 8 fragments of 1 KiB, 2 per wave
constexpr int kRingSlots = 4, kTailChunks = kRingSlots;
constexpr int kChunkF4 = 512;

enum Kind { kL0 = 0, kHidden = 1, kSkip = 2 };

// torch embedding feature carried by embedding k-step t in lane group g
__host__ __device__ inline int e_feature(int t, int g) {
    const int A = t < 8 ? (t >> 1) * 4 + g : 16 + (g >> 1);
    const int is_cos = t < 8 ? (t & 1) : (g & 1);
    return (A / 3) * 6 + is_cos * 3 + A % 3;
}

__host__ __device__ inline int slot_feature(int kind, int t, int g) {
    const int cd = 16 * (t >> 2) + 4 * g + (t & 3);
    switch (kind) {
        case kL0: return t < kKS_E ? kCond + e_feature(t, g) : -1;
        case kHidden: return t < kKS_H ? cd : -1;
        case kSkip:
            if (t < kKS_H) return cd;
            return t < kKS_Skip ? kW + e_feature(t - kKS_H, g) : -1;
    }
    return -1;
}

struct Aux {       // fp32 side data (floats), torch order
    static constexpr int kL0B = 0;              // folded per call
    static constexpr int kHB = 128;             // 3 x 128
    static constexpr int kSkipB = 512;
    static constexpr int kL5B = 640;
    static constexpr int kOutW = 768;           // 3 x 128
    static constexpr int kOutB = 1152;          // 3 (+1 pad)
    static constexpr int kTotal = 1156;
};

struct Stream {    // [chunk stream][tail zeros][aux], offsets in floats
    static constexpr int64_t kChunkFloats = kChunkF4 * 4;
    static constexpr int64_t kL0 = 0;
    static constexpr int64_t kH = kL0 + kC_E * kChunkFloats;
    static constexpr int64_t kSkipW = kH + 3 * kC_H * kChunkFloats;
    static constexpr int64_t kL5 = kSkipW + kC_Skip * kChunkFloats;
    static constexpr int64_t kTail = kL5 + kC_H * kChunkFloats;
    static constexpr int64_t kAux = kTail + kTailChunks * kChunkFloats;
    static constexpr int64_t kTotal = kAux + Aux::kTotal;
};
static_assert(Stream::kTail == (int64_t)kChunks * Stream::kChunkFloats, "chunk stream is contiguous");

__global__ void pack_layer_kernel(const float *__restrict__ W, const float *__restrict__ b, int kind, int in_dim,
                                  int ks, float *__restrict__ Wp, float *__restrict__ Bp) {
    const int total = groups_of(ks) * kOB * 64 * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int rr = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
        const int ob = rest % kOB, G = rest / kOB;
        const int t = 4 * G + rr;
        const int col = t < ks ? slot_feature(kind, t, lane >> 4) : -1;
        Wp[e] = col >= 0 ? W[(size_t)(ob * 16 + (lane & 15)) * in_dim + col] : 0.0f;
    }
    if (Bp)
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kW; e += gridDim.x * blockDim.x) Bp[e] = b[e];
}

__global__ void pack_rows_kernel(const float *__restrict__ W, const float *__restrict__ b, float *__restrict__ Wp,
                                 float *__restrict__ Bp) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 3 * kW; e += gridDim.x * blockDim.x) Wp[e] = W[e];
    if (blockIdx.x == 0 && threadIdx.x < 4) Bp[threadIdx.x] = threadIdx.x < 3 ? b[threadIdx.x] : 0.0f;
}

// layer-0 bias with the frame's condition code folded in (fma chain in k order starting from the bias)
__global__ void fold_bias_kernel(const float *__restrict__ W0, const float *__restrict__ b0,
                                 const float *__restrict__ cond, float *__restrict__ Bp) {
    const int row = threadIdx.x;
    if (row >= kW) return;
    float acc = b0[row];
    for (int k = 0; k < kCond; k++) acc = __fmaf_rn(W0[(size_t)row * (kCond + kEmb) + k], cond[k], acc);
    Bp[row] = acc;
}

struct Params {
    float hann[6];
};

#define NR16_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// rows / n_dev (both nullable): evaluate only the samples rows[0 .. *n_dev) of xyz (the frame's live samples,
// occnerf_live_rows); the launch is sized for N_max and workgroups beyond the device-side count leave at once.
__global__ __launch_bounds__(kWaves * 64, 2) void nonrigid_lds_kernel(const float *xyz_in /* may alias xyz_out (in-place): no __restrict__ */, int64_t N_max,
                                                                      const int32_t *__restrict__ rows,
                                                                      const int32_t *__restrict__ n_dev,
                                                                      const float *__restrict__ pk, Params prm,
                                                                      float *xyz_out) {
    const int64_t N = n_dev ? (int64_t)*n_dev : N_max;
    const int64_t ntiles = (N + 32 * kWaves - 1) / (32 * kWaves);   // a workgroup's tile: 4 waves x 32 samples
    if ((int64_t)blockIdx.x >= ntiles) return;                      // uniform for the workgroup
    // ONE __shared__ object (a second one makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) f32x4 smem[kRingSlots * kChunkF4 + Aux::kTotal / 4];
    f32x4 *ring = smem;
    float *aux = reinterpret_cast<float *>(smem + kRingSlots * kChunkF4);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane & 15, g = lane >> 4;

    for (int i = threadIdx.x; i < Aux::kTotal; i += kWaves * 64) aux[i] = pk[Stream::kAux + i];

    // ---- weight stream: the c-th chunk to enter lives in ring slot c & 3; the stream wraps from chunk 45 to chunk 0 ----
    // The workgroup is PERSISTENT: it walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and the ring keeps running
    // across them (the chunks of the next tile's first layer are requested during this tile's last), so a tile pays neither
    // a workgroup launch nor the first chunk's DMA latency -- 8-10 % of a 128-sample workgroup's 38 us before.
    const f32x4 *stream = reinterpret_cast<const f32x4 *>(pk);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) f32x4 *)ring;
    auto issue1 = [&](int c, int cs, int f) {      // 1 KiB fragment f of this wave's share of stream chunk cs -> slot of c
        const int frag = wave * kFrags + f;
        unsigned keep;      // M0 carries the wave-uniform LDS destination; lane i lands at +16 i
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane * 16), "s"(stream + (size_t)cs * kChunkF4 + frag * 64),
                       "s"(ring_lds + (unsigned)(((c & (kRingSlots - 1)) * kChunkF4 + frag * 64) * 16))
                     : "memory");
    };
    auto issue = [&](int c, int cs) {
#pragma unroll
        for (int f = 0; f < kFrags; f++) issue1(c, cs, f);
    };
    int c = 0;                     // chunks entered so far (ring position)
    int ci = 4;                    // stream chunk the next refill brings (always the (c + 2)-th to enter)
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);

    // Enter the next chunk (3 chunks x kFrags DMAs outstanding; vmcnt(2*kFrags) retires the oldest), rendezvous; after
    // the barrier every wave has finished reading the previous chunk, whose slot the following step refills.
    const f32x4 *slot_;
#define NR16_ENTER()                                                   \
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                                      \
    slot_ = ring + (c & (kRingSlots - 1)) * kChunkF4;                  \
    c++;
#define NR16_READ_HALF(W, HALF)                                        \
    _Pragma("unroll") for (int ob_ = 0; ob_ < 4; ob_++) W[ob_] = slot_[((HALF) * 4 + ob_) * 64 + lane];

    f32x4 wA[4];
    NR16_ENTER()                   // (its lgkmcnt(0) + barrier also publish aux)
    issue(3, 3);
    NR16_READ_HALF(wA, 0)

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = (tile * kWaves + wave) * 32;

    // ---- inputs and embedding: tile T holds sample base + 16 T + s ----
    // (requesting them two tiles ahead was measured neutral: 23.5 ms either way)
    float p[2][3], e[2][12];
    int32_t r_cur[2];
#pragma unroll
    for (int T = 0; T < 2; T++) {
        const int64_t n = base + T * 16 + s;
        const int64_t nm = n < N ? n : N - 1;        // the whole workgroup stays alive for the barriers
        r_cur[T] = rows ? rows[nm] : (int32_t)nm;
#pragma unroll
        for (int c = 0; c < 3; c++) p[T][c] = xyz_in[(int64_t)r_cur[T] * 3 + c];
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const int A = m < 4 ? m * 4 + g : 16 + (g >> 1);
            const int oct = A / 3, c = A - 3 * oct;
            const float pc = c == 0 ? p[T][0] : (c == 1 ? p[T][1] : p[T][2]);
            const float wgt = oct == 0 ? prm.hann[0] : oct == 1 ? prm.hann[1] : oct == 2 ? prm.hann[2]
                            : oct == 3 ? prm.hann[3] : oct == 4 ? prm.hann[4] : prm.hann[5];
            const float a = __fmul_rn(pc, (float)(1 << oct));
            float sa, ca;
#ifdef OCC_NR16_EXP_NO_SINCOS      // (tools/nr16_phases.py: what the ten sincosf per wave pass cost -- wrong offsets, a timing build only)
            sa = a, ca = 1.0f - a;
#else
            sincosf(a, &sa, &ca);                 // one argument reduction for both
#endif
            const float sv = __fmul_rn(wgt, sa), cv = __fmul_rn(wgt, ca);
            if (m < 4) {
                e[T][2 * m] = sv;
                e[T][2 * m + 1] = cv;
            } else {
                e[T][8] = (g & 1) ? cv : sv;
            }
        }
        e[T][9] = e[T][10] = e[T][11] = 0.0f;
    }
#define NR16_HALF(W, HH, CL, KS, BOP)                                                            \
    _Pragma("unroll") for (int rr_ = 0; rr_ < 4; rr_++) {                                        \
        const int t_ = (CL) * 4 + rr_;                                                           \
        if ((HH) == 1 && rr_ < kFrags) {                                                         \
            issue1(c + 2, ci, rr_);                                                              \
            if (rr_ == kFrags - 1) ci = ci + 1 == kChunks ? 0 : ci + 1;                          \
        }                                                                                        \
        if (t_ < (KS)) {                                                                         \
            _Pragma("unroll") for (int ob_ = 0; ob_ < 4; ob_++) {                                \
                _Pragma("unroll") for (int T_ = 0; T_ < 2; T_++)                                 \
                    acc[T_][(HH) * 4 + ob_] = NR16_MFMA(W[ob_][rr_], BOP(T_, t_), acc[T_][(HH) * 4 + ob_]); \
            }                                                                                    \
        }                                                                                        \
    }
#define NR16_LAYER(CHUNKS, KS, BOP)                                                              \
    _Pragma("unroll") for (int c_ = 0; c_ < (CHUNKS); c_++) {                                    \
        f32x4 wB_[4];                                                                            \
        NR16_READ_HALF(wB_, 1)                                                                   \
        NR16_HALF(wA, 0, c_, KS, BOP)                                                            \
        NR16_ENTER()                                                                             \
        NR16_READ_HALF(wA, 0)                                                                    \
        NR16_HALF(wB_, 1, c_, KS, BOP)                                                           \
    }
#define NR16_BIAS(OFF)                                                                           \
    _Pragma("unroll") for (int ob_ = 0; ob_ < kOB; ob_++) {                                      \
        const f32x4 b_ = reinterpret_cast<const f32x4 *>(aux + (OFF))[ob_ * 4 + g];              \
        acc[0][ob_] = b_;                                                                        \
        acc[1][ob_] = b_;                                                                        \
    }
#define NR16_RELU()                                                                              \
    _Pragma("unroll") for (int T_ = 0; T_ < 2; T_++) {                                           \
        _Pragma("unroll") for (int ob_ = 0; ob_ < kOB; ob_++) {                                  \
            _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) act[T_][ob_][r_] = fmaxf(acc[T_][ob_][r_], 0.0f); \
        }                                                                                        \
    }

    f32x4 acc[2][kOB], act[2][kOB];
#define BOP_E(T, t) e[T][t]
#define BOP_A(T, t) act[T][(t) >> 2][(t) & 3]
#define BOP_SKIP(T, t) ((t) < kKS_H ? act[T][((t) >> 2) & 7][(t) & 3] : e[T][((t) - kKS_H) < 0 ? 0 : ((t) - kKS_H)])

    NR16_BIAS(Aux::kL0B)
    NR16_LAYER(kC_E, kKS_E, BOP_E)
    NR16_RELU()
    // the four plain 128 -> 128 layers (1, 2, 3 and 5) share one unrolled body; the skip layer (4) runs
    // before the last of them
#pragma unroll 1
    for (int l = 0; l < 4; l++) {
        if (l == 3) {
            NR16_BIAS(Aux::kSkipB)
            NR16_LAYER(kC_Skip, kKS_Skip, BOP_SKIP)
            NR16_RELU()
        }
        NR16_BIAS(l < 3 ? Aux::kHB + l * 128 : Aux::kL5B)
        NR16_LAYER(kC_H, kKS_H, BOP_A)
        NR16_RELU()
    }
#pragma unroll
    for (int T = 0; T < 2; T++) {
        float off[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const f32x4 *W4 = reinterpret_cast<const f32x4 *>(aux + Aux::kOutW + ch * kW);
            float sum = 0.0f;
#pragma unroll
            for (int ob = 0; ob < kOB; ob++) {
                const f32x4 w = W4[ob * 4 + g];
#pragma unroll
                for (int r = 0; r < 4; r++) sum = __fmaf_rn(w[r], act[T][ob][r], sum);
            }
            sum += __shfl_xor(sum, 16);
            off[ch] = sum + __shfl_xor(sum, 32) + aux[Aux::kOutB + ch];
        }
        const int64_t n = base + T * 16 + s;
        if (g == 0 && n < N) {
            const int64_t nd = r_cur[T];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) xyz_out[nd * 3 + ch] = __fadd_rn(p[T][ch], off[ch]);
        }
    }
    }       // tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunks still in flight: nobody computes with them
#undef BOP_E
#undef BOP_A
#undef BOP_SKIP
#undef NR16_ENTER
#undef NR16_READ_HALF
#undef NR16_HALF
#undef NR16_LAYER
#undef NR16_BIAS
#undef NR16_RELU
}

}  // namespace nr16

int64_t nr_lds_packed_floats() { return nr16::Stream::kTotal; }

int nr_lds_pack(const float *const *h_W, const float *const *h_b, float *packed, hipStream_t st) {
    using namespace nr16;
    OCC_REQUIRE(hipMemsetAsync(packed + Stream::kTail, 0, sizeof(float) * (Stream::kTotal - Stream::kTail), st) ==
                    hipSuccess,
                "nonrigid_pack: memset failed");
    float *aux = packed + Stream::kAux;
    auto layer = [&](int li, int kind, int in_dim, int ks, int64_t woff, int boff) {
        hipLaunchKernelGGL(nr16::pack_layer_kernel, dim3(64), dim3(256), 0, st, h_W[li], h_b[li], kind, in_dim, ks,
                           packed + woff, boff >= 0 ? aux + boff : (float *)nullptr);
    };
    layer(0, kL0, kCond + kEmb, kKS_E, Stream::kL0, -1);                  // bias: folded per call
    for (int l = 0; l < 3; l++)
        layer(1 + l, kHidden, kW, kKS_H, Stream::kH + l * kC_H * Stream::kChunkFloats, Aux::kHB + l * 128);
    layer(4, kSkip, kW + kEmb, kKS_Skip, Stream::kSkipW, Aux::kSkipB);
    layer(5, kHidden, kW, kKS_H, Stream::kL5, Aux::kL5B);
    hipLaunchKernelGGL(nr16::pack_rows_kernel, dim3(2), dim3(256), 0, st, h_W[6], h_b[6], aux + Aux::kOutW,
                       aux + Aux::kOutB);
    return check_launch("nonrigid_pack");
}

int nr_lds_launch(const float *xyz_in, int64_t N, const int32_t *rows, const int32_t *n_dev, const float *cond,
                  const float *h_hann, const float *W0, const float *b0, float *packed, float *xyz_out,
                  hipStream_t st) {
    using namespace nr16;
    hipLaunchKernelGGL(nr16::fold_bias_kernel, dim3(1), dim3(128), 0, st, W0, b0, cond,
                       packed + Stream::kAux + Aux::kL0B);
    Params prm;
    for (int i = 0; i < 6; i++) prm.hann[i] = h_hann[i];
    const int64_t per_block = 32 * kWaves;
    int64_t blocks = (N + per_block - 1) / per_block;
    if (blocks > 2 * (int64_t)kNumCU) blocks = 2 * (int64_t)kNumCU;         // persistent: 2 workgroups per CU
    hipLaunchKernelGGL(nr16::nonrigid_lds_kernel, dim3((unsigned)blocks), dim3(64 * kWaves), 0, st, xyz_in, N, rows,
                       n_dev, packed, prm, xyz_out);
    return check_launch("nonrigid");
}

}  // namespace occ
