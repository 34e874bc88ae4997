// Canonical density/colour MLP -- the default fp32 kernel (row a16, occnerf_mlp.py:183-199):
// 16-sample waves on v_mfma_f32_16x16x4_f32 (exact fp32 = an fmaf chain), weights streamed through LDS.
//
// Why this shape.  A wave carries 16 samples, so its activations + accumulators are 64 + 64 registers
// and TWO waves fit per SIMD (the 32-sample kernel in mlp.hip needs 128 + 128 and runs one): the second
// wave's MFMAs cover the first one's bias/ReLU epilogue, barriers and LDS latency.  Halving the samples
// per wave doubles the weight bytes per sample, so the weights cannot come through L1 any more (measured:
// 232 ms); the workgroup fetches each 16 KiB chunk of the stream ONCE by LDS-DMA into a 4-slot ring and
// its four waves read it with ds_read_b128.  Two things that each cost more than they look:
//   * a vector-memory instruction whose address is a 64-bit VGPR pair per lane holds up the issuing
//     SIMD's matrix pipe for ~40 cycles on gfx950, whatever it moves (4 active lanes cost the same as
//     64): the DMA uses the SGPR-base + 32-bit-lane-offset form (158 -> 150 ms);
//   * the six hidden layers share one unrolled body so that the hot code of the 4 workgroups that share
//     an instruction cache stays small.
// Measured on MI355X, 23.5 M samples: 150 ms = 144.8 TFLOP/s = 92 % of the 157.3 TFLOP/s fp32-MFMA peak
// (the 32-sample direct-load kernel: 165 ms = 84 %).
//
// Layout (16x16x4: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r = row 4*(l>>4)+r, col l&15).
// The layer is computed transposed, D[feature][sample]; lane (g=l>>4, s=l&15) holds in register r
// of output block ob the feature 16*ob + 4*g + r of sample s.  The next layer's k-step
// t = 4*ob + r takes exactly that register as its B operand, i.e. k-step t contracts over the
// features {16*(t>>2) + 4*g + (t&3)}; the A operand of lane (g,i) for the 4 k-steps of one source
// block is W[out = 16*ob' + i][16*G + 4*g .. +3] -- a contiguous float4 of the torch weight row.
// Packed as [G][ob'][lane] float4: one group G of a 16-block layer is exactly one 16 KiB chunk.
#include "common.h"

namespace occ {
namespace m16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWidth = 256;
constexpr int kOB = kWidth / 16;          // 16 output blocks of 16 features
constexpr int kInGeo = 68, kInRgb = 131;
constexpr int kKS_X = 17;                 // k-steps of the 68-wide sample row: 4 groups of 16 features + tail of 4
constexpr int kKS_Hidden = kWidth / 4;    // 64
constexpr int kKS_L0Rgb = 16 + kKS_X;     // 64 geometry features + the sample row

constexpr int groups_of(int ks) { return (ks + 3) / 4; }
enum LayerKind { kL0Geo = 0, kHidden = 1, kGeoHead = 2, kL0Rgb = 3 };

// position in the 68-wide sample row [agg35, var, enc32] carried by x k-step t in lane group g
__host__ __device__ inline int x_slot(int t, int g) { return t < 16 ? 16 * (t >> 2) + 4 * g + (t & 3) : 64 + g; }

// column of the layer's torch-layout input contracted by k-step t in lane group g (-1: zero)
__host__ __device__ inline int slot_feature(int kind, int t, int g) {
    const int cd = 16 * (t >> 2) + 4 * g + (t & 3);
    switch (kind) {
        case kL0Geo: return t < kKS_X ? x_slot(t, g) : -1;
        case kHidden:
        case kGeoHead: return t < kKS_Hidden ? cd : -1;
        case kL0Rgb: {
            if (t < 16) return cd;                       // geometry features h[1:65] -> inputs 0..63
            if (t >= kKS_L0Rgb) return -1;
            const int m = x_slot(t - 16, g);
            if (m < 35) return 64 + m;                   // aggregated point features
            if (m == 35) return -1;                      // var is not an input of the colour trunk
            return 64 + 35 + (m - 36);                   // hash encoding
        }
    }
    return -1;
}

__host__ __device__ inline int out_row(int kind, int row, int out_dim) {
    if (kind == kGeoHead) return row < 64 ? row + 1 : -1;     // row 0 (sigma) is a dot row
    return row < out_dim ? row : -1;
}

__global__ void pack_layer_kernel(const float *__restrict__ W, const float *__restrict__ b, int kind,
                                  int in_dim, int out_dim, int ks, int ob_count, float *__restrict__ Wp,
                                  float *__restrict__ Bp) {
    const int total = groups_of(ks) * ob_count * 64 * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int rr = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
        const int ob = rest % ob_count, G = rest / ob_count;
        const int t = 4 * G + rr;
        const int col = t < ks ? slot_feature(kind, t, lane >> 4) : -1;
        const int row = out_row(kind, ob * 16 + (lane & 15), out_dim);
        Wp[e] = (col >= 0 && row >= 0) ? W[(size_t)row * in_dim + col] : 0.0f;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ob_count * 16; e += gridDim.x * blockDim.x) {
        const int row = out_row(kind, e, out_dim);
        Bp[e] = row >= 0 ? b[row] : 0.0f;
    }
}

__global__ void pack_rows_kernel(const float *__restrict__ W, const float *__restrict__ b, int nrows,
                                 float *__restrict__ Wp, float *__restrict__ Bp) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nrows * kWidth; e += gridDim.x * blockDim.x) Wp[e] = W[e];
    if (blockIdx.x == 0 && threadIdx.x < 4) Bp[threadIdx.x] = (int)threadIdx.x < nrows ? b[threadIdx.x] : 0.0f;
}

#define OCC16_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------
// Weight stream: 16 KiB chunks (one group of 4 k-steps x 16 output blocks x 64 lanes x float4; the
// 4-block geometry head packs 4 groups per chunk), fetched ONCE per workgroup by LDS-DMA into a
// 4-slot ring and read by all four waves with ds_read_b128.  Two workgroups (2 x 4 waves) are
// resident per CU; they drift apart, so one's barriers and epilogues sit under the other's MFMAs.
// ---------------------------------------------------------------------------------------
constexpr int kWaves = 4;                            // waves per workgroup, 16 samples each
constexpr int kFrags = 16 / kWaves;                  // 1 KiB fragments of a chunk issued by each wave
constexpr int kRingSlots = 4;
constexpr int kChunkF4 = 1024;                       // float4 units per chunk
constexpr int kC_L0Geo = groups_of(kKS_X);           // 5
constexpr int kC_Hidden = groups_of(kKS_Hidden);     // 16
constexpr int kC_GeoHead = kC_Hidden / 4;            // 4
constexpr int kC_L0Rgb = groups_of(kKS_L0Rgb);       // 9
constexpr int kChunksTotal = kC_L0Geo + 3 * kC_Hidden + kC_GeoHead + kC_L0Rgb + 3 * kC_Hidden;
constexpr int kTailChunks = kRingSlots;              // zero chunks the prefetch runs into

struct Aux {       // fp32 side data (floats): biases and dot rows in torch order
    static constexpr int kGeoL0B = 0;
    static constexpr int kGeoHB = 256;          // 3 x 256
    static constexpr int kGeoHeadB = 1024;      // 64
    static constexpr int kSigma = 1088;         // 256 weights + bias (+3 pad)
    static constexpr int kRgbL0B = 1348;
    static constexpr int kRgbHB = 1604;         // 3 x 256
    static constexpr int kOut = 2372;           // 3 x 256 weights + 3 biases (+1 pad)
    static constexpr int kTotal = 3144;
};

struct Stream {    // packed blob: [chunk stream][tail zeros][aux], offsets in floats
    static constexpr int64_t kChunkFloats = kChunkF4 * 4;
    static constexpr int64_t kGeoL0 = 0;
    static constexpr int64_t kGeoH = kGeoL0 + kC_L0Geo * kChunkFloats;
    static constexpr int64_t kGeoHead = kGeoH + 3 * kC_Hidden * kChunkFloats;
    static constexpr int64_t kRgbL0 = kGeoHead + kC_GeoHead * kChunkFloats;
    static constexpr int64_t kRgbH = kRgbL0 + kC_L0Rgb * kChunkFloats;
    static constexpr int64_t kTail = kRgbH + 3 * kC_Hidden * kChunkFloats;
    static constexpr int64_t kAux = kTail + kTailChunks * kChunkFloats;
    static constexpr int64_t kTotal = kAux + Aux::kTotal;
};
static_assert(Stream::kTail == (int64_t)kChunksTotal * Stream::kChunkFloats, "chunk stream is contiguous");

template <int OB>
__device__ __forceinline__ void lds_bias(f32x4 (&acc)[OB], const float *aux, int g) {
    const f32x4 *B4 = reinterpret_cast<const f32x4 *>(aux);
#pragma unroll
    for (int ob = 0; ob < OB; ob++) acc[ob] = B4[ob * 4 + g];
}

__device__ __forceinline__ void relu_into(f32x4 (&act)[kOB], const f32x4 (&acc)[kOB]) {
#pragma unroll
    for (int ob = 0; ob < kOB; ob++) {
#pragma unroll
        for (int r = 0; r < 4; r++) act[ob][r] = fmaxf(acc[ob][r], 0.0f);
    }
}

// dot product of a torch-layout weight row (in LDS) with the lane's 64 activations (features
// 16*ob+4*g+r); the other three quarters live in the lanes s+16, s+32, s+48
__device__ __forceinline__ float dot_row(const f32x4 (&act)[kOB], const float *Wrow, int g) {
    const f32x4 *W4 = reinterpret_cast<const f32x4 *>(Wrow);
    float s = 0.0f;
#pragma unroll
    for (int ob = 0; ob < kOB; ob++) {
        const f32x4 w = W4[ob * 4 + g];
#pragma unroll
        for (int r = 0; r < 4; r++) s = __fmaf_rn(w[r], act[ob][r], s);
    }
    s += __shfl_xor(s, 16);
    return s + __shfl_xor(s, 32);
}

static_assert(kFrags == 4, "OCC16_ENTER waits with vmcnt(8) = 2 chunks x 4 DMAs per wave");

// n_dev (nullable): the number of rows actually present, in device memory (the live-sample count of the
// frame, written by occnerf_live_rows on the same stream); the launch is sized for N_max rows and the
// workgroups beyond *n_dev leave at once -- no host round trip to learn the count.
// in_rows (nullable): sample n reads input row in_rows[n] (the list of distinct rows of occnerf_repeat_heads); outputs stay
// compact (row n).
__global__ __launch_bounds__(kWaves * 64, 2) void canonical_mlp_lds_kernel(const float *__restrict__ mlp_in,
                                                                           const int32_t *__restrict__ in_rows,
                                                                           int64_t N_max, const int32_t *__restrict__ n_dev,
                                                                           const float *__restrict__ pk,
                                                                           float *__restrict__ raw) {
    const int64_t N = n_dev ? (int64_t)*n_dev : N_max;
    if ((int64_t)blockIdx.x * (16 * kWaves) >= N) return;          // uniform for the workgroup
    // ONE __shared__ object (a second one makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) f32x4 smem[kRingSlots * kChunkF4 + Aux::kTotal / 4];
    f32x4 *ring = smem;
    float *aux = reinterpret_cast<float *>(smem + kRingSlots * kChunkF4);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
    const int64_t n = tile * 16 + s;
    const int64_t nsrc = n < N ? n : N - 1;      // the whole workgroup stays alive for the barriers

    // ---- side data -> LDS, inputs -> registers (ordinary loads, before any DMA is in flight) ----
    for (int i = threadIdx.x; i < Aux::kTotal; i += kWaves * 64) aux[i] = pk[Stream::kAux + i];
    float x[kKS_X + 3];
    {
        const float *row = mlp_in + (in_rows ? (int64_t)in_rows[nsrc] : nsrc) * kInGeo;
#pragma unroll
        for (int G = 0; G < 4; G++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 16 * G + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; r++) x[4 * G + r] = v[r];
        }
        x[16] = row[64 + g];
        x[17] = x[18] = x[19] = 0.0f;
    }
    __syncthreads();

    // ---- weight stream: chunk c lives in ring slot c & 3 ----
    const f32x4 *stream = reinterpret_cast<const f32x4 *>(pk);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) f32x4 *)ring;
    // LDS-DMA, wave-uniform source base (SGPR pair) + 32-bit lane offset (see the header comment)
    auto issue1 = [&](int c, int f) {      // 1 KiB fragment f of this wave's share of chunk c
        const int frag = wave * kFrags + f;
        unsigned keep;      // M0 carries the wave-uniform LDS destination; lane i lands at +16 i
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane * 16), "s"(stream + (size_t)c * kChunkF4 + frag * 64),
                       "s"(ring_lds + (unsigned)(((c & (kRingSlots - 1)) * kChunkF4 + frag * 64) * 16))
                     : "memory");
    };
    auto issue = [&](int c) {
#pragma unroll
        for (int f = 0; f < kFrags; f++) issue1(c, f);
    };
    int c = 0;                     // next chunk to enter
    issue(0);
    issue(1);
    issue(2);

    // Enter chunk c: wait for it (own quarter landed: 3 chunks x kFrags DMAs are outstanding, vmcnt(2*kFrags)
    // retires the oldest; the two younger chunks stay in flight), rendezvous, point slot_ at it.  After the
    // barrier every wave has finished reading chunk c-1 (its ds_reads completed before the lgkmcnt(0)), so
    // the step that follows may refill that slot.
    const f32x4 *slot_;
#define OCC16_ENTER()                                                  \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                                      \
    slot_ = ring + (c & (kRingSlots - 1)) * kChunkF4;                  \
    c++;
#define OCC16_READ_HALF(W, HALF)                                       \
    _Pragma("unroll") for (int ob_ = 0; ob_ < 8; ob_++) W[ob_] = slot_[((HALF) * 8 + ob_) * 64 + lane];

    // The weight registers are software-pipelined across chunks AND layers: wA holds the first 8 KiB of
    // the chunk being computed on entry to every step; each step reads the second half, computes the
    // first, enters the next chunk, reads ITS first half, computes the second.
    f32x4 wA[8];
    OCC16_ENTER()
    issue(3);
    OCC16_READ_HALF(wA, 0)

    // one chunk (= one group of 4 k-steps) of a 16-block layer
    // (the refill DMA of the chunk entered before a second half -- chunk c+2 after the c++ -- is issued
    // from inside that half)
#define OCC16_DMA(RR) issue1(c + 2, RR);
#define OCC16_HALF(W, HH, CL, KS, ACC, BOP)                                                      \
    _Pragma("unroll") for (int rr_ = 0; rr_ < 4; rr_++) {                                        \
        const int t_ = (CL) * 4 + rr_;                                                           \
        if ((HH) == 1) { OCC16_DMA(rr_) }                                                        \
        if (t_ < (KS)) {                                                                         \
            _Pragma("unroll") for (int ob_ = 0; ob_ < 8; ob_++)                                  \
                ACC[(HH) * 8 + ob_] = OCC16_MFMA(W[ob_][rr_], BOP(t_), ACC[(HH) * 8 + ob_]);     \
        }                                                                                        \
    }
#define OCC16_LAYER(CHUNKS, KS, ACC, BOP)                                                        \
    _Pragma("unroll") for (int c_ = 0; c_ < (CHUNKS); c_++) {                                    \
        f32x4 wB_[8];                                                                            \
        OCC16_READ_HALF(wB_, 1)                                                                  \
        OCC16_HALF(wA, 0, c_, KS, ACC, BOP)                                                      \
        OCC16_ENTER()                                                                            \
        OCC16_READ_HALF(wA, 0)                                                                   \
        OCC16_HALF(wB_, 1, c_, KS, ACC, BOP)                                                     \
    }

    f32x4 acc[kOB], act[kOB];
#define BOP_X(t) x[t]
#define BOP_ACT(t) act[(t) >> 2][(t) & 3]

    // ---------------- geometry trunk ----------------
    lds_bias<kOB>(acc, aux + Aux::kGeoL0B, g);
    OCC16_LAYER(kC_L0Geo, kKS_X, acc, BOP_X)
    relu_into(act, acc);
    // The six hidden layers share ONE unrolled body (keeps the hot code small enough for the instruction
    // cache that 2 CUs x 2 workgroups stream through); the geometry head and the colour trunk's first layer
    // run between the third and the fourth.
    f32x4 geo[4];
    float sigma = 0.0f;
#pragma unroll 1
    for (int l = 0; l < 6; l++) {
        if (l == 3) {
            // geometry head: 64 features on MFMA (4 blocks, no activation); a chunk carries 4 groups [q][ob][lane],
            // i.e. groups q = 0,1 in its first half and q = 2,3 in the second; sigma is a dot row
            lds_bias<4>(geo, aux + Aux::kGeoHeadB, g);
#define OCC16_HEAD_HALF(W, HH, CL)                                                               \
            _Pragma("unroll") for (int qq_ = 0; qq_ < 2; qq_++) {                                        \
                _Pragma("unroll") for (int rr_ = 0; rr_ < 4; rr_++) {                                    \
                    const int t_ = ((CL) * 4 + (HH) * 2 + qq_) * 4 + rr_;                                \
                    if ((HH) == 1 && (rr_ & 1) == 0) { OCC16_DMA(qq_ * 2 + (rr_ >> 1)) }                 \
                    _Pragma("unroll") for (int ob_ = 0; ob_ < 4; ob_++)                                  \
                        geo[ob_] = OCC16_MFMA(W[qq_ * 4 + ob_][rr_], BOP_ACT(t_), geo[ob_]);             \
                }                                                                                        \
            }
#pragma unroll
            for (int c_ = 0; c_ < kC_GeoHead; c_++) {
                f32x4 wB_[8];
                OCC16_READ_HALF(wB_, 1)
                OCC16_HEAD_HALF(wA, 0, c_)
                OCC16_ENTER()
                OCC16_READ_HALF(wA, 0)
                OCC16_HEAD_HALF(wB_, 1, c_)
            }
            sigma = dot_row(act, aux + Aux::kSigma, g) + aux[Aux::kSigma + 256];

            // ---------------- colour trunk ----------------
            lds_bias<kOB>(acc, aux + Aux::kRgbL0B, g);
#define BOP_RGB0(t) ((t) < 16 ? geo[((t) >> 2) & 3][(t) & 3] : x[((t) - 16) < 0 ? 0 : ((t) - 16)])
            OCC16_LAYER(kC_L0Rgb, kKS_L0Rgb, acc, BOP_RGB0)
            relu_into(act, acc);
        }
        lds_bias<kOB>(acc, aux + (l < 3 ? Aux::kGeoHB + l * 256 : Aux::kRgbHB + (l - 3) * 256), g);
        OCC16_LAYER(kC_Hidden, kKS_Hidden, acc, BOP_ACT)
        relu_into(act, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail chunks: nobody computes with them
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) rgb[ch] = dot_row(act, aux + Aux::kOut + ch * kWidth, g) + aux[Aux::kOut + 3 * kWidth + ch];

    if (g == 0 && n < N) {
        float *o = raw + n * 5;
        o[0] = rgb[0];
        o[1] = rgb[1];
        o[2] = rgb[2];
        o[3] = sigma;
    }
#undef BOP_X
#undef BOP_ACT
#undef BOP_RGB0
#undef OCC16_LAYER
#undef OCC16_ENTER
#undef OCC16_READ_HALF
#undef OCC16_HALF
#undef OCC16_DMA
#undef OCC16_HEAD_HALF
}

}  // namespace m16
}  // namespace occ

namespace occ {

int64_t mlp_lds_packed_floats() { return m16::Stream::kTotal; }

int mlp_lds_pack(const float *const *h_W, const float *const *h_b, float *packed, hipStream_t st) {
    using namespace m16;
    OCC_REQUIRE(hipMemsetAsync(packed + Stream::kTail, 0, sizeof(float) * (Stream::kTotal - Stream::kTail), st) ==
                    hipSuccess,
                "canonical_mlp_pack: memset failed");
    float *aux = packed + Stream::kAux;
    auto layer = [&](int li, int kind, int in_dim, int out_dim, int ks, int ob, int64_t woff, int boff) {
        hipLaunchKernelGGL(m16::pack_layer_kernel, dim3(256), dim3(256), 0, st, h_W[li], h_b[li], kind, in_dim,
                           out_dim, ks, ob, packed + woff, aux + boff);
    };
    layer(0, kL0Geo, kInGeo, kWidth, kKS_X, kOB, Stream::kGeoL0, Aux::kGeoL0B);
    for (int l = 0; l < 3; l++)
        layer(1 + l, kHidden, kWidth, kWidth, kKS_Hidden, kOB, Stream::kGeoH + l * kC_Hidden * Stream::kChunkFloats,
              Aux::kGeoHB + l * 256);
    layer(4, kGeoHead, kWidth, 65, kKS_Hidden, 4, Stream::kGeoHead, Aux::kGeoHeadB);
    hipLaunchKernelGGL(m16::pack_rows_kernel, dim3(4), dim3(256), 0, st, h_W[4], h_b[4], 1, aux + Aux::kSigma,
                       aux + Aux::kSigma + 256);
    layer(5, kL0Rgb, kInRgb, kWidth, kKS_L0Rgb, kOB, Stream::kRgbL0, Aux::kRgbL0B);
    for (int l = 0; l < 3; l++)
        layer(6 + l, kHidden, kWidth, kWidth, kKS_Hidden, kOB, Stream::kRgbH + l * kC_Hidden * Stream::kChunkFloats,
              Aux::kRgbHB + l * 256);
    hipLaunchKernelGGL(m16::pack_rows_kernel, dim3(4), dim3(256), 0, st, h_W[9], h_b[9], 3, aux + Aux::kOut,
                       aux + Aux::kOut + 768);
    return check_launch("canonical_mlp_pack");
}

int mlp_lds_launch(const float *mlp_in, const int32_t *in_rows, int64_t N, const int32_t *n_dev, const float *packed,
                   float *raw, hipStream_t st) {
    const int64_t per_block = 16 * m16::kWaves;
    const int64_t blocks = (N + per_block - 1) / per_block;
    OCC_REQUIRE(blocks < (1LL << 31), "canonical_mlp: N too large for one launch");
    hipLaunchKernelGGL(m16::canonical_mlp_lds_kernel, dim3((unsigned)blocks), dim3(64 * m16::kWaves), 0, st, mlp_in, in_rows,
                       N, n_dev, packed, raw);
    return check_launch("canonical_mlp");
}

}  // namespace occ
