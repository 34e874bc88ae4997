// Linear layers of the training step on the matrix pipe (the forward of
// occnerf_mlp.py:183-199 with the activations kept, and what autograd derives from it -- the reference trains
// through nn.Linear, trainer.py:239-249).
//
// Shape of the problem.  A training step pushes M = rays x samples rows (786 432 at 6 144 x 128) through
// ten layers that are at most 256 wide: every GEMM is "tall and skinny" -- M x 256 x 256 -- and in bf16 its
// arithmetic (103 GFLOP, 0.04 ms at the 2.5 PFLOP/s peak) is a third of the time HBM needs to stream the
// M x 256 operand in and the result out (0.8 GB, 0.13 ms).  So each layer is one streaming pass:
//
//   linear_kernel   Y[m,n] = epi(sum_k X[m,k] W[n,k])      forward (W = the layer's weight, bias + ReLU) and
//                                                          dgrad (W = its transpose, epilogue = ReLU mask of
//                                                          the saved activation): both are K-contiguous on
//                                                          both operands, so ONE kernel serves them.
//   wgrad_kernel    dW[n,k] = sum_m dZ[m,n] X[m,k]         contraction over the rows: split over the
//                   db[n]   = sum_m dZ[m,n]                workgroups, partial [256 x 256] tiles reduced by
//   wgrad_reduce_kernel                                    a second (tiny) kernel; no atomics, deterministic.
//
// Two arithmetic flavours of the same code: bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (BASELINE configs[4]) and exact fp32 on v_mfma_f32_32x32x2_f32 (the gradient-parity test against the
// reference's fp32 autograd).  Both read 16 bytes per lane per operand: 8 bf16 = one MFMA, or 4 floats = four.
//
// linear_kernel.  A workgroup of 4 waves owns 128 rows and ALL (<= 256) output columns; the product is formed
// transposed, D'[n][m] = sum_k W[n][k] X[m][k] (A operand = 32 weight rows, B operand = 32 sample rows), so
// that a lane ends up with 4 consecutive output columns of one row and stores them as one 8/16-byte piece.
// K runs in chunks of 256 bytes per row: the weight chunk [N][256 B] is brought into LDS by LDS-DMA (16-byte
// pieces XOR-swizzled by row so that the 16-lane groups of ds_read_b128 cover all 64 banks), the sample
// fragments come straight from global memory (32 rows x 32 contiguous bytes per load instruction); the
// result tile goes back out through LDS in whole rows.  A second K segment lets a layer read its input
// from two buffers (the colour trunk's first layer: geometry features + the sample row) without a concat.
//
// wgrad_kernel.  8 waves hold the whole 256 x 256 fp32 result (64 x 128 per wave = 128 accumulator registers),
// stream 32-row tiles of dZ and X through a double-buffered LDS image (row-major as loaded, 16 B per lane
// coalesced) and read their fragments transposed out of it: fp32 one dword per operand per MFMA; bf16 eight
// 2-byte reads per operand per MFMA (the LDS port, not the matrix pipe, paces that loop -- still under the HBM
// time of the tile).  Column sums of dZ (the bias gradient) ride along in the waves that own k-block 0.
#include "common.h"

namespace occ {
namespace lin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kChunkBytes = 256;           // K bytes of one row per chunk
constexpr int kRowsPerWG = 128;

template <bool BF16>
__device__ __forceinline__ f32x16 mma16(const u32x4 a, const u32x4 b, f32x16 c) {
    if constexpr (BF16) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    } else {
        // (whole-vector casts: element-wise __builtin_bit_cast(float, a[j]) made hipcc 7.2 feed element 0 to all four)
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c, 0, 0, 0);
        return c;
    }
}

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

struct LinearArgs {
    const char *x0;       // segment 0 of the input rows
    int64_t ld0;          // its row pitch, bytes
    int32_t k0;           // its width, bytes (multiple of 32)
    const char *x1;       // optional segment 1
    int64_t ld1;
    int32_t k1;
    const char *W;        // [n_pad][k0 + k1 bytes], row-major
    const float *bias;    // [n_pad] or NULL
    int32_t relu;
    const char *mask;     // [M][>= n_pad] in the element type, or NULL: outputs are zeroed where mask <= 0
    int64_t ldm;          // bytes
    char *y;
    int64_t ldy;          // bytes
    int32_t out_f32;      // store fp32 whatever the element type
    int32_t n_store;      // columns >= n_store are not stored
    float *aux;           // optional: column aux_col also goes, in fp32, to aux[m * aux_stride]
    int32_t aux_col;
    int64_t aux_stride;
    int64_t M;
};

// One output quad of a lane (4 consecutive columns n .. n+3 of row m): bias, ReLU, the optional fp32 side copy.
// (the bias sits in LDS -- zeros when the layer has none: 32 dependent global loads per lane, each waited for,
// were the single largest cost of the first version of this epilogue)
template <int NB>
__device__ __forceinline__ f32x4 out_quad(const f32x16 (&acc)[NB], int nb, int q, int n, const LinearArgs &a, int64_t m,
                                          const float *bias_lds) {
    f32x4 v;
    const f32x4 b = *reinterpret_cast<const f32x4 *>(bias_lds + n);
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = acc[nb][4 * q + j] + b[j];
    if (a.relu) {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = fmaxf(v[j], 0.0f);
    }
    if (a.aux && m < a.M && a.aux_col >= n && a.aux_col < n + 4) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (a.aux_col == n + j) a.aux[m * a.aux_stride] = v[j];
    }
    return v;
}

// Full-width epilogue.  The accumulator layout gives a lane 4 consecutive columns of ONE row, i.e. a wave-store
// of 8/16-byte pieces spread over 32 rows: every 128-byte line of the result would be written in 8 partial
// pieces by 8 different instructions (measured: the whole layer at 2 TB/s).  Instead each wave transposes its
// 32 x N tile through LDS (its own rows only: no workgroup barrier) and writes -- and reads the ReLU mask of the
// input-gradient form -- in whole rows, 16 bytes per lane, 1 KiB contiguous per instruction.
template <bool BF16, int NB, int OSZ>
__device__ __forceinline__ void epilogue_rows(const f32x16 (&acc)[NB], const LinearArgs &a, char *wl, const float *bias_lds,
                                              int lane, int wave) {
    constexpr int ESZ = BF16 ? 2 : 4;
    constexpr int NBP = OSZ == 4 ? (NB < 4 ? NB : 4) : NB;         // 32-column blocks per pass (<= 512 B per row)
    constexpr int OP = NBP * 32 * OSZ + 16;                        // LDS row pitch of the tile
    const int i = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * kRowsPerWG + wave * 32;
    char *ot = wl + wave * 32 * OP;
    constexpr int ITER = (NBP * 32 * OSZ / 16) / 2;                // 16-byte pieces per lane per pass (32 rows)
#pragma unroll
    for (int p0 = 0; p0 < NB; p0 += NBP) {
        const int p1 = p0 + NBP < NB ? p0 + NBP : NB;
        const int ppr = ((p1 - p0) * 32 * OSZ) >> 4;                // 16-byte pieces per row
        // the ReLU-mask rows of this pass, all in flight before the tile is even written
        u32x4 mk[ITER];
        if constexpr (OSZ == ESZ) {
            if (a.mask) {
#pragma unroll
                for (int it = 0; it < ITER; it++) {
                    const int pc = it * 64 + lane, r = pc / ppr, c = pc - r * ppr;
                    mk[it] = u32x4{0u, 0u, 0u, 0u};
                    if (pc < 32 * ppr && m0 + r < a.M)
                        mk[it] = *reinterpret_cast<const u32x4 *>(a.mask + (m0 + r) * a.ldm + (int64_t)p0 * 32 * OSZ + c * 16);
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            if (nb < p0 || nb >= p1) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = nb * 32 + 8 * q + 4 * h;
                const f32x4 v = out_quad<NB>(acc, nb, q, n, a, m0 + i, bias_lds);
                char *dst = ot + i * OP + (n - p0 * 32) * OSZ;
                if constexpr (OSZ == 4) {
                    *reinterpret_cast<f32x4 *>(dst) = v;
                } else {
                    u32x2 o;
                    o[0] = pack_bf16(v[0], v[1]);
                    o[1] = pack_bf16(v[2], v[3]);
                    *reinterpret_cast<u32x2 *>(dst) = o;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's own tile: no barrier needed
#pragma unroll
        for (int it = 0; it < ITER; it++) {
            const int pc = it * 64 + lane, r = pc / ppr, c = pc - r * ppr;
            const int64_t row = m0 + r;
            if (pc >= 32 * ppr || row >= a.M) continue;
            u32x4 v = *reinterpret_cast<const u32x4 *>(ot + r * OP + c * 16);
            if constexpr (OSZ == ESZ) {
                if (a.mask) {
                    const f32x4 mkf = __builtin_bit_cast(f32x4, mk[it]);      // (whole-vector cast, see mma16)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if constexpr (BF16) {
                            if (!((int32_t)(mk[it][e] << 16) > 0)) v[e] &= 0xffff0000u;
                            if (!((int32_t)(mk[it][e] & 0xffff0000u) > 0)) v[e] &= 0x0000ffffu;
                        } else {
                            if (!(mkf[e] > 0.0f)) v[e] = 0u;
                        }
                    }
                }
            }
            *reinterpret_cast<u32x4 *>(a.y + row * a.ldy + (int64_t)p0 * 32 * OSZ + c * 16) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // tile read before the next pass overwrites it
    }
}

template <int NB>
constexpr int linear_lds_bytes() {
    constexpr int w = NB * 32 * kChunkBytes;                              // weight chunk
    constexpr int o = kRowsPerWG * ((NB < 4 ? NB : 4) * 32 * 4 + 16);     // widest epilogue tile (4-byte outputs)
    constexpr int o2 = kRowsPerWG * (NB * 32 * 2 + 16);                   // 2-byte outputs, all columns in one pass
    return w > o ? (w > o2 ? w : o2) : (o > o2 ? o : o2);
}

template <bool BF16, int NB>
__global__ __launch_bounds__(256, 2) void linear_kernel(const LinearArgs a) {
    // ONE __shared__ object: [weight chunk | epilogue tiles][bias]
    __shared__ __attribute__((aligned(16))) char wl[linear_lds_bytes<NB>() + NB * 32 * 4];
    float *bias_lds = reinterpret_cast<float *>(wl + linear_lds_bytes<NB>());
    if (threadIdx.x < NB * 32) bias_lds[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.0f;   // visible after the first barrier
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int64_t m = (int64_t)blockIdx.x * kRowsPerWG + wave * 32 + i;
    const int64_t ms = m < a.M ? m : a.M - 1;
    const int64_t ldw = (int64_t)a.k0 + a.k1;
    const unsigned wl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)wl;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nb][r] = 0.0f;

    int wcol = 0;
#pragma unroll 1
    for (int seg = 0; seg < 2; seg++) {
        const int kb = seg ? a.k1 : a.k0;
        if (kb == 0) continue;
        const char *xrow = (seg ? a.x1 + ms * a.ld1 : a.x0 + ms * a.ld0) + h * 16;
#pragma unroll 1
        for (int c0 = 0; c0 < kb; c0 += kChunkBytes) {
            const int cb = kb - c0 < kChunkBytes ? kb - c0 : kChunkBytes;
            const int ppr = cb >> 4;                              // 16-byte pieces per weight row
            __syncthreads();                                      // the previous chunk has been read
            // Weight chunk -> LDS by LDS-DMA (no registers, one L2 round trip for the whole chunk).  A DMA
            // instruction fills 1 KiB of LDS linearly in lane order = 4 rows x 16 slots of 16 bytes; slot s of
            // row r receives the chunk's piece s ^ (r & 15) (the swizzle is applied on the SOURCE address), so
            // that the 16-lane groups of the ds_read_b128 below -- 16 different rows, one piece -- hit 16
            // different slots.  Pieces beyond a short last chunk are masked off.
#pragma unroll
            for (int it = 0; it < NB * 2; it++) {
                const int rb = wave * (NB * 8) + it * 4;                       // wave-uniform first row
                const int g = (lane & 15) ^ ((rb + (lane >> 4)) & 15);
                if (g < ppr) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"((unsigned)((lane >> 4) * (int)ldw + g * 16)), "s"(a.W + rb * ldw + wcol + c0),
                                   "s"(wl_lds + (unsigned)(rb * kChunkBytes))
                                 : "memory");
                }
            }
            u32x4 xf[8];
#pragma unroll
            for (int kg = 0; kg < 8; kg++)
                if (kg * 32 < cb) xf[kg] = *reinterpret_cast<const u32x4 *>(xrow + c0 + kg * 32);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMAs (invisible to the compiler) and the loads
            __syncthreads();
#pragma unroll
            for (int kg = 0; kg < 8; kg++) {
                if (kg * 32 < cb) {
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const u32x4 w = *reinterpret_cast<const u32x4 *>(
                            wl + (nb * 32 + i) * kChunkBytes + (((2 * kg + h) ^ (i & 15)) << 4));
                        acc[nb] = mma16<BF16>(w, xf[kg], acc[nb]);
                    }
                }
            }
        }
        wcol += kb;
    }

    const bool out4 = !BF16 || a.out_f32;
    if (a.n_store == NB * 32 && !(a.mask && BF16 && a.out_f32)) {
        __syncthreads();                                          // every wave is done with the weight chunk
        if (out4)
            epilogue_rows<BF16, NB, 4>(acc, a, wl, bias_lds, lane, wave);
        else
            epilogue_rows<BF16, NB, 2>(acc, a, wl, bias_lds, lane, wave);
        return;
    }
    // narrow outputs (the 3 colour logits): direct per-lane stores
    constexpr int ESZ = BF16 ? 2 : 4;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n = nb * 32 + 8 * q + 4 * h;
            f32x4 v = out_quad<NB>(acc, nb, q, n, a, m, bias_lds);
            if (m >= a.M) continue;
            if (a.mask) {
                const char *mp = a.mask + m * a.ldm + (int64_t)n * ESZ;
                if constexpr (BF16) {
                    const u32x2 mk = *reinterpret_cast<const u32x2 *>(mp);
                    if (!((int32_t)(mk[0] << 16) > 0)) v[0] = 0.0f;
                    if (!((int32_t)(mk[0] & 0xffff0000u) > 0)) v[1] = 0.0f;
                    if (!((int32_t)(mk[1] << 16) > 0)) v[2] = 0.0f;
                    if (!((int32_t)(mk[1] & 0xffff0000u) > 0)) v[3] = 0.0f;
                } else {
                    const f32x4 mk = *reinterpret_cast<const f32x4 *>(mp);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (!(mk[j] > 0.0f)) v[j] = 0.0f;
                }
            }
            if (n >= a.n_store) continue;
            if (out4) {
                float *yp = reinterpret_cast<float *>(a.y + m * a.ldy) + n;
                if (n + 4 <= a.n_store) {
                    *reinterpret_cast<f32x4 *>(yp) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (n + j < a.n_store) yp[j] = v[j];
                }
            } else {
                u32x2 o;
                o[0] = pack_bf16(v[0], v[1]);
                o[1] = pack_bf16(v[2], v[3]);
                *reinterpret_cast<u32x2 *>(a.y + m * a.ldy + (int64_t)n * 2) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const char *dz;       // [M][n_pad] element type
    int64_t lddz;         // bytes
    int32_t n_pad;        // multiple of 32, <= 256
    const char *x;        // [M][k_pad]
    int64_t ldx;
    int32_t k_pad;
    int64_t M;
    int64_t rows_per_wg;  // multiple of 32
    float *part;          // [gridDim.x][n_pad][k_pad]
    float *dbpart;        // [gridDim.x][n_pad]
};

template <bool BF16>
__global__ __launch_bounds__(512, 1) void wgrad_kernel(const WgradArgs a) {
    constexpr int ESZ = BF16 ? 2 : 4;
    // tile row pitch, bytes: bf16 rows 32 B apart modulo the 256-byte bank row, so that the four rows a 16-lane group of
    // ds_read_b64_tr_b16 touches fall into disjoint banks
    constexpr int TP = BF16 ? 256 * ESZ + 32 : 256 * ESZ + 16;
    constexpr int kTile = 32 * TP;
    constexpr int PIECES = BF16 ? 2 : 4;               // 16-byte pieces per thread per 32 x 256 tile
    __shared__ __attribute__((aligned(16))) char tiles[2 * 2 * kTile];      // [buffer][dz | x]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;           // n blocks 2wn, 2wn+1; k blocks 4wk .. 4wk+3
    const int64_t m_begin = (int64_t)blockIdx.x * a.rows_per_wg;
    const int64_t m_end = m_begin + a.rows_per_wg < a.M ? m_begin + a.rows_per_wg : a.M;
    const int ntiles = m_end > m_begin ? (int)((m_end - m_begin + 31) >> 5) : 0;
    const int ppr_z = (a.n_pad * ESZ) >> 4, ppr_x = (a.k_pad * ESZ) >> 4;

    f32x16 acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[u][v][r] = 0.0f;
    float dbs[2] = {0.0f, 0.0f};

    // (one tile in flight in registers beside the one being multiplied; a second register set, two tiles in flight, measured
    // no gain in round 6: the loop was paced by its LDS operand reads, not by the HBM round trip)
    u32x4 rz[PIECES], rx[PIECES];
    auto fetch = [&](int t) {
        const int64_t r0 = m_begin + (int64_t)t * 32;
#pragma unroll
        for (int e = 0; e < PIECES; e++) {
            const int p = threadIdx.x + e * 512;
            {
                const int row = p / ppr_z, pc = p - row * ppr_z;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < 32 && r0 + row < m_end) v = *reinterpret_cast<const u32x4 *>(a.dz + (r0 + row) * a.lddz + pc * 16);
                rz[e] = v;
            }
            {
                const int row = p / ppr_x, pc = p - row * ppr_x;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < 32 && r0 + row < m_end) v = *reinterpret_cast<const u32x4 *>(a.x + (r0 + row) * a.ldx + pc * 16);
                rx[e] = v;
            }
        }
    };
    auto stash = [&](int buf) {
        char *tz = tiles + buf * 2 * kTile, *tx = tz + kTile;
#pragma unroll
        for (int e = 0; e < PIECES; e++) {
            const int p = threadIdx.x + e * 512;
            {
                const int row = p / ppr_z, pc = p - row * ppr_z;
                if (row < 32) *reinterpret_cast<u32x4 *>(tz + row * TP + pc * 16) = rz[e];
            }
            {
                const int row = p / ppr_x, pc = p - row * ppr_x;
                if (row < 32) *reinterpret_cast<u32x4 *>(tx + row * TP + pc * 16) = rx[e];
            }
        }
    };

    const bool nact[2] = {(2 * wn) * 32 < a.n_pad, (2 * wn + 1) * 32 < a.n_pad};
    bool kact[4];
#pragma unroll
    for (int v = 0; v < 4; v++) kact[v] = (4 * wk + v) * 32 < a.k_pad;

    if (ntiles > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < ntiles; t++) {
        if (t + 1 < ntiles) fetch(t + 1);
        const char *tz = tiles + (t & 1) * 2 * kTile, *tx = tz + kTile;
        if (nact[0] && kact[0]) {
            if constexpr (BF16) {
                // Operand fragments by the LDS TRANSPOSE READ of gfx950 (round 6).  The contraction runs over the tile's ROWS, so a
                // lane's 8 operand values are 8 consecutive rows of ONE column of the row-major tile: eight 2-byte reads + packing
                // per operand before (96 LDS instructions per wave per tile: the loop was paced by them, 178 us per layer), two
                // ds_read_b64_tr_b16 now.  Measured semantics (tools/tr_b16_probe.hip): in a group of 16 lanes, lane p supplies
                // the address of 4 contiguous elements = row p >> 2, column chunk p & 3 of a 4 x 16 block; lane l receives column
                // l & 15 of the block's four rows.  Lane (i, h) of the MFMA operand = group g = lane >> 4: columns 16 (g & 1) ..,
                // rows 8 (g >> 1) + {0..3} in the first read, + {4..7} in the second.  Same fragments as before, bit for bit.
                const int g4 = lane >> 4, p16 = lane & 15;
                const unsigned lane_off = (unsigned)((8 * (g4 >> 1) + (p16 >> 2)) * TP + (16 * (g4 & 1) + 4 * (p16 & 3)) * 2);
                const unsigned tz_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)tz + lane_off;
                const unsigned tx_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)tx + lane_off;
                auto tr = [](unsigned addr) -> u32x2 {
                    u32x2 v;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                    return v;
                };
#pragma unroll
                for (int s = 0; s < 2; s++) {                      // k-steps of 16 rows
                    u32x2 ra[2][2], rb[4][2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
#pragma unroll
                        for (int q = 0; q < 2; q++) ra[u][q] = tr(tz_lds + (unsigned)((s * 16 + 4 * q) * TP + (2 * wn + u) * 64));
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++) {
#pragma unroll
                        for (int q = 0; q < 2; q++) rb[v][q] = tr(tx_lds + (unsigned)((s * 16 + 4 * q) * TP + (4 * wk + v) * 64));
                    }
                    // (the reads are asynchronous and invisible to the compiler: the wait takes every result as an operand, so no
                    // use can be scheduled above it)
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(rb[0][0]), "+v"(rb[0][1]),
                                   "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1])
                                 :
                                 : "memory");
                    u32x4 fa[2], fb[4];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        fa[u] = u32x4{ra[u][0][0], ra[u][0][1], ra[u][1][0], ra[u][1][1]};
                        if (nact[u] && wk == 0) {
#pragma unroll
                            for (int j = 0; j < 4; j++) dbs[u] += bf16_lo(fa[u][j]) + bf16_hi(fa[u][j]);
                        }
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++) fb[v] = u32x4{rb[v][0][0], rb[v][0][1], rb[v][1][0], rb[v][1][1]};
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        if (!nact[u]) continue;
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            if (!kact[v]) continue;
                            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, fa[u]), __builtin_bit_cast(bf16x8, fb[v]), acc[u][v], 0, 0, 0);
                        }
                    }
                }
            } else {
#pragma unroll 4
                for (int s = 0; s < 16; s++) {                     // k-steps of 2 rows
                    const int row = 2 * s + h;
                    float fa[2], fb[4];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        fa[u] = nact[u] ? *reinterpret_cast<const float *>(tz + row * TP + ((2 * wn + u) * 32 + i) * 4) : 0.0f;
                        if (wk == 0) dbs[u] += fa[u];
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++)
                        fb[v] = kact[v] ? *reinterpret_cast<const float *>(tx + row * TP + ((4 * wk + v) * 32 + i) * 4) : 0.0f;
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        if (!nact[u]) continue;
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            if (!kact[v]) continue;
                            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u], fb[v], acc[u][v], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (t + 1 < ntiles) stash((t + 1) & 1);
        __syncthreads();
    }

    float *part = a.part + (int64_t)blockIdx.x * a.n_pad * a.k_pad;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (!nact[u]) continue;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            if (!kact[v]) continue;
            const int k = (4 * wk + v) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = (2 * wn + u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                part[(int64_t)n * a.k_pad + k] = acc[u][v][r];
            }
        }
        if (wk == 0) {
            const float s = dbs[u] + __shfl_xor(dbs[u], 32);
            if (h == 0) a.dbpart[(int64_t)blockIdx.x * a.n_pad + (2 * wn + u) * 32 + i] = s;
        }
    }
}

// dW[row_map[n]][col_map[k]] (+)= sum_g part[g][n][k];  db[row_map[n]] (+)= sum_g dbpart[g][n]
// A 256-thread workgroup reduces 64 consecutive elements: wave w sums the partial tiles g = w (mod 4) -- a 256-byte row piece
// per load, eight independent float64 sums in flight -- and wave 0 adds the four waves' sums in a fixed order.  (The first
// form, one thread per element walking all G tiles, had 4 waves per CU in flight and read its 64 MB at 1.1 TB/s: 60 us per
// layer, a third of the weight-gradient kernel itself.)  The summation order is fixed by (G, element): deterministic.
constexpr int kRedElems = 64, kRedSlices = 4;
__global__ __launch_bounds__(kRedElems * kRedSlices) void wgrad_reduce_kernel(
    const float *__restrict__ part, const float *__restrict__ dbpart, int G, int n_pad, int k_pad,
    const int32_t *__restrict__ row_map, const int32_t *__restrict__ col_map, float *__restrict__ dW, int in_dim,
    float *__restrict__ db, int accumulate) {
    __shared__ double red[kRedSlices][kRedElems];
    const int lane = threadIdx.x & (kRedElems - 1), slice = threadIdx.x / kRedElems;
    const int nk = n_pad * k_pad;
    const int idx = blockIdx.x * kRedElems + lane;            // element of [dW (nk) | db (n_pad)]
    const bool is_w = idx < nk, is_b = !is_w && db && idx < nk + n_pad;
    const float *src = is_w ? part + idx : dbpart + (idx - nk);
    const int64_t pitch = is_w ? nk : n_pad;
    double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (is_w || is_b) {
        int g = slice;
        for (; g + 7 * kRedSlices < G; g += 8 * kRedSlices) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(int64_t)(g + u * kRedSlices) * pitch];
#pragma unroll
            for (int u = 0; u < 8; u++) s8[u] += (double)v[u];
        }
        for (; g < G; g += kRedSlices) s8[0] += (double)src[(int64_t)g * pitch];
    }
    red[slice][lane] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    __syncthreads();
    if (slice != 0 || !(is_w || is_b)) return;
    const double s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (is_w) {
        const int n = idx / k_pad, k = idx - n * k_pad;
        const int rn = row_map[n], ck = col_map[k];
        if (rn >= 0 && ck >= 0) {
            float *o = dW + (int64_t)rn * in_dim + ck;
            *o = (float)(accumulate ? (double)*o + s : s);
        }
    } else {
        const int rn = row_map[idx - nk];
        if (rn >= 0) db[rn] = (float)(accumulate ? (double)db[rn] + s : s);
    }
}

// Wp[n][k] = W[row_map[n]][col_map[k]] (0 where a map entry is -1), Wt = its transpose, bias_p[n] = b[row_map[n]]
template <bool BF16>
__global__ void pack_kernel(const float *__restrict__ W, const float *__restrict__ b, int in_dim,
                            const int32_t *__restrict__ row_map, int n_pad, const int32_t *__restrict__ col_map, int k_pad,
                            void *__restrict__ Wp, void *__restrict__ Wt, float *__restrict__ bias_p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_pad * k_pad) {
        const int n = idx / k_pad, k = idx - n * k_pad;
        const int rn = row_map[n], ck = col_map[k];
        const float w = (rn >= 0 && ck >= 0) ? W[(int64_t)rn * in_dim + ck] : 0.0f;
        if constexpr (BF16) {
            if (Wp) reinterpret_cast<__bf16 *>(Wp)[idx] = (__bf16)w;
            if (Wt) reinterpret_cast<__bf16 *>(Wt)[(int64_t)k * n_pad + n] = (__bf16)w;
        } else {
            if (Wp) reinterpret_cast<float *>(Wp)[idx] = w;
            if (Wt) reinterpret_cast<float *>(Wt)[(int64_t)k * n_pad + n] = w;
        }
    }
    if (bias_p && idx < n_pad) {
        const int rn = row_map[idx];
        bias_p[idx] = (rn >= 0 && b) ? b[rn] : 0.0f;
    }
}

template <bool BF16>
static int launch_linear(const LinearArgs &a, int n_pad, hipStream_t st) {
    const unsigned blocks = (unsigned)((a.M + kRowsPerWG - 1) / kRowsPerWG);
    switch (n_pad / 32) {
#define OCC_LIN_CASE(NB)                                                                                 \
    case NB:                                                                                             \
        hipLaunchKernelGGL((linear_kernel<BF16, NB>), dim3(blocks), dim3(256), 0, st, a);                \
        break;
        OCC_LIN_CASE(1)
        OCC_LIN_CASE(2)
        OCC_LIN_CASE(3)
        OCC_LIN_CASE(4)
        OCC_LIN_CASE(6)
        OCC_LIN_CASE(8)
#undef OCC_LIN_CASE
        default:
            set_error("linear_forward: n_pad=%d is not one of 32, 64, 96, 128, 192, 256", n_pad);
            return 1;
    }
    return check_launch("linear_forward");
}

}  // namespace lin
}  // namespace occ

OCC_API int occnerf_linear_pack(const float *W, const float *b, int32_t out_dim, int32_t in_dim,
                                const int32_t *row_map, int32_t n_pad, const int32_t *col_map, int32_t k_pad,
                                int32_t bf16, void *Wp, void *Wt, float *bias_p, void *stream) {
    using namespace occ;
    OCC_REQUIRE(W && row_map && col_map && (Wp || Wt), "linear_pack: null argument");
    OCC_REQUIRE(out_dim > 0 && in_dim > 0 && n_pad > 0 && k_pad > 0 && n_pad % 32 == 0 && k_pad % 32 == 0,
                "linear_pack: n_pad=%d, k_pad=%d must be positive multiples of 32", n_pad, k_pad);
    const int total = n_pad * k_pad;
    if (bf16)
        hipLaunchKernelGGL(lin::pack_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), W, b,
                           in_dim, row_map, n_pad, col_map, k_pad, Wp, Wt, bias_p);
    else
        hipLaunchKernelGGL(lin::pack_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), W, b,
                           in_dim, row_map, n_pad, col_map, k_pad, Wp, Wt, bias_p);
    return check_launch("linear_pack");
}

OCC_API int occnerf_linear_forward(const void *x0, int64_t ld0, int32_t k0, const void *x1, int64_t ld1, int32_t k1,
                                   const void *W, const float *bias, int32_t relu, const void *mask, int64_t ldm,
                                   void *y, int64_t ldy, int32_t out_f32, int32_t n_store, float *aux, int32_t aux_col,
                                   int64_t aux_stride, int64_t M, int32_t n_pad, int32_t bf16, void *stream) {
    using namespace occ;
    if (M <= 0) return 0;
    OCC_REQUIRE(x0 && W && y, "linear_forward: null argument");
    const int esz = bf16 ? 2 : 4;
    OCC_REQUIRE(k0 > 0 && k0 % 32 == 0 && k1 >= 0 && k1 % 32 == 0 && (k1 == 0 || x1),
                "linear_forward: segment widths k0=%d, k1=%d must be multiples of 32 elements", k0, k1);
    OCC_REQUIRE(n_pad > 0 && n_pad % 32 == 0 && n_pad <= 256, "linear_forward: n_pad=%d", n_pad);
    OCC_REQUIRE(ld0 >= k0 && (k1 == 0 || ld1 >= k1) && ld0 % (16 / esz) == 0 && ld1 % (16 / esz) == 0,
                "linear_forward: row pitches must cover the segment and keep rows 16-byte aligned");
    const int ysz = (bf16 && !out_f32) ? 2 : 4;
    OCC_REQUIRE(n_store > 0 && n_store <= n_pad && ldy >= n_store && (n_store % 4 != 0 || ldy % (16 / ysz) == 0),
                "linear_forward: n_store=%d, ldy=%lld", n_store, (long long)ldy);
    OCC_REQUIRE(!mask || (ldm >= n_pad && ldm % (16 / esz) == 0), "linear_forward: mask pitch %lld", (long long)ldm);
    OCC_REQUIRE(!aux || (aux_col >= 0 && aux_col < n_pad), "linear_forward: aux_col=%d", aux_col);
    lin::LinearArgs a;
    a.x0 = (const char *)x0; a.ld0 = ld0 * esz; a.k0 = k0 * esz;
    a.x1 = (const char *)x1; a.ld1 = ld1 * esz; a.k1 = k1 * esz;
    a.W = (const char *)W; a.bias = bias; a.relu = relu;
    a.mask = (const char *)mask; a.ldm = ldm * esz;
    a.y = (char *)y; a.ldy = ldy * ysz; a.out_f32 = out_f32; a.n_store = n_store;
    a.aux = aux; a.aux_col = aux_col; a.aux_stride = aux_stride; a.M = M;
    return bf16 ? lin::launch_linear<true>(a, n_pad, as_stream(stream)) : lin::launch_linear<false>(a, n_pad, as_stream(stream));
}

OCC_API int32_t occnerf_linear_wgrad_slices(int64_t M) {
    if (M <= 0) return 1;
    const int64_t tiles = (M + 31) / 32;
    return (int32_t)(tiles < occ::kNumCU ? tiles : occ::kNumCU);
}

OCC_API int occnerf_linear_wgrad(const void *dz, int64_t lddz, int32_t n_pad, const void *x, int64_t ldx, int32_t k_pad,
                                 int64_t M, int32_t bf16, float *part, float *dbpart, void *stream) {
    using namespace occ;
    OCC_REQUIRE(dz && x && part && dbpart, "linear_wgrad: null argument");
    OCC_REQUIRE(M > 0, "linear_wgrad: M=%lld", (long long)M);
    const int esz = bf16 ? 2 : 4;
    OCC_REQUIRE(n_pad > 0 && n_pad % 32 == 0 && n_pad <= 256 && k_pad > 0 && k_pad % 32 == 0 && k_pad <= 256,
                "linear_wgrad: n_pad=%d, k_pad=%d", n_pad, k_pad);
    OCC_REQUIRE(lddz >= n_pad && ldx >= k_pad && lddz % (16 / esz) == 0 && ldx % (16 / esz) == 0,
                "linear_wgrad: row pitches must cover the rows and keep them 16-byte aligned");
    const int G = occnerf_linear_wgrad_slices(M);
    const int64_t tiles = (M + 31) / 32;
    lin::WgradArgs a;
    a.dz = (const char *)dz; a.lddz = lddz * esz; a.n_pad = n_pad;
    a.x = (const char *)x; a.ldx = ldx * esz; a.k_pad = k_pad;
    a.M = M; a.rows_per_wg = ((tiles + G - 1) / G) * 32;
    a.part = part; a.dbpart = dbpart;
    if (bf16)
        hipLaunchKernelGGL(lin::wgrad_kernel<true>, dim3(G), dim3(512), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(lin::wgrad_kernel<false>, dim3(G), dim3(512), 0, as_stream(stream), a);
    return check_launch("linear_wgrad");
}

OCC_API int occnerf_linear_wgrad_reduce(const float *part, const float *dbpart, int32_t G, int32_t n_pad, int32_t k_pad,
                                        const int32_t *row_map, const int32_t *col_map, float *dW, int32_t in_dim,
                                        float *db, int32_t accumulate, void *stream) {
    using namespace occ;
    OCC_REQUIRE(part && dbpart && row_map && col_map && dW, "linear_wgrad_reduce: null argument");
    OCC_REQUIRE(G > 0 && n_pad > 0 && k_pad > 0 && in_dim > 0, "linear_wgrad_reduce: bad sizes");
    const int total = n_pad * k_pad + n_pad;
    hipLaunchKernelGGL(lin::wgrad_reduce_kernel, dim3((total + lin::kRedElems - 1) / lin::kRedElems),
                       dim3(lin::kRedElems * lin::kRedSlices), 0, as_stream(stream), part, dbpart,
                       G, n_pad, k_pad, row_map, col_map, dW, in_dim, db, accumulate);
    return check_launch("linear_wgrad_reduce");
}
