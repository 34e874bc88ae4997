// Multi-resolution hash-grid encoder for gfx950: forward (+dy_dx), backward, TV stub.
//
// Operator-level replacement of the reference's `_gridencoder` extension
// (core/nets/occnerf/gridencoder/src/gridencoder.cu:87-369, bindings.cpp:5-9).
// The sample pipeline does not call these kernels for its 33 M samples per frame -- it
// uses the fused occ::encode_level_d4c2 inside sample_features.hip -- but the operator is
// the reference's native seam for this path, and the per-point table build, the training
// backward and the operator-level parity tests go through it.
//
// Bound: HBM/L2 gather.  Algorithmic bytes per (sample, level): 2^D corners x C x 4 B read
// (D=4,C=2: 128 B) + C x 4 B written; the dense levels 0-1 are L2-resident, a hashed level
// is a 4 MiB table (= one XCD L2).  Level is the slow grid axis, as in the reference's
// [L,B,C] output layout, so that concurrently running blocks work on the same level.
#include "common.h"

#include <vector>

namespace occ {

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_forward_kernel(
    const float *__restrict__ inputs, const float *__restrict__ embeddings,
    const int32_t *__restrict__ offsets, float *__restrict__ outputs, uint32_t B, uint32_t L,
    GridLevels lv, float *__restrict__ dy_dx, uint32_t gridtype, bool align_corners,
    uint32_t interp) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;

    const float *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
    const float *x = inputs + (size_t)b * D;
    float *out = outputs + ((size_t)level * B + b) * C;
    float *dyl = dy_dx ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;

    float xin[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        xin[d] = x[d];
        oob |= (xin[d] < 0.f || xin[d] > 1.f);
    }
    if (oob) {  // gridencoder.cu:118-135: rows outside [0,1] encode to zero
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0.f;
        if (dyl) {
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) dyl[i] = 0.f;
        }
        return;
    }

    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];

    float pos[D], pos_deriv[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = __fmaf_rn(xin[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= fl;
        if (interp == 1) {
            pos_deriv[d] = __fmul_rn(__fmul_rn(6.f, pos[d]), __fsub_rn(1.0f, pos[d]));
            pos[d] = __fmul_rn(__fmul_rn(pos[d], pos[d]),
                               __fsub_rn(3.0f, __fmul_rn(2.0f, pos[d])));
        } else {
            pos_deriv[d] = 1.0f;
        }
    }

    float results[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) results[ch] = 0.f;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) {
                w = __fmul_rn(w, __fsub_rn(1.f, pos[d]));
                pl[d] = pg[d];
            } else {
                w = __fmul_rn(w, pos[d]);
                pl[d] = pg[d] + 1;
            }
        }
        const uint32_t index = grid_index<D>(gridtype, align_corners, hashmap_size, resolution, pl) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) results[ch] = __fmaf_rn(w, grid[index + ch], results[ch]);
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];

    if (dyl) {  // gridencoder.cu:201-244
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0.f;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t pl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) {
                        w = __fmul_rn(w, __fsub_rn(1.f, pos[d]));
                        pl[d] = pg[d];
                    } else {
                        w = __fmul_rn(w, pos[d]);
                        pl[d] = pg[d] + 1;
                    }
                }
                pl[gd] = pg[gd];
                const uint32_t il = grid_index<D>(gridtype, align_corners, hashmap_size, resolution, pl) * C;
                pl[gd] = pg[gd] + 1;
                const uint32_t ir = grid_index<D>(gridtype, align_corners, hashmap_size, resolution, pl) * C;
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch++)
                    rg[ch] = __fmaf_rn(__fmul_rn(w, __fsub_rn(grid[ir + ch], grid[il + ch])),
                                       pos_deriv[gd], rg[ch]);
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) dyl[gd * C + ch] = rg[ch];
        }
    }
}

// Round 5 (knob "grid_xcd"): the same encoder with the LEVELS dealt to the XCDs.  Workgroups are
// dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own 4 MiB L2, and a hashed level is a 4 MiB
// table: in the sample-major kernels every XCD's L2 sees all 16 levels (59 MiB).  Here block b serves level pair b % 8 for the
// samples of chunk b / 8 -- one thread per sample, both levels of the pair -- so an XCD's L2 holds 2 levels.  The price: every
// sample's 16-byte input is read by 8 workgroups, and the outputs only come out level-major ([L,B,C], the operator's layout;
// the renderer's feature kernel wants them inside a sample-major 272-byte row).  Same bits.  Measured against the sample-major
// kernel on the benchmark frame's 17.6 M encoder inputs (profiles/r05_xcd_levels.md): 5.31 against 8.59 ms, FETCH_SIZE 5.3
// against 10.5 GB -- so the OPERATOR (training step, per-point table) uses it for large batches; the renderer's fused feature
// kernel keeps its layout (after the centre shortcuts a third of its samples still encode, ~0.5-1 ms of its 7.8: a separate
// XCD-dealt pass for them would cost more than it saves).
__global__ __launch_bounds__(256) void grid_forward_d4c2_xcd_kernel(const float4 *__restrict__ inputs,
                                                                    const float2 *__restrict__ embeddings,
                                                                    const int32_t *__restrict__ offsets,
                                                                    float2 *__restrict__ outputs, uint32_t B, uint32_t L,
                                                                    GridLevels lv, GridModes4 gm) {
    const uint32_t pair = blockIdx.x & 7u;
    const uint32_t b = (blockIdx.x >> 3) * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float4 xv = inputs[b];
    const float x[4] = {xv.x, xv.y, xv.z, xv.w};
    const bool oob = x[0] < 0.f || x[0] > 1.f || x[1] < 0.f || x[1] > 1.f || x[2] < 0.f || x[2] > 1.f || x[3] < 0.f || x[3] > 1.f;
#pragma unroll
    for (uint32_t u = 0; u < 2; u++) {
        const uint32_t l = 2 * pair + u;
        if (l >= L) continue;
        float2 v = make_float2(0.f, 0.f);
        if (!oob) {
            const uint32_t o0 = (uint32_t)offsets[l];
            v = encode_level_d4c2(x, embeddings, (uint32_t)offsets[l + 1] - o0, lv.scale[l], lv.resolution[l], gm.mode[l], o0);
        }
        outputs[(size_t)l * B + b] = v;
    }
}

// The encoder the canonical MLP uses (D = 4, C = 2, hash grid, linear interpolation, no input gradient), shaped for
// this machine instead of the reference's thread-per-(sample, level) grid: 8 lanes share a sample and take 2 levels
// each (occ::encode_level_d4c2: shared partial corner weights, per-level index mode decided on the HOST -- dense /
// power-of-two mask / generic -- so the 35-instruction urem of the generic path never runs for this layout), the
// sample's 16 bytes are read once per lane group instead of once per level, and a wave writes 8 samples x 16 levels.
// Same results bit for bit as grid_forward_kernel (and as the fused copy inside features.hip).
__global__ __launch_bounds__(256) void grid_forward_d4c2_kernel(const float4 *__restrict__ inputs,
                                                                const float2 *__restrict__ embeddings,
                                                                const int32_t *__restrict__ offsets, float2 *__restrict__ outputs,
                                                                uint32_t B, uint32_t L, GridLevels lv, GridModes4 gm) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t >> 3, j = t & 7;
    if (b >= B) return;
    const float4 xv = inputs[b];
    const float x[4] = {xv.x, xv.y, xv.z, xv.w};
    const bool oob = x[0] < 0.f || x[0] > 1.f || x[1] < 0.f || x[1] > 1.f || x[2] < 0.f || x[2] > 1.f || x[3] < 0.f || x[3] > 1.f;
#pragma unroll
    for (uint32_t u = 0; u < 2; u++) {
        const uint32_t l = 2 * j + u;
        if (l >= L) continue;
        float2 v = make_float2(0.f, 0.f);                       // gridencoder.cu:118-135: rows outside [0,1] encode to zero
        if (!oob) {
            const uint32_t o0 = (uint32_t)offsets[l];
            v = encode_level_d4c2(x, embeddings, (uint32_t)offsets[l + 1] - o0, lv.scale[l], lv.resolution[l], gm.mode[l], o0);
        }
        outputs[(size_t)l * B + b] = v;
    }
}

// gridencoder.cu:248-340.  One thread per (sample, level); all C channels of a corner are
// added by the same thread (C <= 8), fp32 atomics into the zero-initialised gradient table.  (Used for small
// batches and the general D/C/gridtype cases; large D = 4, C = 2 batches take the tiled kernel below.)
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_backward_kernel(
    const float *__restrict__ grad, const float *__restrict__ inputs,
    const int32_t *__restrict__ offsets, float *__restrict__ grad_grid, uint32_t B, uint32_t L,
    GridLevels lv, uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    float *gg = grad_grid + (size_t)(uint32_t)offsets[level] * C;
    const float *x = inputs + (size_t)b * D;
    const float *g = grad + ((size_t)level * B + b) * C;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];

    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float xd = x[d];
        if (xd < 0.f || xd > 1.f) return;  // gradient stays zero
        pos[d] = __fmaf_rn(xd, scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= fl;
        if (interp == 1)
            pos[d] = __fmul_rn(__fmul_rn(pos[d], pos[d]), __fsub_rn(3.0f, __fmul_rn(2.0f, pos[d])));
    }
    float gc[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) gc[ch] = g[ch];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) {
                w = __fmul_rn(w, __fsub_rn(1.f, pos[d]));
                pl[d] = pg[d];
            } else {
                w = __fmul_rn(w, pos[d]);
                pl[d] = pg[d] + 1;
            }
        }
        const uint32_t index = grid_index<D>(gridtype, align_corners, hashmap_size, resolution, pl) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) atomicAdd(&gg[index + ch], __fmul_rn(w, gc[ch]));
    }
}

// ---------------------------------------------------------------------------------------
// Tiled backward for the D = 4, C = 2 hash encoder (the one the canonical MLP uses).
//
// Global fp32 atomics are performed on the memory side on this part (8 XCDs with private L2s): the scatter of
// a 786 K-sample batch ran at 1.6 G atomics/s -- 112 ms for the seven fine levels alone.  Here a workgroup
// OWNS a tile of kTileEntries consecutive table entries of one level, holds its gradient in LDS (128 KiB),
// scans ALL samples, recomputes the 16 corner indices of each (a dozen integer ops per corner) and adds the
// contributions that fall into its tile with LDS atomics; at the end it adds the tile to the gradient table
// with plain read-modify-writes -- no other workgroup touches those entries.  Index arithmetic is repeated
// once per tile of the level (<= 64 times), which costs ~2 ms of integer work in total for that batch.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kTileEntries = 8192;         // x 2 channels x 8 B (fp64 accumulators) = 128 KiB of LDS

struct TileJobs {                                // block -> (level, tile, sample slice), arithmetically
    uint32_t first_block[kMaxLevels + 1];        // blocks of level l: [first_block[l], first_block[l+1])
    uint32_t nslices[kMaxLevels];                // sample slices per tile of that level (1: the job owns its tile outright)
    GridModes4 modes;
};

// Per-sample work of a tile-job: cell, corner indices, the corners that fall into [base, base + kTileEntries) as a bit
// mask (no branches), then one LDS atomic pair per hit.  MODE (kGridDense / kGridHashPow2 / kGridGeneric) is a template
// parameter: as a run-time value hipcc replicated the corner loop per mode and branched on every hit test.
template <uint32_t MODE>
struct TileSample {
    float f[4][2];
    uint32_t pg[4], t[4][2];
    bool in;

    __device__ __forceinline__ void setup(const float4 xv, float scale, uint32_t resolution) {
        const float x[4] = {xv.x, xv.y, xv.z, xv.w};
        in = true;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            in = in && !(x[d] < 0.f || x[d] > 1.f);           // rows outside [0,1] keep a zero gradient (gridencoder.cu:262-266)
            float pos = __fmaf_rn(x[d], scale, 0.5f);
            const float fl = floorf(pos);
            pg[d] = (uint32_t)fl;
            pos -= fl;
            f[d][0] = __fsub_rn(1.f, pos);
            f[d][1] = pos;
        }
        if (MODE == kGridDense) {
            uint32_t stride = 1;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                t[d][0] = pg[d] * stride;
                t[d][1] = t[d][0] + stride;
                stride *= resolution + 1;
            }
        } else {
            constexpr uint32_t primes[4] = {1u, 2654435761u, 805459861u, 3674653429u};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                t[d][0] = pg[d] * primes[d];
                t[d][1] = t[d][0] + primes[d];
            }
        }
    }
    __device__ __forceinline__ uint32_t corner_index(uint32_t c, uint32_t size, uint32_t resolution) const {
        const uint32_t t0 = (c & 1) ? t[0][1] : t[0][0], t1 = (c & 2) ? t[1][1] : t[1][0];
        const uint32_t t2 = (c & 4) ? t[2][1] : t[2][0], t3 = (c & 8) ? t[3][1] : t[3][0];
        if (MODE == kGridDense) return t0 + t1 + t2 + t3;
        if (MODE == kGridHashPow2) return (t0 ^ t1 ^ t2 ^ t3) & (size - 1);
        const uint32_t pl[4] = {pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), pg[3] + ((c >> 3) & 1)};
        return grid_index<4>(0, false, size, resolution, pl);
    }
    __device__ __forceinline__ float corner_weight(uint32_t c) const {     // ((1 * a0) * a1) * a2) * a3, as the scatter kernel
        return __fmul_rn(__fmul_rn(__fmul_rn((c & 1) ? f[0][1] : f[0][0], (c & 2) ? f[1][1] : f[1][0]), (c & 4) ? f[2][1] : f[2][0]),
                         (c & 8) ? f[3][1] : f[3][0]);
    }
    // An LDS atomic costs its cycles per wave-instruction however few lanes are active, and with one conditional pair per
    // corner almost every one of the 32 instructions finds SOME lane with a hit: the hits are first collected as a
    // per-lane bit mask and then drained together -- max-over-lanes(hits) = 2-3 pairs per wave instead of 32.
    __device__ __forceinline__ void accumulate(double *s_g, uint32_t base, uint32_t size, uint32_t resolution, float2 gv,
                                               bool live) const {
        uint32_t hits = 0;
#pragma unroll
        for (uint32_t c = 0; c < 16; c++) hits |= (uint32_t)(corner_index(c, size, resolution) - base < kTileEntries) << c;
        hits = (live && in) ? hits : 0u;
        while (hits) {
            const uint32_t c = (uint32_t)__builtin_ctz(hits);
            hits &= hits - 1;
            const uint32_t local = corner_index(c, size, resolution) - base;
            const float w = corner_weight(c);
            atomicAdd(&s_g[local * 2], (double)__fmul_rn(w, gv.x));
            atomicAdd(&s_g[local * 2 + 1], (double)__fmul_rn(w, gv.y));
        }
    }
};

// Plain scan: every tile-job of a level re-hashes every sample (64 times per hashed level).
template <uint32_t MODE>
__device__ __forceinline__ void grid_backward_tile_scan(double *s_g, const float2 *__restrict__ g2,
                                                        const float4 *__restrict__ inputs, uint32_t b_lo, uint32_t B,
                                                        uint32_t tile, uint32_t size, float scale, uint32_t resolution) {
    const uint32_t base = tile * kTileEntries;
    constexpr uint32_t U = 4;                                   // four samples per trip: 8 loads in flight per lane
    for (uint32_t b0 = b_lo + threadIdx.x; b0 < B; b0 += blockDim.x * U) {
        float4 xv4[U];
        float2 gv4[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t bb = b0 + u * blockDim.x < B ? b0 + u * blockDim.x : B - 1;
            xv4[u] = inputs[bb];
            gv4[u] = g2[bb];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const float2 gv = gv4[u];
            TileSample<MODE> ts;
            ts.setup(xv4[u], scale, resolution);
            // exact-zero gradient rows (samples the compositor masks out) add exact zeros
            ts.accumulate(s_g, base, size, resolution, gv, b0 + u * blockDim.x < B && !(gv.x == 0.0f && gv.y == 0.0f));
        }
    }
}

// Masked scan.  A pre-pass (grid_tile_mask_kernel) has written, per (level, sample), the 64-bit set of tiles that the
// sample's 16 corners touch (a hashed level has exactly 64 tiles; 0 for rows outside [0,1] and rows with a zero gradient).
// A tile-job then reads 8 bytes per sample instead of 24 and hashes only the samples whose bit is set -- 22 % on a hashed
// level (1 - (63/64)^16) -- after COMPACTING them: each wave scans 512 samples, packs the offsets of the set ones into a
// small LDS queue by ballot + prefix count, and works the queue off 64 at a time, so the hashing runs with full lanes.
constexpr uint32_t kMaskChunk = 512;                            // samples per wave per trip

template <uint32_t MODE>
__device__ __forceinline__ void grid_backward_tile_scan_masked(double *s_g, uint16_t *queue /*[kMaskChunk] of this wave*/,
                                                               const unsigned long long *__restrict__ masks,
                                                               const float2 *__restrict__ g2,
                                                               const float4 *__restrict__ inputs, uint32_t b_lo, uint32_t B,
                                                               uint32_t tile, uint32_t size, float scale, uint32_t resolution) {
    const uint32_t base = tile * kTileEntries;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (uint32_t c0 = b_lo + wave * kMaskChunk; c0 < B; c0 += nwaves * kMaskChunk) {
        unsigned long long m[kMaskChunk / 64];
#pragma unroll
        for (uint32_t u = 0; u < kMaskChunk / 64; u++) {
            const uint32_t bb = c0 + u * 64 + lane;
            m[u] = bb < B ? masks[bb] : 0ull;
        }
        uint32_t count = 0;
#pragma unroll
        for (uint32_t u = 0; u < kMaskChunk / 64; u++) {
            const bool on = (m[u] >> tile) & 1ull;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(on);
            if (on) queue[count + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(u * 64 + lane);
            count += (uint32_t)__builtin_popcountll(bal);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own queue: no barrier needed
        for (uint32_t q = 0; q < count; q += 128) {              // two rounds of 64: 4 loads in flight per lane
            bool live[2];
            float4 xv[2];
            float2 gv[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                live[r] = q + r * 64 + lane < count;
                uint32_t bb = c0 + (live[r] ? queue[q + r * 64 + lane] : 0);
                bb = bb < B ? bb : B - 1;
                xv[r] = inputs[bb];
                gv[r] = g2[bb];
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (q + r * 64 >= count) break;                   // wave-uniform
                TileSample<MODE> ts;
                ts.setup(xv[r], scale, resolution);
                ts.accumulate(s_g, base, size, resolution, gv[r], live[r]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // queue read before the next trip overwrites it
    }
}

// masks[level][b] = set of tiles (index / kTileEntries) touched by the 16 corners of sample b at that level
__global__ __launch_bounds__(256) void grid_tile_mask_kernel(const float *__restrict__ grad, const float4 *__restrict__ inputs,
                                                             const int32_t *__restrict__ offsets, uint32_t B, GridLevels lv,
                                                             GridModes4 gm, unsigned long long *__restrict__ masks) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, level = blockIdx.y;
    if (b >= B) return;
    const float2 gv = reinterpret_cast<const float2 *>(grad)[(size_t)level * B + b];
    const uint32_t size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t mode = gm.mode[level];
    unsigned long long m = 0ull;
    auto build = [&](auto ts) {
        ts.setup(inputs[b], lv.scale[level], lv.resolution[level]);
#pragma unroll
        for (uint32_t c = 0; c < 16; c++) m |= 1ull << (ts.corner_index(c, size, lv.resolution[level]) / kTileEntries);
        if (!ts.in) m = 0ull;
    };
    if (mode == kGridDense) build(TileSample<kGridDense>());
    else if (mode == kGridHashPow2) build(TileSample<kGridHashPow2>());
    else build(TileSample<kGridGeneric>());
    if (gv.x == 0.0f && gv.y == 0.0f) m = 0ull;                  // exact zeros need not be added
    masks[(size_t)level * B + b] = m;
}

__global__ __launch_bounds__(1024) void grid_backward_tiled_d4c2_kernel(
    const float *__restrict__ grad, const float4 *__restrict__ inputs, const int32_t *__restrict__ offsets,
    float *__restrict__ grad_grid, uint32_t B, GridLevels lv, TileJobs jobs,
    const unsigned long long *__restrict__ masks /*[L][B] or NULL*/) {
    // fp64 accumulators: ds_add_f64 costs 16 cycles per wave-instruction on gfx950, ds_add_f32 190
    // (tools/lds_atomic_rate.hip); the per-cell sums are also more accurate than the scatter kernel's.
    // ONE __shared__ object: [tile: 128 KiB][16 waves x kMaskChunk offsets: 16 KiB]
    __shared__ double smem[kTileEntries * 2 + 16 * kMaskChunk * sizeof(uint16_t) / sizeof(double)];
    double *s_g = smem;
    uint16_t *queue = reinterpret_cast<uint16_t *>(smem + kTileEntries * 2) + (threadIdx.x >> 6) * kMaskChunk;
    uint32_t level = 0;
    while (level + 1 < kMaxLevels && blockIdx.x >= jobs.first_block[level + 1]) level++;
    const uint32_t r = blockIdx.x - jobs.first_block[level], nsl = jobs.nslices[level];
    const uint32_t tile = r / nsl, slice = r - tile * nsl;
    for (uint32_t i = threadIdx.x; i < kTileEntries * 2; i += blockDim.x) s_g[i] = 0.0;
    __syncthreads();
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    const uint32_t mode = jobs.modes.mode[level];
    const float2 *g2 = reinterpret_cast<const float2 *>(grad) + (size_t)level * B;
    // this job's share of the samples (multiples of the wave chunk, so that slices never split a chunk)
    const uint32_t per = ((B + nsl - 1) / nsl + kMaskChunk - 1) / kMaskChunk * kMaskChunk;
    const uint32_t b_lo = slice * per < B ? slice * per : B;
    const uint32_t b_hi = b_lo + per < B ? b_lo + per : B;
    if (masks) {
        const unsigned long long *mk = masks + (size_t)level * B;
        if (mode == kGridDense)
            grid_backward_tile_scan_masked<kGridDense>(s_g, queue, mk, g2, inputs, b_lo, b_hi, tile, size, scale, resolution);
        else if (mode == kGridHashPow2)
            grid_backward_tile_scan_masked<kGridHashPow2>(s_g, queue, mk, g2, inputs, b_lo, b_hi, tile, size, scale, resolution);
        else
            grid_backward_tile_scan_masked<kGridGeneric>(s_g, queue, mk, g2, inputs, b_lo, b_hi, tile, size, scale, resolution);
    } else if (mode == kGridDense) {
        grid_backward_tile_scan<kGridDense>(s_g, g2, inputs, b_lo, b_hi, tile, size, scale, resolution);
    } else if (mode == kGridHashPow2) {
        grid_backward_tile_scan<kGridHashPow2>(s_g, g2, inputs, b_lo, b_hi, tile, size, scale, resolution);
    } else {
        grid_backward_tile_scan<kGridGeneric>(s_g, g2, inputs, b_lo, b_hi, tile, size, scale, resolution);
    }
    __syncthreads();
    const uint32_t n_here = size - tile * kTileEntries < kTileEntries ? size - tile * kTileEntries : kTileEntries;
    float *dst = grad_grid + ((size_t)off0 + (size_t)tile * kTileEntries) * 2;
    if (nsl == 1) {                                  // sole owner of the tile: plain read-modify-write
        for (uint32_t i = threadIdx.x; i < n_here * 2; i += blockDim.x) dst[i] += (float)s_g[i];
    } else {                                         // the tile is shared by the slices of its level: atomics, non-zero entries only
        for (uint32_t i = threadIdx.x; i < n_here * 2; i += blockDim.x) {
            const float v = (float)s_g[i];
            if (v != 0.0f) atomicAdd(&dst[i], v);
        }
    }
}

// gridencoder.cu:343-369
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_input_backward_kernel(
    const float *__restrict__ grad, const float *__restrict__ dy_dx,
    float *__restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *dy = dy_dx + (size_t)b * L * D * C;
    float r = 0.f;
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++)
            r += grad[((size_t)l * B + b) * C + ch] * dy[(l * D + d) * C + ch];
    }
    grad_inputs[t] = r;
}

template <uint32_t D>
int launch_forward_c(uint32_t C, const float *in, const float *emb, const int32_t *off, float *out,
                     uint32_t B, uint32_t L, const GridLevels &lv, float *dy, uint32_t gt, bool ac,
                     uint32_t interp, hipStream_t st) {
    const dim3 grid((B + 255) / 256, L), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL((grid_forward_kernel<D, 1>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 2: hipLaunchKernelGGL((grid_forward_kernel<D, 2>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 4: hipLaunchKernelGGL((grid_forward_kernel<D, 4>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 8: hipLaunchKernelGGL((grid_forward_kernel<D, 8>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return 1;
    }
    return check_launch("grid_encode_forward");
}

template <uint32_t D>
int launch_backward_c(uint32_t C, const float *grad, const float *in, const int32_t *off, float *gg,
                      uint32_t B, uint32_t L, const GridLevels &lv, const float *dy, float *gi,
                      uint32_t gt, bool ac, uint32_t interp, hipStream_t st) {
    const dim3 grid((B + 255) / 256, L), block(256);
    const dim3 grid_in((B * D + 255) / 256);
#define OCC_BWD(CC)                                                                                   \
    hipLaunchKernelGGL((grid_backward_kernel<D, CC>), grid, block, 0, st, grad, in, off, gg, B, L, lv, \
                       gt, ac, interp);                                                               \
    if (dy) hipLaunchKernelGGL((grid_input_backward_kernel<D, CC>), grid_in, block, 0, st, grad, dy, gi, B, L);
    switch (C) {
        case 1: OCC_BWD(1) break;
        case 2: OCC_BWD(2) break;
        case 4: OCC_BWD(4) break;
        case 8: OCC_BWD(8) break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return 1;
    }
#undef OCC_BWD
    return check_launch("grid_encode_backward");
}

}  // namespace occ

static int grid_forward_impl(const float *inputs, const float *embeddings, const int32_t *offsets, const int32_t *h_off,
                             float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             float *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, void *stream) {
    using namespace occ;
    if (B == 0) return 0;
    OCC_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: null tensor");
    OCC_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward: L=%u unsupported (1..%d)", L, kMaxLevels);
    const GridLevels lv = make_grid_levels(L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    if (h_off && D == 4 && C == 2 && gridtype == 0 && !ac && interp == 0 && !dy_dx && B <= (1u << 28)) {
        uint32_t sizes[kMaxLevels] = {0};
        for (uint32_t l = 0; l < L; l++) sizes[l] = (uint32_t)(h_off[l + 1] - h_off[l]);
        const GridModes4 gm = make_grid_modes_d4(L, lv, sizes);
        // level pairs dealt to the XCDs (see the kernel): measured 1.6x faster with half the fabric fetches on large batches
        // (profiles/r05_xcd_levels.md), same bits -- the default from 32 768 samples up (knob grid_xcd: 1 always, 2 never)
        const int xk = knob(kKnobGridXcd);
        if (xk == 1 || (xk == 0 && B >= 32768u)) {
            hipLaunchKernelGGL(grid_forward_d4c2_xcd_kernel, dim3(((B + 255) / 256) * 8), dim3(256), 0, st,
                               reinterpret_cast<const float4 *>(inputs), reinterpret_cast<const float2 *>(embeddings), offsets,
                               reinterpret_cast<float2 *>(outputs), B, L, lv, gm);
            return check_launch("grid_encode_forward");
        }
        const uint32_t threads = B * 8;
        hipLaunchKernelGGL(grid_forward_d4c2_kernel, dim3((threads + 255) / 256), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(inputs), reinterpret_cast<const float2 *>(embeddings), offsets,
                           reinterpret_cast<float2 *>(outputs), B, L, lv, gm);
        return check_launch("grid_encode_forward");
    }
    switch (D) {
        case 2: return launch_forward_c<2>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        case 3: return launch_forward_c<3>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        case 4: return launch_forward_c<4>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        case 5: return launch_forward_c<5>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        default: set_error("GridEncoding: D must be 2, 3, 4, or 5."); return 1;
    }
}

OCC_API int occnerf_grid_encode_forward(const float *inputs, const float *embeddings,
                                        const int32_t *offsets, float *outputs, uint32_t B,
                                        uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        float *dy_dx, uint32_t gridtype, int align_corners,
                                        uint32_t interp, void *stream) {
    return grid_forward_impl(inputs, embeddings, offsets, nullptr, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                             interp, stream);
}

OCC_API int occnerf_grid_encode_forward_h(const float *inputs, const float *embeddings, const int32_t *offsets,
                                          const int32_t *h_offsets, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                                          uint32_t L, float S, uint32_t H, float *dy_dx, uint32_t gridtype,
                                          int align_corners, uint32_t interp, void *stream) {
    OCC_REQUIRE(h_offsets, "grid_encode_forward_h: null host offsets");
    return grid_forward_impl(inputs, embeddings, offsets, h_offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype,
                             align_corners, interp, stream);
}

static int grid_backward_impl(const float *grad, const float *inputs, const int32_t *offsets, const int32_t *h_off,
                              float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                              uint32_t H, const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                              int align_corners, uint32_t interp, void *scratch, int64_t scratch_bytes, void *stream) {
    using namespace occ;
    if (B == 0) return 0;
    OCC_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: null tensor");
    OCC_REQUIRE((dy_dx == nullptr) == (grad_inputs == nullptr),
                "grid_encode_backward: dy_dx and grad_inputs must be given together");
    OCC_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward: L=%u unsupported", L);
    const GridLevels lv = make_grid_levels(L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    if (h_off && D == 4 && C == 2 && gridtype == 0 && !ac && interp == 0 && B >= 32768) {
        // tiled, atomics-free path; the level sizes come from the caller's HOST copy of the offsets
        // Jobs = (level, tile, sample slice).  The encoder's inputs are anything but uniform -- a point projected onto the
        // body surface plus a clamped distance: 87 % of the samples have their level-0 base corner in ONE of its 11 tiles,
        // and on every hashed level some tile holds a few hot cells -- so per-job times measured with wall_clock64 ranged
        // from 0.1 ms to 3.5 ms on a hashed level and 13.5 ms on level 0, and the kernel lasted as long as its hottest
        // tile (whose lanes also serialise on same-address LDS atomics).  Tiles are therefore split over sample slices:
        // 16 per tile on dense levels, 8 on hashed ones (32 / 16 measured slower: more partial tiles to merge).  A slice
        // scans only its share of the tile-set masks, so the total scan work is unchanged; partial tiles meet in the
        // table through fp32 atomics on their non-zero entries, which are few because the hot cells are few.
        TileJobs jobs;
        uint32_t sizes[kMaxLevels] = {0};
        for (uint32_t l = 0; l < L; l++) sizes[l] = (uint32_t)(h_off[l + 1] - h_off[l]);
        jobs.modes = make_grid_modes_d4(L, lv, sizes);
        uint32_t total = 0;
        for (uint32_t l = 0; l < (uint32_t)kMaxLevels; l++) {
            jobs.first_block[l] = total;
            jobs.nslices[l] = 1;
            if (l >= L) continue;
            const uint32_t nt = (sizes[l] + kTileEntries - 1) / kTileEntries;
            jobs.nslices[l] = jobs.modes.mode[l] == kGridDense ? 16u : 8u;
            total += nt * jobs.nslices[l];
        }
        jobs.first_block[kMaxLevels] = total;
        const bool fits = total > 0 && total < 65536;
        if (fits) {
            // with L * B * 8 bytes of scratch from the caller: tile-set pre-pass + masked, compacted scan
            bool max64 = true;
            for (uint32_t l = 0; l < L; l++) max64 = max64 && (sizes[l] + kTileEntries - 1) / kTileEntries <= 64;
            unsigned long long *masks = nullptr;
            if (scratch && max64 && scratch_bytes >= (int64_t)L * B * 8) {
                masks = reinterpret_cast<unsigned long long *>(scratch);
                hipLaunchKernelGGL(grid_tile_mask_kernel, dim3((B + 255) / 256, L), dim3(256), 0, st, grad,
                                   reinterpret_cast<const float4 *>(inputs), offsets, B, lv, jobs.modes, masks);
            }
            hipLaunchKernelGGL(grid_backward_tiled_d4c2_kernel, dim3(total), dim3(1024), 0, st, grad,
                               reinterpret_cast<const float4 *>(inputs), offsets, grad_embeddings, B, lv, jobs, masks);
            if (dy_dx)
                hipLaunchKernelGGL((grid_input_backward_kernel<4, 2>), dim3((B * 4 + 255) / 256), dim3(256), 0, st, grad,
                                   dy_dx, grad_inputs, B, L);
            return check_launch("grid_encode_backward");
        }
    }
    switch (D) {
        case 2: return launch_backward_c<2>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 3: return launch_backward_c<3>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 4: return launch_backward_c<4>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 5: return launch_backward_c<5>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        default: set_error("GridEncoding: D must be 2, 3, 4, or 5."); return 1;
    }
}

namespace occ {

// grad_rows[B][L*C] (what autograd hands grid.py:69-90's backward) -> grad[L][B][C] (what the operator takes), with RUNS
// merged on the way: consecutive samples whose encoder inputs are BITWISE identical -- wherever a sample's motion-weight sum
// is far below the warp's 1e-4 clamp its canonical position collapses onto one point, and with it the encoder input
// (occnerf_mlp.py:144-167) -- touch the same 16 corners of every level with the same weights, so their gradient rows are
// summed (a wave walks 64 samples: suffix sums inside a run by shuffles) into the run's first sample and the others get
// exact zeros, which the tiled backward skips.  It had lasted as long as its hot cells' LDS atomics; this replaces the
// transposing copy torch did in the same place (0.23 ms).  Same sum, fewer additions.
__global__ __launch_bounds__(256) void grid_grad_runs_kernel(const float *__restrict__ rows, const float *__restrict__ inputs,
                                                             int64_t B, int D, int L, int C, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t c0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
    if (c0 >= B) return;
    const int64_t n = c0 + lane;
    const bool live = n < B;
    // is this sample's input the bit pattern of its predecessor's?  (lane 0 always starts a run)
    bool same = live && lane > 0;
    for (int d = 0; d < D && same; d++)
        same = __float_as_uint(inputs[n * D + d]) == __float_as_uint(inputs[(n - 1) * D + d]);
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(live && !same);
    // run id = number of heads at or below this lane; a run is a contiguous lane range
    const int rid = __builtin_popcountll(heads & ((2ull << lane) - 1ull));
    const int LC = L * C;
    for (int l = 0; l < L; l++) {
        for (int c = 0; c < C; c++) {
            float v = live ? rows[n * LC + l * C + c] : 0.0f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_down(v, off);
                const int orid = __shfl_down(rid, off);
                if (lane + off < 64 && orid == rid) v += o;
            }
            if (live) out[((int64_t)l * B + n) * C + c] = same ? 0.0f : v;
        }
    }
}


// The same for L*C = 32 floats per row (the encoder of the path: 16 levels x 2 channels), at the speed of a copy.  The
// general kernel above reads rows[n][j] one float per lane at a 128-byte stride, 32 times: with 32 waves per CU the lines
// fall out of L1 AND the XCD's L2 between two visits (rocprofv3 --pmc, profiles/r05_train_hbm_pmc.json: 1.06 GB fetched
// for a 101 MB operand, 0.35 ms).  Here a wave brings its 64 rows in as ONE contiguous 8 KiB piece (16 bytes per lane,
// 8 loads) into an LDS tile, lane (half, column) walks 32 rows of its column backwards keeping the running sum of the current
// run -- a head row takes the sum, every other row becomes zero; the run that straddles the two halves hands its lower
// part's sum to its head in the upper half -- and the tile leaves level by level, 64 rows x 8 bytes = 512 contiguous bytes
// per store.
constexpr int kRunPitch = 36;                                      // floats per tile row (16-byte aligned, banks spread)
__global__ __launch_bounds__(256) void grid_grad_runs32_kernel(const float *__restrict__ rows, const float *__restrict__ inputs,
                                                               int64_t B, int D, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float tiles[4][64 * kRunPitch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = tiles[wave];
    const int64_t c0 = ((int64_t)blockIdx.x * 4 + wave) * 64;       // (whole workgroup stays for the barriers)
    const int64_t n = c0 + lane;
    const bool live = n < B;
    bool same = live && lane > 0;
    for (int d = 0; d < D && same; d++)
        same = __float_as_uint(inputs[n * D + d]) == __float_as_uint(inputs[(n - 1) * D + d]);
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(live && !same);
    // 64 rows x 128 bytes, contiguous in memory: float4 piece i * 64 + lane of the chunk
    const float4 *src = reinterpret_cast<const float4 *>(rows + c0 * 32);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int piece = i * 64 + lane, r = piece >> 3, q = piece & 7;
        const float4 v = (c0 + r < B) ? src[piece] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(tile + r * kRunPitch + q * 4) = v;
    }
    __syncthreads();
    const int half = lane >> 5, col = lane & 31;
    const unsigned hb = half ? (unsigned)(heads >> 32) : (unsigned)heads;
    float acc = 0.0f;
    bool fresh = true;                                              // no row of the current run seen yet
    for (int r = 31; r >= 0; r--) {
        float *cell = tile + (half * 32 + r) * kRunPitch + col;
        const float v = *cell;
        acc = fresh ? v : acc + v;
        const bool head = (hb >> r) & 1u;
        *cell = head ? acc : 0.0f;
        fresh = head;
    }
    // the upper half's leftover (rows 32.. of a run whose head sits in the lower half) goes to that head
    const float carry = __shfl(fresh ? 0.0f : acc, lane + 32);
    const bool open = !((heads >> 32) & 1ull) && c0 + 32 < B;       // row 32 continues the lower half's last run
    if (half == 0 && open) {
        const int p = 31 - __builtin_clz((unsigned)heads);          // (lane 0 always heads a run: never empty)
        tile[p * kRunPitch + col] += carry;
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int l = 0; l < 16; l++)
            *reinterpret_cast<float2 *>(out + ((int64_t)l * B + n) * 2) =
                *reinterpret_cast<const float2 *>(tile + lane * kRunPitch + 2 * l);
    }
}

}  // namespace occ

/* grad_rows[B][L*C] -> grad[L][B][C] for occnerf_grid_encode_backward, the rows of runs of bitwise identical inputs summed
 * into the run's first sample (zeros elsewhere).  Only valid when no input gradient is wanted (dy_dx == NULL): a per-sample
 * input gradient needs the per-sample rows. */
OCC_API int occnerf_grid_grad_runs(const float *grad_rows, const float *inputs, int64_t B, uint32_t D, uint32_t L, uint32_t C,
                                   float *grad, void *stream) {
    using namespace occ;
    if (B <= 0) return 0;
    OCC_REQUIRE(grad_rows && inputs && grad, "grid_grad_runs: null argument");
    OCC_REQUIRE(D >= 1 && L >= 1 && C >= 1 && (B + 63) / 64 / 4 + 1 < (1ll << 31), "grid_grad_runs: bad size");
    const int64_t waves = (B + 63) / 64;
    if (L == 16 && C == 2 && (reinterpret_cast<uintptr_t>(grad_rows) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad) & 7) == 0)
        hipLaunchKernelGGL(grid_grad_runs32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), grad_rows,
                           inputs, B, (int)D, grad);
    else
        hipLaunchKernelGGL(grid_grad_runs_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), grad_rows,
                           inputs, B, (int)D, (int)L, (int)C, grad);
    return check_launch("grid_grad_runs");
}

OCC_API int occnerf_grid_encode_backward(const float *grad, const float *inputs, const float *embeddings,
                                         const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                                         uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                                         float *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                                         void *stream) {
    (void)embeddings;
    return grid_backward_impl(grad, inputs, offsets, nullptr, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                              gridtype, align_corners, interp, nullptr, 0, stream);
}

OCC_API int occnerf_grid_encode_backward_h(const float *grad, const float *inputs, const float *embeddings,
                                           const int32_t *offsets, const int32_t *h_offsets, float *grad_embeddings,
                                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                           const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                                           int align_corners, uint32_t interp, void *scratch, int64_t scratch_bytes,
                                           void *stream) {
    (void)embeddings;
    OCC_REQUIRE(h_offsets, "grid_encode_backward_h: null host offsets");
    return grid_backward_impl(grad, inputs, offsets, h_offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                              gridtype, align_corners, interp, scratch, scratch_bytes, stream);
}

OCC_API int occnerf_grad_total_variation(const float *, const float *, float *, const int32_t *,
                                         float, uint32_t, uint32_t, uint32_t, uint32_t, float,
                                         uint32_t, uint32_t, int, void *) {
    occ::set_error("grad_total_variation: not implemented (never called by the reference trainer; "
                   "SURVEY.md section 8 row a20)");
    return 3;
}
