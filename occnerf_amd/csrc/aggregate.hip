// Neighbour-feature aggregation for the differentiable (training) path, occnerf_mlp.py:86-126, 176-178:
//     agg[n, :] = sum_j atts[n, j] * feats[knn[n, j], :]                       (atts detached)
// and its gradient with respect to feats,
//     grad_feats[p, :] += sum_{(n, j): knn[n, j] = p} atts[n, j] * grad_agg[n, :].
// The reference materialises feats[knn] as [N, 40, 35] (4.4 GB for one 6 x 32 x 32-patch batch at 128 spp)
// and differentiates the advanced index -- 1.5 s of a 1.9 s training step on MI355X when left to torch's
// index_put backward.  Here: one wave per sample, lane c = feature column c (F <= 64), rows gathered /
// scattered as contiguous 4 F-byte segments; the backward uses native fp32 L2 atomics on the 1 MB table.
#include "common.h"

namespace occ {

// (round 5) A wave takes kAggFwdRun consecutive samples: a sample whose 40 ids AND 40 weights are bitwise its predecessor's -- a
// run of collapsed samples, see the backward below -- has the same sum: it is copied, not gathered again.  (Round 6: the weights
// are compared too.  On the training path they are a function of the ids, but nothing in this entry point's signature says so.)
constexpr int kAggFwdRun = 8;

__global__ __launch_bounds__(256) void agg_forward_kernel(const float *__restrict__ feats, int F,
                                                          const int32_t *__restrict__ knn,
                                                          const float *__restrict__ atts, int64_t N, int K,
                                                          float *__restrict__ agg) {
    const int lane = threadIdx.x & 63;
    const int64_t n0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kAggFwdRun;
    if (n0 >= N) return;
    const int64_t n1 = n0 + kAggFwdRun < N ? n0 + kAggFwdRun : N;
    int prev_id = -2;
    uint32_t prev_w = 0u;
    float acc = 0.0f;
    const bool pairs = F <= 36 && K <= 64;
    for (int64_t n = n0; n < n1; n++) {
        const int32_t *id = knn + n * K;
        const float *w = atts + n * K;
        bool same = K <= 64;
        if (same) {
            const int my = lane < K ? id[lane] : -1;
            const uint32_t myw = lane < K ? __float_as_uint(w[lane]) : 0u;
            same = __builtin_amdgcn_ballot_w64(my != prev_id || myw != prev_w) == 0ull;
            prev_id = my;
            prev_w = myw;
        }
        if (!same && pairs) {
            // (round 6) F <= 36, K <= 64: a row is 18 column PAIRS, so three groups of 18 lanes gather three neighbours' rows per
            // instruction (8 bytes per lane) instead of one row on 35 lanes -- the kernel is bound by the number of gather
            // instructions (>= 16 cycles each on the texture path).  Group g sums neighbours g, g + 3, ...; the three partial sums
            // meet through shuffles.  (Another summation order than j = 0 .. K-1: fp32 reassociation, inside the tolerance the
            // aggregation is held to -- torch's own reduction order is not specified either.)
            struct __attribute__((packed, aligned(4))) F2 { float v[2]; };
            const int grp = lane / 18, c2 = lane - grp * 18;
            const bool two = 2 * c2 + 1 < F, any = grp < 3 && 2 * c2 < F;
            const int my_id = lane < K ? id[lane] : 0;
            const float my_w = lane < K ? w[lane] : 0.0f;
            float s0 = 0.0f, s1 = 0.0f;
            for (int j = 0; j < K; j += 3) {
                const int jj = j + grp < K ? j + grp : K - 1;
                const int p = __shfl(my_id, jj);
                const float wv = __shfl(my_w, jj);                       // (every lane takes part: a masked-off source lane reads as 0)
                const float wj = (grp < 3 && j + grp < K) ? wv : 0.0f;
                if (any) {
                    const char *row = reinterpret_cast<const char *>(feats) + ((uint32_t)p * (uint32_t)F + 2u * (uint32_t)c2) * 4u;
                    if (two) {
                        const F2 v = *reinterpret_cast<const F2 *>(row);
                        s0 = __fadd_rn(s0, __fmul_rn(wj, v.v[0]));
                        s1 = __fadd_rn(s1, __fmul_rn(wj, v.v[1]));
                    } else {
                        s0 = __fadd_rn(s0, __fmul_rn(wj, *reinterpret_cast<const float *>(row)));
                    }
                }
            }
            const int src = lane >> 1;                                   // lane c takes column c: pair c / 2, element c & 1
            const float t0 = __fadd_rn(__fadd_rn(__shfl(s0, src), __shfl(s0, src + 18)), __shfl(s0, src + 36));
            const float t1 = __fadd_rn(__fadd_rn(__shfl(s1, src), __shfl(s1, src + 18)), __shfl(s1, src + 36));
            acc = (lane & 1) ? t1 : t0;
        } else if (!same) {
            acc = 0.0f;
            for (int j0 = 0; j0 < K; j0 += 64) {                       // ids / weights of up to 64 neighbours at once
                const int jj = j0 + lane;
                const int my_id = jj < K ? id[jj] : 0;
                const float my_w = jj < K ? w[jj] : 0.0f;
                const int cnt = K - j0 < 64 ? K - j0 : 64;
                for (int j = 0; j < cnt; j++) {
                    const int p = __shfl(my_id, j);
                    const float wj = __shfl(my_w, j);
                    if (lane < F) acc = __fadd_rn(acc, __fmul_rn(wj, ld32(feats, ((uint32_t)p * (uint32_t)F + (uint32_t)lane) * 4u)));
                }
            }
        }
        if (lane < F) agg[n * F + lane] = acc;
    }
}

// Backward.  Global fp32 atomics are memory-side operations on this part (~20 G/s even on a 1 MB table:
// 54 ms for one training batch), so the scatter is organised by OWNERSHIP instead: workgroup (w, tile) owns
// the gradient rows of `tile_points` consecutive points for the w-th slice of the samples, keeps them in LDS as fp64
// (ds_add_f64 runs at 16 cycles per wave-instruction on gfx950, ds_add_f32 at 190: tools/lds_atomic_rate.hip -- and the
// sums come out more accurate for it), and stores the tile to partial[w] with plain writes; the caller sums the W
// partial tables.  No global atomics.
//
// Round 5 -- two passes.  (Measured: with every (tile, slice) job scanning its slice's ids + weights + gradient rows,
// 460 B per sample, the kernel lasted as long as 14 x that scan -- 2.8 ms -- whatever the atomics did.)
//   1. agg_runs_kernel, once over the samples: wherever a sample's motion-weight sum is far below the warp's 1e-4 clamp
//      its canonical position collapses onto one point (network.py:388; two thirds of a frame's live samples), so
//      consecutive samples of a ray carry the SAME 40 neighbour ids in the same order -- hence the same softmax weights,
//      a function of the ids alone (occnerf_mlp.py:110-125; the kernel compares the weights' bits as well, so arbitrary
//      atts are handled exactly).  A wave walks 64 contiguous samples, sums the gradient rows
//      of each run of identical (ids, weights) lists (fp64 in registers) into the row of the run's FIRST sample, and writes per sample
//      a bit mask of the point tiles its ids touch -- zero for every sample but a run's first, and for runs whose summed
//      row is exactly zero (dead samples: alpha is multiplied by a zero mask).
//   2. agg_backward_tiled_kernel: a (tile, slice) job reads 4 bytes per sample -- the masks, coalesced -- and fetches
//      ids / weights / summed row only for the samples whose mask has its bit: a run's 40 LDS atomics are issued once.
constexpr int kAggTileValues = 18432;            // doubles: 144 KiB of LDS
constexpr int kAggRun = 64;                      // samples a wave walks for runs (half a ray at 128 samples / ray)

__global__ __launch_bounds__(256) void agg_runs_kernel(const float *__restrict__ grad_agg, int F,
                                                       const int32_t *__restrict__ knn, const float *__restrict__ atts,
                                                       int64_t N, int K, int tile_points,
                                                       int tiles, float *__restrict__ gsum /*[N][F]*/,
                                                       uint32_t *__restrict__ mask /*[N]*/) {
    const int lane = threadIdx.x & 63;
    const int64_t c0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kAggRun;
    if (c0 >= N) return;
    const int64_t c1 = c0 + kAggRun < N ? c0 + kAggRun : N;
    constexpr int U = 8;
    int cur_id = -2;                 // the open run's id list (lane j: neighbour j) ...
    uint32_t cur_w = 0u;             // ... and weight bits ...
    double acc = 0.0;                // ... the sum of its gradient rows (lane c: column c) ...
    int head = -1;                   // ... and its first sample (offset in the chunk)
    uint32_t my_mask = 0;            // lane i: mask of sample c0 + i
    auto flush = [&]() {
        const float v = (float)acc;
        if (__builtin_amdgcn_ballot_w64(lane < F && v != 0.0f) == 0ull) return;          // the run adds nothing
        if (lane < F) gsum[(c0 + head) * F + lane] = v;
        const int t = lane < K ? cur_id / tile_points : -1;
        uint32_t m = 0;
        for (int tt = 0; tt < tiles; tt++)
            if (__builtin_amdgcn_ballot_w64(t == tt) != 0ull) m |= 1u << tt;
        if (lane == head) my_mask = m;
    };
    for (int64_t nb = c0; nb < c1; nb += U) {
        float g[U];
        int my_id[U];
        uint32_t my_w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t n = nb + u < c1 ? nb + u : c1 - 1;
            g[u] = lane < F ? grad_agg[n * F + lane] : 0.0f;
            my_id[u] = lane < K ? knn[n * K + lane] : -1;
            my_w[u] = lane < K ? __float_as_uint(atts[n * K + lane]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (nb + u >= c1) break;
            if (head < 0 || __builtin_amdgcn_ballot_w64(my_id[u] != cur_id || my_w[u] != cur_w) != 0ull) {
                if (head >= 0) flush();
                cur_id = my_id[u];
                cur_w = my_w[u];
                acc = 0.0;
                head = (int)(nb + u - c0);
            }
            acc += (double)g[u];
        }
    }
    if (head >= 0) flush();
    if (c0 + lane < c1) mask[c0 + lane] = my_mask;
}

__global__ __launch_bounds__(1024) void agg_backward_tiled_kernel(const float *__restrict__ gsum, int F,
                                                                  const int32_t *__restrict__ knn,
                                                                  const float *__restrict__ atts,
                                                                  const uint32_t *__restrict__ mask, int64_t N, int K,
                                                                  int P, int tile_points, int64_t samples_per_slice,
                                                                  float *__restrict__ partial /*[W][P][F]*/) {
    __shared__ double s_g[kAggTileValues];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int tile = blockIdx.y, tile0 = tile * tile_points;
    const int tile_n = P - tile0 < tile_points ? P - tile0 : tile_points;
    for (int i = threadIdx.x; i < tile_n * F; i += blockDim.x) s_g[i] = 0.0;
    __syncthreads();
    const int64_t n0 = (int64_t)blockIdx.x * samples_per_slice;
    const int64_t n1 = n0 + samples_per_slice < N ? n0 + samples_per_slice : N;
    constexpr int U = 4;             // listed samples fetched together (12 independent loads in flight per wave)
    for (int64_t c0 = n0 + (int64_t)wave * 64; c0 < n1; c0 += (int64_t)nwaves * 64) {
        const uint32_t m = c0 + lane < n1 ? mask[c0 + lane] : 0u;
        unsigned long long todo = __builtin_amdgcn_ballot_w64((m >> tile) & 1u);
        while (todo) {
            int64_t n[U];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                n[u] = -1;
                if (todo) {
                    n[u] = c0 + __builtin_ctzll(todo);
                    todo &= todo - 1;
                    cnt = u + 1;
                }
            }
            float g[U], my_w[U];
            int my_id[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t nn = n[u] >= 0 ? n[u] : n[0];
                g[u] = lane < F ? gsum[nn * F + lane] : 0.0f;
                my_id[u] = lane < K ? knn[nn * K + lane] : -1;
                my_w[u] = lane < K ? atts[nn * K + lane] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (u >= cnt) break;
                const unsigned rel = (unsigned)(my_id[u] - tile0);
                unsigned long long hits = __builtin_amdgcn_ballot_w64(lane < K && rel < (unsigned)tile_n);
                while (hits) {
                    const int j = __builtin_ctzll(hits);                       // wave-uniform: v_readlane, no LDS trip
                    hits &= hits - 1;
                    const int p = __builtin_amdgcn_readlane(my_id[u], j) - tile0;
                    const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w[u]), j));
                    if (lane < F) atomicAdd(&s_g[p * F + lane], (double)wj * (double)g[u]);
                }
            }
        }
    }
    __syncthreads();
    float *dst = partial + ((size_t)blockIdx.x * P + tile0) * F;
    for (int i = threadIdx.x; i < tile_n * F; i += blockDim.x) dst[i] = (float)s_g[i];
}

static int agg_tile_points(int F) { return kAggTileValues / F < 1024 ? kAggTileValues / F : 1024; }

}  // namespace occ

OCC_API int occnerf_agg_forward(const float *feats, int32_t F, const int32_t *knn, const float *atts, int64_t N,
                                int32_t K, float *agg, void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(feats && knn && atts && agg, "agg_forward: null argument");
    OCC_REQUIRE(F >= 1 && F <= 64 && K >= 1, "agg_forward: F=%d (1..64), K=%d", F, K);
    const int64_t blocks = ((N + kAggFwdRun - 1) / kAggFwdRun + 3) / 4;
    OCC_REQUIRE(blocks < (1ll << 31), "agg_forward: N too large");
    hipLaunchKernelGGL(agg_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), feats, F, knn, atts,
                       N, K, agg);
    return check_launch("agg_forward");
}

OCC_API int32_t occnerf_agg_backward_slices(int64_t N) {
    // Sample slices W (x 14 point tiles = workgroups).  Measured in round 5 (tools/debug_agg.py): after the run pre-pass a job's
    // time is its LDS atomics -- ~69 cycles each when they pile onto a few rows (a slice's samples share their neighbours), and
    // the tiles differ 30x in hits (1.07 M pairs in the hottest, 33 K in the lightest) -- so the slices are made small enough
    // for the hardware's workgroup scheduler to balance the hot tile's jobs against everyone else's: up to 256 slices of
    // >= 3 072 samples (786 432 samples: 48 slices 2.15 ms, 96 1.87, 128 1.65, 192 1.44, 256 1.23, partial-table sum included).
    // The experiment knob "agg_slices" (OCCNERF_AGG_SLICES) overrides for A/B runs.
    const int forced = occ::knob(occ::kKnobAggSlices);
    if (forced >= 1) return forced;
    int64_t w = (N + 3071) / 3072;
    return (int32_t)(w < 1 ? 1 : (w > 256 ? 256 : w));
}

/* bytes of scratch occnerf_agg_backward needs: the run sums [N][F] fp32 + the tile masks [N] */
OCC_API int64_t occnerf_agg_backward_scratch_bytes(int64_t N, int32_t F) {
    if (N <= 0 || F < 1) return 0;
    return ((N * F * 4 + 255) & ~(int64_t)255) + N * 4;
}

OCC_API int occnerf_agg_backward(const float *grad_agg, int32_t F, const int32_t *knn, const float *atts, int64_t N,
                                 int32_t K, int32_t P, float *partial, void *scratch, int64_t scratch_bytes, void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(grad_agg && knn && atts && partial && scratch, "agg_backward: null argument");
    OCC_REQUIRE(F >= 1 && F <= 64 && K >= 1 && K <= 64 && P >= 1, "agg_backward: F=%d (1..64), K=%d (1..64), P=%d", F, K, P);
    OCC_REQUIRE(scratch_bytes >= occnerf_agg_backward_scratch_bytes(N, F), "agg_backward: scratch too small");
    const int W = occnerf_agg_backward_slices(N);
    const int tile_points = agg_tile_points(F);
    const int tiles = (P + tile_points - 1) / tile_points;
    OCC_REQUIRE(tiles <= 32, "agg_backward: %d point tiles of %d rows for P=%d, F=%d: at most 32 (the per-sample tile mask is 32 bits "
                "wide: P <= %d at this F)", tiles, tile_points, P, F, 32 * tile_points);
    float *gsum = reinterpret_cast<float *>(scratch);
    uint32_t *mask = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(scratch) + ((N * F * 4 + 255) & ~(int64_t)255));
    const int64_t chunks = (N + kAggRun - 1) / kAggRun;
    OCC_REQUIRE((chunks + 3) / 4 < (1ll << 31), "agg_backward: N too large");
    hipLaunchKernelGGL(agg_runs_kernel, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, as_stream(stream), grad_agg, F, knn, atts, N, K,
                       tile_points, tiles, gsum, mask);
    const int64_t per_slice = (((N + W - 1) / W) + kAggRun - 1) / kAggRun * kAggRun;      // whole chunks: runs never straddle a slice
    hipLaunchKernelGGL(agg_backward_tiled_kernel, dim3(W, tiles), dim3(1024), 0, as_stream(stream), gsum, F, knn, atts, mask,
                       N, K, P, tile_points, per_slice, partial);
    return check_launch("agg_backward");
}
