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

__global__ __launch_bounds__(256) void agg_forward_kernel(const float *__restrict__ feats, int F,
                                                          const int32_t *__restrict__ knn,
                                                          const float *__restrict__ atts, int64_t N, int K,
                                                          float *__restrict__ agg) {
    const int lane = threadIdx.x & 63;
    const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const int32_t *id = knn + n * K;
    const float *w = atts + n * K;
    float acc = 0.0f;
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
    if (lane < F) agg[n * F + lane] = acc;
}

// Backward.  Global fp32 atomics are memory-side operations on this part (~20 G/s even on a 1 MB table:
// 54 ms for one training batch), so the scatter is organised by OWNERSHIP instead: workgroup (w, tile) owns
// the gradient rows of kTilePoints consecutive points for the w-th slice of the samples, keeps them in LDS,
// scans its samples (one wave per sample: the 40 ids in 40 lanes, ballot of the ones in the tile, then one
// LDS atomic instruction per hit with lane c = column c) and finally stores the tile to partial[w] with plain
// writes; the caller sums the W partial tables.  No global atomics.  The LDS accumulators are fp64: ds_add_f64
// runs at 16 cycles per wave-instruction on gfx950, ds_add_f32 at 190 (tools/lds_atomic_rate.hip) -- and the
// sums come out more accurate for it.
constexpr int kAggTileValues = 18432;            // doubles: 144 KiB of LDS

__global__ __launch_bounds__(1024) void agg_backward_tiled_kernel(const float *__restrict__ grad_agg, int F,
                                                                  const int32_t *__restrict__ knn,
                                                                  const float *__restrict__ atts, int64_t N, int K,
                                                                  int P, int tile_points, int64_t samples_per_slice,
                                                                  float *__restrict__ partial /*[W][P][F]*/) {
    __shared__ double s_g[kAggTileValues];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int tile0 = blockIdx.y * tile_points;
    const int tile_n = P - tile0 < tile_points ? P - tile0 : tile_points;
    for (int i = threadIdx.x; i < tile_n * F; i += blockDim.x) s_g[i] = 0.0;
    __syncthreads();
    const int64_t n0 = (int64_t)blockIdx.x * samples_per_slice;
    const int64_t n1 = n0 + samples_per_slice < N ? n0 + samples_per_slice : N;
    // The loop is latency-bound (three dependent-free loads per sample, then a handful of LDS atomics): eight samples
    // per trip keep 24 loads in flight per wave.  K <= 64 (one id per lane) on this path; larger K falls back below.
    constexpr int U = 8;
    if (K <= 64) {
        for (int64_t nb = n0 + (int64_t)wave * U; nb < n1; nb += (int64_t)nwaves * U) {
            float g[U], my_w[U];
            int my_id[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t n = nb + u < n1 ? nb + u : n1 - 1;
                g[u] = lane < F ? grad_agg[n * F + lane] : 0.0f;
                my_id[u] = lane < K ? knn[n * K + lane] : -1;
                my_w[u] = lane < K ? atts[n * K + lane] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (nb + u >= n1) break;
                const unsigned rel = (unsigned)(my_id[u] - tile0);
                unsigned long long hits = __builtin_amdgcn_ballot_w64(lane < K && rel < (unsigned)tile_n);
                while (hits) {
                    const int j = __builtin_ctzll(hits);                       // wave-uniform: v_readlane, no LDS trip
                    hits &= hits - 1;
                    const int p = __builtin_amdgcn_readlane(my_id[u], j) - tile0;
                    const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w[u]), j));
                    if (lane < F) atomicAdd(&s_g[p * F + lane], (double)__fmul_rn(wj, g[u]));
                }
            }
        }
    } else {
        for (int64_t n = n0 + wave; n < n1; n += nwaves) {
            const float g = lane < F ? grad_agg[n * F + lane] : 0.0f;
            for (int j0 = 0; j0 < K; j0 += 64) {
                const int jj = j0 + lane;
                const int my_id = jj < K ? knn[n * K + jj] : -1;
                const float my_w = jj < K ? atts[n * K + jj] : 0.0f;
                const unsigned rel = (unsigned)(my_id - tile0);
                unsigned long long hits = __builtin_amdgcn_ballot_w64(jj < K && rel < (unsigned)tile_n);
                while (hits) {
                    const int j = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    const int p = __builtin_amdgcn_readlane(my_id, j) - tile0;
                    const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
                    if (lane < F) atomicAdd(&s_g[p * F + lane], (double)__fmul_rn(wj, g));
                }
            }
        }
    }
    __syncthreads();
    float *dst = partial + ((size_t)blockIdx.x * P + tile0) * F;
    for (int i = threadIdx.x; i < tile_n * F; i += blockDim.x) dst[i] = (float)s_g[i];
}

}  // namespace occ

OCC_API int occnerf_agg_forward(const float *feats, int32_t F, const int32_t *knn, const float *atts, int64_t N,
                                int32_t K, float *agg, void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(feats && knn && atts && agg, "agg_forward: null argument");
    OCC_REQUIRE(F >= 1 && F <= 64 && K >= 1, "agg_forward: F=%d (1..64), K=%d", F, K);
    const int64_t blocks = (N + 3) / 4;
    OCC_REQUIRE(blocks < (1ll << 31), "agg_forward: N too large");
    hipLaunchKernelGGL(agg_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), feats, F, knn, atts,
                       N, K, agg);
    return check_launch("agg_forward");
}

OCC_API int32_t occnerf_agg_backward_slices(int64_t N) {
    // Sample slices W (x 14 point tiles = workgroups).  Per-job times measured with wall_clock64 at 786 K samples and 24
    // slices: ~1.0 ms for most (point tile, slice) jobs but 3.8 ms for every slice of ONE tile (the points most samples are
    // near), which set the kernel's 4.7 ms.  48 slices halve every job, the hot ones then spread over two rounds of the 256
    // CUs: 2.9 ms.  (18 slices = a single round was slower, 5.4 ms.)
    int64_t w = (N + 16383) / 16384;
    return (int32_t)(w < 1 ? 1 : (w > 48 ? 48 : w));
}

OCC_API int occnerf_agg_backward(const float *grad_agg, int32_t F, const int32_t *knn, const float *atts, int64_t N,
                                 int32_t K, int32_t P, float *partial, void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(grad_agg && knn && atts && partial, "agg_backward: null argument");
    OCC_REQUIRE(F >= 1 && F <= 64 && K >= 1 && P >= 1, "agg_backward: F=%d (1..64), K=%d, P=%d", F, K, P);
    const int W = occnerf_agg_backward_slices(N);
    const int tile_points = kAggTileValues / F < 1024 ? kAggTileValues / F : 1024;
    const int tiles = (P + tile_points - 1) / tile_points;
    const int64_t per_slice = (N + W - 1) / W;
    hipLaunchKernelGGL(agg_backward_tiled_kernel, dim3(W, tiles), dim3(1024), 0, as_stream(stream), grad_agg, F, knn,
                       atts, N, K, P, tile_points, per_slice, partial);
    return check_launch("agg_backward");
}
