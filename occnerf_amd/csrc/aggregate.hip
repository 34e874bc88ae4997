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

__global__ __launch_bounds__(256) void agg_backward_kernel(const float *__restrict__ grad_agg, int F,
                                                           const int32_t *__restrict__ knn,
                                                           const float *__restrict__ atts, int64_t N, int K,
                                                           float *__restrict__ grad_feats) {
    const int lane = threadIdx.x & 63;
    const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const int32_t *id = knn + n * K;
    const float *w = atts + n * K;
    const float g = lane < F ? grad_agg[n * F + lane] : 0.0f;
    for (int j0 = 0; j0 < K; j0 += 64) {
        const int jj = j0 + lane;
        const int my_id = jj < K ? id[jj] : 0;
        const float my_w = jj < K ? w[jj] : 0.0f;
        const int cnt = K - j0 < 64 ? K - j0 : 64;
        for (int j = 0; j < cnt; j++) {
            const int p = __shfl(my_id, j);
            const float wj = __shfl(my_w, j);
            if (lane < F) atomicAdd(grad_feats + (size_t)p * F + lane, __fmul_rn(wj, g));
        }
    }
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

OCC_API int occnerf_agg_backward(const float *grad_agg, int32_t F, const int32_t *knn, const float *atts, int64_t N,
                                 int32_t K, float *grad_feats, void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(grad_agg && knn && atts && grad_feats, "agg_backward: null argument");
    OCC_REQUIRE(F >= 1 && F <= 64 && K >= 1, "agg_backward: F=%d (1..64), K=%d", F, K);
    const int64_t blocks = (N + 3) / 4;
    OCC_REQUIRE(blocks < (1ll << 31), "agg_backward: N too large");
    hipLaunchKernelGGL(agg_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), grad_agg, F, knn,
                       atts, N, K, grad_feats);
    return check_launch("agg_backward");
}
