// Double-precision dispatch case of the reference's grid encoder operator
// (core/nets/occnerf/gridencoder/src/gridencoder.cu:467,500 AT_DISPATCH_FLOATING_TYPES_AND_HALF with scalar_t = double): the
// third and last type the reference's `_gridencoder` accepts.  Nothing on the rendering path reaches it (grid.py keeps float32
// embeddings, half under autocast); it exists so that the operator seam has no hole.
//
// What scalar_t = double changes in the reference's templates: `inputs` stay float (data_ptr<float>(), gridencoder.cu:470) and so
// do the cell position, the corner weights w and pos_deriv (float arithmetic, :141-159,166-180); the embeddings, the outputs, dy_dx
// and the gradients are double, and every product with them is formed in double (usual arithmetic conversions), nvcc contracting
// `r += a * b` into one fma:
//   forward  (:166-191)  results = fma((double)w, grid, results)                              per corner, in corner order
//   dy_dx    (:201-244)  rg = fma(((double)w * (grid_r - grid_l)), (double)pos_deriv, rg)
//   backward (:305-339)  atomicAdd(grad_grid, (double)w * grad)                               (global_atomic_add_f64)
//   input backward (:343-369)  result = fma(grad, dy_dx, result)
// Bound: L2/HBM gathers of 2^D corners x C x 8 B per (sample, level); thread per (sample, level) like the general fp32 kernel.
#include "common.h"

namespace occ {
namespace f64 {

template <uint32_t D>
__device__ __forceinline__ bool cell(const float *__restrict__ x, float scale, bool align_corners, uint32_t interp, float (&pos)[D],
                                     float (&pos_deriv)[D], uint32_t (&pg)[D]) {
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float xd = x[d];
        oob |= (xd < 0.f || xd > 1.f);
        pos[d] = __fmaf_rn(xd, scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= fl;
        if (interp == 1) {
            pos_deriv[d] = __fmul_rn(__fmul_rn(6.f, pos[d]), __fsub_rn(1.0f, pos[d]));
            pos[d] = __fmul_rn(__fmul_rn(pos[d], pos[d]), __fsub_rn(3.0f, __fmul_rn(2.0f, pos[d])));
        } else {
            pos_deriv[d] = 1.0f;
        }
    }
    return oob;
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_forward_f64_kernel(const float *__restrict__ inputs, const double *__restrict__ embeddings,
                                                               const int32_t *__restrict__ offsets, double *__restrict__ outputs,
                                                               uint32_t B, uint32_t L, GridLevels lv, double *__restrict__ dy_dx,
                                                               uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const double *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
    double *out = outputs + ((size_t)level * B + b) * C;
    double *dyl = dy_dx ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    float pos[D], pos_deriv[D];
    uint32_t pg[D];
    if (cell<D>(inputs + (size_t)b * D, scale, align_corners, interp, pos, pos_deriv, pg)) {      // :118-135: zero rows
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0.0;
        if (dyl) {
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) dyl[i] = 0.0;
        }
        return;
    }
    double results[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) results[ch] = 0.0;
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
        for (uint32_t ch = 0; ch < C; ch++) results[ch] = __fma_rn((double)w, grid[index + ch], results[ch]);
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];
    if (dyl) {
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            double rg[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0.0;
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
                    rg[ch] = __fma_rn(__dmul_rn((double)w, __dsub_rn(grid[ir + ch], grid[il + ch])), (double)pos_deriv[gd], rg[ch]);
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) dyl[gd * C + ch] = rg[ch];
        }
    }
}

// gridencoder.cu:248-340 with scalar_t = double: one thread per (sample, level), all C channels (N_C = C is what the reference's
// wrapper picks for C <= 2; for C = 4 / 8 it splits the channels over threads, which changes nothing but the atomics' order).
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_backward_f64_kernel(const double *__restrict__ grad, const float *__restrict__ inputs,
                                                                const int32_t *__restrict__ offsets, double *__restrict__ grad_grid,
                                                                uint32_t B, uint32_t L, GridLevels lv, uint32_t gridtype,
                                                                bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    double *gg = grad_grid + (size_t)(uint32_t)offsets[level] * C;
    const double *g = grad + ((size_t)level * B + b) * C;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    float pos[D], pos_deriv[D];
    uint32_t pg[D];
    if (cell<D>(inputs + (size_t)b * D, lv.scale[level], align_corners, interp, pos, pos_deriv, pg)) return;   // grad is zero-initialised
    double gc[C];
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
        const uint32_t index = grid_index<D>(gridtype, align_corners, hashmap_size, lv.resolution[level], pl) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(gg + index + ch, __dmul_rn((double)w, gc[ch]));
    }
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_input_backward_f64_kernel(const double *__restrict__ grad, const double *__restrict__ dy_dx,
                                                                      double *__restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const double *dy = dy_dx + (size_t)b * L * D * C;
    double r = 0.0;
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) r = __fma_rn(grad[((size_t)l * B + b) * C + ch], dy[(l * D + d) * C + ch], r);
    }
    grad_inputs[t] = r;
}

template <uint32_t D>
int launch_forward(uint32_t C, const float *in, const double *emb, const int32_t *off, double *out, uint32_t B, uint32_t L,
                   const GridLevels &lv, double *dy, uint32_t gt, bool ac, uint32_t interp, hipStream_t st) {
    const dim3 grid((B + 255) / 256, L), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL((grid_forward_f64_kernel<D, 1>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 2: hipLaunchKernelGGL((grid_forward_f64_kernel<D, 2>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 4: hipLaunchKernelGGL((grid_forward_f64_kernel<D, 4>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 8: hipLaunchKernelGGL((grid_forward_f64_kernel<D, 8>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return 1;
    }
    return check_launch("grid_encode_forward_f64");
}

template <uint32_t D>
int launch_backward(uint32_t C, const double *grad, const float *in, const int32_t *off, double *gg, uint32_t B, uint32_t L,
                    const GridLevels &lv, const double *dy, double *gi, uint32_t gt, bool ac, uint32_t interp, hipStream_t st) {
    const dim3 block(256), grid((B + 255) / 256, L), grid_in((B * D + 255) / 256);
#define OCC_BWD64(CC)                                                                                                        \
    hipLaunchKernelGGL((grid_backward_f64_kernel<D, CC>), grid, block, 0, st, grad, in, off, gg, B, L, lv, gt, ac, interp);  \
    if (dy) hipLaunchKernelGGL((grid_input_backward_f64_kernel<D, CC>), grid_in, block, 0, st, grad, dy, gi, B, L);
    switch (C) {
        case 1: OCC_BWD64(1) break;
        case 2: OCC_BWD64(2) break;
        case 4: OCC_BWD64(4) break;
        case 8: OCC_BWD64(8) break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return 1;
    }
#undef OCC_BWD64
    return check_launch("grid_encode_backward_f64");
}

}  // namespace f64
}  // namespace occ

OCC_API int occnerf_grid_encode_forward_f64(const float *inputs, const double *embeddings, const int32_t *offsets, double *outputs,
                                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, double *dy_dx,
                                            uint32_t gridtype, int align_corners, uint32_t interp, void *stream) {
    using namespace occ;
    if (B == 0) return 0;
    OCC_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward_f64: null tensor");
    OCC_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward_f64: L=%u unsupported (1..%d)", L, kMaxLevels);
    const GridLevels lv = make_grid_levels(L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    switch (D) {
        case 2: return f64::launch_forward<2>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        case 3: return f64::launch_forward<3>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        case 4: return f64::launch_forward<4>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        case 5: return f64::launch_forward<5>(C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, st);
        default: set_error("GridEncoding: D must be 2, 3, 4, or 5."); return 1;
    }
}

OCC_API int occnerf_grid_encode_backward_f64(const double *grad, const float *inputs, const double *embeddings,
                                             const int32_t *offsets, double *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                             uint32_t L, float S, uint32_t H, const double *dy_dx, double *grad_inputs,
                                             uint32_t gridtype, int align_corners, uint32_t interp, void *stream) {
    using namespace occ;
    (void)embeddings;
    if (B == 0) return 0;
    OCC_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward_f64: null tensor");
    OCC_REQUIRE((dy_dx == nullptr) == (grad_inputs == nullptr),
                "grid_encode_backward_f64: dy_dx and grad_inputs must be given together");
    OCC_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward_f64: L=%u unsupported", L);
    const GridLevels lv = make_grid_levels(L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    switch (D) {
        case 2: return f64::launch_backward<2>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 3: return f64::launch_backward<3>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 4: return f64::launch_backward<4>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 5: return f64::launch_backward<5>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, st);
        default: set_error("GridEncoding: D must be 2, 3, 4, or 5."); return 1;
    }
}
