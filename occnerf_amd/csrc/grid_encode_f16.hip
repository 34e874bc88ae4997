// Half-precision dispatch case of the reference's grid encoder operator
// (core/nets/occnerf/gridencoder/src/gridencoder.cu:467,500 AT_DISPATCH_FLOATING_TYPES_AND_HALF): what the reference's
// Python feeds it whenever autocast is on (grid.py:44-45 casts the embeddings to torch.half for even C), so a
// `_gridencoder` swap must accept it.  The rendering path of this build never uses it (it computes in fp32).
//
// scalar_t = at::Half in the reference's templates means (c10/util/Half-inl.h: every Half operator computes in float
// and rounds the result to half; `Half += float` converts the float operand to Half first):
//   forward  (gridencoder.cu:166-197): inputs and the cell position / corner weights stay float;
//            results = half(float(results) + float(half(w * float(grid))))                 once per corner, in corner order
//   dy_dx    (:201-244): diff = half(float(grid_r) - float(grid_l));
//            rg = half(float(rg) + float(half((w * float(diff)) * pos_deriv)))
//   backward (:305-339): v = half(w * float(grad)) per channel, channel pairs added with one packed-half atomic
//            (`atomicAdd((__half2*)...)`; here global_atomic_pk_add_f16) -- C must be even, as grid.py guarantees
//   input backward (:343-369): result = half(float(result) + float(half(float(grad) * float(dy_dx))))
// Bound: L2/HBM gathers of 2^D corners x C x 2 B per (sample, level); thread per (sample, level) like the general fp32 kernel.
#include "common.h"

namespace occ {
namespace f16 {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// c10::Half(float): the float VALUE is rounded to half (__float2half_rn).  Left to itself hipcc folds
// `(half)(a * b)` into v_fma_mixlo_f16, which rounds the exact product once (and returns +0 for w * -0): measured on gfx950,
// 1 382 of 2^24 random products differ from multiply-in-float-then-convert.  The empty asm pins the float value first.
__device__ __forceinline__ half_t to_half(float v) {
    asm volatile("" : "+v"(v));
    return (half_t)v;
}
__device__ __forceinline__ half_t hadd(half_t a, half_t b) { return to_half((float)a + (float)b); }   // c10 Half + Half

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_forward_f16_kernel(
    const float *__restrict__ inputs, const half_t *__restrict__ embeddings, const int32_t *__restrict__ offsets,
    half_t *__restrict__ outputs, uint32_t B, uint32_t L, GridLevels lv, half_t *__restrict__ dy_dx, uint32_t gridtype,
    bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const half_t *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
    const float *x = inputs + (size_t)b * D;
    half_t *out = outputs + ((size_t)level * B + b) * C;
    half_t *dyl = dy_dx ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;

    float xin[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        xin[d] = x[d];
        oob |= (xin[d] < 0.f || xin[d] > 1.f);
    }
    if (oob) {  // gridencoder.cu:118-135
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = (half_t)0.f;
        if (dyl) {
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) dyl[i] = (half_t)0.f;
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
            pos[d] = __fmul_rn(__fmul_rn(pos[d], pos[d]), __fsub_rn(3.0f, __fmul_rn(2.0f, pos[d])));
        } else {
            pos_deriv[d] = 1.0f;
        }
    }
    half_t results[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) results[ch] = (half_t)0.f;
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
        for (uint32_t ch = 0; ch < C; ch++)
            results[ch] = hadd(results[ch], to_half(__fmul_rn(w, (float)grid[index + ch])));
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];

    if (dyl) {
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            half_t rg[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = (half_t)0.f;
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
                for (uint32_t ch = 0; ch < C; ch++) {
                    const half_t diff = to_half((float)grid[ir + ch] - (float)grid[il + ch]);
                    rg[ch] = hadd(rg[ch], to_half(__fmul_rn(__fmul_rn(w, (float)diff), pos_deriv[gd])));
                }
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) dyl[gd * C + ch] = rg[ch];
        }
    }
}

// gridencoder.cu:248-340 with scalar_t = at::Half, N_C = 2: one thread per (sample, level, channel pair).
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_backward_f16_kernel(
    const half_t *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
    half_t *__restrict__ grad_grid, uint32_t B, uint32_t L, GridLevels lv, uint32_t gridtype, bool align_corners,
    uint32_t interp) {
    static_assert(C % 2 == 0, "packed-half atomics need an even channel count");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / (C / 2), ch = (t - b * (C / 2)) * 2;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    half_t *gg = grad_grid + (size_t)(uint32_t)offsets[level] * C;
    const float *x = inputs + (size_t)b * D;
    const half_t *g = grad + ((size_t)level * B + b) * C + ch;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float xd = x[d];
        if (xd < 0.f || xd > 1.f) return;
        pos[d] = __fmaf_rn(xd, scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= fl;
        if (interp == 1)
            pos[d] = __fmul_rn(__fmul_rn(pos[d], pos[d]), __fsub_rn(3.0f, __fmul_rn(2.0f, pos[d])));
    }
    const float g0 = (float)g[0], g1 = (float)g[1];
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
        const half2_t v = {to_half(__fmul_rn(w, g0)), to_half(__fmul_rn(w, g1))};
        __builtin_amdgcn_global_atomic_fadd_v2f16(
            reinterpret_cast<__attribute__((address_space(1))) half2_t *>(reinterpret_cast<uintptr_t>(gg + index + ch)), v);
    }
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_input_backward_f16_kernel(const half_t *__restrict__ grad,
                                                                      const half_t *__restrict__ dy_dx,
                                                                      half_t *__restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const half_t *dy = dy_dx + (size_t)b * L * D * C;
    half_t r = (half_t)0.f;
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++)
            r = hadd(r, to_half(__fmul_rn((float)grad[((size_t)l * B + b) * C + ch], (float)dy[(l * D + d) * C + ch])));
    }
    grad_inputs[t] = r;
}

template <uint32_t D>
int launch_forward(uint32_t C, const float *in, const half_t *emb, const int32_t *off, half_t *out, uint32_t B, uint32_t L,
                   const GridLevels &lv, half_t *dy, uint32_t gt, bool ac, uint32_t interp, hipStream_t st) {
    const dim3 grid((B + 255) / 256, L), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL((grid_forward_f16_kernel<D, 1>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 2: hipLaunchKernelGGL((grid_forward_f16_kernel<D, 2>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 4: hipLaunchKernelGGL((grid_forward_f16_kernel<D, 4>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        case 8: hipLaunchKernelGGL((grid_forward_f16_kernel<D, 8>), grid, block, 0, st, in, emb, off, out, B, L, lv, dy, gt, ac, interp); break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return 1;
    }
    return check_launch("grid_encode_forward_f16");
}

template <uint32_t D>
int launch_backward(uint32_t C, const half_t *grad, const float *in, const int32_t *off, half_t *gg, uint32_t B, uint32_t L,
                    const GridLevels &lv, const half_t *dy, half_t *gi, uint32_t gt, bool ac, uint32_t interp, hipStream_t st) {
    const dim3 block(256), grid_in((B * D + 255) / 256);
#define OCC_BWD16(CC)                                                                                                       \
    hipLaunchKernelGGL((grid_backward_f16_kernel<D, CC>), dim3((B * (CC / 2) + 255) / 256, L), block, 0, st, grad, in, off, gg, \
                       B, L, lv, gt, ac, interp);                                                                           \
    if (dy) hipLaunchKernelGGL((grid_input_backward_f16_kernel<D, CC>), grid_in, block, 0, st, grad, dy, gi, B, L);
    switch (C) {
        case 2: OCC_BWD16(2) break;
        case 4: OCC_BWD16(4) break;
        case 8: OCC_BWD16(8) break;
        case 1:
            set_error("grid_encode_backward_f16: C = 1 has no packed-half atomic (the reference's at::Half atomicAdd is an "
                      "empty stub, gridencoder.cu:22-26; grid.py:44 keeps float embeddings when C is odd)");
            return 1;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return 1;
    }
#undef OCC_BWD16
    return check_launch("grid_encode_backward_f16");
}

}  // namespace f16
}  // namespace occ

OCC_API int occnerf_grid_encode_forward_f16(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs,
                                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx,
                                            uint32_t gridtype, int align_corners, uint32_t interp, void *stream) {
    using namespace occ;
    if (B == 0) return 0;
    OCC_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward_f16: null tensor");
    OCC_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward_f16: L=%u unsupported (1..%d)", L, kMaxLevels);
    const GridLevels lv = make_grid_levels(L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    const f16::half_t *emb = static_cast<const f16::half_t *>(embeddings);
    f16::half_t *out = static_cast<f16::half_t *>(outputs), *dy = static_cast<f16::half_t *>(dy_dx);
    switch (D) {
        case 2: return f16::launch_forward<2>(C, inputs, emb, offsets, out, B, L, lv, dy, gridtype, ac, interp, st);
        case 3: return f16::launch_forward<3>(C, inputs, emb, offsets, out, B, L, lv, dy, gridtype, ac, interp, st);
        case 4: return f16::launch_forward<4>(C, inputs, emb, offsets, out, B, L, lv, dy, gridtype, ac, interp, st);
        case 5: return f16::launch_forward<5>(C, inputs, emb, offsets, out, B, L, lv, dy, gridtype, ac, interp, st);
        default: set_error("GridEncoding: D must be 2, 3, 4, or 5."); return 1;
    }
}

OCC_API int occnerf_grid_encode_backward_f16(const void *grad, const float *inputs, const void *embeddings,
                                             const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                             uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                                             uint32_t gridtype, int align_corners, uint32_t interp, void *stream) {
    using namespace occ;
    (void)embeddings;
    if (B == 0) return 0;
    OCC_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward_f16: null tensor");
    OCC_REQUIRE((dy_dx == nullptr) == (grad_inputs == nullptr),
                "grid_encode_backward_f16: dy_dx and grad_inputs must be given together");
    OCC_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward_f16: L=%u unsupported", L);
    const GridLevels lv = make_grid_levels(L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    const f16::half_t *g = static_cast<const f16::half_t *>(grad), *dy = static_cast<const f16::half_t *>(dy_dx);
    f16::half_t *gg = static_cast<f16::half_t *>(grad_embeddings), *gi = static_cast<f16::half_t *>(grad_inputs);
    switch (D) {
        case 2: return f16::launch_backward<2>(C, g, inputs, offsets, gg, B, L, lv, dy, gi, gridtype, ac, interp, st);
        case 3: return f16::launch_backward<3>(C, g, inputs, offsets, gg, B, L, lv, dy, gi, gridtype, ac, interp, st);
        case 4: return f16::launch_backward<4>(C, g, inputs, offsets, gg, B, L, lv, dy, gi, gridtype, ac, interp, st);
        case 5: return f16::launch_backward<5>(C, g, inputs, offsets, gg, B, L, lv, dy, gi, gridtype, ac, interp, st);
        default: set_error("GridEncoding: D must be 2, 3, 4, or 5."); return 1;
    }
}
