// Per-ray alpha compositing (network.py:320-348).
//
// One wavefront per ray.  Lane l owns samples l, l+64, l+128, ...; the exclusive
// transmittance product prod_{j<s}(1 - alpha_j + 1e-10) is a wave64 scan (6 shuffle steps
// per 64-sample chunk, the running product carried across chunks), the three weighted sums
// are wave reductions, and term_point (argmax alpha, first maximum) is a (value, index)
// reduction.  The scan re-associates the fp32 product that torch.cumprod forms
// sequentially; the difference is a few ulp of the transmittance.
//
// Bound: HBM streaming.  Algorithmic bytes per sample: 20 B raw + 4 B mask + 4 B z read;
// per ray 32 B ray record read, 20 B written (+ 4 B/sample when weights are requested).
#include "common.h"

namespace occ {

__device__ __forceinline__ float softplus_t20(float x) {       // F.softplus(beta=1, threshold=20)
    return x > 20.0f ? x : log1pf(expf(x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct CompositeParams {
    float bg[3];
};

__global__ __launch_bounds__(256) void composite_kernel(
    const float *__restrict__ raw, const float *__restrict__ mask, const float *__restrict__ z_vals,
    const float *__restrict__ rays, CompositeParams prm, int64_t n, int S, float *__restrict__ rgb_map,
    float *__restrict__ acc_map, float *__restrict__ depth_map, float *__restrict__ weights,
    int32_t *__restrict__ term, const int64_t *__restrict__ out_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        const float *ry = rays + r * 8;
        const float dn = norm3(ry[3], ry[4], ry[5]);                 // |rays_d|, :328
        float carry = 1.0f;                                          // product of earlier chunks
        float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
        float best_a = -INFINITY;
        int best_s = 0;
        for (int s0 = 0; s0 < S; s0 += kWave) {
            const int s = s0 + lane;
            const bool live = s < S;
            const int64_t i = r * S + (live ? s : S - 1);
            const float z = z_vals[i];
            float alpha = 0.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (live) {
                const float zn = s + 1 < S ? z_vals[i + 1] : 0.0f;
                float dist = s + 1 < S ? __fsub_rn(zn, z) : 1e10f;   // :323-327
                dist = __fmul_rn(dist, dn);
                const float *rw = raw + i * 5;
                const float sp = softplus_t20(rw[3]);
                alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sp, dist)));  // :322
                alpha = __fmul_rn(alpha, mask[i]);                    // :332
                c0 = 1.0f / (1.0f + expf(-rw[0]));                    // sigmoid, :330
                c1 = 1.0f / (1.0f + expf(-rw[1]));
                c2 = 1.0f / (1.0f + expf(-rw[2]));
            }
            // inclusive product scan of t = 1 - alpha + 1e-10 (dead lanes contribute 1)
            const float t = live ? __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0f;
            float incl = t;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const float up = __shfl_up(incl, o);
                if (lane >= o) incl *= up;
            }
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.0f;
            const float T = carry * excl;
            carry *= __shfl(incl, kWave - 1);
            const float w = live ? alpha * T : 0.0f;                  // :334-338
            if (weights && live) weights[i] = w;
            sr += w * c0;
            sg += w * c1;
            sb += w * c2;
            sd += w * z;
            sa += w;
            if (live && alpha > best_a) {                             // per-lane s ascends
                best_a = alpha;
                best_s = s;
            }
        }
        sr = wave_sum(sr);
        sg = wave_sum(sg);
        sb = wave_sum(sb);
        sd = wave_sum(sd);
        sa = wave_sum(sa);
        if (term) {                                                   // first maximum, :340
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float oa = __shfl_xor(best_a, o);
                const int os = __shfl_xor(best_s, o);
                if (oa > best_a || (oa == best_a && os < best_s)) {
                    best_a = oa;
                    best_s = os;
                }
            }
        }
        if (lane == 0) {
            const float rem = 1.0f - sa;                              // :346
            const int64_t o = out_rows ? out_rows[r] : r;             // ray r of the (Morton-ordered) batch is the caller's ray o
            rgb_map[o * 3 + 0] = sr + rem * prm.bg[0] / 255.0f;
            rgb_map[o * 3 + 1] = sg + rem * prm.bg[1] / 255.0f;
            rgb_map[o * 3 + 2] = sb + rem * prm.bg[2] / 255.0f;
            acc_map[o] = sa;
            depth_map[o] = sd;
            if (term) term[o] = best_s;
        }
    }
}

}  // namespace occ

OCC_API int occnerf_composite(const float *raw, const float *mask, const float *z_vals,
                              const float *rays, const float *h_bgcolor, int64_t n, int32_t S,
                              float *rgb, float *acc, float *depth, float *weights, int32_t *term,
                              const int64_t *out_rows, void *stream) {
    using namespace occ;
    if (n <= 0) return 0;
    OCC_REQUIRE(raw && mask && z_vals && rays && h_bgcolor && rgb && acc && depth, "composite: null argument");
    OCC_REQUIRE(S >= 1, "composite: S=%d", S);
    if (n <= 0) return 0;
    CompositeParams prm;
    for (int c = 0; c < 3; c++) prm.bg[c] = h_bgcolor[c];
    int64_t blocks = (n + 3) / 4;                   // 4 waves (rays) per 256-thread block
    if (blocks > (int64_t)kNumCU * 32) blocks = (int64_t)kNumCU * 32;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), raw, mask,
                       z_vals, rays, prm, n, S, rgb, acc, depth, weights, term, out_rows);
    return check_launch("composite");
}
