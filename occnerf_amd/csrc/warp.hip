// Ray sampling + backward warp into canonical space.
//
// One thread per sample.  For each of the 24 bones: pos = R p + T, normalise into the
// motion-weight volume, trilinear tap (F.grid_sample semantics: align_corners=True, zeros
// padding, corners accumulated in ATen's order), then the weight-blended canonical position
// (network.py:351-402).  The [24,N,3] intermediate the reference materialises (1.2 GB per
// ray chunk) never exists.
//
// Bound: L2 gather.  Algorithmic bytes per sample: up to 24 x 8 taps x 4 B = 768 B from the
// 3.1 MB volume (L2-resident) + 32 B of ray record (amortised over S) in, 16 B out
// (x_skel, mask) + 4 B z.  Most bones miss the volume entirely and tap nothing.
#include "common.h"

namespace occ {

struct WarpParams {
    float bmin[3];
    float bscale[3];
};

__device__ __forceinline__ float trilinear_zeros(const float *__restrict__ vol, int G, float gx,
                                                 float gy, float gz) {
    // ATen grid_sampler_unnormalize(align_corners=True): ((g + 1) / 2) * (size - 1)
    const float gm1 = (float)(G - 1);
    const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.0f), 2.0f), gm1);
    const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.0f), 2.0f), gm1);
    const float iz = __fmul_rn(__fdiv_rn(__fadd_rn(gz, 1.0f), 2.0f), gm1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float Gf = (float)G;
    if (!(fx >= -1.0f && fx <= Gf && fy >= -1.0f && fy <= Gf && fz >= -1.0f && fz <= Gf))
        return 0.0f;  // every corner out of bounds (also catches NaN)
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float wx1 = __fsub_rn(ix, fx), wx0 = __fsub_rn((float)x1, ix);
    const float wy1 = __fsub_rn(iy, fy), wy0 = __fsub_rn((float)y1, iy);
    const float wz1 = __fsub_rn(iz, fz), wz0 = __fsub_rn((float)z1, iz);
    const bool vx0 = x0 >= 0 && x0 < G, vx1 = x1 >= 0 && x1 < G;
    const bool vy0 = y0 >= 0 && y0 < G, vy1 = y1 >= 0 && y1 < G;
    const bool vz0 = z0 >= 0 && z0 < G, vz1 = z1 >= 0 && z1 < G;
    float out = 0.0f;
#define OCC_TAP(vz, vy, vx, z, y, x, wz, wy, wx)                                       \
    if ((vz) && (vy) && (vx))                                                          \
        out = __fadd_rn(out, __fmul_rn(vol[((z) * G + (y)) * G + (x)],                 \
                                       __fmul_rn(__fmul_rn(wx, wy), wz)));
    OCC_TAP(vz0, vy0, vx0, z0, y0, x0, wz0, wy0, wx0)
    OCC_TAP(vz0, vy0, vx1, z0, y0, x1, wz0, wy0, wx1)
    OCC_TAP(vz0, vy1, vx0, z0, y1, x0, wz0, wy1, wx0)
    OCC_TAP(vz0, vy1, vx1, z0, y1, x1, wz0, wy1, wx1)
    OCC_TAP(vz1, vy0, vx0, z1, y0, x0, wz1, wy0, wx0)
    OCC_TAP(vz1, vy0, vx1, z1, y0, x1, wz1, wy0, wx1)
    OCC_TAP(vz1, vy1, vx0, z1, y1, x0, wz1, wy1, wx0)
    OCC_TAP(vz1, vy1, vx1, z1, y1, x1, wz1, wy1, wx1)
#undef OCC_TAP
    return out;
}

constexpr int kMaxBones = 32;

__global__ __launch_bounds__(256) void sample_warp_kernel(
    const float *__restrict__ rays, int64_t n, int S, const float *__restrict__ t_vals,
    const float *__restrict__ t_rand, const float *__restrict__ Rs, const float *__restrict__ Ts,
    const float *__restrict__ vol, int nb, int G, WarpParams prm, float *__restrict__ z_vals,
    float *__restrict__ pts_out, float *__restrict__ x_skel, float *__restrict__ mask) {
    // bone transforms: 12 floats x nb, staged once per block
    __shared__ float sR[kMaxBones * 9];
    __shared__ float sT[kMaxBones * 3];
    for (int i = threadIdx.x; i < nb * 9; i += blockDim.x) sR[i] = Rs[i];
    for (int i = threadIdx.x; i < nb * 3; i += blockDim.x) sT[i] = Ts[i];
    __syncthreads();

    const int64_t total = n * (int64_t)S;
    const size_t vsz = (size_t)G * G * G;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / S;
        const int s = (int)(i - r * S);
        const float *ry = rays + r * 8;
        const float near = ry[6], far = ry[7];
        // network.py:416-420: z = near * (1 - t) + far * t, three separate roundings
        auto zlin = [&](int k) {
            const float t = t_vals[k];
            return __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
        };
        float z = zlin(s);
        if (t_rand) {  // network.py:423-432 stratified jitter with injected uniforms
            const float lower = s == 0 ? z : __fmul_rn(0.5f, __fadd_rn(z, zlin(s - 1)));
            const float upper = s == S - 1 ? z : __fmul_rn(0.5f, __fadd_rn(zlin(s + 1), z));
            z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[i]));
        }
        z_vals[i] = z;
        float p[3];
#pragma unroll
        for (int c = 0; c < 3; c++) p[c] = __fadd_rn(ry[c], __fmul_rn(ry[3 + c], z));  // :456
        if (pts_out) {
#pragma unroll
            for (int c = 0; c < 3; c++) pts_out[i * 3 + c] = p[c];
        }
        float wsum = 0.0f, acc[3] = {0.f, 0.f, 0.f};
        for (int b = 0; b < nb; b++) {
            const float *R = sR + b * 9, *T = sT + b * 3;
            float pos[3], g[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                pos[c] = __fadd_rn(__fmaf_rn(R[c * 3 + 2], p[2],
                                             __fmaf_rn(R[c * 3 + 1], p[1], __fmul_rn(R[c * 3], p[0]))),
                                   T[c]);
                g[c] = __fsub_rn(__fmul_rn(__fsub_rn(pos[c], prm.bmin[c]), prm.bscale[c]), 1.0f);
            }
            const float w = trilinear_zeros(vol + (size_t)b * vsz, G, g[0], g[1], g[2]);
            wsum = __fadd_rn(wsum, w);
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] = __fadd_rn(acc[c], __fmul_rn(w, pos[c]));
        }
        const float den = wsum < 0.0001f ? 0.0001f : wsum;  // clamp(min=1e-4), :388
#pragma unroll
        for (int c = 0; c < 3; c++) x_skel[i * 3 + c] = __fdiv_rn(acc[c], den);
        mask[i] = wsum;
    }
}

}  // namespace occ

OCC_API int occnerf_sample_warp(const float *rays, int64_t n, int32_t S, const float *t_vals,
                                const float *t_rand, const float *Rs, const float *Ts,
                                const float *vol, int32_t nb, int32_t G, const float *h_bbox_min,
                                const float *h_bbox_scale, float *z_vals, float *pts, float *x_skel,
                                float *mask, void *stream) {
    using namespace occ;
    if (n <= 0) return 0;
    OCC_REQUIRE(rays && t_vals && Rs && Ts && vol && h_bbox_min && h_bbox_scale && z_vals && x_skel && mask,
                "sample_warp: null argument");
    OCC_REQUIRE(S >= 1 && nb >= 1 && nb <= kMaxBones && G >= 2, "sample_warp: bad sizes S=%d nb=%d G=%d", S, nb, G);
    if (n <= 0) return 0;
    WarpParams prm;
    for (int c = 0; c < 3; c++) {
        prm.bmin[c] = h_bbox_min[c];
        prm.bscale[c] = h_bbox_scale[c];
    }
    const int64_t total = n * (int64_t)S;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)kNumCU * 32) blocks = (int64_t)kNumCU * 32;
    hipLaunchKernelGGL(sample_warp_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), rays,
                       n, S, t_vals, t_rand, Rs, Ts, vol, nb, G, prm, z_vals, pts, x_skel, mask);
    return check_launch("sample_warp");
}
