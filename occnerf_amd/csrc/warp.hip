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

// Support box of every bone's motion-weight channel: the smallest index box [x0,x1] x [y0,y1] x [z0,z1] that holds every
// voxel with a non-zero weight (softmax(decoded + log prior) is exactly 0 wherever the bone's prior is 0: most of the
// 32^3 grid).  boxes[b] = {x0, x1, y0, y1, z0, z1}; an all-zero channel gets x0 > x1.  One block per bone.
__global__ __launch_bounds__(256) void bone_boxes_kernel(const float *__restrict__ vol, int G, int32_t *__restrict__ boxes) {
    const int b = blockIdx.x;
    const float *v = vol + (size_t)b * G * G * G;
    int lo[3] = {G, G, G}, hi[3] = {-1, -1, -1};
    for (int i = threadIdx.x; i < G * G * G; i += blockDim.x) {
        if (v[i] != 0.0f) {
            const int x = i % G, y = (i / G) % G, z = i / (G * G);
            lo[0] = min(lo[0], x), hi[0] = max(hi[0], x);
            lo[1] = min(lo[1], y), hi[1] = max(hi[1], y);
            lo[2] = min(lo[2], z), hi[2] = max(hi[2], z);
        }
    }
    __shared__ int s_lo[3], s_hi[3];
    if (threadIdx.x < 3) s_lo[threadIdx.x] = G, s_hi[threadIdx.x] = -1;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) {
        atomicMin(&s_lo[c], lo[c]);
        atomicMax(&s_hi[c], hi[c]);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        boxes[b * 6 + 2 * threadIdx.x] = s_lo[threadIdx.x];
        boxes[b * 6 + 2 * threadIdx.x + 1] = s_hi[threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void sample_warp_kernel(
    const float *__restrict__ rays, int64_t n, int S, const float *__restrict__ t_vals,
    const float *__restrict__ t_rand, const float *__restrict__ Rs, const float *__restrict__ Ts,
    const float *__restrict__ vol, int nb, int G, WarpParams prm, float *__restrict__ z_vals,
    float *__restrict__ pts_out, float *__restrict__ x_skel, float *__restrict__ mask,
    const int32_t *__restrict__ boxes /*nullable: bone_boxes_kernel's output; used when S % 64 == 0*/) {
    // bone transforms: 12 floats x nb, staged once per block
    __shared__ float sR[kMaxBones * 9];
    __shared__ float sT[kMaxBones * 3];
    for (int i = threadIdx.x; i < nb * 9; i += blockDim.x) sR[i] = Rs[i];
    for (int i = threadIdx.x; i < nb * 3; i += blockDim.x) sT[i] = Ts[i];
    __syncthreads();

    const int64_t total = n * (int64_t)S;
    const size_t vsz = (size_t)G * G * G;
    // BONE CULLING (round 4).  With S a multiple of 64 a wave's 64 samples lie on ONE ray (i is a multiple of 64 plus the lane
    // and the grid stride is one too), and along a ray a bone's grid coordinates are affine in z: lane b computes, once per
    // wave, the z-interval in which bone b's line runs through the bone's support box widened by one cell (a trilinear tap at
    // floor index f reads f and f + 1) plus a 0.02-cell / 1e-5 margin for the different rounding of the per-sample
    // evaluation below.  A bone whose interval misses all 64 samples is skipped by the whole wave: each such (sample, bone)
    // pair would have produced a weight of exactly +0 -- adding it changes no bit of wsum or acc -- so the outputs are
    // bit-identical to the unculled kernel (tested), and the bones that are evaluated go through the very same code.
    // On the benchmark frame 4.6 of the 24 bones have a non-zero weight at a sample on average (9.5 lie inside the grid).
    const bool cull = boxes != nullptr && (S & 63) == 0;
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
        float zlo = -INFINITY, zhi = INFINITY;      // lane b: the z-interval of bone b on this wave's ray
        if (cull) {
            const int lane = threadIdx.x & 63;
            if (lane < nb) {
                const float *R = sR + lane * 9, *T = sT + lane * 3;
                const int32_t *bx = boxes + lane * 6;
                const float half = 0.5f * (float)(G - 1);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    // grid index along axis c: ix(z) = A + B z
                    const float p0 = R[c * 3] * ry[0] + R[c * 3 + 1] * ry[1] + R[c * 3 + 2] * ry[2] + T[c];
                    const float pd = R[c * 3] * ry[3] + R[c * 3 + 1] * ry[4] + R[c * 3 + 2] * ry[5];
                    const float A = (p0 - prm.bmin[c]) * prm.bscale[c] * half, B = pd * prm.bscale[c] * half;
                    const float L = (float)bx[2 * c] - 1.02f, U = (float)bx[2 * c + 1] + 1.02f;
                    if (bx[2 * c] > bx[2 * c + 1]) {
                        zlo = INFINITY, zhi = -INFINITY;                      // empty channel
                    } else if (fabsf(B) < 1e-12f) {
                        if (A < L || A > U) zlo = INFINITY, zhi = -INFINITY;  // parallel to the slab and outside it
                    } else {
                        const float t1 = (L - A) / B, t2 = (U - A) / B;
                        zlo = fmaxf(zlo, fminf(t1, t2));
                        zhi = fminf(zhi, fmaxf(t1, t2));
                    }
                }
                const float eps = 1e-5f * (fabsf(near) + fabsf(far)) + 1e-6f;
                zlo -= eps, zhi += eps;
            }
        }
        float wsum = 0.0f, acc[3] = {0.f, 0.f, 0.f};
        for (int b = 0; b < nb; b++) {
            if (cull) {
                const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zlo), b));
                const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zhi), b));
                if (__builtin_amdgcn_ballot_w64(z >= lo && z <= hi) == 0) continue;      // wave-uniform
            }
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
                       n, S, t_vals, t_rand, Rs, Ts, vol, nb, G, prm, z_vals, pts, x_skel, mask, nullptr);
    return check_launch("sample_warp");
}

OCC_API int occnerf_bone_boxes(const float *vol, int32_t nb, int32_t G, int32_t *boxes, void *stream) {
    using namespace occ;
    OCC_REQUIRE(vol && boxes, "bone_boxes: null argument");
    OCC_REQUIRE(nb >= 1 && nb <= kMaxBones && G >= 2 && G <= 1024, "bone_boxes: bad sizes nb=%d G=%d", nb, G);
    hipLaunchKernelGGL(bone_boxes_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), vol, G, boxes);
    return check_launch("bone_boxes");
}

OCC_API int occnerf_sample_warp_culled(const float *rays, int64_t n, int32_t S, const float *t_vals,
                                       const float *Rs, const float *Ts, const float *vol, int32_t nb, int32_t G,
                                       const int32_t *boxes, const float *h_bbox_min, const float *h_bbox_scale,
                                       float *z_vals, float *x_skel, float *mask, void *stream) {
    using namespace occ;
    if (n <= 0) return 0;
    OCC_REQUIRE(rays && t_vals && Rs && Ts && vol && boxes && h_bbox_min && h_bbox_scale && z_vals && x_skel && mask,
                "sample_warp_culled: null argument");
    OCC_REQUIRE(S >= 1 && nb >= 1 && nb <= kMaxBones && G >= 2, "sample_warp_culled: bad sizes S=%d nb=%d G=%d", S, nb, G);
    WarpParams prm;
    for (int c = 0; c < 3; c++) {
        prm.bmin[c] = h_bbox_min[c];
        prm.bscale[c] = h_bbox_scale[c];
    }
    const int64_t total = n * (int64_t)S;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)kNumCU * 32) blocks = (int64_t)kNumCU * 32;
    hipLaunchKernelGGL(sample_warp_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), rays,
                       n, S, t_vals, nullptr, Rs, Ts, vol, nb, G, prm, z_vals, nullptr, x_skel, mask, boxes);
    return check_launch("sample_warp_culled");
}
