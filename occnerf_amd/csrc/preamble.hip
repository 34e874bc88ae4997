// Per-frame preamble of the renderer: what is left per frame once everything that
// depends on the weights alone has been hoisted (the decoded motion-weight volume logits and the per-point table are
// functions of the checkpoint, not of the frame: occnerf_amd/network.py caches them per weight version).
//
//   pose_motion_bases_kernel   a2 + a3 in one workgroup: pose-refiner MLP 69 -> 256 x4 -> 69 (mlp_delta_body_pose.py:35-41),
//                              Rodrigues (network_util.py:98-124), dst_Rs[1:] @ R_corr (network.py:535-539), forward
//                              kinematics over SMPL_PARENT, inverse, cnl_gtfms @ inverse (network_util.py:166-200).
//                              The reference spends ~100 tiny torch launches on this; it is ~0.5 M MACs.
//   prior_softmax_kernel       a4 tail: softmax over the 25 channels of (decoded + log prior) (deconv_vol_decoder.py:31-33).
//   pack_rays_kernel           rays[2,R,3], near[R], far[R] -> rays8[R,8] in the renderer's (Morton) ray order.
#include "common.h"

namespace occ {

constexpr int kBones = 24;
constexpr int kPoseW = 256;
__constant__ int c_smpl_parent[kBones] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};

struct PoseMlp {
    const float *W[5];
    const float *b[5];
};

// y[j] = act(b[j] + sum_k W[j,k] x[k]) for j < out_dim.  The five layers are a dependent chain inside ONE workgroup, so
// what matters is memory-level parallelism: each of the 16 waves takes 4 rows at a time and has all their loads (16-byte
// pieces of contiguous weight rows, up to 16 per lane) in flight before the first FMA; butterfly sums; x, y in LDS.
__device__ __forceinline__ void dense_layer(const float *__restrict__ W, const float *__restrict__ b, int in_dim, int out_dim,
                                            const float *x, float *y, bool relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j0 = wave * 4; j0 < out_dim; j0 += nw * 4) {
        float w[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = lane + 64 * u, j = j0 + r;
                w[r][u] = (k < in_dim && j < out_dim) ? W[(size_t)j * in_dim + k] : 0.0f;
            }
        }
        float xs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) xs[u] = lane + 64 * u < in_dim ? x[lane + 64 * u] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float s = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; u++) s = __fmaf_rn(w[r][u], xs[u], s);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0 && j0 + r < out_dim) {
                s += b[j0 + r];
                y[j0 + r] = relu ? fmaxf(s, 0.0f) : s;
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void pose_motion_bases_kernel(PoseMlp mlp, const float *__restrict__ posevec, int refine,
                                                                const float *__restrict__ dst_Rs, const float *__restrict__ dst_Ts,
                                                                const float *__restrict__ cnl_gtfms, float *__restrict__ Rs_out,
                                                                float *__restrict__ Ts_out) {
    __shared__ float ha[kPoseW], hb[kPoseW];
    __shared__ float Rl[kBones][9], Tl[kBones][3];       // local transforms (refined rotations)
    __shared__ float Rg[kBones][9], Tg[kBones][3];       // global transforms
    const int t = threadIdx.x;
    if (t < kBones * 9) Rl[t / 9][t % 9] = dst_Rs[t];
    if (t < kBones * 3) Tl[t / 3][t % 3] = dst_Ts[t];
    if (t < 69) ha[t] = posevec[t];
    __syncthreads();
    if (refine) {
        dense_layer(mlp.W[0], mlp.b[0], 69, kPoseW, ha, hb, true);
        dense_layer(mlp.W[1], mlp.b[1], kPoseW, kPoseW, hb, ha, true);
        dense_layer(mlp.W[2], mlp.b[2], kPoseW, kPoseW, ha, hb, true);
        dense_layer(mlp.W[3], mlp.b[3], kPoseW, kPoseW, hb, ha, true);
        dense_layer(mlp.W[4], mlp.b[4], kPoseW, 69, ha, hb, false);
        if (t < kBones - 1) {                                 // bone t+1: R = dst_R @ rodrigues(rvec)
            const float rx = hb[t * 3], ry = hb[t * 3 + 1], rz = hb[t * 3 + 2];
            const float theta = sqrtf(1e-5f + (rx * rx + ry * ry + rz * rz));
            const float x = rx / theta, y = ry / theta, z = rz / theta;
            const float c = cosf(theta), s = sinf(theta), oc = 1.0f - c;
            const float C[9] = {x * x + (1.0f - x * x) * c, x * y * oc - z * s, x * z * oc + y * s,
                                x * y * oc + z * s, y * y + (1.0f - y * y) * c, y * z * oc - x * s,
                                x * z * oc - y * s, y * z * oc + x * s, z * z + (1.0f - z * z) * c};
            float A[9], O[9];
#pragma unroll
            for (int e = 0; e < 9; e++) A[e] = Rl[t + 1][e];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) O[i * 3 + j] = A[i * 3] * C[j] + A[i * 3 + 1] * C[3 + j] + A[i * 3 + 2] * C[6 + j];
#pragma unroll
            for (int e = 0; e < 9; e++) Rl[t + 1][e] = O[e];
        }
        __syncthreads();
    }
    if (t == 0) {                                             // forward kinematics: a 23-step chain of 3x4 products
        for (int e = 0; e < 9; e++) Rg[0][e] = Rl[0][e];
        for (int e = 0; e < 3; e++) Tg[0][e] = Tl[0][e];
        for (int i = 1; i < kBones; i++) {
            const int p = c_smpl_parent[i];
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++)
                    Rg[i][r * 3 + c] = Rg[p][r * 3] * Rl[i][c] + Rg[p][r * 3 + 1] * Rl[i][3 + c] + Rg[p][r * 3 + 2] * Rl[i][6 + c];
                Tg[i][r] = Rg[p][r * 3] * Tl[i][0] + Rg[p][r * 3 + 1] * Tl[i][1] + Rg[p][r * 3 + 2] * Tl[i][2] + Tg[p][r];
            }
        }
    }
    __syncthreads();
    if (t < kBones) {                                         // f = cnl_gtfms @ inverse([Rg Tg; 0 1])
        const float *a = Rg[t];
        const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
        const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
        const float id = 1.0f / det;
        const float Ai[9] = {c00 * id, (a[2] * a[7] - a[1] * a[8]) * id, (a[1] * a[5] - a[2] * a[4]) * id,
                             c01 * id, (a[0] * a[8] - a[2] * a[6]) * id, (a[2] * a[3] - a[0] * a[5]) * id,
                             c02 * id, (a[1] * a[6] - a[0] * a[7]) * id, (a[0] * a[4] - a[1] * a[3]) * id};
        float ti[3];
#pragma unroll
        for (int r = 0; r < 3; r++) ti[r] = -(Ai[r * 3] * Tg[t][0] + Ai[r * 3 + 1] * Tg[t][1] + Ai[r * 3 + 2] * Tg[t][2]);
        const float *G = cnl_gtfms + t * 16;                  // row-major 4x4; its bottom row is (0,0,0,1)
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int c = 0; c < 3; c++)
                Rs_out[t * 9 + r * 3 + c] = G[r * 4] * Ai[c] + G[r * 4 + 1] * Ai[3 + c] + G[r * 4 + 2] * Ai[6 + c];
            Ts_out[t * 3 + r] = G[r * 4] * ti[0] + G[r * 4 + 1] * ti[1] + G[r * 4 + 2] * ti[2] + G[r * 4 + 3];
        }
    }
}

constexpr int kMaxVolCh = 32;

__global__ __launch_bounds__(256) void prior_softmax_kernel(const float *__restrict__ dec, const float *__restrict__ prior,
                                                            int C, int64_t V, float *__restrict__ vol) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float x[kMaxVolCh];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < kMaxVolCh; c++) {
        if (c < C) {
            x[c] = dec[c * V + v] + logf(prior[c * V + v]);
            m = fmaxf(m, x[c]);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < kMaxVolCh; c++) {
        if (c < C) {
            x[c] = expf(x[c] - m);
            s += x[c];
        }
    }
#pragma unroll
    for (int c = 0; c < kMaxVolCh; c++)
        if (c < C) vol[c * V + v] = x[c] / s;
}

__global__ __launch_bounds__(256) void pack_rays_kernel(const float *__restrict__ rays, const float *__restrict__ near,
                                                        const float *__restrict__ far, const int64_t *__restrict__ order, int64_t R,
                                                        float *__restrict__ rays8) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t src = order ? order[r] : r;
    float4 a, b;
    a.x = rays[src * 3];
    a.y = rays[src * 3 + 1];
    a.z = rays[src * 3 + 2];
    a.w = rays[(R + src) * 3];
    b.x = rays[(R + src) * 3 + 1];
    b.y = rays[(R + src) * 3 + 2];
    b.z = near[src];
    b.w = far[src];
    reinterpret_cast<float4 *>(rays8)[r * 2] = a;
    reinterpret_cast<float4 *>(rays8)[r * 2 + 1] = b;
}

}  // namespace occ

OCC_API int occnerf_pose_motion_bases(const float *const *h_W, const float *const *h_b, const float *posevec, int32_t refine,
                                      const float *dst_Rs, const float *dst_Ts, const float *cnl_gtfms, float *Rs, float *Ts,
                                      void *stream) {
    using namespace occ;
    OCC_REQUIRE(posevec && dst_Rs && dst_Ts && cnl_gtfms && Rs && Ts, "pose_motion_bases: null argument");
    PoseMlp mlp;
    for (int l = 0; l < 5; l++) {
        OCC_REQUIRE(!refine || (h_W && h_b && h_W[l] && h_b[l]), "pose_motion_bases: pose-refiner layer %d missing", l);
        mlp.W[l] = refine ? h_W[l] : nullptr;
        mlp.b[l] = refine ? h_b[l] : nullptr;
    }
    hipLaunchKernelGGL(pose_motion_bases_kernel, dim3(1), dim3(1024), 0, as_stream(stream), mlp, posevec, refine, dst_Rs, dst_Ts,
                       cnl_gtfms, Rs, Ts);
    return check_launch("pose_motion_bases");
}

OCC_API int occnerf_prior_softmax(const float *decoded, const float *prior, int32_t C, int64_t V, float *vol, void *stream) {
    using namespace occ;
    OCC_REQUIRE(decoded && prior && vol, "prior_softmax: null argument");
    OCC_REQUIRE(C >= 1 && C <= kMaxVolCh && V > 0, "prior_softmax: C=%d V=%lld", C, (long long)V);
    hipLaunchKernelGGL(prior_softmax_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, as_stream(stream), decoded, prior, C,
                       V, vol);
    return check_launch("prior_softmax");
}

OCC_API int occnerf_pack_rays(const float *rays, const float *near, const float *far, const int64_t *order, int64_t R,
                              float *rays8, void *stream) {
    using namespace occ;
    if (R <= 0) return 0;
    OCC_REQUIRE(rays && near && far && rays8, "pack_rays: null argument");
    hipLaunchKernelGGL(pack_rays_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, as_stream(stream), rays, near, far,
                       order, R, rays8);
    return check_launch("pack_rays");
}
