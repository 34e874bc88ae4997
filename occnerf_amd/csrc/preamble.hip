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

// ---------------------------------------------------------------------------------------------------------------------
// Backward of pose_motion_bases_kernel w.r.t. the pose-refiner MLP (training step, SURVEY 8(f) rows 1 + 4): given
// dL/dRs[24,3,3] and dL/dTs[24,3] (what the warp kernel's backward hands over), the gradients of the five Linear layers.
// torch autograd spends ~460 launches of 2-5 us on this chain per step (inverse, 8 levels of indexed forward kinematics,
// Rodrigues as ~25 elementwise ops, five batch-1 layers); it is < 1 MFLOP.  One workgroup: the forward is recomputed in
// LDS (62 us, nothing to save between the passes), then
//   f = C Ginv (C = cnl_gtfms):    dAi = C3^T dRs,  dti = C3^T dTs
//   Ginv = [A^-1, -A^-1 T]:        dAi -= dti (x) T,  dT = -Ai^T dti,  dA = -Ai^T dAi Ai^T
//   G_i = G_p L_i (reverse order): dRl_i = Rg_p^T dRg_i,  dRg_p += dRg_i Rl_i^T + dTg_i (x) Tl_i,  dTg_p += dTg_i
//   Rl_j = dst_R_j Rod(r_j):       dRod = dst_R_j^T dRl_j,  dr_j = J_Rod(r_j)^T dRod   (theta = sqrt(1e-5 + |r|^2))
//   five layers, batch 1:          dW_l = dz_l (x) h_l,  db_l = dz_l,  dz_{l-1} = (W_l^T dz_l) . [h_l > 0]
// ---------------------------------------------------------------------------------------------------------------------
struct PoseGrads {
    float *dW[5];
    float *db[5];
};

// dz_prev[k] = [h_prev[k] > 0] * sum_j W[j,k] dz[j]   (k < in_dim, j < out_dim; `mask`: apply the ReLU mask of h_prev);
// dW[j,k] = dz[j] * h_prev[k], db[j] = dz[j].  All 1024 threads; part[] is 4 x 256 floats of LDS.
__device__ __forceinline__ void dense_layer_backward(const float *__restrict__ W, int in_dim, int out_dim, const float *dz,
                                                     const float *h_prev, bool mask, float *dz_prev, float *part,
                                                     float *__restrict__ dW, float *__restrict__ db) {
    const int t = threadIdx.x;
    for (int e = t; e < out_dim * in_dim; e += blockDim.x) dW[e] = dz[e / in_dim] * h_prev[e % in_dim];
    if (t < out_dim) db[t] = dz[t];
    if (dz_prev) {
        const int k = t & 255, q = t >> 8;                    // 4 groups of rows per column
        float s = 0.0f;
        if (k < in_dim)
            for (int j = q; j < out_dim; j += 4) s = __fmaf_rn(W[(size_t)j * in_dim + k], dz[j], s);
        part[q * 256 + k] = s;
        __syncthreads();
        if (t < in_dim) {
            const float v = (part[t] + part[256 + t]) + (part[512 + t] + part[768 + t]);
            dz_prev[t] = (!mask || h_prev[t] > 0.0f) ? v : 0.0f;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void pose_motion_bases_backward_kernel(PoseMlp mlp, const float *__restrict__ posevec,
                                                                          const float *__restrict__ dst_Rs,
                                                                          const float *__restrict__ dst_Ts,
                                                                          const float *__restrict__ cnl_gtfms,
                                                                          const float *__restrict__ dRs, const float *__restrict__ dTs,
                                                                          PoseGrads out) {
    __shared__ float h[5][kPoseW];                        // h[0] = posevec (69), h[1..4] = the four ReLU outputs
    __shared__ float rvec[72], drvec[72];
    __shared__ float Rl[kBones][9], Tl[kBones][3], Rg[kBones][9], Tg[kBones][3];
    __shared__ float dRg[kBones][9], dTg[kBones][3], dRl[kBones][9];
    __shared__ float dza[kPoseW], dzb[kPoseW], part[4 * 256];
    const int t = threadIdx.x;
    // ---- forward, as pose_motion_bases_kernel (refine on) ----
    if (t < kBones * 9) Rl[t / 9][t % 9] = dst_Rs[t];
    if (t < kBones * 3) Tl[t / 3][t % 3] = dst_Ts[t];
    if (t < 69) h[0][t] = posevec[t];
    __syncthreads();
    dense_layer(mlp.W[0], mlp.b[0], 69, kPoseW, h[0], h[1], true);
    dense_layer(mlp.W[1], mlp.b[1], kPoseW, kPoseW, h[1], h[2], true);
    dense_layer(mlp.W[2], mlp.b[2], kPoseW, kPoseW, h[2], h[3], true);
    dense_layer(mlp.W[3], mlp.b[3], kPoseW, kPoseW, h[3], h[4], true);
    dense_layer(mlp.W[4], mlp.b[4], kPoseW, 69, h[4], rvec, false);
    if (t < kBones - 1) {
        const float rx = rvec[t * 3], ry = rvec[t * 3 + 1], rz = rvec[t * 3 + 2];
        const float theta = sqrtf(1e-5f + (rx * rx + ry * ry + rz * rz));
        const float x = rx / theta, y = ry / theta, z = rz / theta;
        const float c = cosf(theta), s = sinf(theta), oc = 1.0f - c;
        const float C[9] = {x * x + (1.0f - x * x) * c, x * y * oc - z * s, x * z * oc + y * s,
                            x * y * oc + z * s, y * y + (1.0f - y * y) * c, y * z * oc - x * s,
                            x * z * oc - y * s, y * z * oc + x * s, z * z + (1.0f - z * z) * c};
        float A[9], O[9];
        for (int e = 0; e < 9; e++) A[e] = Rl[t + 1][e];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) O[i * 3 + j] = A[i * 3] * C[j] + A[i * 3 + 1] * C[3 + j] + A[i * 3 + 2] * C[6 + j];
        for (int e = 0; e < 9; e++) Rl[t + 1][e] = O[e];
    }
    __syncthreads();
    if (t == 0) {
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
    // ---- backward of f = C [A^-1, -A^-1 T] ----
    if (t < kBones) {
        const float *a = Rg[t];
        const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
        const float id = 1.0f / (a[0] * c00 + a[1] * c01 + a[2] * c02);
        const float Ai[9] = {c00 * id, (a[2] * a[7] - a[1] * a[8]) * id, (a[1] * a[5] - a[2] * a[4]) * id,
                             c01 * id, (a[0] * a[8] - a[2] * a[6]) * id, (a[2] * a[3] - a[0] * a[5]) * id,
                             c02 * id, (a[1] * a[6] - a[0] * a[7]) * id, (a[0] * a[4] - a[1] * a[3]) * id};
        const float *G = cnl_gtfms + t * 16;
        float dAi[9], dti[3];
        for (int r = 0; r < 3; r++) {                         // C3^T dRs, C3^T dTs
            for (int c = 0; c < 3; c++)
                dAi[r * 3 + c] = G[r] * dRs[t * 9 + c] + G[4 + r] * dRs[t * 9 + 3 + c] + G[8 + r] * dRs[t * 9 + 6 + c];
            dti[r] = G[r] * dTs[t * 3] + G[4 + r] * dTs[t * 3 + 1] + G[8 + r] * dTs[t * 3 + 2];
        }
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) dAi[r * 3 + c] -= dti[r] * Tg[t][c];            // ti = -Ai T
            dTg[t][r] = -(Ai[r] * dti[0] + Ai[3 + r] * dti[1] + Ai[6 + r] * dti[2]);
        }
        float M[9];                                           // dA = -Ai^T dAi Ai^T
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) M[r * 3 + c] = Ai[r] * dAi[c] + Ai[3 + r] * dAi[3 + c] + Ai[6 + r] * dAi[6 + c];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) dRg[t][r * 3 + c] = -(M[r * 3] * Ai[c * 3] + M[r * 3 + 1] * Ai[c * 3 + 1] + M[r * 3 + 2] * Ai[c * 3 + 2]);
    }
    __syncthreads();
    // ---- forward kinematics, reverse order ----
    if (t == 0) {
        for (int i = kBones - 1; i >= 1; i--) {
            const int p = c_smpl_parent[i];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    dRl[i][r * 3 + c] = Rg[p][r] * dRg[i][c] + Rg[p][3 + r] * dRg[i][3 + c] + Rg[p][6 + r] * dRg[i][6 + c];
                    dRg[p][r * 3 + c] += dRg[i][r * 3] * Rl[i][c * 3] + dRg[i][r * 3 + 1] * Rl[i][c * 3 + 1] +
                                         dRg[i][r * 3 + 2] * Rl[i][c * 3 + 2] + dTg[i][r] * Tl[i][c];
                }
            for (int r = 0; r < 3; r++) dTg[p][r] += dTg[i][r];
        }
    }
    __syncthreads();
    // ---- corrected rotation + Rodrigues ----
    if (t < 72) drvec[t] = 0.0f;
    __syncthreads();
    if (t < kBones - 1) {
        const float *A = dst_Rs + (t + 1) * 9;
        float g[9];                                            // dRod = dst_R^T dRl
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) g[r * 3 + c] = A[r] * dRl[t + 1][c] + A[3 + r] * dRl[t + 1][3 + c] + A[6 + r] * dRl[t + 1][6 + c];
        const float rx = rvec[t * 3], ry = rvec[t * 3 + 1], rz = rvec[t * 3 + 2];
        const float theta = sqrtf(1e-5f + (rx * rx + ry * ry + rz * rz));
        const float x = rx / theta, y = ry / theta, z = rz / theta;
        const float c = cosf(theta), s = sinf(theta), oc = 1.0f - c;
        const float dx = g[0] * (2.0f * x * oc) + (g[1] + g[3]) * (y * oc) + (g[2] + g[6]) * (z * oc) + (g[7] - g[5]) * s;
        const float dy = g[4] * (2.0f * y * oc) + (g[1] + g[3]) * (x * oc) + (g[5] + g[7]) * (z * oc) + (g[2] - g[6]) * s;
        const float dz = g[8] * (2.0f * z * oc) + (g[2] + g[6]) * (x * oc) + (g[5] + g[7]) * (y * oc) + (g[3] - g[1]) * s;
        const float dc = g[0] * (1.0f - x * x) + g[4] * (1.0f - y * y) + g[8] * (1.0f - z * z) -
                         ((g[1] + g[3]) * (x * y) + (g[2] + g[6]) * (x * z) + (g[5] + g[7]) * (y * z));
        const float ds = (g[3] - g[1]) * z + (g[2] - g[6]) * y + (g[7] - g[5]) * x;
        const float dtheta = (ds * c - dc * s) - (dx * rx + dy * ry + dz * rz) / (theta * theta);
        drvec[t * 3] = dx / theta + dtheta * rx / theta;
        drvec[t * 3 + 1] = dy / theta + dtheta * ry / theta;
        drvec[t * 3 + 2] = dz / theta + dtheta * rz / theta;
    }
    __syncthreads();
    // ---- the five layers ----
    dense_layer_backward(mlp.W[4], kPoseW, 69, drvec, h[4], true, dza, part, out.dW[4], out.db[4]);
    dense_layer_backward(mlp.W[3], kPoseW, kPoseW, dza, h[3], true, dzb, part, out.dW[3], out.db[3]);
    dense_layer_backward(mlp.W[2], kPoseW, kPoseW, dzb, h[2], true, dza, part, out.dW[2], out.db[2]);
    dense_layer_backward(mlp.W[1], kPoseW, kPoseW, dza, h[1], true, dzb, part, out.dW[1], out.db[1]);
    dense_layer_backward(mlp.W[0], 69, kPoseW, dzb, h[0], false, nullptr, part, out.dW[0], out.db[0]);
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

/* Gradients of the pose refiner's five Linear layers from dL/dRs[24,3,3], dL/dTs[24,3] (the motion bases' cotangents);
 * h_dW[l] / h_db[l]: device buffers shaped like the layers' weights / biases, overwritten.  Refinement on (before the
 * kick-in iteration the refiner does not take part in the graph). */
OCC_API int occnerf_pose_motion_bases_backward(const float *const *h_W, const float *const *h_b, const float *posevec,
                                               const float *dst_Rs, const float *dst_Ts, const float *cnl_gtfms,
                                               const float *dRs, const float *dTs, float *const *h_dW, float *const *h_db,
                                               void *stream) {
    using namespace occ;
    OCC_REQUIRE(h_W && h_b && posevec && dst_Rs && dst_Ts && cnl_gtfms && dRs && dTs && h_dW && h_db,
                "pose_motion_bases_backward: null argument");
    PoseMlp mlp;
    PoseGrads out;
    for (int l = 0; l < 5; l++) {
        OCC_REQUIRE(h_W[l] && h_b[l] && h_dW[l] && h_db[l], "pose_motion_bases_backward: layer %d missing", l);
        mlp.W[l] = h_W[l];
        mlp.b[l] = h_b[l];
        out.dW[l] = h_dW[l];
        out.db[l] = h_db[l];
    }
    hipLaunchKernelGGL(pose_motion_bases_backward_kernel, dim3(1), dim3(1024), 0, as_stream(stream), mlp, posevec, dst_Rs, dst_Ts,
                       cnl_gtfms, dRs, dTs, out);
    return check_launch("pose_motion_bases_backward");
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
