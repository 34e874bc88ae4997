// Backward kernels of the sample pipeline's non-GEMM stages for the training step (SURVEY.md section 8 rows
// a18 / f1): what torch autograd derives from network.py:320-348 (_raw2outputs) and network.py:351-402
// (_sample_motion_fields), plus the (gradient-free) attention weights of simple_agg, occnerf_mlp.py:110-125.
//
//   composite_backward_kernel   d(rgb, acc, depth)/d(raw, mask): wave per ray, the forward is recomputed in
//                               registers (ascending product scan), then a descending suffix-sum scan.
//   warp_backward_kernel        d(mask)/d(vol, Rs, Ts).  mask = sum over bones of a trilinear tap, so the
//                               gradient of the volume is a scatter of 8 corner weights per (sample, bone).
//                               Global fp32 atomics are memory-side operations on this part (~1.6 G/s): a
//                               workgroup instead OWNS half of one bone's 32^3 volume as fp64 in 128 KiB of LDS
//                               (ds_add_f64: 16 cycles per wave-instruction; ds_add_f32: 190), scans a slice of
//                               the samples and writes its tile out with plain stores; slices are summed after.
//   agg_weights_kernel          softmax visibility weights + their unbiased variance per sample.
#include "common.h"

namespace occ {

__device__ __forceinline__ float softplus20(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

constexpr int kMaxChunks = 4;      // S <= 256

struct CompBwdParams {
    float bg[3];
};

__global__ __launch_bounds__(256) void composite_backward_kernel(
    const float *__restrict__ raw, const float *__restrict__ mask, const float *__restrict__ z_vals,
    const float *__restrict__ rays, CompBwdParams prm, int64_t n, int S, const float *__restrict__ g_rgb,
    const float *__restrict__ g_acc, const float *__restrict__ g_depth, float *__restrict__ d_raw,
    float *__restrict__ d_mask) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int nch = (S + kWave - 1) / kWave;
    for (int64_t r = wave; r < n; r += nwaves) {
        const float *ry = rays + r * 8;
        const float dn = norm3(ry[3], ry[4], ry[5]);
        const float gr = g_rgb ? g_rgb[r * 3] : 0.f, gg = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
        const float ga = g_acc ? g_acc[r] : 0.f, gd = g_depth ? g_depth[r] : 0.f;
        // rgb_map = sum w c + (1 - sum w) bg / 255  ->  every weight also carries -g_rgb . bg / 255
        const float gbg = (gr * prm.bg[0] + gg * prm.bg[1] + gb * prm.bg[2]) / 255.0f;

        float alpha[kMaxChunks], T[kMaxChunks], tt[kMaxChunks], em[kMaxChunks], dist[kMaxChunks], x3[kMaxChunks];
        float c0[kMaxChunks], c1[kMaxChunks], c2[kMaxChunks], G[kMaxChunks], mk[kMaxChunks];
        float carry = 1.0f;
#pragma unroll
        for (int ch = 0; ch < kMaxChunks; ch++) {
            alpha[ch] = T[ch] = em[ch] = dist[ch] = x3[ch] = c0[ch] = c1[ch] = c2[ch] = G[ch] = mk[ch] = 0.0f;
            tt[ch] = 1.0f;
            if (ch >= nch) continue;
            const int s = ch * kWave + lane;
            const bool live = s < S;
            const int64_t i = r * S + (live ? s : S - 1);
            const float z = z_vals[i];
            if (live) {
                const float zn = s + 1 < S ? z_vals[i + 1] : 0.0f;
                dist[ch] = (s + 1 < S ? zn - z : 1e10f) * dn;
                const float *rw = raw + i * 5;
                x3[ch] = rw[3];
                em[ch] = expf(-(softplus20(x3[ch]) * dist[ch]));
                mk[ch] = mask[i];
                alpha[ch] = (1.0f - em[ch]) * mk[ch];
                c0[ch] = 1.0f / (1.0f + expf(-rw[0]));
                c1[ch] = 1.0f / (1.0f + expf(-rw[1]));
                c2[ch] = 1.0f / (1.0f + expf(-rw[2]));
                tt[ch] = (1.0f - alpha[ch]) + 1e-10f;
                G[ch] = gr * c0[ch] + gg * c1[ch] + gb * c2[ch] + gd * z + ga - gbg;
            }
            float incl = tt[ch];
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const float up = __shfl_up(incl, o);
                if (lane >= o) incl *= up;
            }
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.0f;
            T[ch] = carry * excl;
            carry *= __shfl(incl, kWave - 1);
        }
        // descending: R_s = sum_{j > s} G_j w_j
        float tail = 0.0f;
#pragma unroll
        for (int ch = kMaxChunks - 1; ch >= 0; ch--) {
            if (ch >= nch) continue;
            const int s = ch * kWave + lane;
            const bool live = s < S;
            const float w = alpha[ch] * T[ch];
            const float gw = live ? G[ch] * w : 0.0f;
            float incl = gw;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const float dnv = __shfl_down(incl, o);
                if (lane + o < kWave) incl += dnv;
            }
            const float R = tail + (incl - gw);
            tail += __shfl(incl, 0);
            if (live) {
                const int64_t i = r * S + s;
                const float dalpha = G[ch] * T[ch] - R / tt[ch];
                const float one_m = 1.0f - em[ch];
                // alpha = (1 - exp(-softplus(x) dist)) mask
                const float dsp = dalpha * mk[ch] * dist[ch] * em[ch];
                const float ex = expf(x3[ch]);
                const float dx3 = x3[ch] > 20.0f ? dsp : dsp * (ex / (ex + 1.0f));
                float *o = d_raw + i * 5;
                o[0] = gr * w * c0[ch] * (1.0f - c0[ch]);
                o[1] = gg * w * c1[ch] * (1.0f - c1[ch]);
                o[2] = gb * w * c2[ch] * (1.0f - c2[ch]);
                o[3] = dx3;
                o[4] = 0.0f;
                if (d_mask) d_mask[i] = dalpha * one_m;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
constexpr int kVolG = 32;                              // the motion-weight volume is 32^3 per bone
constexpr int kHalfVox = (kVolG / 2) * kVolG * kVolG;  // 16384 voxels = 128 KiB of fp64

struct WarpBwdParams {
    float bmin[3];
    float bscale[3];
};

__global__ __launch_bounds__(256, 1) void warp_backward_kernel(
    const float *__restrict__ rays, int64_t n, int S, const float *__restrict__ z_vals,
    const float *__restrict__ g_mask, const float *__restrict__ Rs, const float *__restrict__ Ts,
    const float *__restrict__ vol, int nb, WarpBwdParams prm, int64_t samples_per_slice,
    float *__restrict__ d_vol_part, float *__restrict__ d_rt_part) {
    __shared__ double tile[kHalfVox + 64];              // + 4 waves x 12 partial sums
    const int bone = blockIdx.x >> 1, hz = blockIdx.x & 1;
    const int slice = blockIdx.y;
    for (int v = threadIdx.x; v < kHalfVox + 64; v += blockDim.x) tile[v] = 0.0;
    __syncthreads();

    float R[9], T[3];
#pragma unroll
    for (int c = 0; c < 9; c++) R[c] = Rs[bone * 9 + c];
#pragma unroll
    for (int c = 0; c < 3; c++) T[c] = Ts[bone * 3 + c];
    const float *bv = vol + (size_t)bone * kVolG * kVolG * kVolG;
    const float gm1 = (float)(kVolG - 1);
    const int zlo = hz * (kVolG / 2);

    float dRT[12];
#pragma unroll
    for (int c = 0; c < 12; c++) dRT[c] = 0.0f;

    const int64_t total = n * (int64_t)S;
    const int64_t i0 = slice * samples_per_slice;
    const int64_t i1 = i0 + samples_per_slice < total ? i0 + samples_per_slice : total;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float g = g_mask[i];
        if (g == 0.0f) continue;
        const int64_t r = i / S;
        const float *ry = rays + r * 8;
        const float z = z_vals[i];
        float p[3], pos[3], gi[3];
#pragma unroll
        for (int c = 0; c < 3; c++) p[c] = __fadd_rn(ry[c], __fmul_rn(ry[3 + c], z));
#pragma unroll
        for (int c = 0; c < 3; c++) {
            pos[c] = __fadd_rn(__fmaf_rn(R[c * 3 + 2], p[2], __fmaf_rn(R[c * 3 + 1], p[1], __fmul_rn(R[c * 3], p[0]))), T[c]);
            const float gc = __fsub_rn(__fmul_rn(__fsub_rn(pos[c], prm.bmin[c]), prm.bscale[c]), 1.0f);
            gi[c] = __fmul_rn(__fdiv_rn(__fadd_rn(gc, 1.0f), 2.0f), gm1);
        }
        const float fx = floorf(gi[0]), fy = floorf(gi[1]), fz = floorf(gi[2]);
        const float Gf = (float)kVolG;
        if (!(fx >= -1.0f && fx <= Gf && fy >= -1.0f && fy <= Gf && fz >= -1.0f && fz <= Gf)) continue;
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
        const float wx[2] = {(float)(x0 + 1) - gi[0], gi[0] - fx};
        const float wy[2] = {(float)(y0 + 1) - gi[1], gi[1] - fy};
        const float wz[2] = {(float)(z0 + 1) - gi[2], gi[2] - fz};
        float dix = 0.f, diy = 0.f, diz = 0.f;
#pragma unroll
        for (int cz = 0; cz < 2; cz++) {
            const int zz = z0 + cz;
            if (zz < 0 || zz >= kVolG) continue;
#pragma unroll
            for (int cy = 0; cy < 2; cy++) {
                const int yy = y0 + cy;
                if (yy < 0 || yy >= kVolG) continue;
#pragma unroll
                for (int cx = 0; cx < 2; cx++) {
                    const int xx = x0 + cx;
                    if (xx < 0 || xx >= kVolG) continue;
                    const int lz = zz - zlo;
                    if (lz >= 0 && lz < kVolG / 2)
                        atomicAdd(&tile[(lz * kVolG + yy) * kVolG + xx], (double)(g * (wx[cx] * wy[cy] * wz[cz])));
                    if (hz == 0) {
                        const float v = bv[(zz * kVolG + yy) * kVolG + xx];
                        dix += (cx ? v : -v) * wy[cy] * wz[cz];
                        diy += (cy ? v : -v) * wx[cx] * wz[cz];
                        diz += (cz ? v : -v) * wx[cx] * wy[cy];
                    }
                }
            }
        }
        if (hz == 0) {
            const float dp[3] = {g * dix * prm.bscale[0] * (gm1 * 0.5f), g * diy * prm.bscale[1] * (gm1 * 0.5f),
                                 g * diz * prm.bscale[2] * (gm1 * 0.5f)};
#pragma unroll
            for (int c = 0; c < 3; c++) {
#pragma unroll
                for (int k = 0; k < 3; k++) dRT[c * 3 + k] += dp[c] * p[k];
                dRT[9 + c] += dp[c];
            }
        }
    }
    __syncthreads();
    float *out = d_vol_part + ((size_t)slice * nb + bone) * kVolG * kVolG * kVolG + (size_t)zlo * kVolG * kVolG;
    for (int v = threadIdx.x; v < kHalfVox; v += blockDim.x) out[v] = (float)tile[v];
    if (hz == 0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 12; c++) {
            float s = dRT[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) tile[kHalfVox + wave * 12 + c] = (double)s;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            double s = 0.0;
            for (int w = 0; w < 4; w++) s += tile[kHalfVox + w * 12 + threadIdx.x];
            d_rt_part[((size_t)slice * nb + bone) * 12 + threadIdx.x] = (float)s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// occnerf_mlp.py:110-125 (simple_agg): a = counter[knn]; a += 1 - min; a /= max; var = unbiased variance;
// softmax.  One thread per sample, K <= 40 neighbours; the arithmetic order is features.hip's.
constexpr int kAggMaxK = 40;

__global__ __launch_bounds__(256) void agg_weights_kernel(const float *__restrict__ counter, const int32_t *__restrict__ knn,
                                                          int64_t N, int K, float *__restrict__ atts, float *__restrict__ var_out) {
    // (round 5) a thread owns a sample, but its K ids / K weights are 4 K bytes apart from its neighbour's: the block's
    // 256 x K ids come in -- and the weights go out -- coalesced through an LDS tile (row pitch K + 1 words: the 256 rows' reads
    // spread over the banks).  Same arithmetic, same order.
    __shared__ uint32_t tile[256 * (kAggMaxK + 1)];
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x;
    const int rows = N - i0 < (int64_t)blockDim.x ? (int)(N - i0) : (int)blockDim.x;
    for (int e = threadIdx.x; e < rows * K; e += blockDim.x)
        tile[(e / K) * (kAggMaxK + 1) + e % K] = (uint32_t)knn[i0 * K + e];
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < N;
    uint32_t *mine = tile + threadIdx.x * (kAggMaxK + 1);
    float att[kAggMaxK];
    float amin = INFINITY;
#pragma unroll
    for (int j = 0; j < kAggMaxK; j++) {
        att[j] = (live && j < K) ? counter[(int32_t)mine[j]] : INFINITY;
        amin = fminf(amin, att[j]);
    }
    float amax = -INFINITY;
#pragma unroll
    for (int j = 0; j < kAggMaxK; j++) {
        if (j < K) {
            att[j] = __fadd_rn(att[j], __fsub_rn(1.0f, amin));
            amax = fmaxf(amax, att[j]);
        }
    }
    float mean = 0.0f;
#pragma unroll
    for (int j = 0; j < kAggMaxK; j++) {
        if (j < K) {
            att[j] = __fdiv_rn(att[j], amax);
            mean = __fadd_rn(mean, att[j]);
        }
    }
    mean = __fdiv_rn(mean, (float)K);
    float var = 0.0f, smax = -INFINITY;
#pragma unroll
    for (int j = 0; j < kAggMaxK; j++) {
        if (j < K) {
            const float d = __fsub_rn(att[j], mean);
            var = __fadd_rn(var, __fmul_rn(d, d));
            smax = fmaxf(smax, att[j]);
        }
    }
    if (live) var_out[i] = __fdiv_rn(var, (float)(K - 1));
    float ssum = 0.0f;
#pragma unroll
    for (int j = 0; j < kAggMaxK; j++) {
        if (j < K) {
            att[j] = expf(__fsub_rn(att[j], smax));
            ssum = __fadd_rn(ssum, att[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < kAggMaxK; j++)
        if (j < K) mine[j] = __float_as_uint(__fdiv_rn(att[j], ssum));      // (own row: read above by this thread only)
    __syncthreads();
    for (int e = threadIdx.x; e < rows * K; e += blockDim.x)
        atts[i0 * K + e] = __uint_as_float(tile[(e / K) * (kAggMaxK + 1) + e % K]);
}

// ---------------------------------------------------------------------------------------------------------
// ConvTranspose3d(kernel 4, stride 2, padding 1) of the motion-weight volume decoder (network_util.py:12-50,
// deconv_vol_decoder.py:25-33) as one GEMM + one gather: cols[(co, k), v] = sum_ci W[ci, co, k] x[ci, v] is a plain
// matrix product (the caller's library GEMM); input voxel i reaches outputs o = 2 i - 1 + k per axis, so output voxel
// o collects the (at most 2 per axis, 8 in all) taps k with (o + 1 - k) even: col2im below is that gather -- no
// scatter, no atomics, fixed summation order.  Its adjoint (im2col of the output gradient) is the same index map read
// the other way; the weight and input gradients are then two more plain GEMMs.  MIOpen needs 31 ms for the forward of
// this 2 GFLOP stack and 4.8 ms for its backward.
__global__ __launch_bounds__(256) void convt3d_col2im_kernel(const float *__restrict__ cols, const float *__restrict__ bias,
                                                             int Cout, int D, int H, int W, float *__restrict__ out) {
    const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
    const int64_t V = (int64_t)D * H * W, total = (int64_t)Cout * OD * OH * OW;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int ow = (int)(e % OW), oh = (int)((e / OW) % OH), od = (int)((e / ((int64_t)OW * OH)) % OD);
    const int co = (int)(e / ((int64_t)OW * OH * OD));
    float s = bias ? bias[co] : 0.0f;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int kd = ((od + 1) & 1) + 2 * a, id = (od + 1 - kd) >> 1;
        if (id < 0 || id >= D) continue;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int kh = ((oh + 1) & 1) + 2 * b, ih = (oh + 1 - kh) >> 1;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int kw = ((ow + 1) & 1) + 2 * c, iw = (ow + 1 - kw) >> 1;
                if (iw < 0 || iw >= W) continue;
                s += cols[((int64_t)co * 64 + (kd * 16 + kh * 4 + kw)) * V + ((int64_t)id * H + ih) * W + iw];
            }
        }
    }
    out[e] = s;
}

__global__ __launch_bounds__(256) void convt3d_im2col_kernel(const float *__restrict__ gy, int Cout, int D, int H, int W,
                                                             float *__restrict__ dcols) {
    const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
    const int64_t V = (int64_t)D * H * W, total = (int64_t)Cout * 64 * V;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t v = e % V;
    const int k = (int)((e / V) & 63), co = (int)(e / (V * 64));
    const int iw = (int)(v % W), ih = (int)((v / W) % H), id = (int)(v / ((int64_t)W * H));
    const int od = 2 * id - 1 + (k >> 4), oh = 2 * ih - 1 + ((k >> 2) & 3), ow = 2 * iw - 1 + (k & 3);
    float g = 0.0f;
    if (od >= 0 && od < OD && oh >= 0 && oh < OH && ow >= 0 && ow < OW)
        g = gy[(((int64_t)co * OD + od) * OH + oh) * OW + ow];
    dcols[e] = g;
}

}  // namespace occ

OCC_API int occnerf_convt3d_col2im(const float *cols, const float *bias, int32_t Cout, int32_t D, int32_t H, int32_t W,
                                   float *out, void *stream) {
    using namespace occ;
    OCC_REQUIRE(cols && out, "convt3d_col2im: null argument");
    OCC_REQUIRE(Cout > 0 && D > 0 && H > 0 && W > 0, "convt3d_col2im: bad sizes");
    const int64_t total = (int64_t)Cout * 8 * D * H * W;
    hipLaunchKernelGGL(convt3d_col2im_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), cols,
                       bias, Cout, D, H, W, out);
    return check_launch("convt3d_col2im");
}

OCC_API int occnerf_convt3d_im2col(const float *gy, int32_t Cout, int32_t D, int32_t H, int32_t W, float *dcols,
                                   void *stream) {
    using namespace occ;
    OCC_REQUIRE(gy && dcols, "convt3d_im2col: null argument");
    OCC_REQUIRE(Cout > 0 && D > 0 && H > 0 && W > 0, "convt3d_im2col: bad sizes");
    const int64_t total = (int64_t)Cout * 64 * D * H * W;
    hipLaunchKernelGGL(convt3d_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), gy,
                       Cout, D, H, W, dcols);
    return check_launch("convt3d_im2col");
}

OCC_API int occnerf_composite_backward(const float *raw, const float *mask, const float *z_vals, const float *rays,
                                       const float *h_bgcolor, int64_t n, int32_t S, const float *g_rgb,
                                       const float *g_acc, const float *g_depth, float *d_raw, float *d_mask,
                                       void *stream) {
    using namespace occ;
    if (n <= 0) return 0;
    OCC_REQUIRE(raw && mask && z_vals && rays && h_bgcolor && d_raw, "composite_backward: null argument");
    OCC_REQUIRE(S >= 1 && S <= kMaxChunks * kWave, "composite_backward: S=%d (at most %d samples per ray)", S,
                kMaxChunks * kWave);
    CompBwdParams prm;
    for (int c = 0; c < 3; c++) prm.bg[c] = h_bgcolor[c];
    int64_t blocks = (n + 3) / 4;
    if (blocks > (int64_t)kNumCU * 32) blocks = (int64_t)kNumCU * 32;
    hipLaunchKernelGGL(composite_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), raw, mask,
                       z_vals, rays, prm, n, S, g_rgb, g_acc, g_depth, d_raw, d_mask);
    return check_launch("composite_backward");
}

OCC_API int32_t occnerf_warp_backward_slices(int64_t n_samples) {
    // (bone, half) x slices workgroups, one per CU (128 KiB of LDS each): ~3 rounds over the chip
    if (n_samples <= 0) return 1;
    const int64_t by_work = (n_samples + 16383) / 16384;
    return (int32_t)(by_work < 16 ? by_work : 16);
}

OCC_API int occnerf_warp_backward(const float *rays, int64_t n, int32_t S, const float *z_vals, const float *g_mask,
                                  const float *Rs, const float *Ts, const float *vol, int32_t nb, int32_t G,
                                  const float *h_bbox_min, const float *h_bbox_scale, float *d_vol_part,
                                  float *d_rt_part, void *stream) {
    using namespace occ;
    OCC_REQUIRE(n > 0, "warp_backward: n=%lld", (long long)n);
    OCC_REQUIRE(rays && z_vals && g_mask && Rs && Ts && vol && h_bbox_min && h_bbox_scale && d_vol_part && d_rt_part,
                "warp_backward: null argument");
    OCC_REQUIRE(G == kVolG, "warp_backward: the LDS tiling is built for a %d^3 motion-weight volume, got %d", kVolG, G);
    OCC_REQUIRE(S >= 1 && nb >= 1 && nb <= 32, "warp_backward: bad sizes S=%d nb=%d", S, nb);
    WarpBwdParams prm;
    for (int c = 0; c < 3; c++) {
        prm.bmin[c] = h_bbox_min[c];
        prm.bscale[c] = h_bbox_scale[c];
    }
    const int64_t total = n * (int64_t)S;
    const int slices = occnerf_warp_backward_slices(total);
    const int64_t per = (total + slices - 1) / slices;
    hipLaunchKernelGGL(warp_backward_kernel, dim3(nb * 2, slices), dim3(256), 0, as_stream(stream), rays, n, S, z_vals,
                       g_mask, Rs, Ts, vol, nb, prm, per, d_vol_part, d_rt_part);
    return check_launch("warp_backward");
}

OCC_API int occnerf_agg_weights(const float *counter, const int32_t *knn, int64_t N, int32_t K, float *atts,
                                float *var, void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(counter && knn && atts && var, "agg_weights: null argument");
    OCC_REQUIRE(K >= 2 && K <= kAggMaxK, "agg_weights: K=%d (2..%d)", K, kAggMaxK);
    hipLaunchKernelGGL(agg_weights_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), counter,
                       knn, N, K, atts, var);
    return check_launch("agg_weights");
}
