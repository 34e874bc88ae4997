// Per-frame ray generation on the device: the dataset-side numpy stage
// camera_util.py:133-160 (get_rays_from_KRT) + :163-212 (rays_intersect_3d_bbox) as called at
// tpose.py:155-172 / freeview.py:190-208, one thread per pixel.  numpy's result dtype follows the camera's:
// the reference's synthetic render cameras (tpose.py:66-84, freeview.py) are float32, so pixel -> ray runs in
// float32 there, calibrated dataset cameras are float64; `f32_camera` selects which one is mirrored.  The box
// stage is float64 in both cases (the bbox + [-0.01, 0.01] is float64).  Outputs are float32.
//
// Kept quirks of the reference: directions are not normalised; direction components with |d| < 1e-5 are
// REPLACED by 1e-5 and that clamped direction is what the renderer later receives (the function clamps its
// argument in place, camera_util.py:184); a ray is kept when exactly two of its six slab-plane intersections
// lie inside the box grown by 0.01 (+1e-6); near/far are the two hit distances in units of |d|.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace occ {

struct RayCam {
    double kinv[9];     // K^-1, row major
    double r[9];        // R (world -> camera), row major
    double t[3];
    double o[3];        // camera centre -R^T T
    double lo[3], hi[3];   // bbox already grown by 0.01
};

template <typename F>
__device__ __forceinline__ void pixel_ray(const RayCam &cam, int col, int row, double (&o)[3], double (&d)[3]) {
    const F px = (F)col, py = (F)row;
    F camv[3];
#pragma unroll
    for (int c = 0; c < 3; c++)          // np.dot(xy1, inv(K).T) - T
        camv[c] = ((px * (F)cam.kinv[c * 3] + py * (F)cam.kinv[c * 3 + 1]) + (F)cam.kinv[c * 3 + 2]) - (F)cam.t[c];
#pragma unroll
    for (int c = 0; c < 3; c++) {        // np.dot(., R) - rays_o
        const F w = (camv[0] * (F)cam.r[c] + camv[1] * (F)cam.r[3 + c]) + camv[2] * (F)cam.r[6 + c];
        F dc = w - (F)cam.o[c];
        if (fabs(dc) < (F)1e-5) dc = (F)1e-5;                     // camera_util.py:184
        d[c] = (double)dc;
        o[c] = (double)(F)cam.o[c];
    }
}

template <typename F>
__global__ void gen_rays_kernel(RayCam cam_in, int H, int W, float *__restrict__ rays8,
                                uint8_t *__restrict__ mask) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    RayCam cam = cam_in;
    double d[3];
    pixel_ray<F>(cam, p % W, p / W, cam.o, d);
    // six plane intersections in the reference's order: min x, y, z, max x, y, z
    int hits = 0;
    double dist[2] = {0.0, 0.0};
    const double nrm = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(d[0], d[0]), __dmul_rn(d[1], d[1])), __dmul_rn(d[2], d[2])));
#pragma unroll
    for (int f = 0; f < 6; f++) {
        const int a = f % 3;
        const double bound = f < 3 ? cam.lo[a] : cam.hi[a];
        const double tt = __ddiv_rn(bound - cam.o[a], d[a]);
        double q[3];
        bool in = true;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            q[c] = __dadd_rn(__dmul_rn(tt, d[c]), cam.o[c]);
            in = in && q[c] >= cam.lo[c] - 1e-6 && q[c] <= cam.hi[c] + 1e-6;
        }
        if (in) {
            const double e0 = q[0] - cam.o[0], e1 = q[1] - cam.o[1], e2 = q[2] - cam.o[2];
            const double l = __ddiv_rn(sqrt(__dadd_rn(__dadd_rn(__dmul_rn(e0, e0), __dmul_rn(e1, e1)), __dmul_rn(e2, e2))), nrm);
            if (hits < 2) dist[hits] = l;
            hits++;
        }
    }
    const bool keep = hits == 2;
    mask[p] = keep ? 1 : 0;
    float *o = rays8 + (size_t)p * 8;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o[c] = (float)cam.o[c];
        o[3 + c] = (float)d[c];
    }
    o[6] = keep ? (float)fmin(dist[0], dist[1]) : 0.0f;
    o[7] = keep ? (float)fmax(dist[0], dist[1]) : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// Render order of a frame's rays: a 2-D Morton walk of their directions (occnerf_amd/rayorder.py has the same
// construction in torch ops -- ~35 launches and a stable argsort per frame; a free-view orbit has a new camera every
// frame, so the order is per-frame work).  Here: three small kernels + one radix sort, no host round trip, and
// DETERMINISTIC (fixed-order double-precision reductions: every rank of a sharded render computes the same walk).
//   1. sum of the unit directions (64 partial sums)            -> mean direction m
//   2. basis e1, e2 of the plane normal to m (every block recomputes it from the partials), min / max of
//      (d.e1, d.e2) over the rays (64 partial boxes)
//   3. key = Morton interleave of the two coordinates quantised to 16 bits of the common span
//   4. hipcub::DeviceRadixSort::SortPairs(key, ray index): stable, ascending
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRoBlocks = 64;

__device__ __forceinline__ void ro_unit(const float *__restrict__ d, float (&u)[3]) {
    const float n = fmaxf(norm3(d[0], d[1], d[2]), 1e-20f);
    u[0] = d[0] / n;
    u[1] = d[1] / n;
    u[2] = d[2] / n;
}

__global__ __launch_bounds__(256) void ray_order_sum_kernel(const float *__restrict__ dirs, int64_t R, int64_t stride,
                                                            double *__restrict__ partial) {
    __shared__ double red[256][3];
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < R; i += (int64_t)kRoBlocks * 256) {
        float u[3];
        ro_unit(dirs + i * stride, u);
        s[0] += u[0];
        s[1] += u[1];
        s[2] += u[2];
    }
    for (int c = 0; c < 3; c++) red[threadIdx.x][c] = s[c];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int c = 0; c < 3; c++) red[threadIdx.x][c] += red[threadIdx.x + w][c];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x];
}

// basis of the plane normal to the mean direction: e1 = normalize(m x axis), axis = the coordinate axis m leans on
// least; e2 = normalize(m x e1)
__device__ __forceinline__ void ro_basis(const double *__restrict__ partial, int64_t R, float (&e1)[3], float (&e2)[3]) {
    double md[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < kRoBlocks; b++)
        for (int c = 0; c < 3; c++) md[c] += partial[b * 3 + c];
    float m[3];
    for (int c = 0; c < 3; c++) m[c] = (float)(md[c] / (double)R);
    int a = 0;
    if (fabsf(m[1]) < fabsf(m[a])) a = 1;
    if (fabsf(m[2]) < fabsf(m[a])) a = 2;
    const float ax[3] = {a == 0 ? 1.f : 0.f, a == 1 ? 1.f : 0.f, a == 2 ? 1.f : 0.f};
    float c1[3] = {m[1] * ax[2] - m[2] * ax[1], m[2] * ax[0] - m[0] * ax[2], m[0] * ax[1] - m[1] * ax[0]};
    float n = fmaxf(norm3(c1[0], c1[1], c1[2]), 1e-20f);
    for (int c = 0; c < 3; c++) e1[c] = c1[c] / n;
    float c2[3] = {m[1] * e1[2] - m[2] * e1[1], m[2] * e1[0] - m[0] * e1[2], m[0] * e1[1] - m[1] * e1[0]};
    n = fmaxf(norm3(c2[0], c2[1], c2[2]), 1e-20f);
    for (int c = 0; c < 3; c++) e2[c] = c2[c] / n;
}

__device__ __forceinline__ void ro_uv(const float *__restrict__ d, const float (&e1)[3], const float (&e2)[3], float &u,
                                      float &v) {
    float x[3];
    ro_unit(d, x);
    u = __fmaf_rn(x[2], e1[2], __fmaf_rn(x[1], e1[1], x[0] * e1[0]));
    v = __fmaf_rn(x[2], e2[2], __fmaf_rn(x[1], e2[1], x[0] * e2[0]));
}

__global__ __launch_bounds__(256) void ray_order_box_kernel(const float *__restrict__ dirs, int64_t R, int64_t stride,
                                                            const double *__restrict__ partial, float *__restrict__ boxes) {
    __shared__ float red[256][4];
    float e1[3], e2[3];
    ro_basis(partial, R, e1, e2);
    float lo0 = INFINITY, lo1 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < R; i += (int64_t)kRoBlocks * 256) {
        float u, v;
        ro_uv(dirs + i * stride, e1, e2, u, v);
        lo0 = fminf(lo0, u);
        hi0 = fmaxf(hi0, u);
        lo1 = fminf(lo1, v);
        hi1 = fmaxf(hi1, v);
    }
    red[threadIdx.x][0] = lo0;
    red[threadIdx.x][1] = lo1;
    red[threadIdx.x][2] = hi0;
    red[threadIdx.x][3] = hi1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[threadIdx.x][0] = fminf(red[threadIdx.x][0], red[threadIdx.x + w][0]);
            red[threadIdx.x][1] = fminf(red[threadIdx.x][1], red[threadIdx.x + w][1]);
            red[threadIdx.x][2] = fmaxf(red[threadIdx.x][2], red[threadIdx.x + w][2]);
            red[threadIdx.x][3] = fmaxf(red[threadIdx.x][3], red[threadIdx.x + w][3]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) boxes[blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x];
}

__device__ __forceinline__ uint32_t ro_spread16(uint32_t x) {      // 16 bits -> every second bit
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

__global__ __launch_bounds__(256) void ray_order_keys_kernel(const float *__restrict__ dirs, int64_t R, int64_t stride,
                                                             const double *__restrict__ partial,
                                                             const float *__restrict__ boxes, uint32_t *__restrict__ keys,
                                                             int64_t *__restrict__ index) {
    float e1[3], e2[3];
    ro_basis(partial, R, e1, e2);
    float lo0 = INFINITY, lo1 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY;
    for (int b = 0; b < kRoBlocks; b++) {
        lo0 = fminf(lo0, boxes[b * 4 + 0]);
        lo1 = fminf(lo1, boxes[b * 4 + 1]);
        hi0 = fmaxf(hi0, boxes[b * 4 + 2]);
        hi1 = fmaxf(hi1, boxes[b * 4 + 3]);
    }
    const float span = fmaxf(fmaxf(hi0 - lo0, hi1 - lo1), 1e-20f);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    float u, v;
    ro_uv(dirs + i * stride, e1, e2, u, v);
    const int q0 = min(max((int)((u - lo0) / span * 65535.0f), 0), 65535);
    const int q1 = min(max((int)((v - lo1) / span * 65535.0f), 0), 65535);
    keys[i] = ro_spread16((uint32_t)q0) | (ro_spread16((uint32_t)q1) << 1);
    index[i] = i;
}

static size_t ro_align(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace occ

OCC_API int64_t occnerf_ray_order_temp_bytes(int64_t R) {
    using namespace occ;
    if (R <= 0 || R >= (1ll << 31)) return -1;
    size_t cub = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int64_t *)nullptr,
                                           (int64_t *)nullptr, (int)R, 0, 32, (hipStream_t)0) != hipSuccess)
        return -1;
    return (int64_t)(ro_align(kRoBlocks * 3 * sizeof(double)) + ro_align(kRoBlocks * 4 * sizeof(float)) +
                     2 * ro_align((size_t)R * 4) + ro_align((size_t)R * 8) + ro_align(cub));
}

/* dirs: R ray directions, `stride` floats apart (3 for a [R,3] array, 8 for columns 3..5 of rays8); order[R] int64:
 * the permutation that walks the rays along the Morton curve (order[j] = index of the j-th ray of the walk). */
OCC_API int occnerf_ray_order(const float *dirs, int64_t R, int64_t stride, int64_t *order, void *temp, int64_t temp_bytes,
                              void *stream) {
    using namespace occ;
    if (R == 0) return 0;
    OCC_REQUIRE(dirs && order && temp, "ray_order: null argument");
    OCC_REQUIRE(R > 0 && R < (1ll << 31) && stride >= 3, "ray_order: R=%lld stride=%lld", (long long)R, (long long)stride);
    OCC_REQUIRE(temp_bytes >= occnerf_ray_order_temp_bytes(R), "ray_order: temp too small");
    char *p = reinterpret_cast<char *>(temp);
    double *partial = reinterpret_cast<double *>(p);
    p += ro_align(kRoBlocks * 3 * sizeof(double));
    float *boxes = reinterpret_cast<float *>(p);
    p += ro_align(kRoBlocks * 4 * sizeof(float));
    uint32_t *keys_in = reinterpret_cast<uint32_t *>(p);
    p += ro_align((size_t)R * 4);
    uint32_t *keys_out = reinterpret_cast<uint32_t *>(p);
    p += ro_align((size_t)R * 4);
    int64_t *index = reinterpret_cast<int64_t *>(p);
    p += ro_align((size_t)R * 8);
    size_t cub = (size_t)(reinterpret_cast<char *>(temp) + temp_bytes - p);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(ray_order_sum_kernel, dim3(kRoBlocks), dim3(256), 0, st, dirs, R, stride, partial);
    hipLaunchKernelGGL(ray_order_box_kernel, dim3(kRoBlocks), dim3(256), 0, st, dirs, R, stride, partial, boxes);
    hipLaunchKernelGGL(ray_order_keys_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, dirs, R, stride, partial, boxes,
                       keys_in, index);
    if (int rc = check_launch("ray_order")) return rc;
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(p, cub, (const uint32_t *)keys_in, keys_out, (const int64_t *)index,
                                                            order, (int)R, 0, 32, st);
    OCC_REQUIRE(e == hipSuccess, "ray_order: radix sort: %s", hipGetErrorString(e));
    return 0;
}

OCC_API int occnerf_gen_rays(const double *h_Kinv, const double *h_R, const double *h_T, int32_t f32_camera,
                             int32_t H, int32_t W, const double *h_bbox_min, const double *h_bbox_max,
                             float *rays8, uint8_t *mask, void *stream) {
    using namespace occ;
    OCC_REQUIRE(h_Kinv && h_R && h_T && h_bbox_min && h_bbox_max && rays8 && mask, "gen_rays: null argument");
    OCC_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "gen_rays: bad image size %d x %d", H, W);
    RayCam cam;
    for (int i = 0; i < 9; i++) {
        cam.kinv[i] = h_Kinv[i];
        cam.r[i] = h_R[i];
    }
    for (int c = 0; c < 3; c++) {
        cam.t[c] = h_T[c];
        if (f32_camera)                                                                  // -R^T T in the camera's dtype
            cam.o[c] = -(((float)h_R[c] * (float)h_T[0] + (float)h_R[3 + c] * (float)h_T[1]) + (float)h_R[6 + c] * (float)h_T[2]);
        else
            cam.o[c] = -((h_R[c] * h_T[0] + h_R[3 + c] * h_T[1]) + h_R[6 + c] * h_T[2]);
        cam.lo[c] = h_bbox_min[c] - 0.01;
        cam.hi[c] = h_bbox_max[c] + 0.01;
    }
    const int n = H * W;
    if (f32_camera)
        hipLaunchKernelGGL(gen_rays_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), cam, H, W,
                           rays8, mask);
    else
        hipLaunchKernelGGL(gen_rays_kernel<double>, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), cam, H, W,
                           rays8, mask);
    return check_launch("gen_rays");
}
