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

}  // namespace occ

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
