// Image assembly after the renderer: run.py:46-63 unpack_to_image (scatter of the
// per-ray colours into H x W by ray_mask, background fill) + image_util.py:19-20 to_8b_image
// ((255. * clip(x, 0, 1)).astype(uint8): fp32 product, truncation), as ONE kernel, so that only uint8 pixels cross
// PCIe (3 B per pixel instead of 16 B per ray).
//
// One thread per pixel.  The rays of a frame are the pixels of ray_mask in ascending order, so a pixel finds its ray
// (or learns that it has none) by a binary search in ray_index[R] -- 18 L2-resident probes at 512 x 512; no
// pixel -> ray map has to be built or kept per camera.  Bound: HBM streaming, 16 B per ray in, 3 (+3) B per pixel out.
#include "common.h"

namespace occ {

struct ImageParams {
    float bg[3];       // cfg.bgcolor / 255 as float32 (np.full(..., dtype='float32'))
};

__device__ __forceinline__ uint8_t to_8b(float x) {
    x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);          // np.clip (NaN propagates in numpy; not produced by the renderer)
    return (uint8_t)__fmul_rn(255.0f, x);
}

__global__ __launch_bounds__(256) void assemble_image_kernel(const float *__restrict__ rgb, const float *__restrict__ alpha,
                                                             const int64_t *__restrict__ ray_index, int64_t R, int64_t n_pixels,
                                                             ImageParams prm, uint8_t *__restrict__ out_rgb,
                                                             uint8_t *__restrict__ out_alpha) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    int64_t lo = 0, hi = R;                                 // first ray with ray_index >= p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ray_index[mid] < p) lo = mid + 1;
        else hi = mid;
    }
    const bool hit = lo < R && ray_index[lo] == p;
    float c0 = prm.bg[0], c1 = prm.bg[1], c2 = prm.bg[2], a = 0.0f;
    if (hit) {
        c0 = rgb[lo * 3];
        c1 = rgb[lo * 3 + 1];
        c2 = rgb[lo * 3 + 2];
        a = alpha ? alpha[lo] : 0.0f;
    }
    out_rgb[p * 3] = to_8b(c0);
    out_rgb[p * 3 + 1] = to_8b(c1);
    out_rgb[p * 3 + 2] = to_8b(c2);
    if (out_alpha) {
        const uint8_t q = to_8b(a);
        out_alpha[p * 3] = q;
        out_alpha[p * 3 + 1] = q;
        out_alpha[p * 3 + 2] = q;
    }
}

}  // namespace occ

OCC_API int occnerf_assemble_image(const float *rgb, const float *alpha, const int64_t *ray_index, int64_t R,
                                   int32_t height, int32_t width, const float *h_bgcolor01, uint8_t *out_rgb,
                                   uint8_t *out_alpha, void *stream) {
    using namespace occ;
    OCC_REQUIRE(height > 0 && width > 0 && R >= 0, "assemble_image: bad sizes");
    OCC_REQUIRE(h_bgcolor01 && out_rgb && (R == 0 || (rgb && ray_index)) && (!out_alpha || R == 0 || alpha),
                "assemble_image: null argument");
    ImageParams prm;
    for (int c = 0; c < 3; c++) prm.bg[c] = h_bgcolor01[c];
    const int64_t n = (int64_t)height * width;
    hipLaunchKernelGGL(assemble_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), rgb, alpha,
                       ray_index, R, n, prm, out_rgb, out_alpha);
    return check_launch("assemble_image");
}
