// Live-sample list of a frame, built on the device.
//
// A sample's alpha is multiplied by its motion-weight sum (network.py:330); where that sum is exactly 0 the
// sample cannot contribute and the renderer does not evaluate it.  occnerf_live_rows writes the ascending list
// of the other samples and its length to DEVICE memory (hipCUB DeviceSelect: a utility pass over 4 B/sample),
// and the kernels downstream read the count from there -- the host never learns it, so a frame has no
// device->host round trip.  occnerf_scatter_raw puts the compact raw[M,5] rows back at their sample positions.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace occ {

struct LivePredicate {
    const float *mask;
    __device__ bool operator()(const int &i) const { return mask[i] != 0.0f; }
};

__global__ void scatter_raw_kernel(const float *__restrict__ raw_c, const int32_t *__restrict__ rows,
                                   const int32_t *__restrict__ n_dev, float *__restrict__ raw_full) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (int64_t)*n_dev) return;
    const int64_t n = rows[m];
#pragma unroll
    for (int c = 0; c < 5; c++) raw_full[n * 5 + c] = raw_c[m * 5 + c];
}

}  // namespace occ

OCC_API int64_t occnerf_live_rows_temp_bytes(int64_t N) {
    using namespace occ;
    if (N <= 0 || N >= (1ll << 31)) return 0;
    size_t bytes = 0;
    hipcub::CountingInputIterator<int> it(0);
    if (hipcub::DeviceSelect::If(nullptr, bytes, it, (int *)nullptr, (int *)nullptr, (int)N, LivePredicate{nullptr},
                                 (hipStream_t)0) != hipSuccess)
        return -1;
    return (int64_t)bytes;
}

OCC_API int occnerf_live_rows(const float *mask, int64_t N, int32_t *rows, int32_t *count, void *temp,
                              int64_t temp_bytes, void *stream) {
    using namespace occ;
    OCC_REQUIRE(mask && rows && count && temp, "live_rows: null argument");
    OCC_REQUIRE(N > 0 && N < (1ll << 31), "live_rows: N=%lld out of range", (long long)N);
    size_t bytes = (size_t)temp_bytes;
    hipcub::CountingInputIterator<int> it(0);
    const hipError_t e = hipcub::DeviceSelect::If(temp, bytes, it, rows, count, (int)N, LivePredicate{mask},
                                                  as_stream(stream));
    OCC_REQUIRE(e == hipSuccess, "live_rows: %s", hipGetErrorString(e));
    return check_launch("live_rows");
}

OCC_API int occnerf_scatter_raw(const float *raw_c, const int32_t *rows, const int32_t *n_dev, int64_t N_max,
                                float *raw_full, void *stream) {
    using namespace occ;
    if (N_max <= 0) return 0;
    OCC_REQUIRE(raw_c && rows && n_dev && raw_full, "scatter_raw: null argument");
    const int64_t blocks = (N_max + 255) / 256;
    OCC_REQUIRE(blocks < (1ll << 31), "scatter_raw: N too large");
    hipLaunchKernelGGL(scatter_raw_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), raw_c, rows, n_dev,
                       raw_full);
    return check_launch("scatter_raw");
}
