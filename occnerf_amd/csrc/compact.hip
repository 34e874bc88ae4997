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

// ---- repeated samples -------------------------------------------------------------------------------------------
// Consecutive entries of a sample list often carry bitwise identical inputs: wherever the motion-weight sum is far below
// the reference's clamp (network.py:324, `/ fg_likelihood_mask.clamp(min=1e-4)`) the warped position collapses onto the
// origin, and every stage downstream is a pure per-sample function of it.  occnerf_repeat_heads marks the entries whose
// key (K dwords of a row) differs from the previous entry's, numbers them (inclusive scan), and writes the list of those
// "heads"; the per-sample kernels then run on the heads only and occnerf_scatter_raw_heads hands every entry its head's
// result.  Keys are compared as bit patterns, all K dwords: equal keys give equal results by construction, nothing is
// approximated.
struct RepeatFlag {
    const uint32_t *keys;       // row r at keys + r * stride
    int64_t stride;
    int K;
    const int32_t *rows;        // nullable: entry m is row rows[m]
    const int32_t *n_dev;       // entries beyond *n_dev count 0
    __device__ int operator()(const int &m) const {
        if (m >= *n_dev) return 0;
        if (m == 0) return 1;
        const uint32_t *a = keys + (rows ? (int64_t)rows[m] : (int64_t)m) * stride;
        const uint32_t *b = keys + (rows ? (int64_t)rows[m - 1] : (int64_t)(m - 1)) * stride;
        if (((K | (int)stride) & 3) == 0) {             // 16-byte pieces (rows of occnerf_sample_features)
            for (int k = 0; k < K; k += 4) {
                const uint4 x = *reinterpret_cast<const uint4 *>(a + k), y = *reinterpret_cast<const uint4 *>(b + k);
                if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) return 1;
            }
            return 0;
        }
        for (int k = 0; k < K; k++)
            if (a[k] != b[k]) return 1;
        return 0;
    }
};

// Wide keys (the 272-byte feature rows): 8 lanes per entry read 16-byte pieces of the entry's row and of its predecessor's
// (coalesced 128-byte segments; the functor above walks two rows per thread and reached 2.2 TB/s) and write the flag into
// the scan buffer, which is then summed in place.
__global__ void repeat_flags_wide_kernel(const uint4 *__restrict__ keys, int64_t stride4, int K4,
                                         const int32_t *__restrict__ rows, const int32_t *__restrict__ n_dev,
                                         int64_t N_max, int32_t *__restrict__ flags) {
    const int g = threadIdx.x & 7;
    const int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (m >= N_max) return;                                    // (whole 8-lane groups leave together)
    const int64_t n = *n_dev;
    int diff = 0;
    if (m < n && m > 0) {
        const uint4 *a = keys + (rows ? (int64_t)rows[m] : m) * stride4;
        const uint4 *b = keys + (rows ? (int64_t)rows[m - 1] : m - 1) * stride4;
        for (int k = g; k < K4; k += 8) {
            const uint4 x = a[k], y = b[k];
            diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
        }
    }
    diff |= __shfl_xor(diff, 1, 8);
    diff |= __shfl_xor(diff, 2, 8);
    diff |= __shfl_xor(diff, 4, 8);
    if (g == 0) flags[m] = m >= n ? 0 : (m == 0 ? 1 : diff);
}

// scan[m] = number of heads among entries 0..m.  heads[scan[m]-1] = (rows ? rows[m] : m) for every head m; *head_count =
// scan[n-1]; head_mask (nullable, zero-filled by the caller): 1.0f at the heads' rows.
__global__ void repeat_heads_kernel(const int32_t *__restrict__ scan, const int32_t *__restrict__ rows,
                                    const int32_t *__restrict__ n_dev, int32_t *__restrict__ heads,
                                    int32_t *__restrict__ head_count, float *__restrict__ head_mask) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = *n_dev;
    if (m == 0 && n <= 0) *head_count = 0;
    if (m >= n) return;
    const int32_t s = scan[m], prev = m ? scan[m - 1] : 0;
    if (s != prev) {
        const int32_t r = rows ? rows[m] : (int32_t)m;
        heads[s - 1] = r;
        if (head_mask) head_mask[r] = 1.0f;
    }
    if (m == n - 1) *head_count = s;
}

// raw_full[rows[m]] = (raw_h[head of m][0..3], raw_c[a][4]) with a = scanA ? scanA[m]-1 : m the entry's row in the
// feature kernel's outputs and head = scanB[a]-1 its row in the MLP's.
__global__ void scatter_raw_heads_kernel(const float *__restrict__ raw_h, const float *__restrict__ raw_c,
                                         const int32_t *__restrict__ rows, const int32_t *__restrict__ n_dev,
                                         const int32_t *__restrict__ scanA, const int32_t *__restrict__ scanB,
                                         float *__restrict__ raw_full) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (int64_t)*n_dev) return;
    const int64_t n = rows[m];
    const int64_t a = scanA ? scanA[m] - 1 : m;
    const int64_t h = scanB ? scanB[a] - 1 : a;
#pragma unroll
    for (int c = 0; c < 4; c++) raw_full[n * 5 + c] = raw_h[h * 5 + c];
    raw_full[n * 5 + 4] = raw_c[a * 5 + 4];
}

}  // namespace occ

OCC_API int64_t occnerf_repeat_heads_temp_bytes(int64_t N) {
    using namespace occ;
    if (N <= 0 || N >= (1ll << 31)) return 0;
    size_t bytes = 0;
    hipcub::CountingInputIterator<int> it(0);
    hipcub::TransformInputIterator<int, RepeatFlag, hipcub::CountingInputIterator<int>> flags(it, RepeatFlag{});
    if (hipcub::DeviceScan::InclusiveSum(nullptr, bytes, flags, (int *)nullptr, (int)N, (hipStream_t)0) != hipSuccess)
        return -1;
    size_t bytes2 = 0;
    if (hipcub::DeviceScan::InclusiveSum(nullptr, bytes2, (int *)nullptr, (int *)nullptr, (int)N, (hipStream_t)0) != hipSuccess)
        return -1;
    return (int64_t)(bytes > bytes2 ? bytes : bytes2);
}

OCC_API int occnerf_repeat_heads(const void *keys, int64_t stride_dwords, int32_t key_dwords, const int32_t *rows,
                                 const int32_t *n_dev, int64_t N_max, int32_t *scan, int32_t *heads,
                                 int32_t *head_count, float *head_mask, void *temp, int64_t temp_bytes,
                                 void *stream) {
    using namespace occ;
    OCC_REQUIRE(keys && n_dev && scan && heads && head_count && temp, "repeat_heads: null argument");
    OCC_REQUIRE(N_max > 0 && N_max < (1ll << 31), "repeat_heads: N=%lld out of range", (long long)N_max);
    OCC_REQUIRE(key_dwords > 0 && stride_dwords >= key_dwords, "repeat_heads: key of %d dwords in rows of %lld",
                key_dwords, (long long)stride_dwords);
    size_t bytes = (size_t)temp_bytes;
    hipError_t e;
    if (key_dwords >= 32 && ((key_dwords | stride_dwords) & 3) == 0) {
        const int64_t fblocks = (N_max * 8 + 255) / 256;
        OCC_REQUIRE(fblocks < (1ll << 31), "repeat_heads: N too large");
        hipLaunchKernelGGL(repeat_flags_wide_kernel, dim3((unsigned)fblocks), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const uint4 *>(keys), stride_dwords / 4, key_dwords / 4, rows, n_dev, N_max, scan);
        e = hipcub::DeviceScan::InclusiveSum(temp, bytes, scan, scan, (int)N_max, as_stream(stream));
    } else {
        hipcub::CountingInputIterator<int> it(0);
        hipcub::TransformInputIterator<int, RepeatFlag, hipcub::CountingInputIterator<int>> flags(
            it, RepeatFlag{reinterpret_cast<const uint32_t *>(keys), stride_dwords, key_dwords, rows, n_dev});
        e = hipcub::DeviceScan::InclusiveSum(temp, bytes, flags, scan, (int)N_max, as_stream(stream));
    }
    OCC_REQUIRE(e == hipSuccess, "repeat_heads: %s", hipGetErrorString(e));
    const int64_t blocks = (N_max + 255) / 256;
    hipLaunchKernelGGL(repeat_heads_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), scan, rows, n_dev,
                       heads, head_count, head_mask);
    return check_launch("repeat_heads");
}

OCC_API int occnerf_scatter_raw_heads(const float *raw_h, const float *raw_c, const int32_t *rows, const int32_t *n_dev,
                                      const int32_t *scanA, const int32_t *scanB, int64_t N_max, float *raw_full,
                                      void *stream) {
    using namespace occ;
    if (N_max <= 0) return 0;
    OCC_REQUIRE(raw_h && raw_c && rows && n_dev && raw_full, "scatter_raw_heads: null argument");
    const int64_t blocks = (N_max + 255) / 256;
    OCC_REQUIRE(blocks < (1ll << 31), "scatter_raw_heads: N too large");
    hipLaunchKernelGGL(scatter_raw_heads_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), raw_h, raw_c,
                       rows, n_dev, scanA, scanB, raw_full);
    return check_launch("scatter_raw_heads");
}

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
