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

// ---- distinct samples across the whole list ------------------------------------------------------------------
// Run-length elimination leaves the repeats that are not neighbours in the list (collapsed samples of different rays).
// occnerf_unique_heads finds them with an open-addressing table of list positions: an entry hashes its key, claims the
// first free slot of its probe sequence with one atomicCAS or meets an earlier claimant there, compares the FULL keys as bit
// patterns and, if they are equal, takes that entry as its representative.  Which of several equal entries wins the slot is
// a race, and does not matter: equal keys give equal results, so the pixels do not depend on the winner; the NUMBER of
// representatives is the number of distinct keys (an entry whose 32 probes all meet other keys stays its own
// representative -- exact, merely one row more).
__device__ __forceinline__ uint64_t mix64(uint64_t h) {
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

template <int G /* lanes per entry: 1 for narrow keys, 8 for rows */>
__global__ void unique_insert_kernel(const uint32_t *__restrict__ keys, int64_t stride, int K,
                                     const int32_t *__restrict__ heads, const int32_t *__restrict__ n_dev,
                                     int32_t *__restrict__ table, uint32_t tmask, int32_t *__restrict__ rep) {
    const int g = threadIdx.x % G;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (j >= (int64_t)*n_dev) return;                       // (whole groups leave together)
    const uint32_t *a = keys + (heads ? (int64_t)heads[j] : j) * stride;
    uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(g + 1);
    for (int k = g; k < K; k += G) h = mix64(h ^ ((uint64_t)a[k] + ((uint64_t)(k + 1) << 32)));
    if (G > 1) {
        h ^= __shfl_xor(h, 1, G);
        h ^= __shfl_xor(h, 2, G);
        h ^= __shfl_xor(h, 4, G);
    }
    uint32_t slot = (uint32_t)mix64(h) & tmask;
    int32_t r = (int32_t)j;
    for (int probe = 0; probe < 32; probe++) {
        int32_t prev = 0;
        if (g == 0) prev = atomicCAS(&table[slot], -1, (int32_t)j);
        if (G > 1) prev = __shfl(prev, 0, G);
        if (prev == -1) break;                              // claimed: its own representative
        const uint32_t *b = keys + (heads ? (int64_t)heads[prev] : (int64_t)prev) * stride;
        int diff = 0;
        for (int k = g; k < K; k += G) diff |= a[k] != b[k];
        if (G > 1) {
            diff |= __shfl_xor(diff, 1, G);
            diff |= __shfl_xor(diff, 2, G);
            diff |= __shfl_xor(diff, 4, G);
        }
        if (!diff) {
            r = prev;
            break;
        }
        slot = (slot + 1) & tmask;
    }
    if (g == 0) rep[j] = r;
}

struct IsRepresentative {
    const int32_t *rep, *n_dev;
    __device__ int operator()(const int &j) const { return j < *n_dev && rep[j] == j; }
};

// heads_out[scan2[j]-1] = row of representative j; rep[j] <- position of j's representative in heads_out; *count_out
__global__ void unique_finish_kernel(const int32_t *__restrict__ heads, const int32_t *__restrict__ n_dev,
                                     const int32_t *__restrict__ scan2, int32_t *__restrict__ rep,
                                     int32_t *__restrict__ heads_out, int32_t *__restrict__ count_out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = *n_dev;
    if (j == 0 && n <= 0) *count_out = 0;
    if (j >= n) return;
    const int32_t r = rep[j];
    if (r == (int32_t)j) heads_out[scan2[j] - 1] = heads ? heads[j] : (int32_t)j;
    if (j == n - 1) *count_out = scan2[j];
    rep[j] = scan2[r] - 1;          // (reads rep of no other entry: in place)
}

__global__ void unique_remap_kernel(int32_t *__restrict__ scan, const int32_t *__restrict__ n_scan_dev,
                                    const int32_t *__restrict__ pos) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (int64_t)*n_scan_dev) return;
    scan[m] = pos[scan[m] - 1] + 1;
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

static inline int64_t unique_table_entries(int64_t cap) {
    int64_t t = 1024;
    while (t < cap) t <<= 1;
    return t;
}

OCC_API int64_t occnerf_unique_heads_temp_bytes(int64_t cap) {
    using namespace occ;
    if (cap <= 0 || cap >= (1ll << 30)) return 0;
    size_t bytes = 0;
    hipcub::CountingInputIterator<int> it(0);
    hipcub::TransformInputIterator<int, IsRepresentative, hipcub::CountingInputIterator<int>> flags(it, IsRepresentative{});
    if (hipcub::DeviceScan::InclusiveSum(nullptr, bytes, flags, (int *)nullptr, (int)cap, (hipStream_t)0) != hipSuccess)
        return -1;
    const int64_t words = unique_table_entries(cap) + 2 * ((cap + 63) & ~63ll);
    return words * 4 + (int64_t)((bytes + 255) & ~(size_t)255);
}

OCC_API int occnerf_unique_heads(const void *keys, int64_t stride_dwords, int32_t key_dwords, const int32_t *heads,
                                 const int32_t *n_dev, int64_t cap, int32_t *heads_out, int32_t *count_out,
                                 int32_t *scan, const int32_t *n_scan_dev, int64_t scan_cap, void *temp,
                                 int64_t temp_bytes, void *stream) {
    using namespace occ;
    OCC_REQUIRE(keys && n_dev && heads_out && count_out && temp, "unique_heads: null argument");
    OCC_REQUIRE(cap > 0 && cap < (1ll << 30), "unique_heads: cap=%lld out of range", (long long)cap);
    OCC_REQUIRE(key_dwords > 0 && stride_dwords >= key_dwords, "unique_heads: key of %d dwords in rows of %lld", key_dwords,
                (long long)stride_dwords);
    OCC_REQUIRE(!scan == !n_scan_dev, "unique_heads: scan and its length come together");
    OCC_REQUIRE(temp_bytes >= occnerf_unique_heads_temp_bytes(cap), "unique_heads: temp too small");
    const int64_t T = unique_table_entries(cap), capr = (cap + 63) & ~63ll;
    int32_t *table = reinterpret_cast<int32_t *>(temp), *rep = table + T, *scan2 = rep + capr;
    void *cub_temp = scan2 + capr;
    size_t cub_bytes = (size_t)temp_bytes - (size_t)(T + 2 * capr) * 4;
    hipStream_t st = as_stream(stream);
    OCC_REQUIRE(hipMemsetAsync(table, 0xFF, (size_t)T * 4, st) == hipSuccess, "unique_heads: memset");
    const uint32_t *k32 = reinterpret_cast<const uint32_t *>(keys);
    if (key_dwords >= 32)
        hipLaunchKernelGGL(unique_insert_kernel<8>, dim3((unsigned)((cap * 8 + 255) / 256)), dim3(256), 0, st, k32,
                           stride_dwords, key_dwords, heads, n_dev, table, (uint32_t)(T - 1), rep);
    else
        hipLaunchKernelGGL(unique_insert_kernel<1>, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, k32,
                           stride_dwords, key_dwords, heads, n_dev, table, (uint32_t)(T - 1), rep);
    hipcub::CountingInputIterator<int> it(0);
    hipcub::TransformInputIterator<int, IsRepresentative, hipcub::CountingInputIterator<int>> flags(
        it, IsRepresentative{rep, n_dev});
    const hipError_t e = hipcub::DeviceScan::InclusiveSum(cub_temp, cub_bytes, flags, scan2, (int)cap, st);
    OCC_REQUIRE(e == hipSuccess, "unique_heads: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(unique_finish_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, heads, n_dev, scan2, rep,
                       heads_out, count_out);
    if (scan)
        hipLaunchKernelGGL(unique_remap_kernel, dim3((unsigned)((scan_cap + 255) / 256)), dim3(256), 0, st, scan, n_scan_dev,
                           rep);
    return check_launch("unique_heads");
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
