// Shared host/device helpers for the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstring>

#include "../../include/occnerf_hip.h"

#define OCC_API extern "C" __attribute__((visibility("default")))

namespace occ {

constexpr int kWave = 64;       // CDNA wavefront
constexpr int kNumCU = 256;     // MI355X

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

// Tuning knobs (round 6: the two that select between shipped, bit-identical forms; the experiment variants of earlier
// rounds are gone from the library -- HISTORY.md keeps their numbers).  Each is read from its environment variable ONCE (first
// use), validated and clamped there, and can be switched at run time through the exported occnerf_experiment_knob() --
// launches never call getenv / atoi.
enum Knob : int { kKnobAggSlices = 0, kKnobGridXcd = 1, kKnobCount = 2 };
int knob(Knob k);

#define OCC_REQUIRE(cond, ...)         \
    do {                               \
        if (!(cond)) {                 \
            occ::set_error(__VA_ARGS__); \
            return 1;                  \
        }                              \
    } while (0)

// Load through a wave-uniform base pointer and a 32-bit byte offset.  hipcc then emits the
// SGPR-base form of global_load (address payload 4 B per lane instead of 8); only for tables < 4 GiB.
template <typename T>
__device__ __forceinline__ T ld32(const T *base, uint32_t byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// ---- multi-resolution grid: per-level constants computed on the HOST so that device
// exp2f accuracy never enters the result (oracle: oc_grid_level_params) ---------------
constexpr int kMaxLevels = 16;

struct GridLevels {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

// Per-level index mode of the fused D=4 encoder, decided on the host from the level's table size
// (wave-uniform in the kernel): the generic `index % hashmap_size` of gridencoder.cu:83 costs a
// ~35-instruction 32-bit urem per corner, 256 of them per sample.
//   kDense: all 4 dims fit (stride never exceeds the table) -> index < size, the modulo is the
//           identity;  kHashPow2: hashed level whose size is a power of two -> mask;
//   kGeneric: anything else -> the reference's loop and modulo.
enum GridIndexMode : uint32_t { kGridDense = 0, kGridHashPow2 = 1, kGridGeneric = 2 };
struct GridModes4 {
    uint32_t mode[kMaxLevels];
};
GridModes4 make_grid_modes_d4(uint32_t L, const GridLevels &lv, const uint32_t *h_level_sizes);

GridLevels make_grid_levels(uint32_t L, float S, uint32_t H);

// gridencoder.cu:50-84: dense index while the stride fits, xor-prime hash otherwise;
// uint32 wrap-around arithmetic throughout.
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_index(uint32_t gridtype, bool align_corners,
                                               uint32_t hashmap_size, uint32_t resolution,
                                               const uint32_t (&pos_grid)[D]) {
    constexpr uint32_t primes[7] = {1u,          2654435761u, 805459861u, 3674653429u,
                                    2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos_grid[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t h = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) h ^= pos_grid[d] * primes[d];
        index = h;
    }
    return index % hashmap_size;
}

// One level of the D=4, C=2 encoder used by the canonical MLP (occnerf_mlp.py:45): 16 corners,
// float2 features, explicit fma chain in corner order -- bit-exact with the oracle's
// oc_grid_encode_one and with gridencoder.cu:137-199.  `grid` points at this level's slice.
// The corner weights ((1*a0)*a1)*a2)*a3 are formed through shared partial products (identical
// roundings, 28 instead of 48 multiplies) and the index through per-axis partial terms.
// `grid` + entry0 is this level's slice: with a wave-uniform `grid` (the whole table) and a per-lane
// 32-bit entry0 the gathers take the SGPR-base + 32-bit-offset form of global_load.
// The two halves of a level: `taps` forms the cell fractions and puts the 16 corner gathers in flight, `reduce` forms the
// weights and sums in the reference's corner order.  Kernels that have other loads to issue call them apart.
struct LevelTaps4 {
    float fr[4];          // fractional position in the cell, per axis
    float2 v[16];         // corner values, corner idx = b0 | b1 << 1 | b2 << 2 | b3 << 3
};
__device__ __forceinline__ void encode_level_d4c2_taps(const float (&x)[4], const float2 *grid,
                                                       uint32_t hashmap_size, float scale, uint32_t resolution,
                                                       uint32_t mode, uint32_t entry0, LevelTaps4 &tp) {
    uint32_t pg[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float pos = __fmaf_rn(x[d], scale, 0.5f);
        const float fl = floorf(pos);
        pg[d] = (uint32_t)fl;
        pos -= fl;
        tp.fr[d] = pos;
    }
    // per-axis index terms t[d][lower/upper]
    uint32_t t[4][2];
    if (mode == kGridDense) {
        uint32_t stride = 1;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            t[d][0] = pg[d] * stride;
            t[d][1] = (pg[d] + 1) * stride;
            stride *= resolution + 1;
        }
    } else if (mode == kGridHashPow2) {
        constexpr uint32_t primes[4] = {1u, 2654435761u, 805459861u, 3674653429u};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            t[d][0] = pg[d] * primes[d];
            t[d][1] = (pg[d] + 1) * primes[d];
        }
    }
#pragma unroll
    for (uint32_t idx = 0; idx < 16; idx++) {
        const int b0 = idx & 1, b1 = (idx >> 1) & 1, b2 = (idx >> 2) & 1, b3 = (idx >> 3) & 1;
        uint32_t index;
        if (mode == kGridDense) {
            index = t[0][b0] + t[1][b1] + t[2][b2] + t[3][b3];
        } else if (mode == kGridHashPow2) {
            index = (t[0][b0] ^ t[1][b1] ^ t[2][b2] ^ t[3][b3]) & (hashmap_size - 1);
        } else {
            const uint32_t pl[4] = {pg[0] + b0, pg[1] + b1, pg[2] + b2, pg[3] + b3};
            index = grid_index<4>(0, false, hashmap_size, resolution, pl);
        }
        tp.v[idx] = ld32(grid, (entry0 + index) * 8u);
    }
}
__device__ __forceinline__ float2 encode_level_d4c2_reduce(const LevelTaps4 &tp) {
    float f[4][2];        // per axis: weight of the lower / upper cell
#pragma unroll
    for (int d = 0; d < 4; d++) {
        f[d][0] = __fsub_rn(1.f, tp.fr[d]);
        f[d][1] = tp.fr[d];
    }
    float2 r = make_float2(0.f, 0.f);
    float w01[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w01[i] = __fmul_rn(f[0][i & 1], f[1][i >> 1]);     // (a0 * a1)
#pragma unroll
    for (uint32_t idx = 0; idx < 16; idx++) {
        const int b0 = idx & 1, b1 = (idx >> 1) & 1, b2 = (idx >> 2) & 1, b3 = (idx >> 3) & 1;
        const float w = __fmul_rn(__fmul_rn(w01[b0 | (b1 << 1)], f[2][b2]), f[3][b3]);
        r.x = __fmaf_rn(w, tp.v[idx].x, r.x);
        r.y = __fmaf_rn(w, tp.v[idx].y, r.y);
    }
    return r;
}
__device__ __forceinline__ float2 encode_level_d4c2(const float (&x)[4], const float2 *grid,
                                                    uint32_t hashmap_size, float scale,
                                                    uint32_t resolution, uint32_t mode, uint32_t entry0 = 0) {
    LevelTaps4 tp;
    encode_level_d4c2_taps(x, grid, hashmap_size, scale, resolution, mode, entry0, tp);
    return encode_level_d4c2_reduce(tp);
}

// torch's fp32 2-norm of a 3-vector: sqrt(fma(z,z,fma(y,y,x*x))) (oracle: oc_norm3).
// sqrtf is the correctly rounded one on this toolchain; __fsqrt_rn is NOT (measured on
// gfx950 / ROCm 7.2: 16 % of inputs off by one ulp), despite its name.
__device__ __forceinline__ float norm3(float x, float y, float z) {
    return sqrtf(__fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x))));
}

// LDS-staged fp32 canonical MLP (mlp16.hip); its packed stream sits behind the Blob of mlp.hip in the
// buffer occnerf_canonical_mlp_pack fills.
int64_t mlp_lds_packed_floats();
int mlp_lds_pack(const float *const *h_W, const float *const *h_b, float *packed, hipStream_t st);
int mlp_lds_launch(const float *mlp_in, const int32_t *in_rows, int64_t N, const int32_t *n_dev, const float *packed,
                   float *raw, hipStream_t st);

// LDS-staged fp32 non-rigid MLP (nonrigid16.hip); its packed stream sits behind the NrBlob of nonrigid.hip.
int64_t nr_lds_packed_floats();
int nr_lds_pack(const float *const *h_W, const float *const *h_b, float *packed, hipStream_t st);
int nr_lds_launch(const float *xyz_in, int64_t N, const int32_t *rows, const int32_t *n_dev, const float *cond,
                  const float *h_hann, const float *W0, const float *b0, float *packed, float *xyz_out,
                  hipStream_t st);

}  // namespace occ
