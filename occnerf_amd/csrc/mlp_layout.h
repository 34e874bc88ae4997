// Layout of the canonical MLP's packed operands, shared by the renderer's kernels (mlp.hip) and the training step's fused
// trunk forward (trunks.hip): the register <-> feature map of the transposed 32x32 MFMA scheme, the fp32 blob of biases and
// head rows, the k-step counts of the 16-wide (2-byte operand) kernels and the LDS-DMA helper.
#pragma once

#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWidth = 256;
constexpr int kOB = kWidth / 32;     // 8 output blocks of 32 features
constexpr int kInGeo = 68, kInRgb = 131;
constexpr int kXRegs = 36;           // 34 input k-steps (68 features over two half-waves) + 2 pad
constexpr int kG_L0Geo = kXRegs / 4;             // 9 groups of 4 k-steps
constexpr int kG_Hidden = kWidth / 2 / 4;        // 32
constexpr int kG_L0Rgb = (32 + kXRegs) / 4;      // 17

// packed blob layout, in floats (every offset a multiple of 4 -> 16-byte aligned)
constexpr int64_t wsz(int groups, int ob) { return (int64_t)groups * ob * 64 * 4; }
struct Blob {
    static constexpr int64_t kGeoL0W = 0;
    static constexpr int64_t kGeoL0B = kGeoL0W + wsz(kG_L0Geo, kOB);
    static constexpr int64_t kGeoHW = kGeoL0B + kWidth;                      // 3 x (W, B)
    static constexpr int64_t kHiddenStride = wsz(kG_Hidden, kOB) + kWidth;
    static constexpr int64_t kGeoHeadW = kGeoHW + 3 * kHiddenStride;
    static constexpr int64_t kGeoHeadB = kGeoHeadW + wsz(kG_Hidden, 2);
    static constexpr int64_t kSigmaW = kGeoHeadB + 64;
    static constexpr int64_t kSigmaB = kSigmaW + kWidth;
    static constexpr int64_t kRgbL0W = kSigmaB + 4;
    static constexpr int64_t kRgbL0B = kRgbL0W + wsz(kG_L0Rgb, kOB);
    static constexpr int64_t kRgbHW = kRgbL0B + kWidth;
    static constexpr int64_t kOutW = kRgbHW + 3 * kHiddenStride;
    static constexpr int64_t kOutB = kOutW + 3 * kWidth;
    static constexpr int64_t kTotal = kOutB + 4;
};

// ---------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------
enum LayerKind { kL0Geo = 0, kHidden = 1, kGeoHead = 2, kL0Rgb = 3 };

// feature of the layer's torch-layout input that k-step `t` carries in half-wave `h`
// (-1: zero weight)
__host__ __device__ inline int slot_feature(int kind, int t, int h) {
    const int blk = t >> 4, r = t & 15;
    const int cd = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // 32x32 C/D row of (reg, half)
    switch (kind) {
        case kL0Geo: return t < 34 ? h * 34 + t : -1;
        case kHidden:
        case kGeoHead: return cd;
        case kL0Rgb: {
            if (t < 32) return cd;                 // geometry features h[1:65] -> inputs 0..63
            if (t >= 66) return -1;
            const int m = h * 34 + (t - 32);       // position in [agg35, var, enc32]
            if (m < 35) return 64 + m;             // aggregated point features
            if (m == 35) return -1;                // var is not an input of the colour trunk
            return 64 + 35 + (m - 36);             // hash encoding
        }
    }
    return -1;
}

// torch-layout output row computed in packed row `row` (-1: padding row)
__host__ __device__ inline int out_row(int kind, int row, int out_dim) {
    if (kind == kGeoHead) return row < 64 ? row + 1 : -1;   // row 0 (sigma) handled by dot_rows
    return row < out_dim ? row : -1;
}

// k-steps of 16 (8 per half-wave) of the 2-byte-operand kernels
constexpr int kS_L0Geo = 5;                 // ceil(34 / 8) k-steps of 16 (8 per half-wave)
constexpr int kS_Hidden = kWidth / 16;      // 16
constexpr int kS_L0Rgb = 4 + 5;             // 64 geometry features + 34 x-slots

// LDS-DMA of 64 x 16 B: wave-uniform source base (SGPR pair) + 32-bit lane offset ("saddr" form -- a
// 64-bit VGPR address per lane costs the issuing SIMD ~40 cycles of matrix-pipe time per instruction on
// gfx950), wave-uniform LDS destination in M0 (lane i lands at +16 i).
__device__ __forceinline__ void glds16(const void *gbase, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

}  // namespace occ
