// Precision policies of the split-operand MLP kernels (mlp.hip, nonrigid.hip): how an fp32 operand is cut into two 16-bit
// pieces for the bf16 / fp16 matrix pipe and what the three products are.
#pragma once

#include "common.h"

namespace occ {

typedef float split_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- precision policies of the split kernels ----------------------------------------------------------------------
// Bf16x3: hi = bf16(v), lo = bf16(v - hi): 16 significand bits per operand, 2^-17 of relative residue (opt-in 'bf16x3';
// meets the 1e-4 pixel gate on the random-init checkpoint only).
// F16x3 (round 5, 'f16x3'): the fp32-grade split.  fp16 carries 11 significand bits, so hi + lo = 22 bits and each dropped
// quantity is 2^-22 relative (fp32 itself: 2^-24) -- but only while the pieces stay NORMAL fp16 numbers, which is what the
// scales are for (measured first: tools/mfma_f16_probe.hip -- v_mfma_f32_32x32x16_f16 preserves subnormal inputs of both
// operands, and its fp32 accumulation over K = 256 is 2.6x closer to float64 than an fmaf chain):
//   activations travel scaled by kSx = 16 (exact: biases are scaled at the LDS copy, the two head dot products divide by
//     it): xh = f16(16 x), xl = f16(16 x - xh).  Full 22 bits for 2^-7 <= |x| < 4094; smaller values degrade to an ABSOLUTE
//     error of 2^-25 / 16 = 1.9e-9 (subnormal xl), nothing is flushed; larger ones saturate at 65504 / 16 (the ReLU is a
//     v_med3 with that bound: the documented domain of this mode -- hidden activations below 4 094);
//   weights: Wh = f16(W), and the lo piece is stored SCALED, Wl' = f16((W - Wh) 2^11), so that it is a normal number for
//     every |W| >= 2^-14; its product uses xh 2^-11 (a packed-half multiply per k-step, exact for xh >= 2^-3):
//       acc += Wh xh + Wh xl + Wl' (xh 2^-11)          -- one accumulator, three MFMAs, as bf16x3.
struct Bf16x3 {
    typedef bf16x8 V8;
    typedef __bf16 E;
    static constexpr float kSx = 1.0f;
    static __device__ __forceinline__ split_f32x16 mfma(V8 a, V8 b, split_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ E w_hi(float w) { return (E)w; }
    static __device__ __forceinline__ E w_lo(float w, E hi) { return (E)(w - (float)hi); }
    static __device__ __forceinline__ void split8(const float (&v)[8], V8 &hi, V8 &lo) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const E h = (E)v[i];
            hi[i] = h;
            lo[i] = (E)(v[i] - (float)h);
        }
    }
    static __device__ __forceinline__ V8 third(V8 xh) { return xh; }
    static __device__ __forceinline__ float relu(float a) { return fmaxf(a, 0.0f); }
    static __device__ __forceinline__ float sym(float a) { return a; }
    static constexpr bool kBounded = false;                 // no saturation: nothing to report
    static __device__ __forceinline__ void watch_hi(float &, const V8 &) {}
    static __device__ __forceinline__ void watch_abs(float &, float) {}
};
struct F16x3 {
    typedef f16x8 V8;
    typedef _Float16 E;
    static constexpr float kSx = 16.0f;
    static __device__ __forceinline__ split_f32x16 mfma(V8 a, V8 b, split_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ E w_hi(float w) { return (E)w; }
    static __device__ __forceinline__ E w_lo(float w, E hi) { return (E)((w - (float)hi) * 2048.0f); }
    static __device__ __forceinline__ void split8(const float (&v)[8], V8 &hi, V8 &lo) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const E h = (E)v[i];
            hi[i] = h;
            lo[i] = (E)(v[i] - (float)h);
        }
    }
    static __device__ __forceinline__ V8 third(V8 xh) { return xh * (E)0.00048828125f; }      // 2^-11
    static __device__ __forceinline__ float relu(float a) { return __builtin_amdgcn_fmed3f(a, 0.0f, 65504.0f); }
    static __device__ __forceinline__ float sym(float a) { return __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f); }
    // Round 6: the clamps above are the edge of this mode's DOMAIN (scaled hidden activations below 65 504, i.e. activations
    // below 4 094) and leaving it must not be silent.  A clamped value's hi piece IS 65 504, the largest fp16 number, so the ReLU
    // outputs are watched on their hi pieces as they are produced: a running packed-half maximum, four v_pk_max_f16 per eight
    // values, ONE register (carried as the bits of a float) -- watching the fp32 accumulators instead cost the non-rigid kernel
    // its two-workgroups-per-CU register budget (176 spills).  Signed inputs (sym) are few and watched in fp32.  A wave whose
    // maximum reached the bound sets the caller's flag word (split_report); the host re-renders the frame with the fp32 kernels
    // (Network.forward) -- see DESIGN.md section 3.
    static constexpr bool kBounded = true;
    static constexpr float kBound = 65504.0f;
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    struct Quad { f16x2 q[4]; };
    static __device__ __forceinline__ void watch_hi(float &m, const V8 &hi) {
        f16x2 mm = __builtin_bit_cast(f16x2, m);
        const Quad h = __builtin_bit_cast(Quad, hi);
#pragma unroll
        for (int i = 0; i < 4; i++) mm = __builtin_elementwise_max(mm, h.q[i]);
        m = __builtin_bit_cast(float, mm);
    }
    static __device__ __forceinline__ void watch_abs(float &m, float a) {      // (|a| >= bound -> both halves of the packed maximum)
        if (fabsf(a) >= kBound) m = __builtin_bit_cast(float, f16x2{(_Float16)65504.0f, (_Float16)65504.0f});
    }
    static __device__ __forceinline__ bool left_domain(float m) {
        const f16x2 mm = __builtin_bit_cast(f16x2, m);
        return (float)mm[0] >= kBound || (float)mm[1] >= kBound;
    }
};

// one atomic per wave that left the domain (none on the documented domain); flag may be NULL
template <typename P>
__device__ __forceinline__ void split_report(float amax, uint32_t *flag) {
    if constexpr (P::kBounded) {
        if (flag && __builtin_amdgcn_ballot_w64(P::left_domain(amax)) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
    }
}


}  // namespace occ
