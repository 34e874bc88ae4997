// Optimiser step of the training loop on the device (the reference's
// trainer.py:248-249: torch.nn.utils.clip_grad_norm_(parameters, 1.0) followed by torch.optim.Adam.step() with
// per-group learning rates, optimizer.py:12-43).
//
// One multi-tensor pass instead of torch's per-operation foreach kernels: the ~60 parameter tensors of the
// renderer (45 M decoder weights, 15.5 M hash-table entries, the MLPs) are cut into fixed-size chunks; every
// workgroup owns one chunk of one tensor.  Kernel 1 forms the squared gradient norm per chunk, kernel 2 sums the
// partials (one workgroup: fixed order, deterministic), kernel 3 applies Adam with the clip coefficient
// min(1, max_norm / (norm + 1e-6)) read from device memory -- no host round trip anywhere in the step.
// Bound: HBM streaming, 4 reads + 3 writes of 4 bytes per parameter.
#include "common.h"

namespace occ {
namespace opt {

struct AdamTensor {          // one row of the device-side table (56 bytes)
    float *p;
    const float *g;
    float *m;
    float *v;
    int64_t n;
    float lr;
    float bc1;            // 1 - beta1^t with THIS tensor's step count t (torch.optim.Adam keeps a step per parameter:
    float bc2_sqrt;       // sqrt(1 - beta2^t)      a parameter may start receiving gradients later than the others)
    float pad_;
};
static_assert(sizeof(AdamTensor) == 56, "table row layout is part of the ABI");

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const AdamTensor *__restrict__ tab, const int32_t *__restrict__ chunks,
                                                          int chunk_elems, float *__restrict__ partial) {
    const AdamTensor t = tab[chunks[blockIdx.x * 2]];
    const int64_t begin = (int64_t)chunks[blockIdx.x * 2 + 1] * chunk_elems;
    const int64_t end = begin + chunk_elems < t.n ? begin + chunk_elems : t.n;
    float s = 0.0f;
    const bool vec = (reinterpret_cast<uintptr_t>(t.g) & 15) == 0;
    if (vec) {
        const int64_t n4 = (end - begin) >> 2;
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(t.g + begin);
        for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
            const f32x4 g = g4[i];
            s += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
        }
        for (int64_t i = begin + (n4 << 2) + threadIdx.x; i < end; i += blockDim.x) s += t.g[i] * t.g[i];
    } else {
        for (int64_t i = begin + threadIdx.x; i < end; i += blockDim.x) s += t.g[i] * t.g[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(1024) void sqnorm_final_kernel(const float *__restrict__ partial, int n, float *__restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double ws[16];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; w++) t += ws[w];
        out[0] = (float)t;
    }
}

struct AdamHyper {
    float beta2, one_m_beta1, one_m_beta2, eps;      // 1 - beta formed in double on the host, as torch does
    float max_norm;       // <= 0: no clipping
};

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamTensor &t, float coef, const AdamHyper &h) {
    g *= coef;
    m = m + (g - m) * h.one_m_beta1;                                // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2 + h.one_m_beta2 * g * g;                    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / t.bc2_sqrt + h.eps;
    p -= (t.lr / t.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamTensor *__restrict__ tab, const int32_t *__restrict__ chunks,
                                                   int chunk_elems, const float *__restrict__ norm_sq, AdamHyper h) {
    const AdamTensor t = tab[chunks[blockIdx.x * 2]];
    const int64_t begin = (int64_t)chunks[blockIdx.x * 2 + 1] * chunk_elems;
    const int64_t end = begin + chunk_elems < t.n ? begin + chunk_elems : t.n;
    float coef = 1.0f;
    if (h.max_norm > 0.0f) {                                   // clip_grad_norm_: min(1, max_norm / (norm + 1e-6))
        const float c = h.max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
        coef = c < 1.0f ? c : 1.0f;
    }
    const bool vec = ((reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.m) |
                       reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    int64_t tail = begin;
    if (vec) {
        const int64_t n4 = (end - begin) >> 2;
        f32x4 *p4 = reinterpret_cast<f32x4 *>(t.p + begin), *m4 = reinterpret_cast<f32x4 *>(t.m + begin),
              *v4 = reinterpret_cast<f32x4 *>(t.v + begin);
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(t.g + begin);
        for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
            f32x4 p = p4[i], m = m4[i], v = v4[i];
            const f32x4 g = g4[i];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float pe = p[e], me = m[e], ve = v[e];
                adam1(pe, g[e], me, ve, t, coef, h);
                p[e] = pe;
                m[e] = me;
                v[e] = ve;
            }
            p4[i] = p;
            m4[i] = m;
            v4[i] = v;
        }
        tail = begin + (n4 << 2);
    }
    for (int64_t i = tail + threadIdx.x; i < end; i += blockDim.x) adam1(t.p[i], t.g[i], t.m[i], t.v[i], t, coef, h);
}

}  // namespace opt
}  // namespace occ

OCC_API int32_t occnerf_adam_table_row_bytes(void) { return (int32_t)sizeof(occ::opt::AdamTensor); }

OCC_API int occnerf_adam_step(const void *table, int32_t n_tensors, const int32_t *chunks, int32_t n_chunks,
                              int32_t chunk_elems, double beta1, double beta2, double eps, double max_grad_norm,
                              float *scratch, void *stream) {
    using namespace occ;
    OCC_REQUIRE(table && chunks && scratch, "adam_step: null argument");
    OCC_REQUIRE(n_tensors > 0 && n_chunks > 0 && chunk_elems >= 1024 && chunk_elems % 4 == 0,
                "adam_step: n_tensors=%d n_chunks=%d chunk_elems=%d", n_tensors, n_chunks, chunk_elems);
    const opt::AdamTensor *tab = reinterpret_cast<const opt::AdamTensor *>(table);
    hipStream_t st = as_stream(stream);
    if (max_grad_norm > 0.0) {
        hipLaunchKernelGGL(opt::grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, st, tab, chunks, chunk_elems, scratch + 1);
        hipLaunchKernelGGL(opt::sqnorm_final_kernel, dim3(1), dim3(1024), 0, st, scratch + 1, n_chunks, scratch);
    }
    opt::AdamHyper h{(float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)max_grad_norm};
    hipLaunchKernelGGL(opt::adam_kernel, dim3(n_chunks), dim3(256), 0, st, tab, chunks, chunk_elems, scratch, h);
    return check_launch("adam_step");
}
