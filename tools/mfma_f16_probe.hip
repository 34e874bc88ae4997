// Probe of v_mfma_f32_32x32x16_f16 on gfx950 for the split-fp16 MLP (cfg.mlp_precision = 'f16x3'):
//   (1) are SUBNORMAL fp16 inputs preserved or flushed?  (the lo piece of a split operand is subnormal for small values)
//   (2) how accurate is the fp32 accumulation across 16 chained instructions (K = 256) against float64, compared with an
//       fp32 fmaf chain over the same products?
//   (3) operand layout check with small integers (exact).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/f16probe tools/mfma_f16_probe.hip && /tmp/f16probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A[32][K], B[K][32] row-major halves in global memory; D[32][32] = A x B over `steps` chained MFMAs (K = 16 * steps).
__global__ void mfma_chain(const _Float16 *A, const _Float16 *B, float *D, int steps) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int K = 16 * steps;
    f16v acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    for (int s = 0; s < steps; s++) {
        h8 a, b;
        for (int i = 0; i < 8; i++) {
            a[i] = A[j * K + s * 16 + h * 8 + i];
            b[i] = B[(s * 16 + h * 8 + i) * 32 + j];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

int main() {
    const int steps = 16, K = 16 * steps;
    std::vector<_Float16> A(32 * K), B(K * 32);
    std::vector<float> D(32 * 32);
    _Float16 *dA, *dB;
    float *dD;
    hipMalloc(&dA, A.size() * 2);
    hipMalloc(&dB, B.size() * 2);
    hipMalloc(&dD, D.size() * 4);
    auto run = [&](int st) {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, st);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    };
    // (3) layout: small integers
    srand(1);
    for (auto &v : A) v = (_Float16)(float)(rand() % 7 - 3);
    for (auto &v : B) v = (_Float16)(float)(rand() % 7 - 3);
    run(steps);
    int bad = 0;
    for (int m = 0; m < 32; m++)
        for (int n = 0; n < 32; n++) {
            double s = 0;
            for (int k = 0; k < K; k++) s += (double)(float)A[m * K + k] * (double)(float)B[k * 32 + n];
            bad += (double)D[m * 32 + n] != s;
        }
    printf("layout check (integers, exact): %d mismatches of 1024\n", bad);
    // (1) subnormal inputs
    for (auto &v : A) v = (_Float16)0.f;
    for (auto &v : B) v = (_Float16)0.f;
    A[0] = (_Float16)9.5367431640625e-07f;       // 2^-20: subnormal in fp16 (min normal 2^-14)
    B[0] = (_Float16)1.0f;
    A[1 * 16 + 0] = (_Float16)1.0f;              // row 1 (one step: the kernel's K is 16): a normal A against a subnormal B
    B[0 * 32 + 1] = (_Float16)1.9073486328125e-06f;      // 2^-19
    run(1);
    printf("subnormal A x normal B : D[0][0] = %.10g (2^-20 = %.10g if preserved, 0 if flushed)\n", D[0], 9.5367431640625e-07);
    printf("normal A x subnormal B : D[1][1] = %.10g (2^-19 = %.10g if preserved)\n", D[1 * 32 + 1], 1.9073486328125e-06);
    // (2) accumulation accuracy, K = 256, values like a hidden layer: x ~ |N(0,1)|, w ~ N(0, 0.06)
    srand(7);
    auto gauss = [&]() {
        double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
    };
    for (auto &v : A) v = (_Float16)(float)(0.06 * gauss());
    for (auto &v : B) v = (_Float16)(float)fabs(gauss());
    run(steps);
    double e_mfma = 0, e_fma = 0, scale = 0;
    for (int m = 0; m < 32; m++)
        for (int n = 0; n < 32; n++) {
            double s = 0, sa = 0;
            float c = 0.f;
            for (int k = 0; k < K; k++) {
                const double p = (double)(float)A[m * K + k] * (double)(float)B[k * 32 + n];
                s += p;
                sa += fabs(p);
                c = fmaf((float)A[m * K + k], (float)B[k * 32 + n], c);
            }
            e_mfma = fmax(e_mfma, fabs((double)D[m * 32 + n] - s));
            e_fma = fmax(e_fma, fabs((double)c - s));
            scale = fmax(scale, sa);
        }
    printf("K = 256 dot products of exact fp16 operands: max |mfma chain - f64| = %.3e, max |fp32 fmaf chain - f64| = %.3e "
           "(sum |terms| up to %.3f)\n", e_mfma, e_fma, scale);
    return 0;
}
