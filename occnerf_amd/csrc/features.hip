// Per-point and per-sample feature stages of CanonicalMLP (SURVEY.md section 8 rows a11,
// a13, a14, a15).
//
//  point_sdf_kernel     network.py:263-284   once per frame over the P = 6890 body points
//  point_table_kernel   occnerf_mlp.py:171-175  per-point [hash encoding(32), learnable xyz(3)]
//  sample_features      occnerf_mlp.py:144-181  per sample: neighbour geometry (fp64 where the
//                       reference's float64 normals promote it), 4-D hash encoding, gather of
//                       40 table rows + visibility softmax -> the MLP's 68 inputs
//
// The reference recomputes the per-point block for every 300 000-sample chunk (112x per
// 512^2 frame); it only depends on the weights, so it is hoisted to once per frame.
//
// sample_features is gather-bound: algorithmic bytes per sample = 16 levels x 16 corners x 8 B
// (hash table, 59 MiB) + 40 rows x 140 B (point table, 256-byte row pitch, 1.7 MB, L2-resident) +
// 10 x (12 + 24) B neighbour positions/normals + 160 B of indices in; 272 + 4 B out.  What its time
// actually tracks is the NUMBER of gather instructions per wave (the texture addresser spends ~16+ cycles
// on each, whatever the active lanes or bytes): the 8-lanes-per-sample kernel is organised around that.  Arithmetic mirrors the oracle operation for operation (explicit
// __f*_rn / __d*_rn, no contraction) so that encoder inputs are bit-identical to it.
#include "common.h"

#include <cstdlib>

namespace occ {

constexpr int kKnn = 10;
constexpr int kTableCols = 36;     // 35 features padded to 36 floats
constexpr int kTableStride = 64;   // row pitch in floats (256 B): the 128-byte encoding part of a row is one cache line,
                                   // the 3 learnable-xyz floats start the next -- 2 line lookups per gathered row, not 3

// F.cosine_similarity(float32 dir, float64 normal): dir normalised in fp32, normal in fp64,
// products in fp64 (oracle: oc_cos3).  `un` is the pre-normalised fp64 normal b / max(|b|,eps).
__device__ __forceinline__ double cos3_unit(const float (&a)[3], const double *__restrict__ un) {
    float na = norm3(a[0], a[1], a[2]);
    na = na < 1e-8f ? 1e-8f : na;
    const double t0 = __dmul_rn((double)__fdiv_rn(a[0], na), un[0]);
    const double t1 = __dmul_rn((double)__fdiv_rn(a[1], na), un[1]);
    const double t2 = __dmul_rn((double)__fdiv_rn(a[2], na), un[2]);
    return __dadd_rn(__dadd_rn(t0, t1), t2);
}

// unit[P,3] = n / max(|n|, 1e-8) in fp64, the x2 half of torch's cosine_similarity;
// constant per model, computed once.
__global__ void unit_normals_kernel(const double *__restrict__ normals, int P,
                                    double *__restrict__ unit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const double b0 = normals[i * 3], b1 = normals[i * 3 + 1], b2 = normals[i * 3 + 2];
    double nb = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(b0, b0), __dmul_rn(b1, b1)), __dmul_rn(b2, b2)));
    nb = nb < 1e-8 ? 1e-8 : nb;
    unit[i * 3] = __ddiv_rn(b0, nb);
    unit[i * 3 + 1] = __ddiv_rn(b1, nb);
    unit[i * 3 + 2] = __ddiv_rn(b2, nb);
}

__global__ void point_sdf_kernel(const float *__restrict__ point_cloud,
                                 const float *__restrict__ point_base,
                                 const double *__restrict__ normals,
                                 const double *__restrict__ unit, const int32_t *__restrict__ kidx,
                                 int P, double *__restrict__ knn_base, float *__restrict__ dist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    double num[3] = {0.0, 0.0, 0.0}, den = 0.0;
    float dsum = 0.0f;
    int neg = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int n = kidx[i * 3 + j];
        float dir[3];
#pragma unroll
        for (int c = 0; c < 3; c++) dir[c] = __fsub_rn(point_cloud[i * 3 + c], point_base[n * 3 + c]);
        const double att = fabs(cos3_unit(dir, unit + (size_t)n * 3));           // network.py:275
#pragma unroll
        for (int c = 0; c < 3; c++) num[c] = __dadd_rn(num[c], __dmul_rn(att, (double)point_base[n * 3 + c]));
        den = __dadd_rn(den, att);
        const float n0 = (float)normals[n * 3], n1 = (float)normals[n * 3 + 1], n2 = (float)normals[n * 3 + 2];
        const float dot = __fadd_rn(__fadd_rn(__fmul_rn(dir[0], n0), __fmul_rn(dir[1], n1)), __fmul_rn(dir[2], n2));
        neg += dot < 0.0f;                                                       // :278
        dsum = __fadd_rn(dsum, norm3(dir[0], dir[1], dir[2]));
    }
#pragma unroll
    for (int c = 0; c < 3; c++) knn_base[i * 3 + c] = __ddiv_rn(num[c], den);    // :276
    float d = __fdiv_rn(dsum, 3.0f);                                             // :281
    if (neg > 1) d = -d;                                                         // :279,:282
    dist[i] = d;
}

// Backward of point_sdf_kernel w.r.t. the point offsets (training: network.py:263-284 under autograd; point_cloud =
// point_base + point_dist with the [P,1] offset broadcast over xyz, network.py:119): given d knn_base[P,3] (fp64) and
// d dist[P], the gradient of point_dist[P].  The neighbour ids, the normals and point_base are constants of the graph.
//   dist = +-mean_j |dir_j|                       -> d dir_j += (+-d dist / 3) dir_j / |dir_j|
//   knn_base = sum att_j nbr_j / sum att_j        -> d att_j = d knn_base . (nbr_j - knn_base) / sum att
//   att_j = |c_j|, c_j = (dir_j/|dir_j|) . un_j   -> d dir_j += sign(c_j) d att_j (un_j - c_j dir_j/|dir_j|) / |dir_j|
//   dir_j = pc - nbr_j, pc = base + dist 1        -> d point_dist = sum_c sum_j d dir_j[c]
// torch autograd takes ~70 tiny launches for it per step; fp64 throughout (6 890 threads).
__global__ void point_sdf_backward_kernel(const float *__restrict__ point_cloud, const float *__restrict__ point_base,
                                          const double *__restrict__ normals, const double *__restrict__ unit,
                                          const int32_t *__restrict__ kidx, int P, const double *__restrict__ d_knn_base,
                                          const float *__restrict__ d_dist, float *__restrict__ d_point_dist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    double dir[3][3], nd[3], c[3], att[3], nbr[3][3], num[3] = {0.0, 0.0, 0.0}, den = 0.0;
    int neg = 0;
    for (int j = 0; j < 3; j++) {
        const int n = kidx[i * 3 + j];
        float df[3];
        for (int k = 0; k < 3; k++) {
            df[k] = __fsub_rn(point_cloud[i * 3 + k], point_base[n * 3 + k]);
            dir[j][k] = (double)df[k];
            nbr[j][k] = (double)point_base[n * 3 + k];
        }
        float na = norm3(df[0], df[1], df[2]);
        nd[j] = (double)(na < 1e-8f ? 1e-8f : na);
        c[j] = (dir[j][0] * unit[n * 3] + dir[j][1] * unit[n * 3 + 1] + dir[j][2] * unit[n * 3 + 2]) / nd[j];
        att[j] = fabs(c[j]);
        for (int k = 0; k < 3; k++) num[k] += att[j] * nbr[j][k];
        den += att[j];
        const float n0 = (float)normals[n * 3], n1 = (float)normals[n * 3 + 1], n2 = (float)normals[n * 3 + 2];
        neg += __fadd_rn(__fadd_rn(__fmul_rn(df[0], n0), __fmul_rn(df[1], n1)), __fmul_rn(df[2], n2)) < 0.0f;
    }
    const double gd = (double)d_dist[i] * (neg > 1 ? -1.0 : 1.0) / 3.0;
    double g = 0.0;
    for (int j = 0; j < 3; j++) {
        const int n = kidx[i * 3 + j];
        double datt = 0.0;
        for (int k = 0; k < 3; k++) datt += d_knn_base[i * 3 + k] * (nbr[j][k] - num[k] / den);
        datt /= den;
        const double dc = c[j] < 0.0 ? -datt : (c[j] > 0.0 ? datt : 0.0);
        for (int k = 0; k < 3; k++) {
            const double u = dir[j][k] / nd[j];
            g += gd * u + dc * (unit[n * 3 + k] - c[j] * u) / nd[j];
        }
    }
    d_point_dist[i] = (float)g;
}

__global__ void point_table_kernel(const double *__restrict__ knn_base,
                                   const float *__restrict__ point_sdf,
                                   const float *__restrict__ learnable, int P, float bound,
                                   float two_bound, const float2 *__restrict__ embeddings,
                                   const int32_t *__restrict__ offsets, int L, GridLevels lv,
                                   GridModes4 gm, float *__restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float x[4];
#pragma unroll
    for (int c = 0; c < 3; c++)                                                  // occnerf_mlp.py:171
        x[c] = (float)__ddiv_rn(__dadd_rn(knn_base[i * 3 + c], (double)bound), (double)two_bound);
    float s = __fdiv_rn(__fadd_rn(point_sdf[i], 0.2f), 0.8f);                    // :172
    x[3] = s < 0.0f ? 0.0f : (s > 1.0f ? 1.0f : s);
    bool oob = false;
#pragma unroll
    for (int d = 0; d < 4; d++) oob |= (x[d] < 0.f || x[d] > 1.f);
    float *row = table + (size_t)i * kTableStride;
    for (int l = 0; l < L; l++) {
        float2 v = make_float2(0.f, 0.f);
        if (!oob) {
            const uint32_t o0 = (uint32_t)offsets[l];
            v = encode_level_d4c2(x, embeddings + o0, (uint32_t)offsets[l + 1] - o0, lv.scale[l],
                                  lv.resolution[l], gm.mode[l]);
        }
        row[l * 2] = v.x;
        row[l * 2 + 1] = v.y;
    }
    for (int l = L; l < 16; l++) row[l * 2] = row[l * 2 + 1] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) row[32 + c] = learnable[i * 3 + c];
    row[35] = 0.0f;
}

// host copy of the level offsets -> index modes; without it every level takes the generic path
static GridModes4 modes_from_host_offsets(uint32_t L, const GridLevels &lv, const int32_t *h_offsets) {
    uint32_t sizes[kMaxLevels] = {0};
    if (h_offsets)
        for (uint32_t l = 0; l < L; l++) sizes[l] = (uint32_t)(h_offsets[l + 1] - h_offsets[l]);
    GridModes4 gm = make_grid_modes_d4(h_offsets ? L : 0, lv, sizes);
    return gm;
}

struct FeatParams {
    float bound, two_bound;
    int nscale, L;
    int P;              // rows of the packed point records (8-lanes-per-sample kernel)
};

// Everything sample_features8_kernel needs to know about a level, in one 16-byte record: a lane picks its
// two levels with two 16-byte loads instead of ten dword loads from five arrays.
struct LevelRec4 {
    uint32_t entry0, size;      // first table entry of the level, entries in it
    float scale;
    uint32_t res_mode;          // resolution | mode << 24
};
struct LevelRecs {
    LevelRec4 r[kMaxLevels];
};
// The same for the branch-free index arithmetic of the 8-lanes-per-sample kernel (levels that are dense or hashed
// with a power-of-two size): per-axis multipliers (strides or primes; axis 0 is 1 in both) and the hash mask.
struct LevelRec8 {
    uint32_t entry0, mask;
    float scale;
    uint32_t dense;
    uint32_t mul1, mul2, mul3, pad;
};
struct LevelRecs8 {
    LevelRec8 r[kMaxLevels];
};
// Corner gathers of one level without a branch: a wave holds dense and hashed levels side by side, and the mode
// branches of encode_level_d4c2_taps left every corner's gather behind its own exec-masked block with a full
// s_waitcnt -- 32 serialized round trips per sample group.  Both candidate indices are formed from shared pair terms
// and selected.  Same integers as grid_index (uint32 wrap-around), same fractions as encode_level_d4c2_taps.
__device__ __forceinline__ void level_taps_select(const float (&x)[4], const float2 *grid, const LevelRec8 lr,
                                                  LevelTaps4 &tp) {
    const uint32_t mul[4] = {1u, lr.mul1, lr.mul2, lr.mul3};
    uint32_t t[4][2];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float pos = __fmaf_rn(x[d], lr.scale, 0.5f);
        const float fl = floorf(pos);
        const uint32_t pg = (uint32_t)fl;
        pos -= fl;
        tp.fr[d] = pos;
        t[d][0] = pg * mul[d];
        t[d][1] = t[d][0] + mul[d];
    }
    uint32_t a01[4], x01[4], a23[4], x23[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a01[i] = t[0][i & 1] + t[1][i >> 1];
        x01[i] = t[0][i & 1] ^ t[1][i >> 1];
        a23[i] = t[2][i & 1] + t[3][i >> 1];
        x23[i] = t[2][i & 1] ^ t[3][i >> 1];
    }
#pragma unroll
    for (int idx = 0; idx < 16; idx++) {
        const uint32_t ia = a01[idx & 3] + a23[idx >> 2];
        const uint32_t ix = (x01[idx & 3] ^ x23[idx >> 2]) & lr.mask;
        const uint32_t index = lr.dense ? ia : ix;
#ifdef OCC_FEAT_EXP_RESIDENT_HASH      // (tools/features_fetch_bound.py: every corner gather made an L1 hit -- the bound of any fetch-side change)
        tp.v[idx] = ld32(grid, (lr.entry0 + (index & 63u)) * 8u);
#else
        tp.v[idx] = ld32(grid, (lr.entry0 + index) * 8u);
#endif
    }
}

__global__ __launch_bounds__(256) void sample_features_kernel(
    const float *__restrict__ xyz, int64_t N, const int32_t *__restrict__ knn_idxs,
    const float *__restrict__ point_base, const double *__restrict__ normals,
    const double *__restrict__ unit, const float *__restrict__ counter,
    const float4 *__restrict__ table, const float2 *__restrict__ embeddings,
    const int32_t *__restrict__ offsets, GridLevels lv, GridModes4 gm, FeatParams prm,
    const int32_t *__restrict__ geo_idxs, const float *__restrict__ att_in,
    float *__restrict__ mlp_in, float *__restrict__ raw, float *__restrict__ enc_in_out) {
    const int nk = prm.nscale * kKnn;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t *id = knn_idxs + i * nk;
        const float p[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};

        // ---- neighbour geometry on the finest scale (occnerf_mlp.py:144-167) ----
        int neg = 0;
        float dsum = 0.0f;
        double num[3] = {0.0, 0.0, 0.0}, den = 0.0;
#pragma unroll
        for (int j = 0; j < kKnn; j++) {
            const int n = geo_idxs ? geo_idxs[i * kKnn + j] : id[j];
            float dir[3], nbr[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                nbr[c] = point_base[n * 3 + c];
                dir[c] = __fsub_rn(p[c], nbr[c]);                                 // :147
            }
            double dot = 0.0;                                                     // :152, fp64
#pragma unroll
            for (int c = 0; c < 3; c++) dot = __dadd_rn(dot, __dmul_rn((double)dir[c], normals[(size_t)n * 3 + c]));
            neg += dot < 0.0;
            dsum = __fadd_rn(dsum, norm3(dir[0], dir[1], dir[2]));
            if (j < 3) {                                                          // :164-166
                const double att = fabs(cos3_unit(dir, unit + (size_t)n * 3));
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float pn = __fdiv_rn(__fadd_rn(nbr[c], prm.bound), prm.two_bound);
                    num[c] = __dadd_rn(num[c], __dmul_rn(att, (double)pn));
                }
                den = __dadd_rn(den, att);
            }
        }
        float dist = __fdiv_rn(dsum, (float)kKnn);                                // :155
        if (2 * neg > kKnn) dist = -dist;                                         // :153,:156
        float nd = __fdiv_rn(__fadd_rn(dist, 0.2f), 0.5f);                        // :157
        nd = nd < 0.0f ? 0.0f : (nd > 1.0f ? 1.0f : nd);
        float x[4];
#pragma unroll
        for (int c = 0; c < 3; c++) x[c] = (float)__ddiv_rn(num[c], den);
        x[3] = nd;
        if (enc_in_out) {
#pragma unroll
            for (int c = 0; c < 4; c++) enc_in_out[i * 4 + c] = x[c];
        }
        raw[i * 5 + 4] = dist;

        float *out = mlp_in + i * 68;

        // ---- 4-D multi-resolution hash encoding (gridencoder.cu:87-199) ----
        bool oob = false;
#pragma unroll
        for (int d = 0; d < 4; d++) oob |= (x[d] < 0.f || x[d] > 1.f);
        for (int l = 0; l < prm.L; l++) {
            float2 v = make_float2(0.f, 0.f);
            if (!oob) {
                const uint32_t o0 = (uint32_t)offsets[l];
                v = encode_level_d4c2(x, embeddings + o0, (uint32_t)offsets[l + 1] - o0,
                                      lv.scale[l], lv.resolution[l], gm.mode[l]);
            }
            *reinterpret_cast<float2 *>(out + 36 + l * 2) = v;
        }

        // ---- visibility softmax over the 40 multi-scale neighbours (simple_agg :110-125) ----
        float att[4 * kKnn];
        float amin = INFINITY;
#pragma unroll
        for (int j = 0; j < 4 * kKnn; j++) {
            att[j] = j < nk ? (att_in ? att_in[i * nk + j] : counter[id[j]]) : INFINITY;
            amin = fminf(amin, att[j]);
        }
        float amax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4 * kKnn; j++) {
            if (j < nk) {
                att[j] = __fadd_rn(att[j], __fsub_rn(1.0f, amin));
                amax = fmaxf(amax, att[j]);
            }
        }
        float mean = 0.0f;
#pragma unroll
        for (int j = 0; j < 4 * kKnn; j++) {
            if (j < nk) {
                att[j] = __fdiv_rn(att[j], amax);
                mean = __fadd_rn(mean, att[j]);
            }
        }
        mean = __fdiv_rn(mean, (float)nk);
        float var = 0.0f, smax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4 * kKnn; j++) {
            if (j < nk) {
                const float dlt = __fsub_rn(att[j], mean);
                var = __fadd_rn(var, __fmul_rn(dlt, dlt));
                smax = fmaxf(smax, att[j]);
            }
        }
        var = __fdiv_rn(var, (float)(nk - 1));                                   // unbiased
        float ssum = 0.0f;
#pragma unroll
        for (int j = 0; j < 4 * kKnn; j++) {
            if (j < nk) {
                att[j] = expf(__fsub_rn(att[j], smax));
                ssum = __fadd_rn(ssum, att[j]);
            }
        }
        float agg[kTableCols];
#pragma unroll
        for (int f = 0; f < kTableCols; f++) agg[f] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4 * kKnn; j++) {   // fully unrolled: att[] must stay in registers
            if (j < nk) {
                const float a = __fdiv_rn(att[j], ssum);
                const float4 *row = table + (size_t)id[j] * (kTableStride / 4);
#pragma unroll
                for (int v = 0; v < kTableCols / 4; v++) {
                    const float4 t = row[v];
                    agg[v * 4 + 0] = __fadd_rn(agg[v * 4 + 0], __fmul_rn(a, t.x));
                    agg[v * 4 + 1] = __fadd_rn(agg[v * 4 + 1], __fmul_rn(a, t.y));
                    agg[v * 4 + 2] = __fadd_rn(agg[v * 4 + 2], __fmul_rn(a, t.z));
                    agg[v * 4 + 3] = __fadd_rn(agg[v * 4 + 3], __fmul_rn(a, t.w));
                }
            }
        }
        agg[35] = var;    // mlp input layout: [agg 0..34, var, enc 0..31]
#pragma unroll
        for (int v = 0; v < kTableCols / 4; v++)
            *reinterpret_cast<float4 *>(out + v * 4) =
                make_float4(agg[v * 4], agg[v * 4 + 1], agg[v * 4 + 2], agg[v * 4 + 3]);
    }
}


// Per-point records of the 8-lanes-per-sample kernel (constant per model / per counter version):
//   geo[P]   64 bytes: base xyz (fp32) + pad | normal x, y | normal z, unit-normal x | unit-normal y, z (fp64)
//   tailc[P] 16 bytes: table columns 32..34 (the learnable-xyz tail of the aggregated row) + the visibility count
__global__ void point_pack_kernel(const float *__restrict__ point_base, const double *__restrict__ normals,
                                  const double *__restrict__ unit, const float *__restrict__ counter,
                                  const float *__restrict__ table, int P, float *__restrict__ geo,
                                  float4 *__restrict__ tailc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float *r = geo + (size_t)i * 16;
    r[0] = point_base[i * 3], r[1] = point_base[i * 3 + 1], r[2] = point_base[i * 3 + 2], r[3] = 0.0f;
    double *d = reinterpret_cast<double *>(r + 4);
    d[0] = normals[i * 3], d[1] = normals[i * 3 + 1], d[2] = normals[i * 3 + 2];
    d[3] = unit[i * 3], d[4] = unit[i * 3 + 1], d[5] = unit[i * 3 + 2];
    const float *t = table + (size_t)i * kTableStride + 32;
    tailc[i] = make_float4(t[0], t[1], t[2], counter[i]);
}

// ---------------------------------------------------------------------------------------
// sample_features, 8 lanes per sample.  The thread-per-sample kernel above keeps the memory
// pipeline (TA) 98 % busy with ~700 line look-ups per sample, one per lane per instruction, and
// writes its 272-byte rows in partial lines (3x write amplification, profiles/archive/r01_pmc_hbm.json).
// Here a sample is owned by 8 consecutive lanes:
//   * a table row's 32 encoding features are one float4 per lane = 128 contiguous bytes (2 line
//     look-ups instead of 8), the 3-float tail goes to lane j & 7;
//   * lane g encodes hash levels 2g and 2g+1 and ends up holding exactly the float4 it stores;
//   * the 10 finest neighbours are spread over the lanes and combined in the reference's
//     sequential order through shuffles, so the encoder input stays bit-identical to the oracle;
//   * outputs are written as 128-byte (8 x float4) runs.
// Used for the renderer's normal call (4 scales, counters gathered through knn_idxs).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float grp_sum8(float v) {
    v += __shfl_xor(v, 1, 8);
    v += __shfl_xor(v, 2, 8);
    v += __shfl_xor(v, 4, 8);
    return v;
}
__device__ __forceinline__ float grp_min8(float v) {
    v = fminf(v, __shfl_xor(v, 1, 8));
    v = fminf(v, __shfl_xor(v, 2, 8));
    return fminf(v, __shfl_xor(v, 4, 8));
}
__device__ __forceinline__ float grp_max8(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 8));
    v = fmaxf(v, __shfl_xor(v, 2, 8));
    return fmaxf(v, __shfl_xor(v, 4, 8));
}

// quad broadcasts: lane Q of every aligned group of four lanes, one DPP move per dword
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q * 0x55, 0xf, 0xf, true));
}
template <int Q>
__device__ __forceinline__ double quad_bcast_f64(float lo, float hi) {
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, hi), Q * 0x55, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, lo), Q * 0x55, 0xf, 0xf, true));
}

constexpr int kLdsTailPoints = 9216;       // (tail, count) records that fit the workgroup's LDS image (144 KiB)

template <bool GENERIC /* some level is neither dense nor power-of-two hashed: the reference's loop + modulo */,
          bool LDS_TAIL /* one 768-thread workgroup per CU keeps the (tail, count) records of all points in LDS */>
__global__ __launch_bounds__(LDS_TAIL ? 768 : 256, 3) void sample_features8_kernel(
    const float *__restrict__ xyz, int64_t N_max, const int32_t *__restrict__ knn_idxs,
    const float4 *__restrict__ geo /*[P] 64-byte GeoRec*/, const float4 *__restrict__ tailc /*[P] (table cols 32..34, counter)*/,
    const float4 *__restrict__ table, const float2 *__restrict__ embeddings,
    LevelRecs levels, LevelRecs8 levels8, FeatParams prm, const int32_t *__restrict__ rows /*nullable: compact list of samples*/,
    const int32_t *__restrict__ n_dev /*nullable: device-side count of rows*/, float *__restrict__ mlp_in, float *__restrict__ raw, float *__restrict__ enc_in_out,
    const float *__restrict__ center /*nullable: occnerf_knn_center's [4] (c, r^2)*/,
    const float *__restrict__ center_agg /*with center: columns 0..35 of mlp_in for a sample that has c's neighbour lists*/) {
    constexpr int NK = 4 * kKnn;                       // 40 neighbours over 4 scales
    // CENTRE AGGREGATE (round 4).  The kNN kernel hands every query inside the centre cache's radius the SAME 40 neighbour ids
    // (csrc/knn.hip: the samples that collapse onto the frame's point c, two thirds of the live ones), and columns 0..35 of
    // mlp_in -- the visibility-softmax aggregate of those 40 table rows, its three tail columns and the variance -- are a
    // function of the ids alone: for such a sample they are the 36 numbers the caller computed once for c (this kernel on a
    // sample at c).  A trip whose samples ALL lie inside the radius (the same |p - c|^2 < r^2 test as the kNN kernel's)
    // therefore skips the count / softmax / 40-row phase, 46 % of a trip, and copies them: same bits (tested).
    float ccx = 0.f, ccy = 0.f, ccz = 0.f, cr2 = 0.f;
    if (center) ccx = center[0], ccy = center[1], ccz = center[2], cr2 = center[3];
    // The texture path is this kernel's bound (TA busy 80-89 %, ~16 cycles per gather instruction of 8+ bytes per lane,
    // 64 when every lane has its own line): the five (tail, count) gathers per sample group -- 64 lines each -- are LDS
    // reads instead when the records of all points fit.
    __shared__ float4 s_tail[LDS_TAIL ? kLdsTailPoints : 1];
    // Output timing (round 4).  A sample's 272-byte row of mlp_in is written in three 16-byte-per-lane pieces; the encoding
    // half used to leave half-way through the trip and the aggregate at its end, ~1 us apart, and L2 evicted half-written
    // lines in between: WRITE_SIZE was 1.26x the 276 B/sample the kernel produces (profiles/archive/r03_pmc_hbm.json).  The
    // encoding piece now waits in a lane-private LDS slot (no registers held across the row phase) and the three stores
    // leave back to back at the end of the trip.
    __shared__ float4 s_out[LDS_TAIL ? 768 : 256];
    static_assert(sizeof(float4) * (size_t)kLdsTailPoints + sizeof(float4) * 768 <= 160 * 1024,
                  "sample_features8_kernel<LDS_TAIL>: (tail, count) image + per-lane encoding slots exceed gfx950's 160 KiB of LDS");
    if constexpr (LDS_TAIL) {
        for (int i = threadIdx.x; i < prm.P; i += blockDim.x) s_tail[i] = tailc[i];
        __syncthreads();
    }
    const int g = threadIdx.x & 7;
    const int lane64 = threadIdx.x & 63;
    const int tr = ((lane64 & 7) << 3) | (lane64 >> 3);          // partner lane of the 8x8 transposes: (s, g) <-> (g, s)
    const int64_t N = n_dev ? (int64_t)*n_dev : N_max;
    if (N <= 0) return;
    const int64_t group0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> 3;
    const int64_t iters = (N + ngroups - 1) / ngroups;

    // A sample's journey is a chain of dependent gathers (row list -> position + neighbour ids -> point records ->
    // hash corners; ids -> counts -> table rows) and a SIMD holds only 4 such waves, so the chain's latency, not the
    // texture path's throughput, set the kernel's time.  The loop is software-pipelined: what the NEXT sample group
    // needs from the streamed arrays (StageA) is requested while this group's hash corners are in flight, the row index
    // one step further ahead, and the table rows are fetched one owner lane ahead of their use.
    struct StageA {
        float p[3];             // canonical position
        int idg, id89;          // neighbour g; neighbour 8 (lanes 0..3) / 9 (lanes 4..7)
        int id5[5];             // neighbours 5g..5g+4 of the 40
    };
    auto out_row = [&](int64_t it) {                    // output row of this lane's sample group; clamped: every lane
        const int64_t i_raw = group0 + it * ngroups;    // stays in the shuffles and loads valid memory
        return i_raw < N ? i_raw : N - 1;
    };
    auto load_a = [&](int64_t i, StageA &a) {
        const int32_t *id = knn_idxs + i * NK;
        struct __attribute__((packed, aligned(4))) F3 { float v[3]; };
        struct __attribute__((packed, aligned(4))) I4 { int v[4]; };
        const F3 pq = *reinterpret_cast<const F3 *>(xyz + i * 3);               // one 12-byte load
        a.p[0] = pq.v[0], a.p[1] = pq.v[1], a.p[2] = pq.v[2];
        a.idg = id[g];
        a.id89 = id[8 + (g >> 2)];
        const I4 q = *reinterpret_cast<const I4 *>(id + g * 5);                 // 20 contiguous bytes: 16 + 4
        a.id5[0] = q.v[0], a.id5[1] = q.v[1], a.id5[2] = q.v[2], a.id5[3] = q.v[3];
        a.id5[4] = id[g * 5 + 4];
    };
    StageA cur, nxt;
    int32_t i1;                                         // input row of iteration it + 1
    {
        const int64_t o0 = out_row(0), o1 = out_row(1);
        load_a(rows ? (int64_t)rows[o0] : o0, cur);
        i1 = rows ? rows[o1] : (int32_t)o1;
    }
    for (int64_t it = 0; it < iters; it++) {
        const bool live = group0 + it * ngroups < N;
        const int64_t o = out_row(it);
        const int64_t o2 = out_row(it + 2);
        float *out = mlp_in + o * 68;

        // every live sample of this trip inside the centre's radius: its columns 0..35 are the centre's (wave-uniform)
        bool cached_agg = false;
        if (cr2 > 0.0f) {
            const float dx = cur.p[0] - ccx, dy = cur.p[1] - ccy, dz = cur.p[2] - ccz;
            const bool inside = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) < cr2;
            cached_agg = __builtin_amdgcn_ballot_w64(live && !inside) == 0;
        }
        // (tail, count) of the lane's five rows: in flight with the point records
        float4 tl[5];
#pragma unroll
        for (int k = 0; k < 5; k++) tl[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!cached_agg) {
#pragma unroll
            for (int k = 0; k < 5; k++) tl[k] = LDS_TAIL ? s_tail[cur.id5[k]] : ld32(tailc, (uint32_t)cur.id5[k] * 16u);
        }

        // ---- neighbour geometry: the four lanes of a quad fetch one neighbour's 64-byte record together ----
        // A gather costs one L1 look-up per run of adjacent lanes on the same line (tools/gather_rate.hip), never
        // less than 16 cycles.  Lane (h = g >> 2, q = g & 3) reads 16-byte piece q of neighbour 2k + h in trip k: five
        // instructions of 16 runs each for the ten neighbours of the wave's 8 samples (a lane per neighbour reading its
        // record alone: seven instructions of 64 runs).  The pieces meet through quad-broadcast DPP moves and every lane
        // of the quad evaluates its neighbour (redundant VALU work: the kernel is bound by the texture path).
        float nrm[5];
        int negf[5];
        double t_att[2] = {0.0, 0.0}, t_num[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        {
            float4 pc[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int nid = k < 4 ? __shfl(cur.idg, 2 * k + (g >> 2), 8) : cur.id89;
                pc[k] = ld32(geo, (uint32_t)nid * 64u + (uint32_t)(g & 3) * 16u);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const float nbr[3] = {quad_bcast<0>(pc[k].x), quad_bcast<0>(pc[k].y), quad_bcast<0>(pc[k].z)};
                const double nx = quad_bcast_f64<1>(pc[k].x, pc[k].y), ny = quad_bcast_f64<1>(pc[k].z, pc[k].w);
                const double nz = quad_bcast_f64<2>(pc[k].x, pc[k].y);
                float dir[3];
#pragma unroll
                for (int c = 0; c < 3; c++) dir[c] = __fsub_rn(cur.p[c], nbr[c]);
                double dot = __dadd_rn(0.0, __dmul_rn((double)dir[0], nx));
                dot = __dadd_rn(dot, __dmul_rn((double)dir[1], ny));
                dot = __dadd_rn(dot, __dmul_rn((double)dir[2], nz));
                negf[k] = dot < 0.0;
                nrm[k] = norm3(dir[0], dir[1], dir[2]);
                if (k < 2) {            // neighbours 0, 1 (trip 0) and 2 (trip 1, quad 0) enter the projection
                    const double un[3] = {quad_bcast_f64<2>(pc[k].z, pc[k].w), quad_bcast_f64<3>(pc[k].x, pc[k].y),
                                          quad_bcast_f64<3>(pc[k].z, pc[k].w)};
                    t_att[k] = fabs(cos3_unit(dir, un));
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float pn = __fdiv_rn(__fadd_rn(nbr[c], prm.bound), prm.two_bound);
                        t_num[k][c] = __dmul_rn(t_att[k], (double)pn);
                    }
                }
            }
        }
        // sequential combination in neighbour order j = 0..9 (fp32 sum) / 0..2 (fp64 sums): neighbour j sits in trip
        // j >> 1, quad j & 1
        float dsum = 0.0f;
        int neg = 0;
#pragma unroll
        for (int j = 0; j < kKnn; j++) {
            dsum = __fadd_rn(dsum, __shfl(nrm[j >> 1], 4 * (j & 1), 8));
            neg += __shfl(negf[j >> 1], 4 * (j & 1), 8);
        }
        double num[3] = {0.0, 0.0, 0.0}, den = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
#pragma unroll
            for (int c = 0; c < 3; c++) num[c] = __dadd_rn(num[c], __shfl(t_num[j >> 1][c], 4 * (j & 1), 8));
            den = __dadd_rn(den, __shfl(t_att[j >> 1], 4 * (j & 1), 8));
        }
        float dist = __fdiv_rn(dsum, (float)kKnn);
        if (2 * neg > kKnn) dist = -dist;
        float nd = __fdiv_rn(__fadd_rn(dist, 0.2f), 0.5f);
        nd = nd < 0.0f ? 0.0f : (nd > 1.0f ? 1.0f : nd);
        float x[4];
#pragma unroll
        for (int c = 0; c < 3; c++) x[c] = (float)__ddiv_rn(num[c], den);
        x[3] = nd;
        if (live && g == 0) {
            raw[o * 5 + 4] = dist;
            if (enc_in_out) *reinterpret_cast<float4 *>(enc_in_out + o * 4) = make_float4(x[0], x[1], x[2], x[3]);
        }

        // ---- visibility softmax: lane g owns neighbours 5g..5g+4 ----
        float a5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        float var = 0.0f;
        float tail[3] = {0.f, 0.f, 0.f};
        if (!cached_agg) {
            float lmin = INFINITY;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                a5[k] = tl[k].w;
                lmin = fminf(lmin, a5[k]);
            }
            const float amin = grp_min8(lmin);
            float lmax = -INFINITY;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                a5[k] = __fadd_rn(a5[k], __fsub_rn(1.0f, amin));
                lmax = fmaxf(lmax, a5[k]);
            }
            const float amax = grp_max8(lmax);
            float lsum = 0.0f;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                a5[k] = __fdiv_rn(a5[k], amax);
                lsum += a5[k];
            }
            const float mean = __fdiv_rn(grp_sum8(lsum), (float)NK);
            float lvar = 0.0f, lsm = -INFINITY;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const float dl = a5[k] - mean;
                lvar += dl * dl;
                lsm = fmaxf(lsm, a5[k]);
            }
            var = __fdiv_rn(grp_sum8(lvar), (float)(NK - 1));
            const float smax = grp_max8(lsm);
            float le = 0.0f;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                a5[k] = expf(__fsub_rn(a5[k], smax));
                le += a5[k];
            }
            const float ssum = grp_sum8(le);
#pragma unroll
            for (int k = 0; k < 5; k++) a5[k] = __fdiv_rn(a5[k], ssum);

#pragma unroll
            for (int k = 0; k < 5; k++) {
                tail[0] += a5[k] * tl[k].x;
                tail[1] += a5[k] * tl[k].y;
                tail[2] += a5[k] * tl[k].z;
            }
        }

        // ---- hash encoding ----
        // Lane (s, g) of the wave -- sample slot s = lane >> 3, g = lane & 7 -- encodes levels 2s, 2s+1 of the sample in
        // slot g, i.e. the 8x8 (sample, level pair) assignment is TRANSPOSED for this phase: the 8 adjacent lanes of a
        // gather instruction then look up the SAME level for 8 consecutive samples of a ray, whose encoder inputs (a point
        // projected onto the body surface + a clamped distance) mostly fall into the same or neighbouring cells, so that
        // they share cache lines (adjacent lanes on one line cost one look-up) instead of touching 8 different levels'
        // tables.  Two 8x8 lane transposes (the sample's input out, the two level results back) pay for it.
        // Centre's encoding: inside the radius the encoder input x (projection onto the surface + clamped distance) is,
        // for almost every sample, bitwise the centre's -- fp64 sums of the same ten records rounded to fp32 -- and the 32
        // encoded columns are a function of x alone: a trip whose live samples all carry the centre's x (compared as bit
        // patterns) copies the centre's columns 36..67 and skips the 32 corner gathers.
        bool cached_enc = false;
        if (cached_agg) {
            const bool eq = __float_as_uint(x[0]) == __float_as_uint(center_agg[36]) && __float_as_uint(x[1]) == __float_as_uint(center_agg[37]) &&
                            __float_as_uint(x[2]) == __float_as_uint(center_agg[38]) && __float_as_uint(x[3]) == __float_as_uint(center_agg[39]);
            cached_enc = __builtin_amdgcn_ballot_w64(live && !eq) == 0;
        }
        float2 ev[2];
        int32_t i2;
        if (!cached_enc) {
            LevelTaps4 tp[2];
            bool oob = false;
            float xt[4];
#pragma unroll
            for (int d = 0; d < 4; d++) xt[d] = __shfl(x[d], tr);
#pragma unroll
            for (int d = 0; d < 4; d++) oob |= (xt[d] < 0.f || xt[d] > 1.f);
            __builtin_amdgcn_sched_barrier(0);
            if (oob) {                      // result is 0 (gridencoder.cu:117-126); keep the (discarded) gathers inside the table
#pragma unroll
                for (int d = 0; d < 4; d++) xt[d] = 0.5f;
            }
#pragma unroll
            for (int a = 0; a < 2; a++) {
                if constexpr (GENERIC) {
                    const LevelRec4 lr = levels.r[2 * (lane64 >> 3) + a];
                    encode_level_d4c2_taps(xt, embeddings, lr.size, lr.scale, lr.res_mode & 0xFFFFFFu, lr.res_mode >> 24,
                                           lr.entry0, tp[a]);
                } else {
                    level_taps_select(xt, embeddings, levels8.r[2 * (lane64 >> 3) + a], tp[a]);
                }
            }
            // the next group's streamed inputs: in flight behind the corners
            load_a(i1, nxt);
            i2 = rows ? rows[o2] : (int32_t)o2;
            __builtin_amdgcn_sched_barrier(0);
            float2 evt[2];
#pragma unroll
            for (int a = 0; a < 2; a++) {
                evt[a] = encode_level_d4c2_reduce(tp[a]);
                if (oob) evt[a] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int a = 0; a < 2; a++) ev[a] = make_float2(__shfl(evt[a].x, tr), __shfl(evt[a].y, tr));
        } else {
            load_a(i1, nxt);
            i2 = rows ? rows[o2] : (int32_t)o2;
            const float4 ce = *reinterpret_cast<const float4 *>(center_agg + 40 + 4 * g);
            ev[0] = make_float2(ce.x, ce.y), ev[1] = make_float2(ce.z, ce.w);
        }
        s_out[threadIdx.x] = make_float4(ev[0].x, ev[0].y, ev[1].x, ev[1].y);      // (stored with the rest of the row, below)

        // ---- gather the 40 rows: 128 B of encoding per row across the 8 lanes, one owner lane's five rows ahead ----
        float agg[4] = {0.f, 0.f, 0.f, 0.f};
        if (!cached_agg) {
            // The table rows come in four chunks of ten (two owner lanes' rows); three chunks are in flight or in use at a
            // time: with one chunk ahead every trip waited out most of an L2 round trip and the row phase was 46 % of the
            // kernel (clock64 marks per phase).
            const uint32_t piece = (uint32_t)g * 16u;
            auto issue_chunk = [&](int c, float4 (&t)[10]) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
#pragma unroll
                    for (int k = 0; k < 5; k++)
#ifdef OCC_FEAT_EXP_RESIDENT_ROWS      // (tools/features_fetch_bound.py: the 40 table rows of every sample taken from 16 rows)
                        t[h * 5 + k] = ld32(table, (uint32_t)(__shfl(cur.id5[k], 2 * c + h, 8) & 15) * (uint32_t)(kTableStride * 4) + piece);
#else
                        t[h * 5 + k] = ld32(table, (uint32_t)__shfl(cur.id5[k], 2 * c + h, 8) * (uint32_t)(kTableStride * 4) + piece);
#endif
                }
            };
            float4 b0[10], b1[10], b2[10];
            issue_chunk(0, b0);
            issue_chunk(1, b1);
    #pragma unroll 1
            for (int c = 0; c < 4; c++) {
                const int cn = c + 2 < 4 ? c + 2 : 3;       // (the last two trips re-read chunk 3: L1 hits, no branch)
                issue_chunk(cn, b2);
#pragma unroll
                for (int h = 0; h < 2; h++) {
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const float w = __shfl(a5[k], 2 * c + h, 8);
                        const float4 t = b0[h * 5 + k];
                        agg[0] = __fadd_rn(agg[0], __fmul_rn(w, t.x));
                        agg[1] = __fadd_rn(agg[1], __fmul_rn(w, t.y));
                        agg[2] = __fadd_rn(agg[2], __fmul_rn(w, t.z));
                        agg[3] = __fadd_rn(agg[3], __fmul_rn(w, t.w));
                    }
                }
#pragma unroll
                for (int k = 0; k < 10; k++) b0[k] = b1[k], b1[k] = b2[k];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) tail[c] = grp_sum8(tail[c]);
        } else {
            const float4 ca = *reinterpret_cast<const float4 *>(center_agg + 4 * g);
            agg[0] = ca.x, agg[1] = ca.y, agg[2] = ca.z, agg[3] = ca.w;
            tail[0] = center_agg[32], tail[1] = center_agg[33], tail[2] = center_agg[34], var = center_agg[35];
        }
        if (live) {
            *reinterpret_cast<float4 *>(out + 4 * g) = make_float4(agg[0], agg[1], agg[2], agg[3]);
            if (g == 0) *reinterpret_cast<float4 *>(out + 32) = make_float4(tail[0], tail[1], tail[2], var);
            *reinterpret_cast<float4 *>(out + 36 + 4 * g) = s_out[threadIdx.x];
        }
        cur = nxt;
        i1 = i2;
    }
}


}  // namespace occ

// The unit normals are a per-model constant; they are cached in a small device buffer
// owned by the caller: occnerf_point_sdf(...) fills `unit` when asked to.
OCC_API int occnerf_unit_normals(const double *normals, int32_t P, double *unit, void *stream) {
    using namespace occ;
    OCC_REQUIRE(normals && unit, "unit_normals: null argument");
    if (P <= 0) return 0;
    hipLaunchKernelGGL(unit_normals_kernel, dim3((P + 255) / 256), dim3(256), 0, as_stream(stream), normals, P, unit);
    return check_launch("unit_normals");
}

OCC_API int occnerf_point_sdf(const float *point_cloud, const float *point_base,
                              const double *normals, const double *unit_normals,
                              const int32_t *kidx, int32_t P, double *knn_base, float *dist,
                              void *stream) {
    using namespace occ;
    OCC_REQUIRE(point_cloud && point_base && normals && unit_normals && kidx && knn_base && dist,
                "point_sdf: null argument");
    if (P <= 0) return 0;
    hipLaunchKernelGGL(point_sdf_kernel, dim3((P + 255) / 256), dim3(256), 0, as_stream(stream), point_cloud,
                       point_base, normals, unit_normals, kidx, P, knn_base, dist);
    return check_launch("point_sdf");
}

/* Gradient of point_dist[P] (the [P,1] offset added to every coordinate of point_base, network.py:119) through
 * occnerf_point_sdf, from d_knn_base[P,3] (fp64) and d_dist[P]; kidx = the same 3-NN ids the forward used. */
OCC_API int occnerf_point_sdf_backward(const float *point_cloud, const float *point_base, const double *normals,
                                       const double *unit_normals, const int32_t *kidx, int32_t P, const double *d_knn_base,
                                       const float *d_dist, float *d_point_dist, void *stream) {
    using namespace occ;
    OCC_REQUIRE(point_cloud && point_base && normals && unit_normals && kidx && d_knn_base && d_dist && d_point_dist,
                "point_sdf_backward: null argument");
    if (P <= 0) return 0;
    hipLaunchKernelGGL(point_sdf_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, as_stream(stream), point_cloud, point_base,
                       normals, unit_normals, kidx, P, d_knn_base, d_dist, d_point_dist);
    return check_launch("point_sdf_backward");
}

OCC_API int32_t occnerf_point_table_stride(void) { return occ::kTableStride; }

OCC_API int occnerf_point_table(const double *knn_base, const float *point_sdf,
                                const float *learnable, int32_t P, float bound, float two_bound,
                                const float *embeddings, const int32_t *offsets,
                                const int32_t *h_offsets, uint32_t L, float S, uint32_t H, float *table,
                                void *stream) {
    using namespace occ;
    OCC_REQUIRE(knn_base && point_sdf && learnable && embeddings && offsets && table, "point_table: null argument");
    OCC_REQUIRE(L >= 1 && L <= 16, "point_table: L=%u unsupported", L);
    if (P <= 0) return 0;
    const GridLevels lv = make_grid_levels(L, S, H);
    const GridModes4 gm = modes_from_host_offsets(L, lv, h_offsets);
    hipLaunchKernelGGL(point_table_kernel, dim3((P + 255) / 256), dim3(256), 0, as_stream(stream), knn_base,
                       point_sdf, learnable, P, bound, two_bound, reinterpret_cast<const float2 *>(embeddings),
                       offsets, (int)L, lv, gm, table);
    return check_launch("point_table");
}

OCC_API int occnerf_point_pack(const float *point_base, const double *normals, const double *unit_normals,
                               const float *counter, const float *table, int32_t P, float *point_geo,
                               float *point_tail, void *stream) {
    using namespace occ;
    OCC_REQUIRE(point_base && normals && unit_normals && counter && table && point_geo && point_tail,
                "point_pack: null argument");
    if (P <= 0) return 0;
    hipLaunchKernelGGL(point_pack_kernel, dim3((P + 255) / 256), dim3(256), 0, as_stream(stream), point_base, normals,
                       unit_normals, counter, table, P, point_geo, reinterpret_cast<float4 *>(point_tail));
    return check_launch("point_pack");
}

OCC_API int occnerf_sample_features(const float *xyz, int64_t N, const int32_t *knn_idxs,
                                    int32_t nscale, const float *point_base, const double *normals,
                                    const double *unit_normals, const float *counter,
                                    const float *table, float bound, float two_bound,
                                    const float *embeddings, const int32_t *offsets,
                                    const int32_t *h_offsets, uint32_t L, float S, uint32_t H,
                                    const int32_t *geo_idxs,
                                    const float *att_in, const int32_t *rows, const int32_t *n_dev,
                                    const float *point_geo, const float *point_tail, int32_t P,
                                    float *mlp_in, float *raw, float *enc_in, void *stream) {
    return occnerf_sample_features_centered(xyz, N, knn_idxs, nscale, point_base, normals, unit_normals, counter, table, bound,
                                            two_bound, embeddings, offsets, h_offsets, L, S, H, geo_idxs, att_in, rows, n_dev,
                                            point_geo, point_tail, P, nullptr, nullptr, mlp_in, raw, enc_in, stream);
}

OCC_API int occnerf_sample_features_centered(const float *xyz, int64_t N, const int32_t *knn_idxs,
                                             int32_t nscale, const float *point_base, const double *normals,
                                             const double *unit_normals, const float *counter,
                                             const float *table, float bound, float two_bound,
                                             const float *embeddings, const int32_t *offsets,
                                             const int32_t *h_offsets, uint32_t L, float S, uint32_t H,
                                             const int32_t *geo_idxs,
                                             const float *att_in, const int32_t *rows, const int32_t *n_dev,
                                             const float *point_geo, const float *point_tail, int32_t P,
                                             const float *center, const float *center_agg,
                                             float *mlp_in, float *raw, float *enc_in, void *stream) {
    using namespace occ;
    OCC_REQUIRE(!center == !center_agg, "sample_features: center and center_agg come together");
    if (N <= 0) return 0;
    OCC_REQUIRE(xyz && knn_idxs && point_base && normals && unit_normals && (counter || att_in) && table &&
                    embeddings && offsets && mlp_in && raw, "sample_features: null argument");
    OCC_REQUIRE(nscale >= 1 && nscale <= 4, "sample_features: nscale=%d unsupported", nscale);
    OCC_REQUIRE(L == 16, "sample_features: built for the 16-level encoder of occnerf_mlp.py:45 (L=%u)", L);
    if (N <= 0) return 0;
    const GridLevels lv = make_grid_levels(L, S, H);
    const GridModes4 gm = modes_from_host_offsets(L, lv, h_offsets);
    FeatParams prm{bound, two_bound, nscale, (int)L, (int)P};
    OCC_REQUIRE(!n_dev || rows, "sample_features: a device-side count needs the row list");
    OCC_REQUIRE(!point_geo == !point_tail, "sample_features: point_geo and point_tail come together (occnerf_point_pack)");
    const bool lanes8 = nscale == 4 && !geo_idxs && !att_in && counter && h_offsets && point_geo;
    OCC_REQUIRE(!rows || lanes8,
                "sample_features: a row list is only supported on the renderer's path (4 scales, per-point inputs, "
                "packed point records)");
    if (lanes8) {      // the renderer's call: 8 lanes per sample
        LevelRecs levels;
        for (uint32_t l = 0; l < kMaxLevels; l++)
            levels.r[l] = LevelRec4{(uint32_t)h_offsets[l], (uint32_t)(h_offsets[l + 1] - h_offsets[l]), lv.scale[l],
                                    lv.resolution[l] | (gm.mode[l] << 24)};
        LevelRecs8 levels8;
        bool generic = false;
        for (uint32_t l = 0; l < kMaxLevels; l++) {
            const uint32_t size = (uint32_t)(h_offsets[l + 1] - h_offsets[l]), r1 = lv.resolution[l] + 1;
            const bool dense = gm.mode[l] == kGridDense;
            generic |= gm.mode[l] == kGridGeneric;
            levels8.r[l] = LevelRec8{(uint32_t)h_offsets[l], dense ? 0xFFFFFFFFu : size - 1, lv.scale[l], dense ? 1u : 0u,
                                     dense ? r1 : 2654435761u, dense ? r1 * r1 : 805459861u,
                                     dense ? r1 * r1 * r1 : 3674653429u, 0u};
        }
        OCC_REQUIRE(P > 0, "sample_features: P=%d rows of packed point records", P);
        const bool lds_tail = P <= kLdsTailPoints && N >= 96 * 64;      // (small calls: not worth a block-wide LDS image -- kLdsTailPoints x 16 B of (tail, count) records + 12 KiB of encoding slots, 156 of 160 KiB)
        const int threads = lds_tail ? 768 : 256;
        int64_t blocks8 = (N * 8 + threads - 1) / threads;
        const int64_t cap = lds_tail ? (int64_t)kNumCU : (int64_t)kNumCU * 32;
        if (blocks8 > cap) blocks8 = cap;
        auto kern = generic ? (lds_tail ? sample_features8_kernel<true, true> : sample_features8_kernel<true, false>)
                            : (lds_tail ? sample_features8_kernel<false, true> : sample_features8_kernel<false, false>);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks8), dim3(threads), 0, as_stream(stream), xyz,
                           N, knn_idxs, reinterpret_cast<const float4 *>(point_geo),
                           reinterpret_cast<const float4 *>(point_tail),
                           reinterpret_cast<const float4 *>(table), reinterpret_cast<const float2 *>(embeddings),
                           levels, levels8, prm, rows, n_dev, mlp_in, raw, enc_in, center, center_agg);
        return check_launch("sample_features");
    }
    OCC_REQUIRE(!center, "sample_features: the centre aggregate is only supported on the renderer's path");
    int64_t blocks = (N + 255) / 256;
    if (blocks > (int64_t)kNumCU * 16) blocks = (int64_t)kNumCU * 16;
    hipLaunchKernelGGL(sample_features_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), xyz, N,
                       knn_idxs, point_base, normals, unit_normals, counter,
                       reinterpret_cast<const float4 *>(table), reinterpret_cast<const float2 *>(embeddings),
                       offsets, lv, gm, prm, geo_idxs, att_in, mlp_in, raw, enc_in);
    return check_launch("sample_features");
}
