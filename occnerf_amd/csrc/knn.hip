// Exact k-nearest-neighbour search on gfx950.
//
// Semantics (oracle: oc_knn / oc_msknn; reference: knn.py:46-83 through pykeops):
// distance = sqrt(fma(dz,dz,fma(dy,dy,dx*dx))) in fp32, the k smallest per query in
// ascending order, candidates visited in ascending row order and inserted on strict '<'
// (a tie keeps the lower row first).  Results are bit-exact integers.
//
// msknn kernel: brute force, 4 queries per lane as two packed-fp32 pairs (v_pk_add/mul/fma:
// the 157 TFLOP/s fp32 vector rate needs packed ops).  Every lane of a wave tests the same
// support point at the same time, so the point stream is wave-uniform and is fetched with
// SCALAR loads (s_load_dwordx16 = 4 points) straight into SGPR operands: no LDS, no VGPRs
// for points.  The k-best lists live in registers (static indexing only).
//   The sqrt is taken only for candidates that pass a conservative squared-distance
// filter; the exact strict-'<' decision is made on the rounded sqrt, as pykeops does.
//   Scales are searched coarse -> fine: when scale s is a superset of scale s+1, the
// 10th-best distance found at s+1 bounds the search radius at s, which removes almost all
// list insertions (the expensive, divergent part) without changing the result.
//
// Bound: fp32 VALU.  Algorithmic work: 9152 distance evaluations/sample (6 packed-pair
// instructions each); support set 146 KB streamed through the scalar cache per wave.
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

#include <cmath>
#include <vector>

namespace occ {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kK = 10;

struct KBest {
    float s[kK];
    int i[kK];
};

__device__ __forceinline__ void kbest_reset(KBest &b) {
#pragma unroll
    for (int p = 0; p < kK; p++) {
        b.s[p] = INFINITY;
        b.i[p] = 0;
    }
}

// strict '<' insertion from the tail: the newcomer moves up only past strictly larger
// entries, so equal distances keep the earlier (lower) row first.
__device__ __forceinline__ void kbest_insert(KBest &b, float s, int row) {
    b.s[kK - 1] = s;
    b.i[kK - 1] = row;
#pragma unroll
    for (int p = kK - 1; p > 0; p--) {
        const bool sw = b.s[p] < b.s[p - 1];
        const float ts = b.s[p - 1];
        const int ti = b.i[p - 1];
        b.s[p - 1] = sw ? b.s[p] : ts;
        b.i[p - 1] = sw ? b.i[p] : ti;
        b.s[p] = sw ? ts : b.s[p];
        b.i[p] = sw ? ti : b.i[p];
    }
}

// Squared-distance bound that admits every d2 whose rounded sqrt can be <= s.
__device__ __forceinline__ float filter_bound(float s) { return s * s * 1.0000005f; }

__device__ __forceinline__ void consider(KBest &b, float &thr, float d2, int row) {
    if (d2 < thr) {
        const float s = sqrtf(d2);
        if (s < b.s[kK - 1]) {
            kbest_insert(b, s, row);
            thr = fminf(thr, filter_bound(b.s[kK - 1]));
        }
    }
}

struct MsKnnScales {
    int begin[5];   // row range of each scale inside the padded point array
    int end[5];     // real rows end here; rows up to the next multiple of 4 are +inf pads
    int seed[4];    // scale s may start from the bound found at scale s+1
    int nscale;
};

constexpr int kQ = 4;  // queries per lane

__global__ __launch_bounds__(256) void msknn_kernel(const float *__restrict__ xyz, int64_t N,
                                                    const float4 *__restrict__ points,
                                                    const int32_t *__restrict__ index_map,
                                                    MsKnnScales sc,
                                                    int32_t *__restrict__ knn_idxs) {
    const int64_t tile = (int64_t)blockDim.x * kQ;
    for (int64_t base = (int64_t)blockIdx.x * tile; base < N; base += (int64_t)gridDim.x * tile) {
        // lane owns queries base + t + {0,1,2,3} * blockDim: consecutive lanes -> consecutive
        // samples (coalesced loads, coherent neighbourhoods along a ray)
        int64_t qi[kQ];
        f32x2 qx[2], qy[2], qz[2];
#pragma unroll
        for (int a = 0; a < kQ; a++) {
            qi[a] = base + threadIdx.x + (int64_t)a * blockDim.x;
            const int64_t src = qi[a] < N ? qi[a] : N - 1;
            qx[a >> 1][a & 1] = xyz[src * 3 + 0];
            qy[a >> 1][a & 1] = xyz[src * 3 + 1];
            qz[a >> 1][a & 1] = xyz[src * 3 + 2];
        }
        KBest best[kQ];
        float thr[kQ];
#pragma unroll
        for (int a = 0; a < kQ; a++) thr[a] = INFINITY;

        for (int l = sc.nscale - 1; l >= 0; l--) {
#pragma unroll
            for (int a = 0; a < kQ; a++) {
                // radius carried over from the coarser (subset) scale, else unbounded
                const bool carry = (l < sc.nscale - 1) && sc.seed[l];
                thr[a] = carry ? filter_bound(best[a].s[kK - 1]) : INFINITY;
                kbest_reset(best[a]);
            }
            const int jb = sc.begin[l], je = sc.end[l];
            for (int j = jb; j < je; j += 4) {
                // wave-uniform addresses: 4 points = one 64-byte scalar load
                const float4 p0 = points[j], p1 = points[j + 1], p2 = points[j + 2],
                             p3 = points[j + 3];
#define OCC_PT(P, OFF)                                                                     \
    {                                                                                      \
        _Pragma("unroll") for (int h = 0; h < 2; h++) {                                    \
            const f32x2 dx = qx[h] - P.x, dy = qy[h] - P.y, dz = qz[h] - P.z;              \
            const f32x2 d2 = __builtin_elementwise_fma(                                    \
                dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));                       \
            consider(best[2 * h], thr[2 * h], d2[0], j + OFF - jb);                        \
            consider(best[2 * h + 1], thr[2 * h + 1], d2[1], j + OFF - jb);                \
        }                                                                                  \
    }
                OCC_PT(p0, 0)
                OCC_PT(p1, 1)
                OCC_PT(p2, 2)
                OCC_PT(p3, 3)
#undef OCC_PT
            }
#pragma unroll
            for (int a = 0; a < kQ; a++) {
                if (qi[a] < N) {
                    int32_t *out = knn_idxs + (qi[a] * sc.nscale + l) * kK;
#pragma unroll
                    for (int p = 0; p < kK; p++) out[p] = index_map[jb + best[a].i[p]];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Cluster-culled search.  The support points of every scale are grouped by their nearest
// coarsest-scale point (108 clusters for the SMPL body) and stored cluster by cluster; a
// wave owns a compact tile of queries -- 64 neighbouring rays (an 8x8 pixel patch when the host
// orders rays along a Morton curve, Network.forward) x 4 consecutive samples, a box of roughly
// 4 x 4 x 10 cm in observation space -- and skips a whole cluster when, for every
// one of its 256 queries, the triangle inequality puts all of the cluster's points outside the
// current search radius:  |q - c| - r_cluster > radius(q).  Measured on the benchmark frame
// this leaves ~1 670 of the 9 152 distance evaluations per sample.  Exactness is unchanged:
// the test is conservative (fp32 slack included), and because points are no longer visited in
// row order the k-best lists order equal distances by original row explicitly.  (The brute-force
// kernel above keeps its own strict-'<' arrival-order insertion: an independent check of the keyed lists.)
// ---------------------------------------------------------------------------------------
// k-best list as ONE fp64 key per entry: high word = the bits of the (non-negative, correctly rounded)
// fp32 distance, low word = the row.  For such pairs the order of the 64-bit patterns as IEEE doubles is
// exactly the lexicographic (distance, row) order -- the pattern is never a NaN (a finite or infinite fp32
// puts at most 0x7F8 into the 11 exponent bits), and fp64 denormals (distance < 2^-119 or 0) are preserved
// by the kernel's float mode.  Insertion into the sorted list is then a branch-free chain of v_min_f64 /
// v_max_f64 compare-exchanges: 20 instructions, no compares, no selects, rows carried for free.
struct KBest64 {
    double k[kK];
};

__device__ __forceinline__ double key64(float s, int row) { return __hiloint2double(__float_as_int(s), row); }
__device__ __forceinline__ float key_dist(double k) { return __int_as_float(__double2hiint(k)); }
__device__ __forceinline__ int key_row(double k) { return __double2loint(k); }

__device__ __forceinline__ void kbest64_reset(KBest64 &b) {
#pragma unroll
    for (int p = 0; p < kK; p++) b.k[p] = key64(INFINITY, 0x7fffffff);
}

__device__ __forceinline__ void kbest64_insert(KBest64 &b, double t) {
#pragma unroll
    for (int p = 0; p < kK; p++) {
        double hi;          // the list entry is updated in place: no copies where the divergent branch rejoins
        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(b.k[p]), "v"(t));
        asm("v_min_f64 %0, %0, %1" : "+v"(b.k[p]) : "v"(t));
        t = hi;
    }
}

// sb: search radius in distance units (seed or current 10th best), thr: its squared filter
__device__ __forceinline__ void consider_lex(KBest64 &b, float &thr, float &sb, float d2, int row) {
    if (d2 < thr) {
        kbest64_insert(b, key64(sqrtf(d2), row));
        sb = fminf(sb, key_dist(b.k[kK - 1]));
        thr = filter_bound(sb);
    }
}

// ---------------------------------------------------------------------------------------
// CENTRE CACHE (round 4).  Two thirds of a frame's live samples sit within a micrometre of ONE point: wherever a sample's
// motion-weight sum is far below the reference's 1e-4 clamp (network.py:388) its warped position collapses onto the origin, and
// after the non-rigid offset onto c = offset(0).  knn_center_kernel searches c once per frame -- the 11 nearest points of every
// scale, by the very distance formula and (distance, row) order of the search kernels -- and derives a radius r inside which
// EVERY query provably has c's neighbours in c's order: a query q moves every true distance by at most |q - c| and the computed
// fp32 distance (correctly rounded sqrt of an fma chain) is within 2e-7 relative of the true one, so with g = the smallest gap
// between consecutive ones of c's 11 computed distances over all scales (d_1 .. d_11; every other point is at least d_11 away),
// r = 0.9 (g / 2 - 2e-6 (d_11max + 1)) keeps every pair (a, b) with d_c(a) < d_c(b) in that order at q.  A tie among c's 11
// (g = 0) disables the cache.  msknn_clustered_kernel then takes the queries with |q - c|^2 < r^2 out of the search, writes c's
// indices for them, and skips tiles that hold no other query: same indices as the search, bit for bit (tested against the
// brute-force kernel with queries just inside and just outside r, and on the tie model).
__global__ __launch_bounds__(256) void knn_center_kernel(const float *__restrict__ c, const float4 *__restrict__ points,
                                                         const int32_t *__restrict__ index_map, MsKnnScales sc,
                                                         float *__restrict__ center_out /*[4]: c, r^2*/,
                                                         int32_t *__restrict__ idx_out /*[nscale][10]*/) {
    // One wave per scale (<= 4 scales, 4 waves), no block-wide rendezvous inside the search: a lane keeps the 11 smallest
    // (distance, row) keys of its share of the scale's points in registers -- key = distance bits << 32 | row: for non-negative
    // floats the unsigned order of the 64-bit patterns is the lexicographic order -- and the wave then pops the global minimum
    // 11 times (a shuffle reduction; the owning lane advances its head).
    __shared__ unsigned long long s_key[4][11];
    const int lane = threadIdx.x & 63, l = threadIdx.x >> 6;
    const float cx = c[0], cy = c[1], cz = c[2];
    if (l < sc.nscale) {
        const int jb = sc.begin[l], je = sc.end[l];
        unsigned long long best[11];
#pragma unroll
        for (int p = 0; p < 11; p++) best[p] = ~0ull;
        for (int j = jb + lane; j < je; j += 64) {
            const float4 P = points[j];
            const float dx = cx - P.x, dy = cy - P.y, dz = cz - P.z;
            const float d = sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
            unsigned long long t = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(j - jb);
            if (!(d < INFINITY)) t = ~0ull;                       // +inf pad rows (and NaN) never enter
#pragma unroll
            for (int p = 0; p < 11; p++) {                        // sorted insertion: carry the larger key down the list
                const unsigned long long lo = t < best[p] ? t : best[p];
                t = t < best[p] ? best[p] : t;
                best[p] = lo;
            }
        }
        for (int round = 0; round < 11; round++) {
            unsigned long long m = best[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long other = ((unsigned long long)__shfl_xor((unsigned)(m >> 32), o) << 32) | __shfl_xor((unsigned)m, o);
                m = other < m ? other : m;
            }
            if (best[0] == m && m != ~0ull) {                     // (keys are unique: one owner) pop
#pragma unroll
                for (int p = 0; p < 10; p++) best[p] = best[p + 1];
                best[10] = ~0ull;
            }
            if (lane == 0) s_key[l][round] = m;
        }
    }
    __syncthreads();
    if (threadIdx.x < sc.nscale * 10) {
        const int ls = threadIdx.x / 10, j = threadIdx.x % 10;
        idx_out[ls * 10 + j] = index_map[sc.begin[ls] + (int)(unsigned)s_key[ls][j]];
    }
    if (threadIdx.x == 0) {
        float g = INFINITY, dmax = 0.0f;
        for (int ls = 0; ls < sc.nscale; ls++) {
            if (s_key[ls][10] == ~0ull) g = 0.0f;                  // a scale with < 11 real points: no radius
            for (int j = 0; j < 10; j++)
                g = fminf(g, __uint_as_float((unsigned)(s_key[ls][j + 1] >> 32)) - __uint_as_float((unsigned)(s_key[ls][j] >> 32)));
            if (s_key[ls][10] != ~0ull) dmax = fmaxf(dmax, __uint_as_float((unsigned)(s_key[ls][10] >> 32)));
        }
        const float r = 0.9f * (0.5f * g - 2e-6f * (dmax + 1.0f));
        center_out[0] = cx, center_out[1] = cy, center_out[2] = cz;
        center_out[3] = (r > 0.0f && r < INFINITY) ? r * r : 0.0f;
    }
}

struct ClusteredScales {
    int nscale, ncl, ngrp;
    int coarse_begin, coarse_end;   // rows of the coarsest scale (original order, padded to 4)
    int seed[4];
};

// ---------------------------------------------------------------------------------------
// ONE query per lane (round 6): the form small launches use for their SPREAD tiles.  A tile whose 256 queries are spread over the
// body -- 64 neighbouring rays whose samples warp to different bones -- makes its wave scan most of the four point sets for all
// four queries of every lane: 8 x the average tile (per-tile stamps, profiles/r06_knn_split_tiles.md).  On a full frame such tiles
// hide among 30 tiles per wave; on a rank's eighth of a frame (3.7 tiles per wave) or a training batch (ONE tile per wave) the
// longest tile IS the kernel.  There a spread tile is handed out as FOUR jobs, sample slot a of its 64 rays each, to four waves,
// which search with this scalar form: half the arithmetic per point, and the clusters in reach of 64 queries that share a depth
// slab.  Same distance formula, same keys, same conservative sphere tests as the four-query form: identical indices.
// ---------------------------------------------------------------------------------------
struct OneQuery {
    float x, y, z;
    bool live;
};

#define OCC_PT1(P)                                                                          \
    {                                                                                       \
        const float dx = q.x - P.x, dy = q.y - P.y, dz = q.z - P.z;                         \
        const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));           \
        consider_lex(best, thr, sb, d2, __float_as_int(P.w));                               \
    }
#define OCC_SCAN1(JB, JE)                                                                   \
    {                                                                                       \
        const int je_ = (JE);                                                               \
        int j = (JB);                                                                       \
        float4 n0 = points[j], n1 = points[j + 1], n2 = points[j + 2], n3 = points[j + 3];  \
        for (; j < je_; j += 4) {                                                           \
            const float4 p0 = n0, p1 = n1, p2 = n2, p3 = n3;                                \
            const int jn = j + 4 < je_ ? j + 4 : j;                                         \
            n0 = points[jn], n1 = points[jn + 1], n2 = points[jn + 2], n3 = points[jn + 3]; \
            OCC_PT1(p0) OCC_PT1(p1) OCC_PT1(p2) OCC_PT1(p3)                                 \
        }                                                                                   \
    }
#define OCC_EMIT1(L)                                                                        \
    if (q.live) {                                                                           \
        struct __attribute__((packed, aligned(4))) I4 { int v[4]; };                        \
        struct __attribute__((packed, aligned(4))) I2 { int v[2]; };                        \
        int32_t *out = knn_idxs + (qi * sc.nscale + (L)) * kK;                              \
        int r_[kK];                                                                         \
        _Pragma("unroll") for (int p = 0; p < kK; p++) r_[p] = key_row(best.k[p]) & 0xFFFF; \
        *reinterpret_cast<I4 *>(out) = I4{{r_[0], r_[1], r_[2], r_[3]}};                    \
        *reinterpret_cast<I4 *>(out + 4) = I4{{r_[4], r_[5], r_[6], r_[7]}};                \
        *reinterpret_cast<I2 *>(out + 8) = I2{{r_[8], r_[9]}};                              \
    }

__device__ __forceinline__ void search_one_query(const OneQuery q, const int64_t qi, const int lane,
                                                 const float4 *__restrict__ points, const float4 *__restrict__ centers,
                                                 const int2 *__restrict__ ranges, const float *__restrict__ radius,
                                                 const float4 *__restrict__ gcenters, const int2 *__restrict__ granges,
                                                 const float *__restrict__ gradius, const ClusteredScales &sc,
                                                 int32_t *__restrict__ knn_idxs) {
    KBest64 best;
    float thr = q.live ? INFINITY : -1.0f, sb = INFINITY;
    kbest64_reset(best);
    OCC_SCAN1(sc.coarse_begin, sc.coarse_end)
    OCC_EMIT1(sc.nscale - 1)
    // reference point of the "nearest cluster first" choice: the first live query of the wave
    const unsigned long long lv = __builtin_amdgcn_ballot_w64(q.live);
    const int first = lv ? __builtin_ctzll(lv) : 0;
    const float rx = __shfl(q.x, first), ry = __shfl(q.y, first), rz = __shfl(q.z, first);
    for (int l = sc.nscale - 2; l >= 0; l--) {
        sb = sc.seed[l] ? key_dist(best.k[kK - 1]) : INFINITY;
        thr = q.live ? filter_bound(sb) : -1.0f;
        kbest64_reset(best);
        const int2 *rg = ranges + (size_t)l * sc.ncl;
        const float *rd = radius + (size_t)l * sc.ncl;
        int k_first;
        {
            float bd = INFINITY;
            int bk = 0;
            for (int k = lane; k < sc.ncl; k += kWave) {
                const float4 c = centers[k];
                const float dx = rx - c.x, dy = ry - c.y, dz = rz - c.z;
                const float d = dx * dx + dy * dy + dz * dz;
                const bool has = rg[k].x < rg[k].y;
                if (has && d < bd) {
                    bd = d;
                    bk = k;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float od = __shfl_xor(bd, o);
                const int ok = __shfl_xor(bk, o);
                if (od < bd || (od == bd && ok < bk)) {
                    bd = od;
                    bk = ok;
                }
            }
            k_first = __builtin_amdgcn_readfirstlane(bk);
        }
        {
            const int2 range = rg[k_first];
            OCC_SCAN1(range.x, range.y)
        }
        auto in_reach = [&](const float4 c, const float r) -> bool {
            const float dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const float lim = (sb + r) * 1.00001f;                   // (1e-5 relative slack >> fp32 error)
            return __builtin_amdgcn_ballot_w64(q.live && !(d2 > lim * lim)) != 0;
        };
        const int ngrp = sc.ngrp > 0 ? sc.ngrp : 1;
        for (int gi = 0; gi < ngrp; gi++) {
            int k_lo = 0, k_hi = sc.ncl;
            if (sc.ngrp > 0) {
                const float gr = gradius[(size_t)l * sc.ngrp + gi];
                if (gr < 0.0f) continue;
                if (!in_reach(gcenters[gi], gr)) continue;
                const int2 gk = granges[gi];
                k_lo = gk.x, k_hi = gk.y;
            }
            for (int k = k_lo; k < k_hi; k++) {
                if (k == k_first) continue;
                const int2 range = rg[k];
                if (range.x >= range.y) continue;
                if (!in_reach(centers[k], rd[k])) continue;
                OCC_SCAN1(range.x, range.y)
            }
        }
        OCC_EMIT1(l)
    }
}
#undef OCC_PT1
#undef OCC_SCAN1
#undef OCC_EMIT1

// squared extent (bounding-box diagonal of the searched queries, m^2) above which a small launch hands a tile to four waves
// (0.2 m; splitting EVERY tile of a training batch -- one tile per resident wave -- measured no better: 0.95 against 0.93 ms, the
// longest one-query job, 64 queries spread over the body, is the floor then)
constexpr float kSplitExtent2 = 0.04f;

template <bool SPLIT /* small launches: four tickets per tile, spread tiles searched as four one-query jobs */>
__global__ __launch_bounds__(256) void msknn_clustered_kernel(
    const float *__restrict__ xyz, const float *__restrict__ mask /*nullable*/, int64_t n_rays, int S,
    const float4 *__restrict__ points, const float4 *__restrict__ centers,
    const int2 *__restrict__ ranges /*[nscale-1][ncl]*/, const float *__restrict__ radius /*[nscale-1][ncl]*/,
    const float4 *__restrict__ gcenters /*[ngrp]*/, const int2 *__restrict__ granges /*[ngrp]: cluster index range*/,
    const float *__restrict__ gradius /*[nscale-1][ngrp], < 0: empty*/,
    ClusteredScales sc, int32_t *__restrict__ knn_idxs, unsigned *__restrict__ ticket,
    const int32_t *__restrict__ qrows /*nullable: ascending list of the samples to query*/,
    const int32_t *__restrict__ ray_start /*with qrows: [n_rays + 1] first list entry of every ray*/,
    const float *__restrict__ center /*nullable: knn_center_kernel's [4] (c, r^2)*/,
    const int32_t *__restrict__ center_idx /*with center: c's indices [nscale][10]*/,
    const float split_extent2 /*SPLIT: tiles wider than this (squared) are searched as four one-query jobs*/) {
    const int lane = threadIdx.x & 63;
    float ccx = 0.f, ccy = 0.f, ccz = 0.f, cr2 = 0.f;      // wave-uniform (scalar loads)
    if (center) ccx = center[0], ccy = center[1], ccz = center[2], cr2 = center[3];
    const int tiles_per_chunk = (S + 3) / 4;
    const int64_t n_tiles = ((n_rays + 63) / 64) * tiles_per_chunk;
    // Tiles differ widely in cost (a tile of dead samples is skipped after its mask loads, a tile near the body
    // visits many clusters), so the waves draw tiles from a counter instead of striding over them: with ~6 tiles per
    // resident wave a static assignment left a third of the wave slots idle in the kernel's tail.
    for (;;) {
        unsigned t32 = 0;
        if (lane == 0) t32 = atomicAdd(ticket, 1u);
        const int64_t job = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)t32);
        // SPLIT: every tile owns four consecutive tickets (tile, part 0 .. 3).  Part 0 searches a compact tile whole; the other
        // parts re-derive "compact" from the same loads and leave.  A spread tile is searched by all four parts, on four waves at
        // about the same time, sample slot `part` of its 64 rays each.
        const int64_t tile = SPLIT ? job >> 2 : job;
        const int part = SPLIT ? (int)(job & 3) : 0;
        if (tile >= n_tiles) break;
        // lane = one of 64 neighbouring rays (the host orders rays in compact pixel patches),
        // its 4 queries = 4 consecutive samples of that ray
        // Ticket -> tile.  With a query list the tickets walk the sample chunks OUTERMOST (chunk j of every ray block, then
        // chunk j + 1): the first chunks are full tiles -- every ray has a few listed samples -- and the last ones hold only
        // the longest rays' remainders, so the heavy tiles are drawn first and the kernel's tail is made of light ones
        // (a rank's eighth of a frame has < 4 tiles per resident wave).
        const int64_t n_blocks = (n_rays + 63) / 64;
        const int64_t blk = qrows ? tile % n_blocks : tile / tiles_per_chunk;
        const int64_t ray = blk * 64 + lane;
        const int s0 = (int)(qrows ? tile / n_blocks : tile % tiles_per_chunk) * 4;
        int64_t qi[kQ];
        bool live[kQ];
        f32x2 qx[2], qy[2], qz[2];
        // With a query list the lane's 4 queries are the NEXT FOUR LISTED samples of its ray (entries ray_start[ray] + s0 ..),
        // not 4 fixed sample slots: a tile is full wherever its rays still have listed samples, and the tiles beyond a ray
        // block's longest list are skipped.  (The arithmetic is this kernel's bound -- VALU 84 % busy -- and a lane evaluates
        // its four packed distances whether the queries are listed or not.)
        int32_t hs = 0, hn = 0;
        if (qrows) {
            const int64_t r = ray < n_rays ? ray : n_rays - 1;
            hs = ray_start[r];
            hn = ray < n_rays ? ray_start[r + 1] - hs : 0;
        }
#pragma unroll
        for (int a = 0; a < kQ; a++) {
            live[a] = ray < n_rays && (s0 + a) < S;
            const int64_t r = ray < n_rays ? ray : n_rays - 1;
            const int sm = (s0 + a) < S ? (s0 + a) : S - 1;
            qi[a] = r * S + sm;
            if (qrows) {
                live[a] = (s0 + a) < hn;
                qi[a] = live[a] ? (int64_t)qrows[hs + s0 + a] : r * S;
            }
            // samples whose motion-weight sum is exactly 0 cannot contribute to the pixel (their alpha is
            // multiplied by it, network.py:330): their neighbours are never read
            if (mask && mask[qi[a]] == 0.0f) live[a] = false;
            qx[a >> 1][a & 1] = xyz[qi[a] * 3 + 0];
            qy[a >> 1][a & 1] = xyz[qi[a] * 3 + 1];
            qz[a >> 1][a & 1] = xyz[qi[a] * 3 + 2];
        }
        // centre cache: queries provably inside the radius in which c's neighbour lists hold leave the search with c's indices
        if (cr2 > 0.0f) {
#pragma unroll
            for (int a = 0; a < kQ; a++) {
                const float dx = qx[a >> 1][a & 1] - ccx, dy = qy[a >> 1][a & 1] - ccy, dz = qz[a >> 1][a & 1] - ccz;
                const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (live[a] && d2 < cr2) {
                    live[a] = false;
                    struct __attribute__((packed, aligned(4))) I4 { int v[4]; };
                    struct __attribute__((packed, aligned(4))) I2 { int v[2]; };
                    for (int l = 0; part == 0 && l < sc.nscale; l++) {
                        int32_t *out = knn_idxs + (qi[a] * sc.nscale + l) * kK;
                        const int32_t *ci = center_idx + l * kK;
                        *reinterpret_cast<I4 *>(out) = I4{{ci[0], ci[1], ci[2], ci[3]}};
                        *reinterpret_cast<I4 *>(out + 4) = I4{{ci[4], ci[5], ci[6], ci[7]}};
                        *reinterpret_cast<I2 *>(out + 8) = I2{{ci[8], ci[9]}};
                    }
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(live[0] || live[1] || live[2] || live[3]) == 0) continue;   // whole tile dead (or served by the cache)
        if constexpr (SPLIT) {
            // extent of the searched queries: squared diagonal of their bounding box (wave-wide; the same for the tile's four parts)
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int a = 0; a < kQ; a++) {
                if (live[a]) {
                    const float v[3] = {qx[a >> 1][a & 1], qy[a >> 1][a & 1], qz[a >> 1][a & 1]};
#pragma unroll
                    for (int d = 0; d < 3; d++) lo[d] = fminf(lo[d], v[d]), hi[d] = fmaxf(hi[d], v[d]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
                    hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
                }
            }
            const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
            const float ext2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ex * ex + ey * ey + ez * ez)));
            const bool spread = ext2 > split_extent2;
            if (!spread && part != 0) continue;
            if (spread) {              // a lane's query = sample slot `part` of its ray
                OneQuery q1{0.f, 0.f, 0.f, false};
                int64_t q1i = 0;
#pragma unroll
                for (int a = 0; a < kQ; a++) {
                    if (a == part) {
                        q1 = OneQuery{qx[a >> 1][a & 1], qy[a >> 1][a & 1], qz[a >> 1][a & 1], live[a]};
                        q1i = qi[a];
                    }
                }
                if (__builtin_amdgcn_ballot_w64(q1.live) != 0)
                    search_one_query(q1, q1i, lane, points, centers, ranges, radius, gcenters, granges, gradius, sc, knn_idxs);
                continue;
            }
        }
        KBest64 best[kQ];
        float thr[kQ], sb[kQ];

    /* one point against the lane's 4 queries: a single guard for "any of the four may enter its list" */ \
#define OCC_PT4(P)                                                                          \
    {                                                                                       \
        const f32x2 dx0 = qx[0] - P.x, dy0 = qy[0] - P.y, dz0 = qz[0] - P.z;                \
        const f32x2 dx1 = qx[1] - P.x, dy1 = qy[1] - P.y, dz1 = qz[1] - P.z;                \
        const f32x2 da = __builtin_elementwise_fma(dz0, dz0, __builtin_elementwise_fma(dy0, dy0, dx0 * dx0)); \
        const f32x2 db = __builtin_elementwise_fma(dz1, dz1, __builtin_elementwise_fma(dy1, dy1, dx1 * dx1)); \
        if ((da[0] < thr[0]) | (da[1] < thr[1]) | (db[0] < thr[2]) | (db[1] < thr[3])) {    \
            const int row = __float_as_int(P.w);                                            \
            consider_lex(best[0], thr[0], sb[0], da[0], row);                               \
            consider_lex(best[1], thr[1], sb[1], da[1], row);                               \
            consider_lex(best[2], thr[2], sb[2], db[0], row);                               \
            consider_lex(best[3], thr[3], sb[3], db[1], row);                               \
        }                                                                                   \
    }
    /* the next four points are fetched (scalar loads) while the current four are tested */    \
#define OCC_SCAN(JB, JE)                                                                    \
    {                                                                                       \
        const int je_ = (JE);                                                               \
        int j = (JB);                                                                       \
        float4 n0 = points[j], n1 = points[j + 1], n2 = points[j + 2], n3 = points[j + 3];  \
        for (; j < je_; j += 4) {                                                           \
            const float4 p0 = n0, p1 = n1, p2 = n2, p3 = n3;                                \
            const int jn = j + 4 < je_ ? j + 4 : j;                                         \
            n0 = points[jn], n1 = points[jn + 1], n2 = points[jn + 2], n3 = points[jn + 3]; \
            OCC_PT4(p0) OCC_PT4(p1) OCC_PT4(p2) OCC_PT4(p3)                                 \
        }                                                                                   \
    }
    /* .w of a point = original row << 16 | base-point index: the key's low word orders ties by row and   \
       carries the reported index, so no index_map gather; 10 indices = 40 contiguous bytes = 3 stores */   \
#define OCC_EMIT(L)                                                                         \
    _Pragma("unroll") for (int a = 0; a < kQ; a++) {                                        \
        if (live[a]) {                                                                      \
            struct __attribute__((packed, aligned(4))) I4 { int v[4]; };                    \
            struct __attribute__((packed, aligned(4))) I2 { int v[2]; };                    \
            int32_t *out = knn_idxs + (qi[a] * sc.nscale + (L)) * kK;                       \
            int r_[kK];                                                                     \
            _Pragma("unroll") for (int p = 0; p < kK; p++) r_[p] = key_row(best[a].k[p]) & 0xFFFF; \
            *reinterpret_cast<I4 *>(out) = I4{{r_[0], r_[1], r_[2], r_[3]}};                \
            *reinterpret_cast<I4 *>(out + 4) = I4{{r_[4], r_[5], r_[6], r_[7]}};            \
            *reinterpret_cast<I2 *>(out + 8) = I2{{r_[8], r_[9]}};                          \
        }                                                                                   \
    }

        // coarsest scale: every point, original order
#pragma unroll
        for (int a = 0; a < kQ; a++) {
            thr[a] = live[a] ? INFINITY : -1.0f;      // a dead query accepts no candidate (d2 >= 0)
            sb[a] = INFINITY;
            kbest64_reset(best[a]);
        }
        OCC_SCAN(sc.coarse_begin, sc.coarse_end)
        OCC_EMIT(sc.nscale - 1)

        for (int l = sc.nscale - 2; l >= 0; l--) {
#pragma unroll
            for (int a = 0; a < kQ; a++) {
                sb[a] = sc.seed[l] ? key_dist(best[a].k[kK - 1]) : INFINITY;
                thr[a] = live[a] ? filter_bound(sb[a]) : -1.0f;
                kbest64_reset(best[a]);
            }
            const int2 *rg = ranges + (size_t)l * sc.ncl;
            const float *rd = radius + (size_t)l * sc.ncl;
            // Visit the cluster nearest to the tile first: it holds most of the true neighbours,
            // so the search radius collapses to its final value before the other clusters are
            // tested -- far fewer (divergent) list insertions and more clusters skipped.
            int k_first;
            {
                const float rx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qx[0][0])));
                const float ry = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qy[0][0])));
                const float rz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qz[0][0])));
                float bd = INFINITY;
                int bk = 0;
                for (int k = lane; k < sc.ncl; k += kWave) {
                    const float4 c = centers[k];
                    const float dx = rx - c.x, dy = ry - c.y, dz = rz - c.z;
                    const float d = dx * dx + dy * dy + dz * dz;
                    const bool has = rg[k].x < rg[k].y;
                    if (has && d < bd) {
                        bd = d;
                        bk = k;
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float od = __shfl_xor(bd, o);
                    const int ok = __shfl_xor(bk, o);
                    if (od < bd || (od == bd && ok < bk)) {
                        bd = od;
                        bk = ok;
                    }
                }
                k_first = __builtin_amdgcn_readfirstlane(bk);
            }
            {
                const int2 range = rg[k_first];
                OCC_SCAN(range.x, range.y)
            }
            // sphere test shared by groups and clusters: reject only if surely |q - c| - r > sb for every live query of the wave
            auto in_reach = [&](const float4 c, const float r) -> bool {
                bool want = false;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const f32x2 dx = qx[h] - c.x, dy = qy[h] - c.y, dz = qz[h] - c.z;
                    const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        // (1e-5 relative slack >> fp32 error)
                        const float lim = (sb[2 * h + e] + r) * 1.00001f;
                        want |= live[2 * h + e] && !(d2[e] > lim * lim);
                    }
                }
                return __builtin_amdgcn_ballot_w64(want) != 0;           // wave-uniform
            };
            // Clusters are listed group by group (geometry.build_knn_clusters): a group's sphere bounds everything its ~8
            // clusters hold at this scale, so a tile tests the ~14 group spheres and only the clusters of the groups in reach
            // instead of all 108 cluster spheres (those tests were a fifth of the kernel's instructions).
            const int ngrp = sc.ngrp > 0 ? sc.ngrp : 1;
            for (int gi = 0; gi < ngrp; gi++) {
                int k_lo = 0, k_hi = sc.ncl;
                if (sc.ngrp > 0) {
                    const float gr = gradius[(size_t)l * sc.ngrp + gi];
                    if (gr < 0.0f) continue;
                    if (!in_reach(gcenters[gi], gr)) continue;
                    const int2 gk = granges[gi];
                    k_lo = gk.x, k_hi = gk.y;
                }
                for (int k = k_lo; k < k_hi; k++) {
                    if (k == k_first) continue;
                    const int2 range = rg[k];
                    if (range.x >= range.y) continue;
                    if (!in_reach(centers[k], rd[k])) continue;
                    OCC_SCAN(range.x, range.y)
                }
            }
            OCC_EMIT(l)
        }
#undef OCC_PT4
#undef OCC_SCAN
#undef OCC_EMIT
    }
}

// Small generic kNN (k <= 16): LQ lanes per query (8, or a whole wave), lists in registers via a fixed-size unrolled
// insertion; used for the per-point k=3 search and the k=10 visibility update.  Lane g scans support rows g, g + LQ, ...
// (four rows' loads in flight per trip) into its own sorted list of fp64 (distance, row) keys (see KBest64), then the LQ
// lists are merged by butterfly exchange.  The key order is the (distance, row) order, i.e. exactly "strict '<' in row
// order" of the serial scan: the result does not depend on LQ.  (Round 5: with 8 lanes per query the P x P search of the
// training step's per-point block was 862 single-wave blocks of 861 dependent trips -- 0.26 ms; a wave per query: 6 890 waves
// of 108 trips.)
template <int K, int LQ>
__global__ __launch_bounds__(256) void knn_small_kernel(const float *__restrict__ q, int nq,
                                                        const float *__restrict__ s, int ns,
                                                        int32_t *__restrict__ idx) {
    const int g = threadIdx.x & (LQ - 1);
    const int i_raw = (blockIdx.x * blockDim.x + threadIdx.x) / LQ;
    const int i = i_raw < nq ? i_raw : nq - 1;            // keep all lanes for the shuffles
    const float qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
    double best[K];
#pragma unroll
    for (int p = 0; p < K; p++) best[p] = key64(INFINITY, 0x7fffffff);
    for (int j0 = g; j0 < ns; j0 += 4 * LQ) {
        float sx[4], sy[4], sz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * LQ < ns ? j0 + u * LQ : ns - 1;
            sx[u] = s[j * 3];
            sy[u] = s[j * 3 + 1];
            sz[u] = s[j * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * LQ;
            if (j >= ns) break;
            const float dx = qx - sx[u], dy = qy - sy[u], dz = qz - sz[u];
            double t = key64(sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)))), j);
            if (t < best[K - 1]) {
#pragma unroll
                for (int p = 0; p < K; p++) {
                    const double lo = fmin(best[p], t);
                    t = fmax(best[p], t);
                    best[p] = lo;
                }
            }
        }
    }
#pragma unroll
    for (int o = 1; o < LQ; o <<= 1) {
        double other[K];
#pragma unroll
        for (int p = 0; p < K; p++) other[p] = __shfl_xor(best[p], o, LQ);
#pragma unroll
        for (int e = 0; e < K; e++) {
            double t = other[e];
#pragma unroll
            for (int p = 0; p < K; p++) {
                const double lo = fmin(best[p], t);
                t = fmax(best[p], t);
                best[p] = lo;
            }
        }
    }
    if (g == 0 && i_raw < nq) {
#pragma unroll
        for (int p = 0; p < K; p++) idx[i * K + p] = key_row(best[p]);
    }
}

}  // namespace occ

OCC_API int occnerf_msknn(const float *xyz, int64_t N, const float *points,
                          const int32_t *index_map, const int32_t *h_scale_begin,
                          const int32_t *h_seed_from_coarser, int32_t nscale, int32_t *knn_idxs,
                          void *stream) {
    using namespace occ;
    if (N <= 0) return 0;
    OCC_REQUIRE(xyz && points && index_map && h_scale_begin && knn_idxs, "msknn: null argument");
    OCC_REQUIRE(nscale >= 1 && nscale <= 4, "msknn: nscale=%d unsupported (1..4)", nscale);
    if (N <= 0) return 0;
    MsKnnScales sc;
    sc.nscale = nscale;
    for (int l = 0; l < nscale; l++) {
        sc.begin[l] = h_scale_begin[l];
        sc.end[l] = h_scale_begin[l + 1];
        OCC_REQUIRE(sc.begin[l] % 4 == 0, "msknn: scale %d must start at a multiple of 4 rows "
                    "(pad each scale with +inf rows)", l);
        OCC_REQUIRE(sc.end[l] - sc.begin[l] >= kK, "msknn: scale %d has fewer than %d points", l, kK);
        sc.seed[l] = h_seed_from_coarser ? h_seed_from_coarser[l] : 0;
    }
    const int64_t tile = 256 * kQ;
    int64_t blocks = (N + tile - 1) / tile;
    if (blocks > (int64_t)kNumCU * 8) blocks = (int64_t)kNumCU * 8;
    hipLaunchKernelGGL(msknn_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), xyz, N,
                       reinterpret_cast<const float4 *>(points), index_map, sc, knn_idxs);
    return check_launch("msknn");
}

namespace occ {
// ray_start[r] = first entry of the ascending sample list that belongs to ray r or a later one (entry = sample index,
// ray = index / S); ray_start[n_rays] = the list's length.
__global__ void ray_list_ranges_kernel(const int32_t *__restrict__ qrows, const int32_t *__restrict__ n_dev, int64_t n_rays,
                                       int S, int32_t *__restrict__ ray_start) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rays) return;
    const int32_t n = *n_dev;
    const int64_t key = r * S;
    int32_t lo = 0, hi = n;                 // first entry >= key
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if ((int64_t)qrows[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    ray_start[r] = lo;
}
}  // namespace occ

OCC_API int occnerf_knn_center(const float *c, const float *points, const int32_t *index_map,
                               const int32_t *h_scale_begin, int32_t nscale, float *center_out, int32_t *idx_out, void *stream) {
    using namespace occ;
    OCC_REQUIRE(c && points && index_map && h_scale_begin && center_out && idx_out, "knn_center: null argument");
    OCC_REQUIRE(nscale >= 1 && nscale <= 4, "knn_center: nscale=%d unsupported (1..4)", nscale);
    MsKnnScales sc;
    sc.nscale = nscale;
    for (int l = 0; l < nscale; l++) {
        sc.begin[l] = h_scale_begin[l];
        sc.end[l] = h_scale_begin[l + 1];
        sc.seed[l] = 0;
        OCC_REQUIRE(sc.end[l] - sc.begin[l] >= kK, "knn_center: scale %d has fewer than %d points", l, kK);
    }
    hipLaunchKernelGGL(knn_center_kernel, dim3(1), dim3(256), 0, as_stream(stream), c, reinterpret_cast<const float4 *>(points),
                       index_map, sc, center_out, idx_out);
    return check_launch("knn_center");
}

OCC_API int occnerf_msknn_clustered(const float *xyz, const float *mask, int64_t n_rays, int32_t samples_per_ray,
                                    const float *points,
                                    const float *centers, const int32_t *cluster_ranges,
                                    const float *cluster_radius, int32_t ncl, const float *group_centers,
                                    const int32_t *group_ranges, const float *group_radius, int32_t ngrp,
                                    const int32_t *h_coarse_rows,
                                    const int32_t *h_seed_from_coarser, int32_t nscale,
                                    const int32_t *query_rows, const int32_t *n_query_dev, int32_t *ray_start,
                                    int32_t *knn_idxs, void *stream) {
    return occnerf_msknn_clustered_centered(xyz, mask, n_rays, samples_per_ray, points, centers, cluster_ranges, cluster_radius, ncl,
                                            group_centers, group_ranges, group_radius, ngrp, h_coarse_rows, h_seed_from_coarser,
                                            nscale, query_rows, n_query_dev, ray_start, nullptr, nullptr, knn_idxs, stream);
}

OCC_API int occnerf_msknn_clustered_centered(const float *xyz, const float *mask, int64_t n_rays, int32_t samples_per_ray,
                                             const float *points,
                                             const float *centers, const int32_t *cluster_ranges,
                                             const float *cluster_radius, int32_t ncl, const float *group_centers,
                                             const int32_t *group_ranges, const float *group_radius, int32_t ngrp,
                                             const int32_t *h_coarse_rows,
                                             const int32_t *h_seed_from_coarser, int32_t nscale,
                                             const int32_t *query_rows, const int32_t *n_query_dev, int32_t *ray_start,
                                             const float *center, const int32_t *center_idx,
                                             int32_t *knn_idxs, void *stream) {
    using namespace occ;
    OCC_REQUIRE(!center == !center_idx, "msknn_clustered: center and center_idx come together (occnerf_knn_center)");
    if (n_rays <= 0 || samples_per_ray <= 0) return 0;
    OCC_REQUIRE((!query_rows && !n_query_dev && !ray_start) || (query_rows && n_query_dev && ray_start),
                "msknn_clustered: query_rows, n_query_dev and the ray_start scratch come together");
    OCC_REQUIRE(!(query_rows && mask), "msknn_clustered: a query list replaces the mask");
    OCC_REQUIRE(xyz && points && centers && cluster_ranges && cluster_radius && h_coarse_rows && knn_idxs,
                "msknn_clustered: null argument");
    OCC_REQUIRE(nscale >= 2 && nscale <= 4, "msknn_clustered: nscale=%d unsupported (2..4)", nscale);
    OCC_REQUIRE(ncl >= 1, "msknn_clustered: ncl=%d", ncl);
    OCC_REQUIRE((ngrp == 0 && !group_centers && !group_ranges && !group_radius) ||
                    (ngrp > 0 && group_centers && group_ranges && group_radius),
                "msknn_clustered: group_centers, group_ranges, group_radius and ngrp come together");
    ClusteredScales sc;
    sc.nscale = nscale;
    sc.ncl = ncl;
    sc.ngrp = ngrp;
    sc.coarse_begin = h_coarse_rows[0];
    sc.coarse_end = h_coarse_rows[1];
    OCC_REQUIRE(sc.coarse_begin % 4 == 0 && (sc.coarse_end - sc.coarse_begin) % 4 == 0 &&
                    sc.coarse_end - sc.coarse_begin >= kK, "msknn_clustered: bad coarse row range");
    for (int l = 0; l < 4; l++) {
        sc.seed[l] = (l < nscale && h_seed_from_coarser) ? h_seed_from_coarser[l] : 0;
    }
    const int64_t tiles = ((n_rays + 63) / 64) * ((samples_per_ray + 3) / 4);
    OCC_REQUIRE(tiles < (1ll << 31), "msknn_clustered: too many tiles for one launch");
    // Small launches (at most 4 tiles per resident wave: a rank's eighth of a frame, a training batch) end with their longest
    // tile: they run the SPLIT form.  Large ones keep one ticket per tile (four would load every dead or cached tile four times:
    // +1.5 ms on the benchmark frame, profiles/r06_knn_split_tiles.md).
    const bool split = tiles <= (int64_t)kNumCU * 12 * 4;
    int64_t blocks = ((split ? 4 * tiles : tiles) + 3) / 4;      // one wave per ticket, up to the resident limit
    if (blocks > (int64_t)kNumCU * 3) blocks = (int64_t)kNumCU * 3;      // 12 resident waves per CU at this register count
    // Ticket counter of this launch: ONE 64-byte line per (device, stream), zeroed on the launch stream.  Launches on a stream
    // are in order, so the memset of launch n + 1 cannot pass the kernel of launch n, and launches on different streams never
    // share a line (round 5's 64-slot per-device ring could hand a line still in use on stream A to a launch on stream B, whose
    // tickets the older kernel would then have drawn).  The lines come from one pool per device allocated at the first call
    // (no allocation later, so a launch inside a stream capture is legal once any launch has happened outside one); a captured
    // launch replays its own memset node and owns the capture stream's line.
    constexpr int kTicketLines = 1024;
    static std::mutex mu;
    static std::map<int, unsigned *> pools;
    static std::map<std::pair<int, void *>, unsigned *> lines;
    int dev = 0;
    OCC_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0, "msknn_clustered: device id");
    unsigned *ticket;
    {
        std::lock_guard<std::mutex> lock(mu);
        unsigned *&pool = pools[dev];
        if (!pool)
            OCC_REQUIRE(hipMalloc(&pool, kTicketLines * 64) == hipSuccess,
                        "msknn_clustered: hipMalloc of the ticket lines (the first call must not be inside a stream capture)");
        auto it = lines.find(std::make_pair(dev, stream));
        if (it == lines.end()) {
            int used = 0;
            for (auto &kv : lines) used += kv.first.first == dev;
            OCC_REQUIRE(used < kTicketLines, "msknn_clustered: more than %d streams on device %d", kTicketLines, dev);
            it = lines.emplace(std::make_pair(dev, stream), pool + 16 * used).first;
        }
        ticket = it->second;
    }
    OCC_REQUIRE(hipMemsetAsync(ticket, 0, sizeof(unsigned), as_stream(stream)) == hipSuccess, "msknn_clustered: memset");
    if (query_rows)
        hipLaunchKernelGGL(ray_list_ranges_kernel, dim3((unsigned)((n_rays + 256) / 256)), dim3(256), 0, as_stream(stream),
                           query_rows, n_query_dev, n_rays, samples_per_ray, ray_start);
    auto kern = split ? msknn_clustered_kernel<true> : msknn_clustered_kernel<false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), xyz, mask,
                       n_rays, samples_per_ray, reinterpret_cast<const float4 *>(points),
                       reinterpret_cast<const float4 *>(centers), reinterpret_cast<const int2 *>(cluster_ranges),
                       cluster_radius, reinterpret_cast<const float4 *>(group_centers),
                       reinterpret_cast<const int2 *>(group_ranges), group_radius, sc, knn_idxs, ticket, query_rows, ray_start,
                       center, center_idx, kSplitExtent2);
    return check_launch("msknn_clustered");
}

OCC_API int occnerf_knn_small(const float *q, int32_t nq, const float *s, int32_t ns, int32_t k,
                              int32_t *idx, void *stream) {
    using namespace occ;
    if (nq <= 0) return 0;
    OCC_REQUIRE(q && s && idx, "knn_small: null argument");
    OCC_REQUIRE(ns >= k, "knn_small: fewer support points (%d) than k (%d)", ns, k);
    if (nq <= 0) return 0;
    // a few thousand queries at most: a whole wave per query when the support set is large enough to feed 64 lanes
    // (one-wave blocks spread over the CUs), 8 lanes per query otherwise
    hipStream_t st = as_stream(stream);
    const bool wide = ns >= 1024 && nq <= (1 << 20);
    const dim3 block(64), grid(wide ? nq : (nq + 7) / 8);
#define OCC_KS(KK)                                                                                      \
    case KK:                                                                                            \
        if (wide) hipLaunchKernelGGL((knn_small_kernel<KK, 64>), grid, block, 0, st, q, nq, s, ns, idx); \
        else hipLaunchKernelGGL((knn_small_kernel<KK, 8>), grid, block, 0, st, q, nq, s, ns, idx);       \
        break;
    switch (k) {
        OCC_KS(1)
        OCC_KS(3)
        OCC_KS(10)
        OCC_KS(16)
        default: set_error("knn_small: k=%d not built (1, 3, 10, 16)", k); return 1;
    }
#undef OCC_KS
    return check_launch("knn_small");
}
