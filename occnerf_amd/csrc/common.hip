// Library-wide host helpers: error reporting, ABI version, grid level constants.
#include "common.h"

#include <cmath>
#include <cstdarg>

namespace occ {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// gridencoder.cu:137-139: scale = exp2f(level * S) * H - 1 (one fma, as nvcc contracts it),
// resolution = ceil(scale) + 1.  Host libm, shared bit-for-bit with the oracle.
GridLevels make_grid_levels(uint32_t L, float S, uint32_t H) {
    GridLevels lv;
    for (uint32_t l = 0; l < (uint32_t)kMaxLevels; l++) {
        if (l < L) {
            lv.scale[l] = std::fmaf(exp2f((float)l * S), (float)H, -1.0f);
            lv.resolution[l] = (uint32_t)std::ceil(lv.scale[l]) + 1;
        } else {
            lv.scale[l] = 0.f;
            lv.resolution[l] = 0;
        }
    }
    return lv;
}

// Index mode per level for D = 4 (see GridIndexMode).  Dense iff the reference's stride loop
// (gridencoder.cu:70-74) consumes all four dims without exceeding the level's size.
GridModes4 make_grid_modes_d4(uint32_t L, const GridLevels &lv, const uint32_t *h_level_sizes) {
    GridModes4 m;
    for (uint32_t l = 0; l < (uint32_t)kMaxLevels; l++) {
        m.mode[l] = kGridGeneric;
        if (l >= L) continue;
        const uint64_t size = h_level_sizes[l];
        uint64_t stride = 1;
        bool all = true;
        for (int d = 0; d < 4; d++) {
            if (stride > size) { all = false; break; }
            stride *= (uint64_t)lv.resolution[l] + 1;
        }
        if (all && stride <= size) m.mode[l] = kGridDense;               // hashing never kicks in
        else if (!(all && stride <= size) && size > 0 && (size & (size - 1)) == 0 && stride > size)
            m.mode[l] = kGridHashPow2;
    }
    return m;
}

}  // namespace occ

OCC_API int occnerf_abi_version(void) { return OCCNERF_ABI_VERSION; }
OCC_API const char *occnerf_last_error(void) { return occ::g_err; }
