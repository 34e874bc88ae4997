// Library-wide host helpers: error reporting, ABI version, grid level constants.
#include "common.h"

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdlib>

namespace occ {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- experiment knobs ----------------------------------------------------------------------------------------------
struct KnobSpec {
    const char *name, *env;
    int lo, hi;
};
static const KnobSpec kKnobSpecs[kKnobCount] = {{"agg_slices", "OCCNERF_AGG_SLICES", 0, 1024},
                                                {"grid_xcd", "OCCNERF_GRID_XCD", 0, 2}};
static std::atomic<int> g_knob[kKnobCount];
static std::atomic<bool> g_knob_read[kKnobCount];

static int clamp_knob(int k, long v) {
    return (int)(v < kKnobSpecs[k].lo ? kKnobSpecs[k].lo : (v > kKnobSpecs[k].hi ? kKnobSpecs[k].hi : v));
}

int knob(Knob k) {
    if (!g_knob_read[k].load(std::memory_order_acquire)) {
        const char *e = getenv(kKnobSpecs[k].env);
        g_knob[k].store(e ? clamp_knob(k, strtol(e, nullptr, 10)) : 0);
        g_knob_read[k].store(true, std::memory_order_release);
    }
    return g_knob[k].load(std::memory_order_relaxed);
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

OCC_API int occnerf_experiment_knob(const char *name, int value) {
    for (int k = 0; k < occ::kKnobCount; k++) {
        if (name && strcmp(name, occ::kKnobSpecs[k].name) == 0) {
            const int prev = occ::knob((occ::Knob)k);
            if (value >= 0) occ::g_knob[k].store(occ::clamp_knob(k, value));
            return prev;
        }
    }
    occ::set_error("occnerf_experiment_knob: unknown knob '%s'", name ? name : "(null)");
    return -1;
}
