// TEST INFRASTRUCTURE ONLY — run-time support of the reference harness (oracle/_ref build).
// Supplies the linker-wrapped rand()/srand() that feed the reference's unifRand()
// (/root/reference/main.cpp:19-22) and the function-entry/exit hooks that tell the wrap
// which pixel of Renderer::iterate() (/root/reference/main.cpp:349-365,
// next_event.cpp:374-473) is being sampled.  Compiled WITHOUT -finstrument-functions.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ptor_rng.h"
#include "ptor_rt.h"

#define PTOR_NOINST __attribute__((no_instrument_function))

namespace ptor {
scene_hook_fn scene_hook_target = 0;
}

namespace {

struct RtState {
    int mode;  // PTOR_MODE_*
    uint64_t seed;
    ptor_rng rng;       // current stream
    uint32_t iteration; // PIXEL mode
    uint32_t row;       // next row whose x jitter has not been drawn yet
    uint32_t pixel;     // next running pixel index
    bool frame_jitter_pending;
    bool in_pixel;
    std::vector<int> script;
    size_t script_pos;
    long long draws;
};

thread_local RtState g;

void* g_addr_getray[4] = {0, 0, 0, 0};
void* g_addr_iadd[4] = {0, 0, 0, 0};

}  // namespace

extern "C" {

PTOR_NOINST void ptor_rt_register(int prog, void* getray, void* iadd) {
    g_addr_getray[prog] = getray;
    g_addr_iadd[prog] = iadd;
}

PTOR_NOINST void ptor_rt_mode_stream(uint64_t seed, uint32_t iteration, uint32_t domain, uint32_t index) {
    g.mode = PTOR_MODE_SEQ;
    g.seed = seed;
    ptor_rng_init(&g.rng, seed, iteration, domain, index);
    g.draws = 0;
}

PTOR_NOINST void ptor_rt_mode_seq(uint64_t seed, uint32_t stream) {
    ptor_rt_mode_stream(seed, 0, PTOR_DOM_SEQ, stream);
}

PTOR_NOINST void ptor_rt_mode_script(const int* k, int n) {
    g.mode = PTOR_MODE_SCRIPT;
    g.script.assign(k, k + n);
    g.script_pos = 0;
    g.draws = 0;
}

PTOR_NOINST void ptor_rt_mode_pixel(uint64_t seed) {
    g.mode = PTOR_MODE_PIXEL;
    g.seed = seed;
    g.draws = 0;
}

// Call before each Renderer::iterate() in PIXEL mode.
PTOR_NOINST void ptor_rt_begin_iteration(uint32_t iteration) {
    g.iteration = iteration;
    g.row = 0;
    g.pixel = 0;
    g.frame_jitter_pending = true;
    g.in_pixel = false;
}

PTOR_NOINST long long ptor_rt_draws(void) { return g.draws; }

PTOR_NOINST int __wrap_rand(void) {
    g.draws++;
    switch (g.mode) {
        case PTOR_MODE_SCRIPT: {
            if (g.script.empty()) return 0;
            int k = g.script[g.script_pos];
            g.script_pos = (g.script_pos + 1) % g.script.size();
            return k;
        }
        case PTOR_MODE_PIXEL: {
            if (!g.in_pixel) {
                // Outside a pixel the only draws iterate() makes are the y jitter (once,
                // first) and one x jitter per row, in row order.
                ptor_rng j;
                if (g.frame_jitter_pending) {
                    g.frame_jitter_pending = false;
                    ptor_rng_init(&j, g.seed, g.iteration, PTOR_DOM_FRAME, 0);
                } else {
                    ptor_rng_init(&j, g.seed, g.iteration, PTOR_DOM_ROW, g.row++);
                }
                return ptor_rng_next(&j);
            }
            return ptor_rng_next(&g.rng);
        }
        default:
            return ptor_rng_next(&g.rng);
    }
}

PTOR_NOINST void __wrap_srand(unsigned) {}

// -finstrument-functions hooks (only the hooked build of ref_harness.cpp emits calls).
PTOR_NOINST void __cyg_profile_func_enter(void* fn, void*) {
    if (g.mode != PTOR_MODE_PIXEL) return;
    if (fn == g_addr_getray[0] || fn == g_addr_getray[1] || fn == g_addr_getray[2] || fn == g_addr_getray[3]) {
        // Camera::getRay is called exactly once per pixel, first thing in the pixel body.
        ptor_rng_init(&g.rng, g.seed, g.iteration, PTOR_DOM_PIXEL, g.pixel++);
        g.in_pixel = true;
    }
}

PTOR_NOINST void __cyg_profile_func_exit(void* fn, void*) {
    if (g.mode != PTOR_MODE_PIXEL) return;
    if (fn == g_addr_iadd[0] || fn == g_addr_iadd[1] || fn == g_addr_iadd[2] || fn == g_addr_iadd[3]) {
        // V3::iadd is called exactly once per pixel, last thing in the pixel body.
        g.in_pixel = false;
    }
}

}  // extern "C"
