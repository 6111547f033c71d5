/* TEST INFRASTRUCTURE ONLY — interface between the reference harness and its rand() wrap. */
#ifndef PTOR_RT_H
#define PTOR_RT_H
#include <stdint.h>

/* Random-number modes shared by both oracle builds. */
enum {
    PTOR_MODE_SEQ = 0,    /* one sequential stream, like the reference's global rand() */
    PTOR_MODE_SCRIPT = 1, /* cyclic list of scripted draws (known-answer tests) */
    PTOR_MODE_PIXEL = 2   /* per-(iteration,pixel) streams + per-row / per-frame jitter streams */
};

#ifdef __cplusplus
/* How the harness borrows the Scene that a reference program's main() builds on its stack: the
 * program's own call `save(buffer, iterations, width, height)` at the end of main() is compiled, for
 * main() only, as a call of ptor::scene_hook(&scene) (a macro set between the definition of save() and
 * main(); see ref_harness.cpp). */
namespace ptor {
typedef void (*scene_hook_fn)(void* scene);
extern scene_hook_fn scene_hook_target;
inline void scene_hook(void* scene) {
    if (scene_hook_target) scene_hook_target(scene);
}
}  // namespace ptor

extern "C" {
#endif
void ptor_rt_register(int prog, void* getray, void* iadd);
void ptor_rt_mode_seq(uint64_t seed, uint32_t stream);
void ptor_rt_mode_stream(uint64_t seed, uint32_t iteration, uint32_t domain, uint32_t index);
void ptor_rt_mode_script(const int* k, int n);
void ptor_rt_mode_pixel(uint64_t seed);
void ptor_rt_begin_iteration(uint32_t iteration);
long long ptor_rt_draws(void);
#ifdef __cplusplus
}
#endif
#endif
