/* TEST INFRASTRUCTURE ONLY — part of the parity oracle, never linked into the product.
 *
 * Counter-based per-stream random number contract shared by the two oracle builds
 * (the harness that drives the reference's own code, and the plain-C restatement).
 * The product's HIP code restates the same contract independently; tests compare them.
 *
 * The reference draws from libc:  unifRand() = rand() / double(RAND_MAX)
 * (/root/reference/main.cpp:19-22, next_event.cpp:19-22), one global sequential stream.
 * A sequential global stream cannot be reproduced by a parallel renderer, so every
 * (iteration, pixel) gets its own stream and the reference is driven with those streams
 * through a linker wrap of rand().  A draw is an integer k in [0, 2^31-1] (= glibc's
 * RAND_MAX, both ends reachable) and u = (double)k / 2147483647.0, exactly the mapping
 * the reference applies to rand().
 *
 * Stream key: (seed, iteration, domain, index).
 *   domain 0 PIXEL : index = running pixel index i of iterate() (main.cpp:350,362)
 *   domain 1 ROW   : index = row number; its first draw is that row's x jitter (main.cpp:355)
 *   domain 2 FRAME : index = 0; its first draw is the iteration's y jitter (main.cpp:352)
 *   domain 3 SEQ   : index = stream id; one long sequential stream (reference-like mode)
 * State: xoshiro128+ (Blackman/Vigna), seeded by three rounds of the splitmix64 finaliser.
 */
#ifndef PTOR_RNG_H
#define PTOR_RNG_H
#include <stdint.h>

enum { PTOR_DOM_PIXEL = 0, PTOR_DOM_ROW = 1, PTOR_DOM_FRAME = 2, PTOR_DOM_SEQ = 3 };

typedef struct ptor_rng {
    uint32_t s[4];
} ptor_rng;

static inline uint64_t ptor_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline void ptor_rng_init(ptor_rng* r, uint64_t seed, uint32_t iteration, uint32_t domain,
                                 uint32_t index) {
    uint64_t a = ptor_mix64(seed + 0x9E3779B97F4A7C15ULL * ((((uint64_t)domain << 32) | iteration) + 1ULL));
    uint64_t b = ptor_mix64(a ^ (0xD1B54A32D192ED03ULL * ((uint64_t)index + 1ULL)));
    uint64_t c = ptor_mix64(b + 0x9E3779B97F4A7C15ULL);
    r->s[0] = (uint32_t)b;
    r->s[1] = (uint32_t)(b >> 32);
    r->s[2] = (uint32_t)c;
    r->s[3] = (uint32_t)(c >> 32);
    if ((r->s[0] | r->s[1] | r->s[2] | r->s[3]) == 0u) r->s[0] = 0x9E3779B9u;
}

/* Next draw k in [0, 2^31-1]. */
static inline int32_t ptor_rng_next(ptor_rng* r) {
    uint32_t result = r->s[0] + r->s[3];
    uint32_t t = r->s[1] << 9;
    r->s[2] ^= r->s[0];
    r->s[3] ^= r->s[1];
    r->s[1] ^= r->s[2];
    r->s[0] ^= r->s[3];
    r->s[2] ^= t;
    r->s[3] = (r->s[3] << 11) | (r->s[3] >> 21);
    return (int32_t)(result >> 1);
}

#endif
