// Device-side building blocks of the estimator (gfx950, FP64 VALU, no FMA contraction).
//
// Every function states the reference lines it reproduces.  The reference is x86-64 g++ code:
// IEEE double without fused multiply-add, evaluated left to right.  This translation unit must
// be compiled with -ffp-contract=off; the only fused operations are explicit __builtin_fma calls
// whose result is proven equal to the unfused reference expression.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_x87.hpp"

namespace ptd {

#define PTD_INLINE __device__ __forceinline__
#define PTD_HOST_DEVICE __host__ __device__

// ---- optional section timing (development builds only: -DPT_PROFILE_SECTIONS) -------------
// PT_PROF(k) times the enclosing scope with the wave's cycle counter and adds the time to a
// per-wave LDS slot; the fused kernel flushes the slots to g_prof when it ends.  Nested scopes
// are counted in both.  Without the macro nothing is emitted.
#ifdef PT_PROFILE_SECTIONS
enum {
    kProfTotal, kProfSetup, kProfScanPlanes, kProfScanScreen, kProfScanExact, kProfShade, kProfGlass, kProfDiffuse,
    kProfChrome, kProfNormalSphere, kProfNeePrep, kProfNeeVertex, kProfRecords,
    kProfRejectTrips,  // not a time: trips of the hemisphere rejection loop per wave (slot k) and per lane (slot kProfN + k)
    kProfN
};
constexpr int kProfWaves = 4;
// per wave: slot k = cycles of section k, slot kProfN + k = the same cycles times the lanes active
// when the section ended (lanes that take part in the section; loops inside may run with fewer)
constexpr int kProfSlots = 2 * kProfN;
__shared__ long long s_prof[kProfWaves * kProfSlots];
__device__ unsigned long long g_prof[kProfSlots];
struct ProfScope {
    int k;
    long long t0;
    PTD_INLINE ProfScope(int kk) : k(kk), t0(__builtin_amdgcn_s_memtime()) {}
    PTD_INLINE ~ProfScope() {
        long long dt = (long long)__builtin_amdgcn_s_memtime() - t0;
        unsigned long long act = __builtin_amdgcn_ballot_w64(true);
        int lane = (int)(threadIdx.x & 63);
        if (lane == __builtin_ctzll(act)) {
            s_prof[(threadIdx.x >> 6) * kProfSlots + k] += dt;
            s_prof[(threadIdx.x >> 6) * kProfSlots + kProfN + k] += dt * (long long)__builtin_popcountll(act);
        }
    }
};
// one trip of a loop that lanes leave one by one: counted once per wave and once per active lane
PTD_INLINE void prof_trip(int k) {
    unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(act)) {
        s_prof[(threadIdx.x >> 6) * kProfSlots + k] += 1;
        s_prof[(threadIdx.x >> 6) * kProfSlots + kProfN + k] += (long long)__builtin_popcountll(act);
    }
}
#define PT_PROF_TRIP(k) prof_trip(k)
#define PT_PROF_CAT2(a, b) a##b
#define PT_PROF_CAT(a, b) PT_PROF_CAT2(a, b)
#define PT_PROF(k) ProfScope PT_PROF_CAT(prof_scope_, __LINE__)(k)
#else
#define PT_PROF(k) \
    do {           \
    } while (0)
#define PT_PROF_TRIP(k) \
    do {                \
    } while (0)
#endif

// ---- optional duplicate-work builds (development only: -DPT_DUP_REJECT / _SLAB / _SCREEN) -------
// One part of the estimator is executed twice per use and the duplicate's results are kept alive
// but discarded: paths, trip counts and images are unchanged, and the slow-down of the kernel is
// that part's share of its time (scripts/ko_bench.sh).  Without the macros nothing is emitted.
PTD_INLINE void keep_alive(double v) { asm volatile("" ::"v"(v)); }
PTD_INLINE void keep_alive(int v) { asm volatile("" ::"v"(v)); }

constexpr int kMaxDepth = 5;  // trace() returns at n > 4: main.cpp:368
constexpr int kGroupBodies = 8;  // member slots per group of the grouped scan
constexpr int kSuperGroups = 8;  // groups per super-group
constexpr int kBrdfN = 90 * 90 * 180;  // MatusikBRDF.h:24-26

enum { kSphere = 0, kPlane = 1, kPlaneX = 2, kPlaneY = 3, kPlaneZ = 4, kRect = 5, kRectMT = 6 };  // 2..4: device-only, normal = +-e_k; 5: raytrace3's Rectangle; 6: light_tracing's
enum { kDiffuse = 0, kChrome = 1, kGlass = 2, kMeasured = 3 };
enum { kEstPT = 0, kEstNEE = 1, kEstBidir = 2 };
enum { kDomPixel = 0, kDomRow = 1, kDomFrame = 2, kDomSeq = 3 };

// ---- V3: main.cpp:50-103 ---------------------------------------------------------------
struct V3 {
    double x, y, z;
};
PTD_INLINE V3 mk(double x, double y, double z) {
    V3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
PTD_INLINE V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PTD_INLINE V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PTD_INLINE V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PTD_INLINE V3 muls(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
// dot: x*v.x + y*v.y + z*v.z, left to right (main.cpp:96-98)
PTD_INLINE double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// normalize = divs(sqrt(dot)) = muls(1.0L / s) (main.cpp:92-102): the reciprocal is formed in
// x87 extended precision and narrowed to the double parameter of muls.
PTD_INLINE V3 normalize(V3 a) { return muls(a, ptx::x87_recip(sqrt(dot(a, a)))); }
PTD_INLINE bool same_point(V3 a, V3 b) { return (a.x == b.x) & (a.y == b.y) & (a.z == b.z); }
// The only NaN the estimator can produce from finite scenes is the flux of a lit vertex that
// coincides with the previous one (see nee_vertex); flux is an fabs(), so the reference's pixel
// holds the quiet NaN with the sign clear, which then propagates unchanged.  Samples are stored
// with exactly that pattern so that accumulators match the reference bit for bit there too.
PTD_INLINE double flux_nan() { return __longlong_as_double((long long)0x7FF8000000000000ULL); }
PTD_INLINE double canonical_nan(double v) { return v != v ? flux_nan() : v; }
PTD_INLINE V3 canonical_nan(V3 v) { return mk(canonical_nan(v.x), canonical_nan(v.y), canonical_nan(v.z)); }

// ---- random numbers --------------------------------------------------------------------
// Stream contract: xoshiro128+ state seeded from (seed, iteration, domain, index) by three
// splitmix64 finalisers; a draw is k = (s0 + s3) >> 1 in [0, 2^31-1] and
// u = k / 2147483647.0, the mapping unifRand() applies to rand() (main.cpp:19-22).
struct Rng {
    uint32_t s0, s1, s2, s3;
};

PTD_INLINE uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

PTD_INLINE Rng rng_init(uint64_t seed, uint32_t iteration, uint32_t domain, uint32_t index) {
    uint64_t a = mix64(seed + 0x9E3779B97F4A7C15ULL * ((((uint64_t)domain << 32) | iteration) + 1ULL));
    uint64_t b = mix64(a ^ (0xD1B54A32D192ED03ULL * ((uint64_t)index + 1ULL)));
    uint64_t c = mix64(b + 0x9E3779B97F4A7C15ULL);
    Rng r;
    r.s0 = (uint32_t)b;
    r.s1 = (uint32_t)(b >> 32);
    r.s2 = (uint32_t)c;
    r.s3 = (uint32_t)(c >> 32);
    if ((r.s0 | r.s1 | r.s2 | r.s3) == 0u) r.s0 = 0x9E3779B9u;
    return r;
}

// One xoshiro128+ step; returns the 32-bit sum whose upper 31 bits are the draw.  The textbook form
//   t = s1 << 9; s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 11)
// makes every new word the exclusive-or of three old ones (s1' = s1^s2^s0, s0' = s0^s3^s1,
// s2' = s2^s0^t), which gfx950 does in one v_bitop3_b32 each (truth table 0x96): 7 VALU
// instructions a step instead of 8.
PTD_INLINE uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
PTD_INLINE uint32_t rng_next_sum(Rng& r) {
    const uint32_t result = r.s0 + r.s3;
    const uint32_t t = r.s1 << 9;
    const uint32_t b = r.s3 ^ r.s1;
    const uint32_t n1 = xor3(r.s1, r.s2, r.s0);
    const uint32_t n0 = xor3(r.s0, r.s3, r.s1);
    const uint32_t n2 = xor3(r.s2, r.s0, t);
    r.s0 = n0;
    r.s1 = n1;
    r.s2 = n2;
    r.s3 = __builtin_amdgcn_alignbit(b, b, 21u);  // rotl(b, 11)
    return result;
}
PTD_INLINE int32_t rng_next(Rng& r) { return (int32_t)(rng_next_sum(r) >> 1); }

// rand() / double(RAND_MAX) with RAND_MAX = 2147483647, i.e. the correctly rounded quotient
// k / 2147483647.0.  One multiplication by the rounded reciprocal plus one exact-residual
// correction yields that quotient for every k in [0, 2^31-1] (checked exhaustively over all 2^31
// draws, tests/test_rng_exhaustive.py), at 3 VALU operations instead of a ~15-instruction division.
PTD_INLINE double k_to_unit(int32_t k) {
    const double D = 2147483647.0;
    const double c = 0x1.00000002p-31;  // RN(1 / D)
    double kd = (double)k;
    double q0 = kd * c;
    double r = __builtin_fma(-q0, D, kd);  // exact residual
    return __builtin_fma(r, c, q0);
}

// k_to_unit(k) * 2.0 - 1.0, the coordinate of a hemisphere candidate (main.cpp:108), with the doubling
// folded into the constants: scaling by two commutes with every rounding of k_to_unit, so
// 2 q0 = kd (2c), the residual kd - (2 q0)(D / 2) is the same number, and fma(r, 2c, 2 q0) is twice
// the unit value exactly (D / 2 = 1073741823.5 is a double).
PTD_INLINE double k_to_coord(int32_t k) {
    const double Dh = 1073741823.5;
    const double c2 = 0x1.00000002p-30;
    double kd = (double)k;
    double q0 = kd * c2;
    double r = __builtin_fma(-q0, Dh, kd);
    return __builtin_fma(r, c2, q0) - 1.0;
}

struct Draws {
    Rng rng;
    int count;
};
PTD_INLINE double unif(Draws& d) {
    d.count++;
    return k_to_unit(rng_next(d.rng));
}
// unifRand()*2.0L-1.0L narrowed to double (main.cpp:108) is written u * 2.0 - 1.0: for
// u >= 2^-13 the extended-precision difference is exact, and for the 2^18 smaller draws an
// exhaustive check shows that the plain double expression rounds to the same value.

// Lane masks: a compare taken as a 64-bit wave mask (the SGPR pair v_cmp writes).
typedef unsigned long long lanemask;
#define PTD_MASK(cond) __builtin_amdgcn_ballot_w64(cond)

// ---- scene records ---------------------------------------------------------------------
struct DGeom {  // 64 bytes, what the nearest-hit scan reads
    double cx, cy, cz;
    double nx;  // plane: stored (normalised) normal x; sphere: radius*radius
    union {
        struct {
            double ny, nz;  // plane normal y, z
        };
        struct {
            float fcx, fcy, fcz, fr2;  // sphere: centre and radius^2 rounded to float (screening only)
        };
    };
    int32_t shape;
    int32_t pad[3];  // scan lists: pad[0] = body index; pad[1..2] = a double, see origin0_term
};
// Scan-list records carry the part of the test that depends only on the ray origin, evaluated
// for the origin (0,0,0) where the shadow rays of the reference start (next_event.cpp:167,406-407):
// plane: n.(c - 0) = nx*cx + ny*cy + nz*cz;  sphere: |0 - c|^2 - r^2.  Same operations in the same
// order as the scan would perform them, so using the stored value changes no bit.
PTD_INLINE double origin0_term(const DGeom& g) {
    return __hiloint2double(g.pad[2], g.pad[1]);
}

// Rectangle (raytrace3.cpp:244-276) as its constructor stores it: p1, p3, v1 = normalize(p2 - p1),
// v3 = normalize(p4 - p3), normal = normalize((p3 - p1) x (p2 - p1)).  The body's DGeom carries the
// normal (shape kRect) and, in pad[0], the index of its DRect.
struct DRect {  // 128 bytes
    double p1[3], p3[3], v1[3], v3[3], n[3];
    int32_t body;
    int32_t pad;
};

// Rectangle of the light-tracing programs (light_tracing.cpp:477-495) as its constructor stores it:
// p1 and the edges e1 = p2 - p1, e2 = p3 - p1, e3 = p4 - p1.  The body's DGeom carries the normal
// (shape kRectMT) and, in pad[0], the index of its DRectMT.
struct DRectMT {  // 128 bytes
    double p1[3], e1[3], e2[3], e3[3];
    int32_t body;
    int32_t pad[7];
};

// Per-ray data of the single-precision sphere screen (see make_rayf).
struct RayF {
    float ox, oy, oz, dx, dy, dz;
    float e_disc;  // |float disc - exact disc| is below this for every sphere of the scene
    float e_b;     // likewise for b
};
struct DMat {  // 96 bytes
    double colr, colg, colb;
    double emr, emg, emb;
    double ior, refl;
    double eta;  // 1.0 / ior (Glass::bounce main.cpp:291)
    int32_t material;
    int32_t brdf_id;
    int32_t emits;  // any emission channel != 0
    // an emitter of colour exactly (0, 0, 0) in a scene whose next-event direct term is dead and whose
    // colours and emissions are small enough for every radiance to stay finite
    int32_t absorber;
    int32_t pad[2];
};

// Screening records of the grouped scan: sphere centre and radius^2 in float, two consecutive
// bodies interleaved per 32-byte record so that a pair is screened with packed FP32 instructions
// (v_pk_add/mul/fma_f32).  For every non-sphere (and the padding slot of an odd body count) r2 is
// NaN, which makes the screen inconclusive so that the full record is examined.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct DScreen2 {
    f32x2 cx, cy, cz, r2;  // .x: body 2k, .y: body 2k+1
};

// The exact record of a member slot of the grouped scan: what Sphere::intersect needs, and the
// body's index (-1: a padding slot), side by side so that one round trip fetches both.
struct DSlotExact {  // 48 bytes
    double cx, cy, cz, r2;
    int32_t body;
    int32_t pad[3];
};

struct DScene {
    const DGeom* geom;
    // Grouped layout of the large-scene scan: n_groups groups of kGroupBodies member slots
    // (n_groups is even); bounds and members are float (centre, r^2) records, two per DScreen2;
    // grp_exact[slot] is the FP64 record of a member slot.
    const DScreen2* grp_supers;  // bounds of kSuperGroups consecutive groups (n_groups is a multiple of 2 kSuperGroups)
    const DScreen2* grp_halves;  // per section of kSectionGroups groups: the bounds of its two halves
    const float* grp_half4;      // the same bounds one by one: (cx, cy, cz, R) of half h at [4 h]
    // Visiting order of the halves, nearest first, for rays starting in each of the kOrderCells^3 cells
    // of the sort box: entry [cell * 32 + k] packs the halves of step k as h0 | h1 << 8.  Null when the
    // scene has more than 64 halves (they are then visited in index order).
    const uint32_t* grp_half_order;
    const DScreen2* grp_bounds;
    const DScreen2* grp_members;
    const DSlotExact* grp_exact;
    int32_t n_groups;
    float sort_mid[3], sort_inv[3];  // origin cells of the ray sort: cell coordinate = (o - mid) * inv in [-1, 1]
    // Compact lists of the fused scan (body index in pad[0]): the planes that belong to no slab;
    // the spheres padded to an even count with a record whose screen can never pass, and the
    // spheres' paired screen records.
    const DGeom* scan_planes;
    const DGeom* scan_spheres;
    const DScreen2* scan_screen;
    const DScreen2* scan_screen0;  // the same pairs with |c|^2 - r^2 in place of r^2: the screen from the origin (0,0,0)
    const DRect* scan_rects;       // rectangles in ascending body order
    const DRectMT* scan_rects_mt;  // Moller-Trumbore rectangles in ascending body order
    // |det| below this counts as parallel: det > -EPSILON && det < EPSILON with the long double literal
    // 0.000001L (light_tracing.cpp:20,542) is, for a double det, |det| < the smallest double not below it
    double mt_eps;
    int32_t n_planes, n_spheres, n_rects, n_rects_mt;
    // A hit needs t > hit_eps: 0 (main.cpp:378, next_event.cpp:502) or, for raytrace3.cpp:776, the
    // largest double below 0.0000001L (t > 0.0000001L for a double t is t > that double).
    double hit_eps;
    // Slab of axis k (bit k of slab_mask): the two axis-aligned planes of that axis with the
    // smallest and the largest coordinate, bodies slab_ilo[k] and slab_ihi[k].  slab_big[k] bounds
    // |o_k| for the one-division form (see scan_slab).
    int32_t slab_mask;
    int32_t slab_ilo[3], slab_ihi[3];
    double slab_lo[3], slab_hi[3], slab_big[3];
    const DMat* mat;
    const int32_t* lights;  // body indices with any emission channel >= 1 (next_event.cpp:362-367)
    const double* brdf[8];
    const double* brdf_bins;  // bin edges of the table indices as thresholds (see brdf_index_of_vectors)
    float sphere_c1max;  // max over spheres of |cx|+|cy|+|cz|   (error bound of the float screen)
    float sphere_r2max;  // max over spheres of radius^2
    int32_t n_bodies;
    int32_t n_lights;
    int32_t width, height;
    int32_t estimator;
    uint32_t flags;
    int32_t nee_emission_only;  // see trace_path_nee_compact
    int32_t flux_finite;        // geometry and colours are finite and moderate: see nee_vertex
    uint64_t seed;
    // Camera (main.cpp:144-152): xd = topright - topleft, yd = bottomleft - topleft
    double cam_o[3], cam_tl[3], cam_xd[3], cam_yd[3];
};

PTD_INLINE V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }

// A scene record at a wave-uniform address, read through the constant address space: the scan
// lists are never written while a kernel runs, and a load from that address space is a scalar
// load whatever stands between it and the start of the kernel.  (Plain global loads are scalar only
// while the compiler can show that nothing may have written the memory; a fence or an atomic,
// even one confined to LDS, ends that, and every later record load becomes a vector load.)
template <class T>
PTD_INLINE T load_record(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "records are whole dwords");
    typedef const __attribute__((address_space(4))) uint32_t* const_dwords;
    const const_dwords q = (const_dwords)(uintptr_t)p;
    T out;
    uint32_t* o = (uint32_t*)&out;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) o[i] = q[i];
    return out;
}

// The wave-uniform controls of the scan (list lengths, slab bits, the shadow-origin flag) as fresh
// scalar values.  An estimator runs its scans in phases — the gather loop, then the loop over shadow
// or light-path rays — and lanes that leave the first phase early come back for the second.  The
// compiler may park a uniform boolean such as "the scene has an x slab" in a vector register and read
// it back, inside the first loop, as a mask of the lanes active there; reused by the second phase it
// is wrong for the lanes that had left.  Values that pass through this barrier cannot be traced back
// to the ones the first phase tested, so the second phase derives its controls anew.
PTD_INLINE void refresh_scan_controls(DScene& S) {
    asm volatile("" : "+s"(S.n_planes), "+s"(S.n_spheres), "+s"(S.slab_mask), "+s"(S.flags));
}

// Camera::getPoint next_event.cpp:159-161: topleft.add(xd.muls(x)).add(yd.muls(y))
PTD_INLINE V3 cam_point(const DScene& S, double x, double y) {
    return add(add(ld3(S.cam_tl), muls(ld3(S.cam_xd), x)), muls(ld3(S.cam_yd), y));
}

// ---- intersection ----------------------------------------------------------------------
// Plane::intersect main.cpp:181-191.  The reference compares against the long double literals
// +-0.00001L; for a double operand that is the same as comparing with the double 0.00001
// (the 53-bit round-up of 0.00001L equals the double literal).
PTD_INLINE double intersect_plane(const DGeom& g, V3 o, V3 d) {
    double n_dot_u = g.nx * d.x + g.ny * d.y + g.nz * d.z;
    double n_dot_p0 = g.nx * (g.cx - o.x) + g.ny * (g.cy - o.y) + g.nz * (g.cz - o.z);
    double t = n_dot_p0 / n_dot_u;
    return ((n_dot_u > -0.00001) && (n_dot_u < 0.00001)) ? -1.0 : t;
}

// Sphere::intersect main.cpp:210-222 (near root only, assumes |d| = 1).
PTD_INLINE double intersect_sphere(const DGeom& g, V3 o, V3 d) {
    double dx = o.x - g.cx, dy = o.y - g.cy, dz = o.z - g.cz;
    double b = dx * d.x + dy * d.y + dz * d.z;
    double c = (dx * dx + dy * dy + dz * dz) - g.nx;
    double disc = (b * b) - c;
    double t = -1.0;
    if (disc > 0.0) t = -b - sqrt(disc);
    return t;
}

// Planes whose stored normal is exactly +-e_k take a one-division form in the scan: with
// n = s*e_k the reference's two dot products are exactly s*d_k and s*(c_k - o_k) (the other terms
// are signed zeros), and their quotient is (c_k - o_k) / d_k bit for bit whenever it is non-zero;
// zero or negative results never win the scan.  Valid for finite rays.

// Rectangle::intersect raytrace3.cpp:283-308: the plane test, `t < 1.0L` counts as a miss, then
// the hit point must lie on the positive side of v1 seen from p1 and of v3 seen from p3 (both
// differences normalised first, as the reference does — a hit exactly on p1 or p3 normalises a zero
// vector and misses).  Returns false where the reference returns -1.
// The two side tests compare v . normalize(w) with 0.  Scaling w by the positive factor 1/|w| and
// rounding cannot move the dot product across zero unless it is already within rounding error of
// it: with s = v . w and m = sum |v_k w_k|, the normalised form differs from s/|w| by less than
// 2^-50 m/|w|, so |s| > 2^-48 m decides the sign of both (m above 2^-400 keeps |w|^2 and the
// products clear of underflow, below 2^400 clear of overflow).  Undecided lanes — a hit within rounding
// error of the edge line, or a degenerate w — take the reference's own sequence of operations.
PTD_INLINE bool side_decided(V3 v, V3 w, bool& positive) {
    const double s = dot(v, w);
    const double m = (__builtin_fabs(v.x * w.x) + __builtin_fabs(v.y * w.y)) + __builtin_fabs(v.z * w.z);
    positive = s > 0.0;
    return (__builtin_fabs(s) > 0x1p-48 * m) && (m > 0x1p-400) && (m < 0x1p400);
}
PTD_INLINE bool scan_rect(const DRect& r, V3 o, V3 d, double& t) {
    const V3 n = ld3(r.n), p1 = ld3(r.p1);
    const double n_dot_u = dot(n, d);
    const double n_dot_p0 = dot(n, sub(p1, o));
    t = n_dot_p0 / n_dot_u;
    bool ok = !((n_dot_u > -0.00001) && (n_dot_u < 0.00001)) && !(t < 1.0);
    if (__builtin_amdgcn_ballot_w64(ok) != 0) {
        const V3 hit = add(o, muls(d, t));
        const V3 w1 = sub(hit, p1), w3 = sub(hit, ld3(r.p3));
        const V3 v1 = ld3(r.v1), v3 = ld3(r.v3);
        bool pos1, pos3;
        const bool dec1 = side_decided(v1, w1, pos1), dec3 = side_decided(v3, w3, pos3);
        if (__builtin_amdgcn_ballot_w64(ok && !(dec1 && dec3)) != 0) {
            const V3 v4 = normalize(w1);
            const V3 v5 = normalize(w3);
            if (!dec1) pos1 = dot(v1, v4) > 0.0;
            if (!dec3) pos3 = dot(v3, v5) > 0.0;
        }
        ok = ok && pos1 && pos3;
    }
    return ok;
}

// V3::cross light_tracing.cpp:146-151
PTD_INLINE V3 cross(V3 a, V3 b) { return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); }

// Rectangle::tri_intersect2 light_tracing.cpp:537-563 (Moller-Trumbore): false where the reference
// returns -1 from one of its tests.  inv_det = 1.0L / det is the x87 quotient narrowed to double.
// NaN operands fail none of the reference's comparisons and fall through to the value, as here.
PTD_INLINE bool tri_mt(V3 o, V3 d, V3 p0, V3 edge1, V3 edge2, double eps, double& t) {
    const V3 pvec = cross(d, edge2);
    const double det = dot(edge1, pvec);
    const double inv_det = ptx::x87_recip(det);
    const V3 tvec = sub(o, p0);
    const double u = dot(tvec, pvec) * inv_det;
    const V3 qvec = cross(tvec, edge1);
    const double v = dot(d, qvec) * inv_det;
    t = dot(edge2, qvec) * inv_det;
    return !((det > -eps) & (det < eps)) & !((u < 0.0) | (u > 1.0)) & !((v < 0.0) | ((u + v) > 1.0));
}
// Rectangle::intersect light_tracing.cpp:501-535: the triangle (p1, e1, e2), and only when that
// returns exactly -1.0 (a miss, or a genuine t of -1) the triangle (p1, e2, e3).  Returns false where
// the reference returns -1 from both.
PTD_INLINE bool scan_rect_mt(const DRectMT& r, V3 o, V3 d, double eps, double& t) {
    const V3 p1 = ld3(r.p1), e2 = ld3(r.e2);
    double t1;
    const bool ok1 = tri_mt(o, d, p1, ld3(r.e1), e2, eps, t1);
    const bool first = ok1 & (t1 != -1.0);
    t = t1;
    bool ok = first;
    if (__builtin_amdgcn_ballot_w64(!first) != 0) {
        double t2;
        const bool ok2 = tri_mt(o, d, p1, e2, ld3(r.e3), eps, t2);
        t = first ? t1 : t2;
        ok = first | ok2;
    }
    return ok;
}

// Exact Shape::intersect value, signed zeros included (diagnostics).
PTD_INLINE double intersect_body(const DScene& S, const DGeom& g, V3 o, V3 d) {
    if (g.shape == kRectMT) {
        double t;
        return scan_rect_mt(S.scan_rects_mt[g.pad[0]], o, d, S.mt_eps, t) ? t : -1.0;
    }
    if (g.shape == kRect) {
        double t;
        return scan_rect(S.scan_rects[g.pad[0]], o, d, t) ? t : -1.0;
    }
    return g.shape == kSphere ? intersect_sphere(g, o, d) : intersect_plane(g, o, d);
}

// The scan's view of Shape::intersect: `t` plus a flag that is false where the reference returns
// -1 (parallel plane, sphere without a positive discriminant) — a value that can never pass
// (t > 0).  Carrying the flag instead of the -1 saves the 64-bit selects.
PTD_INLINE bool scan_plane(const DGeom& g, V3 o, V3 d, double& t) {
    double n_dot_u = g.nx * d.x + g.ny * d.y + g.nz * d.z;
    double n_dot_p0 = g.nx * (g.cx - o.x) + g.ny * (g.cy - o.y) + g.nz * (g.cz - o.z);
    t = n_dot_p0 / n_dot_u;
    return !((n_dot_u > -0.00001) && (n_dot_u < 0.00001));
}
PTD_INLINE bool scan_plane_axis(double ck, double ok, double dk, double& t) {
    t = (ck - ok) / dk;
    return !((dk > -0.00001) && (dk < 0.00001));
}
// Single-precision screen for Sphere::intersect.  A sphere can only be hit when disc > 0 and b < 0;
// both are first evaluated in float (13 operations, half of them packed) together with a rigorous
// bound on the float error, and the double-precision evaluation is skipped for the whole wave when
// no lane can possibly satisfy them.  Lanes are never decided by float arithmetic: whenever any
// lane is uncertain the wave takes the exact FP64 path, so results are unchanged.
//
// Error bound (eps = 2^-24): with M = 2*(|o|_1 + max|c|_1) >= |D|_1 + |o|_1 + |c|_1 and n = |d|_1,
//   |b_f - b| <= 12 eps M n,   |disc_f - disc| <= 32 eps (M^2 (n^2 + 1) + r^2);
// e_b and e_disc carry a further factor 2..4 of safety.  NaN/inf compare false and fall through.
PTD_INLINE RayF make_rayf(V3 o, V3 d, float c1max, float r2max) {
    RayF r;
    r.ox = (float)o.x;
    r.oy = (float)o.y;
    r.oz = (float)o.z;
    r.dx = (float)d.x;
    r.dy = (float)d.y;
    r.dz = (float)d.z;
    float M = 2.0f * ((__builtin_fabsf(r.ox) + __builtin_fabsf(r.oy) + __builtin_fabsf(r.oz)) + c1max) * 1.0001f;
    float n = (__builtin_fabsf(r.dx) + __builtin_fabsf(r.dy) + __builtin_fabsf(r.dz)) * 1.0001f;
    r.e_disc = 0x1p-18f * (M * M * (n * n + 1.0f) + r2max);
    r.e_b = 0x1p-19f * M * n;
    return r;
}

// Packed FP32 operations with one operand broadcast from a single float: the packed instructions can
// read the low half of a register pair for both results (op_sel_hi), so the ray's components need
// no second copy in the high halves (the compiler's own form spends a v_mov per component and scan).
// `lo` carries the value in .x; .y is never read.
PTD_INLINE f32x2 lo_half(float v) {
    f32x2 t;
    t.x = v;
    asm("" : "+v"(t));  // .y stays undefined: no instruction is spent on it
    return t;
}
PTD_INLINE f32x2 pk_bcast_sub(f32x2 lo, f32x2 b) {  // lo.x - b
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(lo), "s"(b));
    return r;
}
PTD_INLINE f32x2 pk_mul_bcast(f32x2 a, f32x2 lo) {  // a * lo.x
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(lo));
    return r;
}
PTD_INLINE f32x2 pk_fma_bcast(f32x2 a, f32x2 lo, f32x2 c) {  // a * lo.x + c
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(lo), "v"(c));
    return r;
}
PTD_INLINE f32x2 pk_smul_bcast(f32x2 s_ab, f32x2 lo) {  // s_ab (a scalar register pair) * lo.x
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(s_ab), "v"(lo));
    return r;
}
PTD_INLINE f32x2 pk_sfma_bcast(f32x2 s_ab, f32x2 lo, f32x2 c) {  // s_ab * lo.x + c
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "s"(s_ab), "v"(lo), "v"(c));
    return r;
}
// The ray of the screen with its components in the low halves of register pairs.
struct RayF2 {
    f32x2 ox, oy, oz, dx, dy, dz;
};

// The screen for two bodies at once (packed FP32), as wave masks: m0 / m1 = lanes that might hit
// body 0 / 1 (NaN compares as "might").  `sc` must sit in scalar registers (load_record).
PTD_INLINE void sphere_possible2_masks(const RayF2& r, float e_disc, float e_b, const DScreen2& sc, lanemask& m0,
                                       lanemask& m1) {
    f32x2 Dx = pk_bcast_sub(r.ox, sc.cx), Dy = pk_bcast_sub(r.oy, sc.cy), Dz = pk_bcast_sub(r.oz, sc.cz);
    f32x2 bf = pk_fma_bcast(Dz, r.dz, pk_fma_bcast(Dy, r.dy, pk_mul_bcast(Dx, r.dx)));
    f32x2 cf = __builtin_elementwise_fma(Dz, Dz, __builtin_elementwise_fma(Dy, Dy, Dx * Dx)) - sc.r2;
    f32x2 discf = __builtin_elementwise_fma(bf, bf, -cf);
    m0 = PTD_MASK(!(discf.x < -e_disc)) & PTD_MASK(!(bf.x > e_b));
    m1 = PTD_MASK(!(discf.y < -e_disc)) & PTD_MASK(!(bf.y > e_b));
}

PTD_INLINE bool scan_sphere_exact(const DGeom& g, V3 o, V3 d, double& t) {
    t = 0.0;
    double dx = o.x - g.cx, dy = o.y - g.cy, dz = o.z - g.cz;
    double b = dx * d.x + dy * d.y + dz * d.z;
    double c = (dx * dx + dy * dy + dz * dz) - g.nx;
    double disc = (b * b) - c;
    // -b - sqrt(disc) can only be positive when b < 0; most (wave, sphere) pairs have no such lane
    // and skip the square root altogether.
    bool ok = (disc > 0.0) && (b < 0.0);
    if (__builtin_amdgcn_ballot_w64(ok) != 0) {
        double dd = disc;
        asm("" : "+v"(dd));  // keeps the square root inside this wave-uniform branch
        t = -b - sqrt(dd);
    }
    return ok;
}

// Nearest-hit scan main.cpp:372-383: (t > 0) && (t <= mint) over the bodies in index order, i.e.
// the smallest positive t with ties going to the larger index; `skip` is the light excluded by
// next_event.cpp:415-419 (-1: none).  That result does not depend on the order of evaluation, so
// the scan is organised by shape (lists built at upload, body index in pad[0]):
//  1. planes in ascending order with the reference's own `t <= mint` rule;
//  2. slabs: per axis, two axis-aligned planes c_lo < c_hi tested with one division (scan_slab),
//     merged with the explicit rule (t < mint) || (t == mint && index > hit);
//  3. spheres, two per step: the packed single-precision screen decides per wave whether either
//     needs its exact FP64 evaluation, merged with the same explicit rule.
// Records are read at wave-uniform addresses (scalar loads); the shape switch is a scalar branch.
PTD_INLINE void merge_hit(bool ok, double t, int i, int skip, double eps, int& hit, double& mint) {
    // bitwise on purpose: plain compares combined on the scalar unit, no divergent branches
    const bool take = ok & (t > eps) & ((t < mint) | ((t == mint) & (i > hit))) & (i != skip);
    mint = take ? t : mint;
    hit = take ? i : hit;
}
// Lane masks: a compare is taken as a 64-bit wave mask (the SGPR pair v_cmp writes), combined with
// scalar logic, and applied with v_cndmask.  Written out because the compiler otherwise
// materialises selected booleans as 0/1 in vector registers.
PTD_INLINE uint32_t sel32(lanemask m, uint32_t if_set, uint32_t if_clear) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
PTD_INLINE double sel64(lanemask m, double if_set, double if_clear) {
    uint32_t lo = sel32(m, (uint32_t)__double2loint(if_set), (uint32_t)__double2loint(if_clear));
    uint32_t hi = sel32(m, (uint32_t)__double2hiint(if_set), (uint32_t)__double2hiint(if_clear));
    return __hiloint2double((int)hi, (int)lo);
}
// mint = t and hit = i in the lanes of `take` (a subset of the active lanes), i being the same for
// the whole wave: the execution mask is narrowed to `take` around two moves — one 64-bit, one from
// the scalar register that holds i — where selects would take three v_cndmask and a copy of i into
// a vector register.  The mask is restored inside the same statement.
PTD_INLINE void assign_under(lanemask take, double t, int i, double& mint, int& hit) {
    lanemask saved;
    asm("s_and_saveexec_b64 %[sv], %[tk]\n\t"
        "v_mov_b64 %[m], %[t]\n\t"
        "v_mov_b32 %[h], %[i]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m] "+v"(mint), [h] "+v"(hit), [sv] "=&s"(saved)
        : [tk] "s"(take), [t] "v"(t), [i] "s"(i)
        : "scc");
}
// Lanes where body i (wave-uniform) replaces the hit so far: (t < mint) || (t == mint && i > hit).
// Equal distances are rare, so the index compare sits behind a wave-uniform branch.
PTD_INLINE lanemask wins_mask(double t, double mint, int i, int hit) {
    lanemask wins = PTD_MASK(t < mint);
    const lanemask eq = PTD_MASK(t == mint);
    if (eq != 0) wins |= eq & PTD_MASK(i > hit);
    return wins;
}
// merge_hit for a wave-uniform body index.
PTD_INLINE void merge_hit_uniform(bool ok, double t, int i, int skip, double eps, int& hit, double& mint) {
    const lanemask take = PTD_MASK(ok) & PTD_MASK(t > eps) & wins_mask(t, mint, i, hit) & PTD_MASK(i != skip);
    assign_under(take, t, i, mint, hit);
}
// (t > 0) && ((t < mint) || (t == mint && i > hit)) && (i != skip), within `valid`, for a per-lane index.
template <bool HAS_SKIP>
PTD_INLINE void merge_masked(lanemask valid, double t, int i, int skip, int& hit, double& mint) {
    lanemask wins = PTD_MASK(t < mint);
    const lanemask eq = PTD_MASK(t == mint);
    if (eq != 0) wins |= eq & PTD_MASK(i > hit);  // equal distances are rare: wave-uniform branch
    lanemask take = valid & PTD_MASK(t > 0.0) & wins;
    if (HAS_SKIP) take &= PTD_MASK(i != skip);
    mint = sel64(take, t, mint);
    hit = (int)sel32(take, (uint32_t)i, (uint32_t)hit);
}
// Two parallel axis-aligned planes, n_lo = c_lo - o_k < n_hi = c_hi - o_k, share the divisor d_k.
// The quotient that can win the scan is the smallest positive one: for d_k > 0 that of n_lo when
// n_lo > 0, else that of n_hi; for d_k < 0 that of n_hi when n_hi < 0, else that of n_lo.  Only
// that division is made.  The other quotient matters in two cases, both only when the two have
// the same sign (`both`): the chosen plane is the excluded light, or the two quotients could
// round to the same value (excluded by |o_k| < big, which keeps n_hi - n_lo above 2^-41 |n|).
// Then (wave-uniform, rare) the second quotient is formed and merged as well.
template <bool HAS_SKIP>
PTD_INLINE void scan_slab(double lo, double hi, double big, int ilo, int ihi, double ok_, double dk, int skip, int& hit,
                          double& mint) {
    const double n_lo = lo - ok_, n_hi = hi - ok_;
    const lanemask pos = PTD_MASK(dk > 0.0), a = PTD_MASK(n_lo > 0.0), b = PTD_MASK(n_hi < 0.0);
    const lanemask pick_lo = (pos & a) | ~(pos | b);
    const lanemask both = (pos & a) | (~pos & b);
    const lanemask valid = PTD_MASK(!(__builtin_fabs(dk) < 0.00001));  // == !((dk > -1e-5) && (dk < 1e-5))
    const double t = sel64(pick_lo, n_lo, n_hi) / dk;
    const int i = (int)sel32(pick_lo, (uint32_t)ilo, (uint32_t)ihi);
    merge_masked<HAS_SKIP>(valid, t, i, skip, hit, mint);
    lanemask second = both & ~PTD_MASK(__builtin_fabs(ok_) < big);
    if (HAS_SKIP) second |= both & PTD_MASK(i == skip);
    if (second != 0) {
        const double t2 = sel64(pick_lo, n_hi, n_lo) / dk;
        merge_masked<HAS_SKIP>(valid, t2, (int)sel32(pick_lo, (uint32_t)ihi, (uint32_t)ilo), skip, hit, mint);
    }
}
// Steps 1 and 2 of the scan: the planes outside the slabs, then the slabs.  ORIGIN0: the ray starts
// at exactly (0,0,0) — `o` must be that — and the origin-dependent terms come from the records.
template <bool ORIGIN0, bool HAS_SKIP = true>
PTD_INLINE void scan_all_planes(const DScene& S, const DGeom* __restrict__ planes, V3 o, V3 d, int skip, int& hit,
                                double& mint) {
    PT_PROF(kProfScanPlanes);
    if (ORIGIN0) o = mk(0.0, 0.0, 0.0);  // c - 0.0 folds to c
    const int n_planes = S.n_planes;
    const double eps = S.hit_eps;  // slabs exist only in scenes with hit_eps == 0
    for (int k = 0; k < n_planes; ++k) {
        const DGeom g = load_record(&planes[k]);
        double t;
        bool ok;
        switch (g.shape) {
            case kPlaneX: ok = scan_plane_axis(g.cx, o.x, d.x, t); break;
            case kPlaneY: ok = scan_plane_axis(g.cy, o.y, d.y, t); break;
            case kPlaneZ: ok = scan_plane_axis(g.cz, o.z, d.z, t); break;
            default:
                if (ORIGIN0) {
                    const double n_dot_u = g.nx * d.x + g.ny * d.y + g.nz * d.z;
                    t = origin0_term(g) / n_dot_u;
                    ok = !((n_dot_u > -0.00001) && (n_dot_u < 0.00001));
                } else {
                    ok = scan_plane(g, o, d, t);
                }
                break;
        }
        const int i = g.pad[0];
        lanemask take = PTD_MASK(ok) & PTD_MASK(t > eps) & PTD_MASK(t <= mint);
        if (HAS_SKIP) take &= PTD_MASK(i != skip);
        assign_under(take, t, i, mint, hit);
    }
    if (S.slab_mask & 1)
        scan_slab<HAS_SKIP>(S.slab_lo[0], S.slab_hi[0], S.slab_big[0], S.slab_ilo[0], S.slab_ihi[0], o.x, d.x, skip, hit, mint);
    if (S.slab_mask & 2)
        scan_slab<HAS_SKIP>(S.slab_lo[1], S.slab_hi[1], S.slab_big[1], S.slab_ilo[1], S.slab_ihi[1], o.y, d.y, skip, hit, mint);
    if (S.slab_mask & 4)
        scan_slab<HAS_SKIP>(S.slab_lo[2], S.slab_hi[2], S.slab_big[2], S.slab_ilo[2], S.slab_ihi[2], o.z, d.z, skip, hit, mint);
}
// Sphere::intersect from the origin (0,0,0): D = -c, c-term stored in the record.
PTD_INLINE bool scan_sphere_exact_origin0(const DGeom& g, V3 d, double& t) {
    t = 0.0;
    const double dx = 0.0 - g.cx, dy = 0.0 - g.cy, dz = 0.0 - g.cz;
    const double b = dx * d.x + dy * d.y + dz * d.z;
    const double disc = (b * b) - origin0_term(g);
    const bool ok = (disc > 0.0) && (b < 0.0);
    if (__builtin_amdgcn_ballot_w64(ok) != 0) {
        double dd = disc;
        asm("" : "+v"(dd));  // keeps the square root inside this wave-uniform branch
        t = -b - sqrt(dd);
    }
    return ok;
}
// The same screen in the compiler's own form (ray components as plain floats): the grouped scan of
// the large-scene pipeline uses it.
PTD_INLINE void sphere_possible2_masks(float ox, float oy, float oz, float dx, float dy, float dz, float e_disc,
                                       float e_b, const DScreen2& sc, lanemask& m0, lanemask& m1) {
    f32x2 Dx = ox - sc.cx, Dy = oy - sc.cy, Dz = oz - sc.cz;
    f32x2 bf = __builtin_elementwise_fma(Dz, (f32x2)(dz), __builtin_elementwise_fma(Dy, (f32x2)(dy), Dx * dx));
    f32x2 cf = __builtin_elementwise_fma(Dz, Dz, __builtin_elementwise_fma(Dy, Dy, Dx * Dx)) - sc.r2;
    f32x2 discf = __builtin_elementwise_fma(bf, bf, -cf);
    m0 = PTD_MASK(!(discf.x < -e_disc)) & PTD_MASK(!(bf.x > e_b));
    m1 = PTD_MASK(!(discf.y < -e_disc)) & PTD_MASK(!(bf.y > e_b));
}

// The sphere screen from the origin (0,0,0): the record holds the centre and, in place of r^2,
// the float c-term |c|^2 - r^2.  Same error bounds as sphere_possible2_masks.
PTD_INLINE void sphere_possible2_masks_origin0(const RayF2& r, float e_disc, float e_b, const DScreen2& sc, lanemask& m0,
                                               lanemask& m1) {
    f32x2 bf = -pk_sfma_bcast(sc.cz, r.dz, pk_sfma_bcast(sc.cy, r.dy, pk_smul_bcast(sc.cx, r.dx)));
    f32x2 discf = __builtin_elementwise_fma(bf, bf, -sc.r2);
    m0 = PTD_MASK(!(discf.x < -e_disc)) & PTD_MASK(!(bf.x > e_b));
    m1 = PTD_MASK(!(discf.y < -e_disc)) & PTD_MASK(!(bf.y > e_b));
}
// RECTS: the scene may hold rectangles (plain path tracing only; the next-event kernels are built
// without them).
template <bool ORIGIN0, bool RECTS = false, bool HAS_SKIP = true>
PTD_INLINE int nearest_hit_lists(const DScene& S, const DGeom* __restrict__ planes, const DGeom* __restrict__ spheres,
                                 const DScreen2* __restrict__ screen, V3 o, V3 d, int skip, double& tmin) {
    int hit = -1;
    double mint = 1.7976931348623157e308;  // DBL_MAX
#ifdef PT_DUP_SLAB
    {
        int h2 = -1;
        double m2 = 1.7976931348623157e308;
        V3 dd = d;
        asm volatile("" : "+v"(dd.x));
        scan_all_planes<ORIGIN0, HAS_SKIP>(S, planes, o, dd, skip, h2, m2);
        keep_alive(m2);
        keep_alive(h2);
    }
#endif
    scan_all_planes<ORIGIN0, HAS_SKIP>(S, planes, o, d, skip, hit, mint);
    if (RECTS) {
        const DRect* __restrict__ rects = S.scan_rects;
        const int n_rects = S.n_rects;
        for (int k = 0; k < n_rects; ++k) {
            double t;
            const bool ok = scan_rect(rects[k], o, d, t);
            merge_hit(ok, t, rects[k].body, skip, S.hit_eps, hit, mint);
        }
        const DRectMT* __restrict__ rects_mt = S.scan_rects_mt;
        const int n_rects_mt = S.n_rects_mt;
        for (int k = 0; k < n_rects_mt; ++k) {
            double t;
            const bool ok = scan_rect_mt(rects_mt[k], o, d, S.mt_eps, t);
            merge_hit(ok, t, rects_mt[k].body, skip, S.hit_eps, hit, mint);
        }
    }
    const int n_spheres = S.n_spheres;
#ifdef PT_DUP_SCREEN
    for (int rep_ = 0; rep_ < 2; ++rep_) {
    const int hit_save = hit;
    const double mint_save = mint;
    asm volatile("" : "+v"(d.x));
#endif
    if (n_spheres > 0) {
        PT_PROF(kProfScanScreen);
        const RayF rf = make_rayf(ORIGIN0 ? mk(0.0, 0.0, 0.0) : o, d, S.sphere_c1max, S.sphere_r2max);
        RayF2 r2;
        r2.ox = lo_half(rf.ox), r2.oy = lo_half(rf.oy), r2.oz = lo_half(rf.oz);
        r2.dx = lo_half(rf.dx), r2.dy = lo_half(rf.dy), r2.dz = lo_half(rf.dz);
        const float e_disc = rf.e_disc, e_b = rf.e_b;
        for (int k = 0; 2 * k < n_spheres; ++k) {
            const DScreen2 sc = load_record(&screen[k]);
            lanemask m0, m1;
            if (ORIGIN0)
                sphere_possible2_masks_origin0(r2, e_disc, e_b, sc, m0, m1);
            else
                sphere_possible2_masks(r2, e_disc, e_b, sc, m0, m1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((h ? m1 : m0) == 0) continue;
                PT_PROF(kProfScanExact);
                const DGeom g = load_record(&spheres[2 * k + h]);
                double t;
                const bool ok = ORIGIN0 ? scan_sphere_exact_origin0(g, d, t) : scan_sphere_exact(g, o, d, t);
                merge_hit_uniform(ok, t, g.pad[0], skip, S.hit_eps, hit, mint);
            }
        }
    }
#ifdef PT_DUP_SCREEN
    if (rep_ == 0) {
        keep_alive(mint);
        keep_alive(hit);
        hit = hit_save;
        mint = mint_save;
    }
    }
#endif
    tmin = mint;
    return hit;
}
template <bool RECTS = false>
PTD_INLINE int nearest_hit(const DScene& S, V3 o, V3 d, int skip, double& tmin) {
    return nearest_hit_lists<false, RECTS>(S, S.scan_planes, S.scan_spheres, S.scan_screen, o, d, skip, tmin);
}
// The scan of a gather ray: no body is left out (main.cpp:372-383).
template <bool RECTS = false>
PTD_INLINE int nearest_hit_any(const DScene& S, V3 o, V3 d, double& tmin) {
    return nearest_hit_lists<false, RECTS, false>(S, S.scan_planes, S.scan_spheres, S.scan_screen, o, d, -1, tmin);
}
// The scan of a shadow ray towards `d` with `light` left out (next_event.cpp:405-426).  Unless the
// scene asks for the lights' real centres every shadow ray starts at (0,0,0).
PTD_INLINE int nearest_hit_shadow(const DScene& S, V3 lc, V3 d, int light, double& tmin) {
    if (S.flags & 1u) return nearest_hit_lists<false>(S, S.scan_planes, S.scan_spheres, S.scan_screen, lc, d, light, tmin);
    return nearest_hit_lists<true>(S, S.scan_planes, S.scan_spheres, S.scan_screen0, lc, d, light, tmin);
}

// ---- materials -------------------------------------------------------------------------
// getRandomNormalInHemisphere main.cpp:105-117.  Draw order z, y, x (g++ evaluates the
// constructor arguments right to left); the normalised copy is discarded by the reference.
// The rejection test dot(v,v) > 1.0 is first screened in float: a candidate whose float length
// is outside 1 +- 1e-5 is decided without forming the doubles; only candidates inside that shell
// take the exact FP64 test.  The accepted triple is then converted once.  Decisions and results are
// identical to the plain loop.
// The float coordinate of a draw comes straight from the generator's 32-bit sum s (the draw is
// k = s >> 1): the loop carries s + 2^31 (mod 2^32) — the constant rides in the generator's own
// addition (v_add3_u32) — which read as a signed integer is s - 2^31, and one v_cvt_f32_i32 makes it
// the coordinate scaled by 2^31.  Against the exact coordinate 2 k / (2^31 - 1) - 1 the scaled value
// (s - 2^31) / 2^31 is off by less than 2^-29 and its rounding to float by at most 2^-25: 3.2e-8, so the
// squared length differs by less than 3 * 2 * 3.2e-8 = 2e-7, plus 5.4e-7 of float rounding in the three
// squares and two sums (fewer with fma): 7.4e-7, well inside the shell's 1e-5.
PTD_INLINE uint32_t rng_next_sum_flipped(Rng& r) {  // rng_next_sum + 2^31 (mod 2^32)
    const uint32_t a = r.s0, b = r.s3;
    rng_next_sum(r);
    uint32_t t;  // written out: the compiler would form a + b and flip the top bit with a second instruction
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "s"(0x80000000u));
    return t;
}
PTD_INLINE V3 random_in_hemisphere(Draws& dr, V3 n) {
    uint32_t tz, ty, tx;  // flipped sums
    for (;;) {
        PT_PROF_TRIP(kProfRejectTrips);
        tz = rng_next_sum_flipped(dr.rng);
        ty = rng_next_sum_flipped(dr.rng);
        tx = rng_next_sum_flipped(dr.rng);
        dr.count += 3;
        const float fz = (float)(int32_t)tz, fy = (float)(int32_t)ty, fx = (float)(int32_t)tx;  // coordinates * 2^31
        const float s2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));                // length^2 * 2^62
        if (s2 > 1.00001f * 0x1p62f) continue;
        if (s2 < 0.99999f * 0x1p62f) break;
        const int32_t kz = (int32_t)((tz ^ 0x80000000u) >> 1), ky = (int32_t)((ty ^ 0x80000000u) >> 1),
                      kx = (int32_t)((tx ^ 0x80000000u) >> 1);
        V3 e = mk(k_to_coord(kx), k_to_coord(ky), k_to_coord(kz));
        if (!(dot(e, e) > 1.0)) break;
    }
    const uint32_t sz = tz ^ 0x80000000u, sy = ty ^ 0x80000000u, sx = tx ^ 0x80000000u;
    const int32_t kz = (int32_t)(sz >> 1), ky = (int32_t)(sy >> 1), kx = (int32_t)(sx >> 1);
    V3 v = mk(k_to_coord(kx), k_to_coord(ky), k_to_coord(kz));
    if (dot(v, n) < 0.0) v = muls(v, -1.0);
    return v;
}

// Chrome::bounce main.cpp:261-265
PTD_INLINE V3 bounce_chrome(V3 d, V3 n) {
    double theta1 = fabs(dot(d, n));
    return add(d, muls(n, theta1 * 2.0));
}

// Glass::bounce main.cpp:279-305.  theta1 = fabs(..) >= 0 always, so internal = ior,
// external = 1.  theta2 is an x87 extended-precision expression (ptx::x87_glass_theta2).
PTD_INLINE V3 bounce_glass(Draws& dr, V3 d, V3 n, double ior, double eta, double refl) {
    double theta1 = fabs(dot(d, n));
    // eta = externalIndex / internalIndex = 1.0 / ior, divided once on the host.  (A NaN theta1 would
    // take the reference's other branch; every output is NaN either way.)
    const double internalIndex = ior, externalIndex = 1.0;
    double theta2 = ptx::x87_glass_theta2(eta, theta1);
    double rs = (externalIndex * theta1 - internalIndex * theta2) / (externalIndex * theta1 + internalIndex * theta2);
    double rp = (internalIndex * theta1 - externalIndex * theta2) / (internalIndex * theta1 + externalIndex * theta2);
    double reflectance = (rs * rs + rp * rp);
    if (unif(dr) < reflectance + refl) return add(d, muls(n, theta1 * 2.0));
    return add(muls(add(d, muls(n, theta1)), eta), muls(n, -theta2));
}

PTD_INLINE V3 bounce(const DMat& m, Draws& dr, V3 d, V3 n) {
    if (m.material == kChrome) {
        PT_PROF(kProfChrome);
        return bounce_chrome(d, n);
    }
    if (m.material == kGlass) {
        PT_PROF(kProfGlass);
        return bounce_glass(dr, d, n, m.ior, m.eta, m.refl);
    }
    PT_PROF(kProfDiffuse);
#ifdef PT_DUP_REJECT
    {
        Draws cp = dr;
        asm volatile("" : "+v"(cp.rng.s0));
        V3 w = random_in_hemisphere(cp, n);
        keep_alive(w.x + w.y + w.z);
    }
#endif
    return random_in_hemisphere(dr, n);  // Material::bounce main.cpp:250-253
}

// getNormal: Plane main.cpp:192-194, Sphere main.cpp:224-226
PTD_INLINE V3 shape_normal(const DGeom& g, V3 point) {
    if (g.shape != kSphere) return mk(g.nx, g.ny, g.nz);
    PT_PROF(kProfNormalSphere);
    return normalize(sub(point, mk(g.cx, g.cy, g.cz)));
}

// ---- measured BRDF: MatusikBRDF.h ------------------------------------------------------
PTD_INLINE void brdf_normalize(double* v) {  // :124-130
    double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] = v[0] / len;
    v[1] = v[1] / len;
    v[2] = v[2] / len;
}

PTD_INLINE void brdf_rotate(const double* vector, const double* axis, double angle, double* out) {  // :133-156
    double cos_ang = cos(angle);
    double sin_ang = sin(angle);
    out[0] = vector[0] * cos_ang;
    out[1] = vector[1] * cos_ang;
    out[2] = vector[2] * cos_ang;
    double temp = axis[0] * vector[0] + axis[1] * vector[1] + axis[2] * vector[2];
    temp = temp * (1.0 - cos_ang);
    out[0] += axis[0] * temp;
    out[1] += axis[1] * temp;
    out[2] += axis[2] * temp;
    double cx = axis[1] * vector[2] - axis[2] * vector[1];
    double cy = axis[2] * vector[0] - axis[0] * vector[2];
    double cz = axis[0] * vector[1] - axis[1] * vector[0];
    out[0] += cx * sin_ang;
    out[1] += cy * sin_ang;
    out[2] += cz * sin_ang;
}

// double -> int as the x86 reference does it: NaN and out-of-range give INT_MIN.
PTD_INLINE int to_int_x86(double v) {
    if (!(v == v) || v >= 2147483648.0 || v <= -2147483649.0) return (int)0x80000000;
    return (int)v;
}

constexpr double kPi = 0x1.921fb54442d18p+1;       // M_PI
constexpr double kHalfPi = 0x1.921fb54442d18p+0;   // M_PI / 2.0 == M_PI * 0.5

PTD_INLINE int theta_half_index(double theta_half) {  // :211-223
    if (theta_half <= 0.0) return 0;
    double theta_half_deg = ((theta_half / kHalfPi) * 90);
    double temp = theta_half_deg * 90;
    temp = sqrt(temp);
    int ret_val = to_int_x86(temp);
    if (ret_val < 0) ret_val = 0;
    if (ret_val >= 90) ret_val = 89;
    return ret_val;
}
PTD_INLINE int theta_diff_index(double theta_diff) {  // :228-237
    int tmp = to_int_x86(theta_diff / kHalfPi * 90);
    if (tmp < 0) return 0;
    if (tmp < 89) return tmp;
    return 89;
}
PTD_INLINE int phi_diff_index(double phi_diff) {  // :241-257
    if (phi_diff < 0.0) phi_diff += kPi;
    int tmp = to_int_x86(phi_diff / kPi * 360 / 2);
    if (tmp < 0) return 0;
    if (tmp < 179) return tmp;
    return 179;
}

// std_coords_to_half_diff_coords :159-205 and the index of lookup_brdf_val :63-66
PTD_INLINE int brdf_index(double theta_in, double fi_in, double theta_out, double fi_out) {
    double in_vec_z = cos(theta_in);
    double proj_in_vec = sin(theta_in);
    double in_vec_x = proj_in_vec * cos(fi_in);
    double in_vec_y = proj_in_vec * sin(fi_in);
    double in[3] = {in_vec_x, in_vec_y, in_vec_z};
    brdf_normalize(in);
    double out_vec_z = cos(theta_out);
    double proj_out_vec = sin(theta_out);
    double out_vec_x = proj_out_vec * cos(fi_out);
    double out_vec_y = proj_out_vec * sin(fi_out);
    // the reference also normalises `out`, but only in_vec_*/out_vec_* and `in` are used afterwards
    double half[3] = {(in_vec_x + out_vec_x) / 2.0, (in_vec_y + out_vec_y) / 2.0, (in_vec_z + out_vec_z) / 2.0};
    brdf_normalize(half);
    double theta_half = acos(half[2]);
    double fi_half = atan2(half[1], half[0]);
    const double bi_normal[3] = {0.0, 1.0, 0.0};
    const double normal[3] = {0.0, 0.0, 1.0};
    double temp[3], diff[3];
    brdf_rotate(in, normal, -fi_half, temp);
    brdf_rotate(temp, bi_normal, -theta_half, diff);
    double theta_diff = acos(diff[2]);
    double fi_diff = atan2(diff[1], diff[0]);
    return phi_diff_index(fi_diff) + theta_diff_index(theta_diff) * 360 / 2 + theta_half_index(theta_half) * 360 / 2 * 90;
}

// The values of lookup_brdf_val :68-79 for a table index.  The device copy of a table keeps the
// red, green and blue value of an index next to each other (24 bytes).
// The channel scales and the "below the horizon" rule of lookup_brdf_val (:73-78) depend on the
// table entry alone and are applied once, at upload (brdf_entry): the device table holds final values.
PTD_HOST_DEVICE inline void brdf_entry(double red, double green, double blue, double* out3) {
    double r = red * (1.0 / 1500.0);
    double g = green * (1.15 / 1500.0);
    double b = blue * (1.66 / 1500.0);
    if (r < 0.0 || g < 0.0 || b < 0.0) r = g = b = 0.0;
    out3[0] = r, out3[1] = g, out3[2] = b;
}
PTD_INLINE V3 brdf_rgb(const double* __restrict__ table, int ind) {
    const double* e = table + 3 * (size_t)ind;
    return mk(e[0], e[1], e[2]);
}
// lookup_brdf_val :50-80 from the four angles.
PTD_INLINE V3 brdf_lookup(const double* __restrict__ table, double theta_in, double fi_in, double theta_out,
                          double fi_out, int* ind_out) {
    int ind = brdf_index(theta_in, fi_in, theta_out, fi_out);
    if (ind_out) *ind_out = ind;
    return brdf_rgb(table, ind);
}

// vector_to_spherical raytrace3.cpp:725-728 (pow(x, 2) is x*x)
PTD_INLINE void vector_to_spherical(V3 in, double& theta, double& phi) {
    theta = atan(in.y / in.x);
    phi = acos(in.z / sqrt(in.x * in.x + in.y * in.y + in.z * in.z));
}

// Bounding-sphere screen of two groups (records hold the centre and, in r2, the RADIUS R, rounded
// outwards): a member can only be hit when the ray starts inside the bound (c < 0) or meets it ahead
// (disc >= 0 and b < 0); evaluated in float with the error bounds of make_rayf (e_disc also bounds the
// error of c and of |D|^2).  R = +inf always passes, -inf never.
// Distance cut: a sphere (c, r) that the reference's formula hits from outside, D = o - c, A = |D|,
// has t = C / (beta + sqrt(beta^2 - C)) with C = A^2 - r^2 and beta = -D.d <= s A, s = |d|; t falls
// as beta grows, and at beta = s A the inequality t >= (A - r) / s reduces to
// A^2 (1 - s^2) >= r^2 (1 - s^2): true for s <= 1.  A sphere inside the bound has
// A - r >= delta = |o - c_B| - R, so t >= delta / s, and a lane whose nearest hit so far is mint cannot
// improve on it (nor tie with it) inside a bound with delta > s mint.  That is for the formula in
// exact arithmetic; evaluated in FP64 its t is lower by at most sqrt(2e-15 (A^2 + r^2)) < 7e-8 (A + r)
// (the square root of the rounding error of the discriminant, reached at grazing incidence), which is
// what the margins are for: the stored R is 2.5e-4 too large and tau is passed as 1.0004 s mint (+inf
// disables the cut), so a bound is cut only when delta - s mint > 2e-4 (R + s mint), far above the
// error wherever the cut can apply (A + r < 2 (R + s mint) + delta).
PTD_INLINE void group_possible2_masks(float ox, float oy, float oz, float dx, float dy, float dz, float e_disc,
                                      float e_b, float tau, const DScreen2& sc, lanemask& m0, lanemask& m1) {
    f32x2 Dx = ox - sc.cx, Dy = oy - sc.cy, Dz = oz - sc.cz;
    f32x2 bf = __builtin_elementwise_fma(Dz, (f32x2)(dz), __builtin_elementwise_fma(Dy, (f32x2)(dy), Dx * dx));
    f32x2 D2 = __builtin_elementwise_fma(Dz, Dz, __builtin_elementwise_fma(Dy, Dy, Dx * Dx));
    f32x2 cf = D2 - sc.r2 * sc.r2;
    f32x2 discf = __builtin_elementwise_fma(bf, bf, -cf);
    f32x2 reach = sc.r2 + tau;
    f32x2 far2 = reach * reach;
    m0 = (PTD_MASK(!(cf.x > e_disc)) | (PTD_MASK(!(discf.x < -e_disc)) & PTD_MASK(!(bf.x > e_b)))) & PTD_MASK(!(D2.x - e_disc > far2.x));
    m1 = (PTD_MASK(!(cf.y > e_disc)) | (PTD_MASK(!(discf.y < -e_disc)) & PTD_MASK(!(bf.y > e_b)))) & PTD_MASK(!(D2.y - e_disc > far2.y));
}

// ---- the estimator ---------------------------------------------------------------------
// Gather-path records live in LDS: point component c of vertex k of thread tid is at
// pts[(k*3 + c) * stride + tid], its body at bod[k * stride + tid].
struct Records {
    double* pts;
    int8_t* bod;  // the fused kernels handle at most 127 bodies
    int stride;
    int tid;
    // row * stride + t with a 24-bit multiply (full rate; the 32-bit one is quarter rate): rows < 16, stride <= 1024
    PTD_INLINE int idx(int row, int t) const { return (int)__umul24((unsigned)row, (unsigned)stride) + t; }
    PTD_INLINE void put_body(int k, int body) const { bod[idx(k, tid)] = (int8_t)body; }
    PTD_INLINE void put(int k, V3 p, int body) const {
        pts[idx(k * 3 + 0, tid)] = p.x;
        pts[idx(k * 3 + 1, tid)] = p.y;
        pts[idx(k * 3 + 2, tid)] = p.z;
        bod[idx(k, tid)] = (int8_t)body;
    }
    PTD_INLINE V3 point(int k) const {
        return mk(pts[idx(k * 3 + 0, tid)], pts[idx(k * 3 + 1, tid)], pts[idx(k * 3 + 2, tid)]);
    }
    PTD_INLINE int body(int k) const { return bod[idx(k, tid)]; }
    // records of another thread of the block
    PTD_INLINE V3 point_of(int k, int t) const {
        return mk(pts[idx(k * 3 + 0, t)], pts[idx(k * 3 + 1, t)], pts[idx(k * 3 + 2, t)]);
    }
    PTD_INLINE int body_of(int k, int t) const { return bod[idx(k, t)]; }
};

struct PathResult {
    V3 rgb;
    int nrec;
    int visible;
};

// One (pixel, iteration) sample.
//  PT : Renderer::trace main.cpp:367-405 — the recursion `trace(next)*color + emission` is a Horner
//       evaluation from the deepest hit outwards.
//  NEE: pixel body of Renderer::iterate next_event.cpp:384-470 with trace() next_event.cpp:491-530.
// ---- stages shared by the fused kernel and the wavefront pipeline -------------------------

// One trace() level after the scan (main.cpp:391-403 / next_event.cpp:515-528): hit point, normal,
// bounce, then the hit point moved by the factor 1 +- 1e-7 along the ray.  Updates the ray in
// place and returns the recorded (offset) point.
PTD_INLINE V3 shade_hit(const DGeom& g, const DMat& m, Draws& dr, V3& o, V3& d, double mint) {
    PT_PROF(kProfShade);
    V3 point = add(o, muls(d, mint));
    V3 normal = shape_normal(g, point);
    V3 dir = bounce(m, dr, d, normal);
    // main.cpp:394-401 (the comparison is against the float 0.0f)
    bool fwd = dot(dir, d) > 0.0;
    double tt = ptx::x87_mul_const(mint, fwd ? 0x1.000001ad7f29bp+0 : 0x1.fffffca501acbp-1,
                                   fwd ? -0x1.0d8p-54 : -0x1.e58p-55);
    point = add(o, muls(d, tt));
    o = point;
    d = dir;
    return point;
}

// lights[unifRand(lights.size() - 1)] next_event.cpp:390; unifRand(long) main.cpp:36-47.
// Returns the body index of the chosen light (lights[0] is never chosen when there are several).
PTD_INLINE int pick_light(const DScene& S, Draws& dr) {
    int li = 0;
    int nl = S.n_lights - 1;
    if (nl < 0) nl = -nl;
    if (nl != 0) {
        long guard = (long)(unif(dr) * (double)nl) + 1;
        li = guard > nl ? nl : (int)guard;
    }
    return S.lights[li];
}

// Origin of the shadow rays towards `light`: light->shape->center through a Shape* is the
// never-assigned base member (0,0,0) (next_event.cpp:167,406-407) unless the scene asks for the
// real centre.
PTD_INLINE V3 light_origin(const DScene& S, int light) {
    V3 lc = mk(0.0, 0.0, 0.0);
    if (S.flags & 1u) {
        DGeom lg = S.geom[light];
        lc = mk(lg.cx, lg.cy, lg.cz);
    }
    return lc;
}
// emission * colour of the light (next_event.cpp:452).  When it is exactly zero (every light of
// the reference's own scenes has colour 0, next_event.cpp:660,674), (Le*Lc)*flux is a zero of
// known sign whatever the finite flux is, and nothing after the NEE loop reads the random stream:
// the fake-BRDF bounce that produces flux is dead and may be skipped.  Results are bit-identical.
// The term is re-read per lit vertex instead of being kept in registers across the shadow scans.
PTD_INLINE V3 light_term(const DScene& S, int light) {
    DMat lm = S.mat[light];
    return mul(mk(lm.emr, lm.emg, lm.emb), mk(lm.colr, lm.colg, lm.colb));
}

// ---- the table index straight from the two direction vectors -------------------------------
// measured_term's angles come from vectors (vector_to_spherical) and go straight back into sines
// and cosines (std_coords_to_half_diff_coords), and the half/difference angles are only ever
// binned.  This form computes the same quantities without a single trigonometric call:
//   theta = atan(y/x):  cos theta = |x|/h, sin theta = sign(x) y/h,  h = sqrt(x^2 + y^2);
//   phi = acos(c), c = z/len:  cos phi = c, sin phi = sqrt((1-c)(1+c));
//   fi_half = atan2(hy, hx):  cos = hx/r, sin = hy/r, r = sqrt(hx^2 + hy^2);  sin(theta_half) = r;
// and bins theta_half = acos(hz), theta_diff = acos(dz), fi_diff = atan2(dy, dx) by comparing hz, dz
// and dx/dy with the cosines / cotangents of the bin edges (tables built at upload, descending;
// uniform binary search).  Each replaced call is accurate to an ulp or two, like the library
// calls of the reference's libm and of the device's, which already differ from each other in the
// last place; an index changes only when an angle sits within a few ulps of a bin edge, or on the
// pole of the parametrisation where fi_diff is rounding noise in every implementation.
// NaN inputs give index 0 in every term, as the reference's double -> int conversions do.
constexpr int kBinsHalf = 0, kBinsDiff = 128, kBinsPhi = 256, kBinsTotal = 512;

// Number of edges passed: the largest pos with v <= T[pos] (T[0] counts as passed, T descending,
// padded with values no v reaches).  NaN passes nothing.
// The position is kept as a 32-bit byte offset whose bits the steps set one by one, so that a step is
// a load at (uniform base + offset + constant), a compare, a select and an or.
template <int SIZE>
PTD_INLINE uint32_t bins_passed8(const double* __restrict__ T, double v) {  // 8 * the number of edges passed
    const char* base = reinterpret_cast<const char*>(T);
    uint32_t off = 0;
#pragma unroll
    for (uint32_t step = SIZE / 2 * 8; step >= 8; step >>= 1) {
        const double edge = *reinterpret_cast<const double*>(base + (off + step));
        off |= v <= edge ? step : 0u;
    }
    return off;
}

// 1/sqrt(x) and 1/x to about two ulps: the hardware estimate (relative error 2^-26) and one Newton
// step (1.5 e^2 = 3.3e-16, plus the step's own rounding); a second step changed no table index of
// the tests and costs four instructions per call, seven calls per lookup.
// 0 gives NaN (0 * inf) where the quotient forms would give 0/0 = NaN or x/0 = inf; callers treat
// those poles separately.
#ifndef PT_NR_STEPS
#define PT_NR_STEPS 1
#endif
PTD_INLINE double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int k = 0; k < PT_NR_STEPS; ++k) {
        const double g = x * y, hh = 0.5 * y;
        const double r = __builtin_fma(-g, hh, 0.5);
        y = __builtin_fma(y, r, y);
    }
    return y;
}
PTD_INLINE double rcp_nr(double d) {
    double y = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int k = 0; k < PT_NR_STEPS; ++k) {
        const double e = __builtin_fma(-d, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

// The unit vector std_coords_to_half_diff_coords :161-167 builds from the angles vector_to_spherical
// (raytrace3.cpp:725-728) gives for v: theta = atan(y/x), phi = acos(z/len), and
//   (sin theta cos phi, sin theta sin phi, cos theta) = (s y z / (h len), s y / len, |x| / h),
// h = sqrt(x^2 + y^2), s = sign(x)  (sin phi = h/len cancels one h).  The sums and products of this
// function and the next are fused where the operation count allows: the lookup is exact to the table
// bin, not to the bit, and a fused operation is the more accurate one.
PTD_INLINE void spherical_unit(V3 v, double& a0, double& a1, double& a2) {
    const double h2 = __builtin_fma(v.y, v.y, v.x * v.x);
    const double ih = rsqrt_nr(h2), il = rsqrt_nr(__builtin_fma(v.z, v.z, h2));
    const double c = v.z * il;  // the reference's correctly rounded z / len never leaves [-1, 1]
    const double st = __builtin_copysign(1.0, v.x) * (v.y * ih);
    a0 = st * c;
    a1 = (st * h2) * (ih * il);  // NaN with st when h = 0, as atan(0/0) makes every term
    a2 = fabs(v.x) * ih;
}

PTD_INLINE int brdf_index_of_vectors(const double* __restrict__ bins, V3 vin, V3 vout) {
    double in[3], out[3];
    spherical_unit(vin, in[0], in[1], in[2]);
    spherical_unit(vout, out[0], out[1], out[2]);
    // std_coords_to_half_diff_coords :159-205.  `in` has unit length up to rounding; the reference's
    // normalisation of it (:169) changes the last place only and is left out.
    // half = normalize((in + out) / 2): the factor 1/2 cancels, and the rotation angles are taken
    // from the sum before it is normalised
    double hx = in[0] + out[0];
    const double hy = in[1] + out[1], hz = in[2] + out[2];
    const double r2 = __builtin_fma(hy, hy, hx * hx);
    const double ihl = rsqrt_nr(__builtin_fma(hz, hz, r2));
    // rotation by -fi_half about (0,0,1); on the pole (r2 == 0, hx and hy zeros) atan2 gives 0 or pi:
    // cos = the sign of hx, sin = 0.  There the rotation is taken from (sign of hx, hy) and length 1.
    const bool pole = r2 == 0.0;
    if (pole) hx = __builtin_copysign(1.0, hx);
    const double ir = rsqrt_nr(pole ? 1.0 : r2);
    const double cos1 = hx * ir, sin1 = -hy * ir;
    const double r = (r2 * ir) * ihl;  // sin(theta_half)
    double half[3];
    half[2] = hz * ihl;
    // rotate_vector :133-156 about (0,0,1) by (cos1, sin1), then about (0,1,0) by (half.z, -r);
    // the terms that multiply the zero components of the axes are dropped
    double temp[3];
    // (not fused: where the two products cancel, as for a direction in a coordinate plane, the
    // result must be the exact zero that the separate roundings of the reference's form give)
    temp[0] = in[0] * cos1 - in[1] * sin1;
    temp[1] = in[1] * cos1 + in[0] * sin1;
    temp[2] = in[2];
    const double cos2 = half[2], sin2 = -r;
    double diff[3];
    diff[0] = temp[0] * cos2 + temp[2] * sin2;
    diff[1] = temp[1];
    diff[2] = temp[2] * cos2 - temp[0] * sin2;
    // theta_half_index(acos(half.z)) :211-223
    const uint32_t i_half8 = bins_passed8<128>(bins + kBinsHalf, half[2]);
    // theta_diff_index(acos(diff.z)) :228-237; acos is NaN outside [-1, 1] and NaN indexes bin 0
    uint32_t i_diff8 = bins_passed8<128>(bins + kBinsDiff, diff[2]);
    if (diff[2] < -1.0) i_diff8 = 0;
    // phi_diff_index(atan2(diff.y, diff.x)) :241-257: the angle is taken modulo pi, so it depends on
    // cot = x/y alone; y == 0 gives 0 or pi, and pi stays pi (bin 179) only for y = +0 with x
    // negative (or -0), -pi being moved to 0.
    uint32_t i_phi8;
    if (diff[1] == 0.0)
        i_phi8 = (!__builtin_signbit(diff[1]) && __builtin_signbit(diff[0]) && !(diff[0] != diff[0])) ? 179 * 8 : 0;
    else
        i_phi8 = bins_passed8<256>(bins + kBinsPhi, diff[0] * rcp_nr(diff[1]));
    return (int)((i_phi8 + i_diff8 * (360 / 2) + i_half8 * (360 / 2 * 90)) >> 3);
}

// The measured-material term of a lit vertex (the incoming and outgoing directions in the roles
// vector_to_spherical, raytrace3.cpp:725-728, and lookup_brdf_val, MatusikBRDF.h:50-80, give them).
PTD_INLINE V3 measured_term(const double* __restrict__ table, const double* __restrict__ bins, V3 bd, V3 nprime,
                            int* ind_out = nullptr) {
    const int ind = brdf_index_of_vectors(bins, bd, nprime);
    if (ind_out) *ind_out = ind;
    return brdf_rgb(table, ind);
}

// One lit vertex of the NEE loop (next_event.cpp:442-463): L = ((L + (Le*Lc)*flux) * c) + e.
// `bd` is the shadow-ray direction, `next_point` the previous vertex (or the camera-plane point
// for vertex 0, next_event.cpp:443-447).
template <bool MEASURED, bool ELIDE>
PTD_INLINE V3 nee_vertex(const DScene& S, const DGeom& g, const DMat& m, Draws& dr, int light, V3 P, V3 bd,
                         V3 next_point, V3 L, int* brdf_ind_out = nullptr) {
    V3 scale;
    const V3 lterm = light_term(S, light);
    const bool direct_is_zero = (lterm.x == 0.0) & (lterm.y == 0.0) & (lterm.z == 0.0);
    // The term (Le*Lc)*flux is a zero of known sign when Le*Lc is exactly zero and flux is finite:
    // true for diffuse and chrome, and for glass with ior >= 1, in a scene of finite, moderate
    // coordinates — unless a chrome or glass vertex coincides with the previous one (a ray that
    // re-hits the surface it left at t ~ 1e-21): then the reference reflects about a normalised
    // zero vector, flux is NaN and so is the pixel.  That case takes the full path below.  (The
    // diffuse bounce only compares against that vector; its flux stays finite.)
    const bool twin = (m.material != kDiffuse) & same_point(next_point, P);
    if (ELIDE && direct_is_zero && S.flux_finite && !twin && m.material != kMeasured &&
        (m.material != kGlass || m.ior >= 1.0)) {
        scale = muls(lterm, 0.0);
        return add(mul(add(L, scale), mk(m.colr, m.colg, m.colb)), mk(m.emr, m.emg, m.emb));
    }
    V3 nprime = normalize(sub(next_point, P));
    if (MEASURED && m.material == kMeasured) {
        // Defined by this build: the table RGB replaces the scalar flux (the reference's own
        // lookups, raytrace3.cpp:646-676, are never executed).
        V3 f = measured_term(S.brdf[m.brdf_id], S.brdf_bins, bd, nprime, brdf_ind_out);
        scale = mul(lterm, f);
    } else {
        V3 light_brdf = bounce(m, dr, bd, nprime);
        double light_flux = fabs(dot(light_brdf, shape_normal(g, P)));
        scale = muls(lterm, light_flux);
    }
    return add(mul(add(L, scale), mk(m.colr, m.colg, m.colb)), mk(m.emr, m.emg, m.emb));
}

// One (pixel, iteration) sample, all stages fused.
//  PT : Renderer::trace main.cpp:367-405 — the recursion `trace(next)*color + emission` is a Horner
//       evaluation from the deepest hit outwards.
//  NEE: pixel body of Renderer::iterate next_event.cpp:384-470 with trace() next_event.cpp:491-530.
template <int EST, bool MEASURED, bool ELIDE>
PTD_INLINE PathResult trace_path(const DScene& S, Draws& dr, double x, double y, const Records& rec,
                                 int* brdf_ind = nullptr) {
    const DGeom* __restrict__ geom = S.geom;
    const DMat* __restrict__ mat = S.mat;
    V3 o = ld3(S.cam_o);
    V3 campt = cam_point(S, x, y);
    V3 d = normalize(sub(campt, o));  // Camera::getRay main.cpp:154-157
    int n = 0;
#pragma unroll 1
    for (; n < kMaxDepth; ++n) {
        double mint;
        int hit = nearest_hit_any<EST == kEstPT>(S, o, d, mint);
        if (hit < 0) break;
        V3 point = shade_hit(geom[hit], mat[hit], dr, o, d, mint);
        PT_PROF(kProfRecords);
        if (EST == kEstPT && ELIDE)
            rec.put_body(n, hit);  // plain path tracing reads only the bodies back (the render kernel keeps no points)
        else
            rec.put(n, point, hit);
    }
    PathResult R;
    R.nrec = n;
    R.visible = 0;
    V3 L = mk(0.0, 0.0, 0.0);
    if (EST == kEstPT) {
#pragma unroll 1
        for (int k = n - 1; k >= 0; --k) {
            DMat m = mat[rec.body(k)];
            // raytrace3.cpp:805,840: color * (trace(next) + direct_color) + emission with
            // direct_color = V3(): the added zero turns a -0 into +0
            if (S.flags & 2u) L = add(L, mk(0.0, 0.0, 0.0));
            L = add(mul(L, mk(m.colr, m.colg, m.colb)), mk(m.emr, m.emg, m.emb));
        }
        R.rgb = L;
        return R;
    }
    const int light = pick_light(S, dr);
    DScene T = S;  // the second phase: see refresh_scan_controls
    refresh_scan_controls(T);
#pragma unroll 1
    for (int gi = n - 1; gi >= 0; --gi) {
        V3 P, bd, lc;
        int target;
        {
            PT_PROF(kProfNeePrep);
            P = rec.point(gi);
            target = rec.body(gi);
            lc = light_origin(T, light);
            bd = normalize(sub(P, lc));
        }
        double mint;
        int hit = nearest_hit_shadow(T, lc, bd, light, mint);
        if (hit != target) continue;  // occluded: the vertex is skipped entirely, next_event.cpp:428-430
        R.visible |= 1 << gi;
        PT_PROF(kProfNeeVertex);
        V3 next_point = campt;
        if (gi >= 1) next_point = rec.point(gi - 1);
        L = nee_vertex<MEASURED, ELIDE>(S, geom[hit], mat[hit], dr, light, P, bd, next_point, L,
                                        brdf_ind ? brdf_ind + gi : nullptr);
    }
    R.rgb = L;
    return R;
}

// The double `v` of lane `src` (0..63) of the wave, through ds_bpermute: no LDS is allocated.
PTD_INLINE double lane_read(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// NEE sample for scenes whose direct term is dead for every light and every material
// (DScene::nee_emission_only: emission*colour of each light is exactly zero, no measured material,
// every glass ior >= 1, finite moderate geometry and colours).  The pixel value is then the recurrence
// L = L*c + e over the lit vertices, deepest first, starting from 0: vertices deeper than the
// deepest emitting one leave L at zero whether they are lit or not, so their shadow scans are not
// needed.  (One exception is kept: a lit vertex that coincides with the one before it makes the
// reference's pixel NaN, see nee_vertex; such "twin" vertices count like emitting ones.)  The remaining (lane, vertex) shadow tasks of a wave are dealt to its 64 lanes, 64 at a
// time: an owner publishes (lane, vertex, light) in a 16-bit LDS slot at the task's rank, lane q
// takes slot q, reads the vertex from the owner's LDS records, runs the shadow scan, and the
// owners pick their visibility bits out of the ballot.  Every lane of the wave must call this
// (`valid` false for lanes without a pixel).  Sample values equal trace_path<kEstNEE,false,true>'s
// except that a zero result may differ in sign, which no sum of samples can observe.
PTD_INLINE V3 trace_path_nee_compact(const DScene& S, Draws& dr, bool valid, double x, double y, const Records& rec,
                                     uint16_t* __restrict__ wave_slots) {
    int n = 0, need = 0, light = 0;
    int first_abs = -1;  // the shallowest vertex on an absorbing emitter
    V3 p_last = mk(0.0, 0.0, 0.0);  // position of vertex kMaxDepth - 1
    unsigned twin = 0;  // bit k: chrome or glass vertex k coincides with the vertex before it (the camera-plane point for k = 0)
    if (valid) {
        V3 o = ld3(S.cam_o);
        V3 prev = cam_point(S, x, y);
        V3 d = normalize(sub(prev, o));  // Camera::getRay main.cpp:154-157
#pragma unroll 1
        for (; n < kMaxDepth; ++n) {
            double mint;
            int hit = nearest_hit_any(S, o, d, mint);
            if (hit < 0) break;
            const DMat& m = S.mat[hit];
            V3 point = shade_hit(S.geom[hit], m, dr, o, d, mint);  // o = point on return
            if (n < kMaxDepth - 1) {
                rec.put(n, point, hit);
            } else {  // the records hold kMaxDepth - 1 positions: the deepest vertex keeps its own
                rec.put_body(n, hit);
                p_last = point;
            }
            const bool tw = (m.material != kDiffuse) & same_point(point, prev);
            twin |= (tw ? 1u : 0u) << n;
            if (m.emits | tw) need = n + 1;  // a lit twin vertex turns the pixel into NaN: its visibility matters
            first_abs = ((m.absorber != 0) & (first_abs < 0)) ? n : first_abs;
            prev = point;
        }
        light = pick_light(S, dr);
    }
    const int lane = rec.tid & 63;
    const int wave_tid0 = rec.tid & ~63;
    DScene T = S;  // the second phase: see refresh_scan_controls
    refresh_scan_controls(T);
    // An absorbing emitter (colour exactly 0) that is lit sets L to its emission whatever the deeper
    // vertices gave — (L + 0) * 0 + e with L finite — so the deeper shadow scans are needed only when that
    // vertex turns out to be in shadow (0.15 % of them in the reference's room), or when a deeper twin
    // vertex could make L NaN.  First pass: the vertices up to the shallowest absorbing one; a second
    // pass, for the lanes whose absorber was not lit, runs all of theirs (rarely entered, wave-uniform).
    const int abs_end = first_abs + 1;
    const bool prune = (first_abs >= 0) & (abs_end < need) & ((twin >> abs_end) == 0u);
    int stop = prune ? abs_end : need;
    unsigned vis = 0;
    lanemask owners[kMaxDepth];
    int first[kMaxDepth + 1];
    first[0] = 0;
#pragma unroll
    for (int g = 0; g < kMaxDepth; ++g) {
        owners[g] = PTD_MASK(stop > g);
        first[g + 1] = first[g] + __builtin_popcountll(owners[g]);
    }
    int total = first[kMaxDepth];
    bool second = false;
#pragma unroll 1
    for (int base = 0;; base += 64) {
        if (base >= total) {
            // end of a pass: the second one only for lanes whose absorber was not lit
            const bool redo = prune & !((vis >> (first_abs & 7)) & 1u);
            if (second || PTD_MASK(redo) == 0) break;
            second = true;
            stop = redo ? need : 0;
#pragma unroll
            for (int g = 0; g < kMaxDepth; ++g) {
                owners[g] = PTD_MASK(stop > g);
                first[g + 1] = first[g] + __builtin_popcountll(owners[g]);
            }
            total = first[kMaxDepth];
            base = 0;
        }
        PT_PROF(kProfNeeVertex);
#pragma unroll
        for (int g = 0; g < kMaxDepth; ++g) {
            if (first[g + 1] > base && first[g] < base + 64) {
                const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(owners[g] >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)owners[g], 0u));
                const int q = first[g] + below - base;
                if ((stop > g) & ((unsigned)q < 64u)) wave_slots[q] = (uint16_t)(lane | (g << 6) | (light << 9));
            }
        }
        // The slots are written and read by lanes of one wave, whose LDS instructions execute in
        // program order; the fence pair (wavefront scope, LDS address space only) makes that order part
        // of the program: the stores above may not sink below it nor the loads below rise above it.
        // (The compiler no longer takes plain global loads behind a fence for scalar loads; the
        // shadow scans below keep theirs because the scan reads its records with load_record.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        bool lit = false;
        // every lane reads a slot (those past the count a stale one) and fetches the owner's deepest
        // position through the lane crossbar, which needs owner and taker both active
        const unsigned slot = wave_slots[lane];
        const V3 p_deep = mk(lane_read(p_last.x, (int)(slot & 63u)), lane_read(p_last.y, (int)(slot & 63u)),
                             lane_read(p_last.z, (int)(slot & 63u)));
        if (base + lane < total) {
            const int g = (int)((slot >> 6) & 7u), lt = (int)(slot >> 9);
            const int owner = wave_tid0 | (int)(slot & 63u);
            V3 P, bd, lc;
            {
                PT_PROF(kProfNeePrep);
                P = rec.point_of(g < kMaxDepth - 1 ? g : 0, owner);
                if (g == kMaxDepth - 1) P = p_deep;
                lc = light_origin(T, lt);
                bd = normalize(sub(P, lc));  // next_event.cpp:405-407
            }
            double mint;
            int hit = nearest_hit_shadow(T, lc, bd, lt, mint);
            lit = hit == rec.body_of(g, owner);  // next_event.cpp:428-430
        }
        const lanemask lit_mask = PTD_MASK(lit);
#pragma unroll
        for (int g = 0; g < kMaxDepth; ++g) {
            if (first[g + 1] > base && first[g] < base + 64) {
                const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(owners[g] >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)owners[g], 0u));
                const int q = first[g] + below - base;
                if ((stop > g) & ((unsigned)q < 64u)) vis |= (unsigned)((lit_mask >> q) & 1ull) << g;
            }
        }
    }
    V3 L = mk(0.0, 0.0, 0.0);
    if (need > 0) {
        const V3 scale = muls(light_term(S, light), 0.0);
#pragma unroll 1
        for (int g = need - 1; g >= 0; --g) {
            if (!((vis >> g) & 1u)) continue;
            const DMat& m = S.mat[rec.body(g)];
            // a twin vertex: the reference normalises a zero vector here, flux and L become NaN
            const V3 sc = ((twin >> g) & 1u) ? mk(flux_nan(), flux_nan(), flux_nan()) : scale;
            L = add(mul(add(L, sc), mk(m.colr, m.colg, m.colb)), mk(m.emr, m.emg, m.emb));  // next_event.cpp:452-462
        }
    }
    return L;
}

// Pixel body of bidirectional.cpp:392-494 for scenes whose direct term is dead everywhere
// (DScene::nee_emission_only, which pt_scene_create demands of this estimator): the gather path
// (:394, trace() :515-554), the light pick (:398), a light path started at the chosen light in a random
// direction of the hemisphere about its normal (:402-406), and for every (gather vertex, light-path
// vertex) pair a scan from the light-path vertex towards the gather vertex with the light left out
// (:415-437).  A visible pair applies L = (L + (Le*Lc)*flux) * c + e of the gather vertex's body
// (:486-487).  With Le*Lc exactly zero and flux finite the added term is a zero, nothing after the
// light path reads the random stream, and the two bounce() calls that make flux (:463-474) are dead:
// the pixel depends only on how many light-path vertices see each gather vertex.  Gather vertices
// deeper than the deepest emitting one leave L at zero and are not connected at all.  As in
// trace_path_nee_compact, a chrome or glass gather vertex that coincides with the vertex before it
// ("twin") makes the reference's flux NaN (it reflects about a normalised zero vector): such vertices
// count like emitting ones and turn L into NaN when they are seen.  (flux also carries
// flux_lookup[shoot_hits.size()][shoot_i], :484, which the reference reads past the end of the array
// when the light path has five vertices; any finite value there gives these results.)
// The connections are made while the light path is traced, vertex by vertex, so its positions are
// never stored: the order of the pairs does not matter, only the count per gather vertex.
// A lane has connections to make only while its light path lives and only to its gather vertices up to
// the deepest emitting one — none at all for most paths — so after every light-path step the
// (lane, gather vertex) connections of the wave are dealt to its 64 lanes as in trace_path_nee_compact:
// an owner publishes (lane, vertex, light) in a 16-bit LDS slot at the connection's rank, lane q takes
// slot q, fetches the owner's light-path vertex through the lane crossbar and the gather vertex from
// the owner's LDS records, runs the scan, and the owners count their hits out of the ballot.  Every
// lane of the wave must call this (`valid` false for lanes without a pixel).
PTD_INLINE V3 trace_path_bidir(const DScene& S, Draws& dr, bool valid, double x, double y, const Records& rec,
                               uint16_t* __restrict__ wave_slots) {
    int n = 0, need = 0, light = 0;
    unsigned twin = 0;
    V3 o = mk(0.0, 0.0, 0.0), d = mk(0.0, 0.0, 0.0);
    if (valid) {
        o = ld3(S.cam_o);
        V3 prev = cam_point(S, x, y);
        d = normalize(sub(prev, o));  // Camera::getRay main.cpp:154-157
#pragma unroll 1
        for (; n < kMaxDepth; ++n) {
            double mint;
            int hit = nearest_hit_any(S, o, d, mint);
            if (hit < 0) break;
            const DMat& m = S.mat[hit];
            V3 point = shade_hit(S.geom[hit], m, dr, o, d, mint);
            rec.put(n, point, hit);
            const bool tw = (m.material != kDiffuse) & same_point(point, prev);
            twin |= (tw ? 1u : 0u) << n;
            if (m.emits | tw) need = n + 1;
            prev = point;
        }
        light = pick_light(S, dr);
        // light->shape->center through a Shape* is (0,0,0) unless the scene asks for the real centre;
        // getNormal is virtual and uses the shape's own centre (bidirectional.cpp:402-404)
        o = light_origin(S, light);
        d = random_in_hemisphere(dr, shape_normal(S.geom[light], o));
    }
    const int lane = rec.tid & 63;
    const int wave_tid0 = rec.tid & ~63;
    // a path without an emitting vertex is black whatever its light path does, and nothing reads the
    // random stream after the light path: such lanes trace none
    bool shooting = valid & (need > 0);
    unsigned seen = 0;  // three bits per gather vertex: the number of light-path vertices that see it
    DScene T = S;  // the light-path scans: see refresh_scan_controls
    refresh_scan_controls(T);
#pragma unroll 1
    for (int ns = 0; ns < kMaxDepth; ++ns) {
        // the connection scans are run by other lanes than the light-path scans, and by more lanes in a
        // later step than in the last round of this one: their controls are derived anew in every step,
        // here, where the whole wave is present
        DScene C = S;
        refresh_scan_controls(C);
        V3 sp = mk(0.0, 0.0, 0.0);
        if (shooting) {
            double mint;
            int hit = nearest_hit_any(T, o, d, mint);
            shooting = hit >= 0;
            if (shooting) sp = shade_hit(S.geom[hit], S.mat[hit], dr, o, d, mint);
        }
        if (PTD_MASK(shooting) == 0) break;
        const int stop = shooting ? need : 0;
        lanemask owners[kMaxDepth];
        int first[kMaxDepth + 1];
        first[0] = 0;
#pragma unroll
        for (int g = 0; g < kMaxDepth; ++g) {
            owners[g] = PTD_MASK(stop > g);
            first[g + 1] = first[g] + __builtin_popcountll(owners[g]);
        }
        const int total = first[kMaxDepth];
#pragma unroll 1
        for (int base = 0; base < total; base += 64) {
#pragma unroll
            for (int g = 0; g < kMaxDepth; ++g) {
                if (first[g + 1] > base && first[g] < base + 64) {
                    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(owners[g] >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)owners[g], 0u));
                    const int q = first[g] + below - base;
                    if ((stop > g) & ((unsigned)q < 64u)) wave_slots[q] = (uint16_t)(lane | (g << 6) | (light << 9));
                }
            }
            // slots written and read by lanes of one wave, in program order: see trace_path_nee_compact
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            bool lit = false;
            // every lane reads a slot (those past the count a stale one) and fetches the owner's light-path
            // vertex through the lane crossbar, which needs owner and taker both active
            const unsigned slot = wave_slots[lane];
            const V3 sp_o = mk(lane_read(sp.x, (int)(slot & 63u)), lane_read(sp.y, (int)(slot & 63u)), lane_read(sp.z, (int)(slot & 63u)));
            if (base + lane < total) {
                const int g = (int)((slot >> 6) & 7u), lt = (int)(slot >> 9);
                const int owner = wave_tid0 | (int)(slot & 63u);
                const V3 bd = normalize(sub(rec.point_of(g, owner), sp_o));  // bi_ray :415-417
                double t;
                const int h = nearest_hit<false>(C, sp_o, bd, lt, t);
                lit = h == rec.body_of(g, owner);  // :435-437
            }
            const lanemask lit_mask = PTD_MASK(lit);
#pragma unroll
            for (int g = 0; g < kMaxDepth; ++g) {
                if (first[g + 1] > base && first[g] < base + 64) {
                    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(owners[g] >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)owners[g], 0u));
                    const int q = first[g] + below - base;
                    if ((stop > g) & ((unsigned)q < 64u)) seen += (unsigned)((lit_mask >> q) & 1ull) << (3 * g);
                }
            }
            // the slots are rewritten by the next round: its stores must stay behind this round's loads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        }
    }
    V3 L = mk(0.0, 0.0, 0.0);
    const V3 scale = muls(light_term(S, light), 0.0);
#pragma unroll 1
    for (int g = need - 1; g >= 0; --g) {
        const DMat& m = S.mat[rec.body(g)];
        const V3 sc = ((twin >> g) & 1u) ? mk(flux_nan(), flux_nan(), flux_nan()) : scale;
#pragma unroll 1
        for (unsigned k = (seen >> (3 * g)) & 7u; k > 0; --k)
            L = add(mul(add(L, sc), mk(m.colr, m.colg, m.colb)), mk(m.emr, m.emg, m.emb));  // :486-487
    }
    return L;
}

}  // namespace ptd
