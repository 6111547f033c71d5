// Bit-exact emulation, in IEEE double arithmetic, of the places where the reference computes in
// x87 extended precision (64-bit significand) and then narrows to double.
//
// The reference is x86-64 g++ code; every expression containing an `L` literal is evaluated as
// `long double` and rounded twice: first to 64 significant bits by the x87 operation, then to 53
// bits when the result is stored in a double parameter/variable.  That "double rounding"
// differs from a single IEEE rounding about once in 4096 operations, by one ulp.  One ulp is
// enough to flip a nearest-hit decision on degenerate paths (a ray re-hitting the plane it
// just left at t ~ 1e-21), so the estimator reproduces the x87 results exactly:
//   x87_recip        (double)(1.0L / s)                      V3::divs            main.cpp:92-94
//   x87_2u_minus_1   (double)(u * 2.0L - 1.0L)               hemisphere sampling main.cpp:108
//   x87_mul_1p0000001 / x87_mul_0p9999999
//                    (double)(t * 1.0000001L), (double)(t * 0.9999999L)          main.cpp:396,400
//   x87_glass_theta2 (double)sqrtl(1.0L - (eta*eta) * (1.0L - (theta1*theta1)))   main.cpp:292
//
// Method: carry the exact (or 2^-100-accurate) value as an unevaluated sum hi + lo with
// hi = RN53(value), round the tail lo to the 64-bit grid of hi (rn64_tail), and let one final
// IEEE addition hi + lo' perform the second rounding, ties-to-even included.
//
// Usable from host and device code; must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace ptx {

#define PTX_INLINE __host__ __device__ __forceinline__

PTX_INLINE uint64_t bits_of(double v) {
    uint64_t b;
    memcpy(&b, &v, sizeof b);
    return b;
}
PTX_INLINE double from_bits(uint64_t b) {
    double v;
    memcpy(&v, &b, sizeof v);
    return v;
}
PTX_INLINE double fma_exact(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Spacing of doubles in the binade of |v| (v normal and >= 2^-969 in magnitude).
PTX_INLINE double ulp_of(double v) {
    uint64_t e = bits_of(v) & 0x7FF0000000000000ULL;
    return from_bits(e - (52ULL << 52));
}

// v = hi + lo with hi = RN53(v), |lo| <= ulp(hi)/2.  Returns lo rounded (ties to even) to the
// spacing of 64-bit-significand numbers around v, so that hi + result == RN64(v) exactly.
PTX_INLINE double rn64_tail(double hi, double lo) {
    double g = ulp_of(hi) * 0x1p-11;
    // Just below a power of two the grid is twice as fine.
    bool pow2 = (bits_of(hi) & 0x000FFFFFFFFFFFFFULL) == 0;
    bool toward_zero = (lo < 0.0) == (hi > 0.0);
    if (pow2 && toward_zero && lo != 0.0) g *= 0.5;
    double M = g * 0x1.8p+52;  // ulp(M) == g, and |lo| < 2^51 g
    double r = (lo + M) - M;
    return r;
}

// hi + lo, narrowed the x87 way: RN53(RN64(hi + lo)).
PTX_INLINE double narrow_x87(double hi, double lo) {
    // Double rounding can only matter when the tail is within 2^-11 of half an ulp.
    if (!(__builtin_fabs(lo) >= 0x1.ff8p-1 * (ulp_of(hi) * 0.5))) return hi;
    return hi + rn64_tail(hi, lo);
}

// (double)(1.0L / (long double)s)
PTX_INLINE double x87_recip(double s) {
    double q = 1.0 / s;
    double r = fma_exact(-q, s, 1.0);  // exact residual 1 - q*s
    double lo = r * q;                 // (1 - q*s)/s up to a relative 2^-52
    uint64_t eb = bits_of(q) & 0x7FF0000000000000ULL;
    if (eb == 0 || eb == 0x7FF0000000000000ULL || eb < (70ULL << 52)) return q;  // 0, inf, nan, tiny
    return narrow_x87(q, lo);
}

// (double)((long double)u * 2.0L - 1.0L) for u = k / 2147483647.0, 0 <= k <= 2^31 - 1.
// For u >= 2^-13 the x87 difference is exact, so one IEEE rounding gives the same double.
PTX_INLINE double x87_2u_minus_1(double u, int32_t k) {
    double v = u * 2.0 - 1.0;
    if (k >= (1 << 18) || k == 0) return v;
    double lo = u * 2.0 - (v + 1.0);  // exact: 2u - 1 == v + lo
    return v + rn64_tail(v, lo);
}

// (double)((long double)t * C) for a 64-bit-significand constant C == chi + clo exactly.
PTX_INLINE double x87_mul_const(double t, double chi, double clo) {
    double p = t * chi;
    double e1 = fma_exact(t, chi, -p);  // exact error of the leading product
    double e = e1 + t * clo;
    double hi = p + e;
    double lo = (p - hi) + e;
    uint64_t eb = bits_of(hi) & 0x7FF0000000000000ULL;
    if (eb == 0x7FF0000000000000ULL || eb < (70ULL << 52)) return hi;
    return narrow_x87(hi, lo);
}
PTX_INLINE double x87_mul_1p0000001(double t) { return x87_mul_const(t, 0x1.000001ad7f29bp+0, -0x1.0d8p-54); }
PTX_INLINE double x87_mul_0p9999999(double t) { return x87_mul_const(t, 0x1.fffffca501acbp-1, -0x1.e58p-55); }

struct dd {
    double hi, lo;
};
PTX_INLINE dd two_sum(double a, double b) {
    double s = a + b;
    double bb = s - a;
    dd r;
    r.hi = s;
    r.lo = (a - (s - bb)) + (b - bb);
    return r;
}
PTX_INLINE dd fast_two_sum(double a, double b) {  // |a| >= |b| or a == 0
    double s = a + b;
    dd r;
    r.hi = s;
    r.lo = b - (s - a);
    return r;
}
PTX_INLINE dd rn64(dd v) {  // v.hi = RN53(v.hi + v.lo) required
    if (v.hi != 0.0 && v.lo != 0.0) v.lo = rn64_tail(v.hi, v.lo);
    return v;
}

// Glass::bounce's theta2 (main.cpp:292):
//   a = theta1*theta1 (double), b = 1.0L - a, c = eta*eta (double), d = c*b, e = 1.0L - d,
//   f = sqrtl(e), theta2 = (double)f — b, d, e, f each rounded to 64 bits.
PTX_INLINE double x87_glass_theta2(double eta, double theta1) {
    double a = theta1 * theta1;
    dd b = rn64(two_sum(1.0, -a));
    double c = eta * eta;
    double p = c * b.hi;
    double pe = fma_exact(c, b.hi, -p);
    dd d = rn64(two_sum(p, pe + c * b.lo));
    dd e = two_sum(1.0, -d.hi);
    e = rn64(two_sum(e.hi, e.lo - d.lo));
    if (!(e.hi > 0.0)) return __builtin_sqrt(e.hi);  // 0, negative (NaN) or NaN
    double hf = __builtin_sqrt(e.hi);
    double r = fma_exact(-hf, hf, e.hi);  // exact e.hi - hf*hf
    double lf = (r + e.lo) / (2.0 * hf);
    dd f = rn64(two_sum(hf, lf));
    return f.hi + f.lo;
}

}  // namespace ptx
