/* TEST INFRASTRUCTURE ONLY — parity oracle, never linked into or called by the product.
 *
 * Plain-C, single-purpose CPU restatement of the reference's per-pixel radiance estimator:
 * /root/reference/main.cpp (plain path tracing), /root/reference/next_event.cpp (path
 * tracing + next-event estimation) and /root/reference/bidirectional.cpp (gather path x light
 * path connections), plus /root/reference/MatusikBRDF.h (measured BRDF lookup).
 * Every function cites the reference lines it follows.  It is pinned against the golden vectors
 * under tests/golden/, which the reference's own code produced: oracle/_ref, the reference's
 * programs compiled here without Boost and without a stand-in for it (the lines of thread glue are
 * left out, nothing is put in their place; oracle/Makefile, tests/test_reference_build.py), driven by
 * the generator scripts tests/golden/make_golden*.py.  That includes the Moller-Trumbore Rectangle of
 * light_tracing.cpp, whose class compiles from the head of that file although the program does not
 * (kdtree.h is absent).  Not pinned, for want of anything to pin them to: values of real MERL tables
 * (the .binary files are absent from the reference checkout), and the row range / row split of
 * light_tracing.cpp (inside its photon-map Renderer and main(), which need the kd-tree): "parity
 * unpinned" for those two, which closed-form known answers cover instead.
 *
 * Floating point: the reference is x86-64 g++ code — IEEE double with SSE2, no FMA
 * contraction, and x87 80-bit arithmetic wherever an `L` literal makes an expression
 * `long double`.  This file mirrors the expression types one for one, and must be built
 * with gcc on x86-64 with -ffp-contract=off (oracle/Makefile does).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ptor_rng.h"
#include "ptor_rt.h"
#include "ptor_types.h"

#define POR_MAX_DEPTH 5 /* trace() returns at n > 4: main.cpp:368 */

/* Scene flags */
#define POR_F_LIGHT_TRUE_CENTER 1u /* shadow rays start at the light's real centre instead of the
                                      base-class Shape::center == (0,0,0) that
                                      next_event.cpp:167,406-407 actually reads */

#define POR_F_RAYTRACE3 2u /* raytrace3.cpp's variant of main.cpp's trace(): a hit needs
                              t > 0.0000001L (raytrace3.cpp:776) and the radiance is
                              color*(trace(next) + V3()) + emission (raytrace3.cpp:805,840) */

#define POR_F_PIXEL_GRID 4u /* samples on the pixel grid, as Renderer::ray_iterate takes them
                               (light_tracing.cpp:785-790): no jitter draws, every row full */

typedef struct por_scene {
    int32_t width, height;
    ptor_camera camera;
    const ptor_body* bodies;
    int32_t body_count;
    int32_t estimator; /* PTOR_EST_PT (main.cpp), PTOR_EST_NEE (next_event.cpp) or PTOR_EST_BIDIR (bidirectional.cpp) */
    uint32_t flags;
    uint32_t pad;
    uint64_t seed;
    const double* brdf_tables[8]; /* measured-BRDF tables by brdf_id, 3*1458000 doubles each */
} por_scene;

typedef struct v3 {
    double x, y, z;
} v3;

/* ---- random draws ------------------------------------------------------------------- */

typedef struct draws {
    int mode;
    ptor_rng rng;
    const int* script;
    int script_len;
    size_t pos;
    long long count;
} draws;

static inline int draw_k(draws* d) {
    d->count++;
    if (d->mode == PTOR_MODE_SCRIPT) {
        if (d->script_len <= 0) return 0;
        int k = d->script[d->pos];
        d->pos = (d->pos + 1) % (size_t)d->script_len;
        return k;
    }
    return ptor_rng_next(&d->rng);
}

/* unifRand(): main.cpp:19-22 — rand() / double(RAND_MAX), RAND_MAX = 2147483647 (glibc). */
static inline double unif(draws* d) { return draw_k(d) / (double)2147483647; }

/* unifRand(long n): main.cpp:36-47. */
static long unif_long(draws* d, long n) {
    if (n < 0) n = -n;
    if (n == 0) return 0;
    long guard = (long)(unif(d) * n) + 1;
    return (guard > n) ? n : guard;
}

/* ---- V3: main.cpp:50-103 ------------------------------------------------------------ */

static inline v3 V(double x, double y, double z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
/* divs: muls(1.0L / s) — the quotient is formed in long double, then narrowed to the double
 * parameter of muls (main.cpp:92-94). */
static inline v3 divs(v3 a, double s) { return muls(a, (double)(1.0L / s)); }
static inline double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 normalize(v3 a) { return divs(a, sqrt(dot(a, a))); }

/* getRandomNormalInHemisphere: main.cpp:105-117. g++ evaluates the three constructor
 * arguments right to left, so the first draw of a trip lands in z, the third in x.  The
 * result of v2.normalize() is discarded by the reference, so the direction is NOT unit. */
static v3 random_in_hemisphere(draws* d, v3 n) {
    v3 v2;
    do {
        double z = (double)(unif(d) * 2.0L - 1.0L);
        double y = (double)(unif(d) * 2.0L - 1.0L);
        double x = (double)(unif(d) * 2.0L - 1.0L);
        v2 = V(x, y, z);
    } while (dot(v2, v2) > 1.0L);
    if (dot(v2, n) < 0.0L) return muls(v2, (double)-1.0L);
    return v2;
}

/* ---- camera: main.cpp:144-157, next_event.cpp:159-161 ------------------------------- */

typedef struct cam {
    v3 origin, topleft, xd, yd;
} cam;

static v3 ld3(const double* p) { return V(p[0], p[1], p[2]); }

static cam make_cam(const ptor_camera* c) {
    cam r;
    r.origin = ld3(c->origin);
    r.topleft = ld3(c->topleft);
    r.xd = sub(ld3(c->topright), r.topleft);
    r.yd = sub(ld3(c->bottomleft), r.topleft);
    return r;
}

static v3 cam_point(const cam* c, double x, double y) { return add(add(c->topleft, muls(c->xd, x)), muls(c->yd, y)); }

static v3 cam_ray_dir(const cam* c, double x, double y) { return normalize(sub(cam_point(c, x, y), c->origin)); }

/* ---- shapes ------------------------------------------------------------------------- */

/* V3::cross raytrace3.cpp:128-133 */
static inline v3 cross(v3 a, v3 b) {
    return V((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}

/* Rectangle ctor raytrace3.cpp:254-265: normal = normalize((p3-p1) x (p2-p1)). */
static v3 rect_normal(const double* c) {
    v3 p1 = ld3(c), p2 = ld3(c + 3), p3 = ld3(c + 6);
    return normalize(cross(sub(p3, p1), sub(p2, p1)));
}

/* Rectangle::intersect raytrace3.cpp:283-308 (v1, v3 as the ctor stores them, :264-265). */
static double intersect_rectangle(const ptor_body* b, v3 o, v3 dir) {
    v3 p1 = ld3(b->corners), p2 = ld3(b->corners + 3), p3 = ld3(b->corners + 6), p4 = ld3(b->corners + 9);
    v3 normal = ld3(b->normal);
    v3 v1 = normalize(sub(p2, p1));
    v3 v3_ = normalize(sub(p4, p3));
    double n_dot_u = dot(normal, dir);
    if ((n_dot_u > -0.00001L) && (n_dot_u < 0.00001L)) return (double)-1.0L;
    double n_dot_p0 = dot(normal, sub(p1, o));
    double t = n_dot_p0 / n_dot_u;
    if (t < 1.0L) return (double)-1.0L;
    v3 hit = add(o, muls(dir, t));
    v3 v4 = normalize(sub(hit, p1));
    v3 v5 = normalize(sub(hit, p3));
    if ((dot(v1, v4) > 0.0L) && (dot(v3_, v5) > 0.0L)) return t;
    return (double)-1.0L;
}

/* Rectangle::tri_intersect2 light_tracing.cpp:537-563 (Moller-Trumbore), EPSILON = 0.000001L (:20).
 * PARITY UNPINNED: every program that holds this class needs kdtree.h, which is absent, so no build
 * of the reference can be run against this restatement; it follows the cited lines one for one. */
static double tri_intersect2(v3 o, v3 dir, v3 p0, v3 edge1, v3 edge2) {
    v3 pvec = cross(dir, edge2);
    double det = dot(edge1, pvec);
    if (det > -0.000001L && det < 0.000001L) return (double)-1.0L;
    double inv_det = (double)(1.0L / det);
    v3 tvec = sub(o, p0);
    double u = dot(tvec, pvec) * inv_det;
    if ((u < 0.0L) || (u > 1.0L)) return (double)-1.0L;
    v3 qvec = cross(tvec, edge1);
    double v = dot(dir, qvec) * inv_det;
    if ((v < 0.0L) || ((u + v) > 1.0L)) return (double)-1.0L;
    return dot(edge2, qvec) * inv_det;
}

/* Rectangle::intersect light_tracing.cpp:501-535 with the edges of the ctor (:490-492). */
static double intersect_rectangle_mt(const ptor_body* b, v3 o, v3 dir) {
    v3 p1 = ld3(b->corners), p2 = ld3(b->corners + 3), p3 = ld3(b->corners + 6), p4 = ld3(b->corners + 9);
    v3 e1 = sub(p2, p1), e2 = sub(p3, p1), e3 = sub(p4, p1);
    double first_t = tri_intersect2(o, dir, p1, e1, e2);
    if (first_t != -1.0L) return first_t;
    first_t = tri_intersect2(o, dir, p1, e2, e3);
    if (first_t != -1.0L) return first_t;
    return (double)-1.0L;
}

/* Plane::intersect main.cpp:181-191; Sphere::intersect main.cpp:210-222. */
static double intersect(const ptor_body* b, v3 o, v3 dir) {
    if (b->shape == PTOR_RECTANGLE) return intersect_rectangle(b, o, dir);
    if (b->shape == PTOR_RECTANGLE_MT) return intersect_rectangle_mt(b, o, dir);
    if (b->shape == PTOR_PLANE) {
        v3 n = ld3(b->normal);
        double n_dot_u = dot(n, dir);
        if ((n_dot_u > -0.00001L) && (n_dot_u < 0.00001L)) return (double)-1.0L;
        double n_dot_p0 = dot(n, sub(ld3(b->center), o));
        return n_dot_p0 / n_dot_u;
    } else {
        v3 distance = sub(o, ld3(b->center));
        double radius2 = b->radius * b->radius; /* Sphere ctor main.cpp:204-208 */
        double bb = dot(distance, dir);
        double c = dot(distance, distance) - radius2;
        double d = (bb * bb) - c;
        if (d > 0.0L) return -bb - sqrt(d);
        return (double)-1.0L;
    }
}

/* getNormal: Plane main.cpp:192-194, Sphere main.cpp:224-226. */
static v3 shape_normal(const ptor_body* b, v3 point) {
    if (b->shape == PTOR_PLANE || b->shape == PTOR_RECTANGLE || b->shape == PTOR_RECTANGLE_MT)
        return ld3(b->normal); /* raytrace3.cpp:278-281, light_tracing.cpp:497-499 */
    return normalize(sub(point, ld3(b->center)));
}

/* Nearest-hit scan: main.cpp:372-383; `skip` is the light excluded by next_event.cpp:415-419
 * (-1 = none).  Ties go to the larger index (<=). */
static int nearest(const por_scene* s, v3 o, v3 dir, int skip, double* t_out) {
    int hit = -1;
    double mint = DBL_MAX;
    for (int i = 0; i < s->body_count; i++) {
        if (i == skip) continue;
        double t = intersect(&s->bodies[i], o, dir);
        int positive = (s->flags & POR_F_RAYTRACE3) ? (t > 0.0000001L) : (t > 0);
        if (positive && (t <= mint)) {
            mint = t;
            hit = i;
        }
    }
    *t_out = mint;
    return hit;
}

/* ---- materials: main.cpp:250-253 (diffuse), 261-265 (Chrome), 279-305 (Glass) ------- */

static v3 bounce(const ptor_body* b, draws* d, v3 dir, v3 normal) {
    if (b->material == PTOR_CHROME) {
        double theta1 = fabs(dot(dir, normal));
        return add(dir, muls(normal, (double)(theta1 * 2.0L)));
    }
    if (b->material == PTOR_GLASS) {
        double theta1 = fabs(dot(dir, normal));
        double internalIndex, externalIndex;
        if (theta1 >= 0.0L) {
            internalIndex = b->ior;
            externalIndex = (double)1.0L;
        } else {
            internalIndex = (double)1.0L;
            externalIndex = b->ior;
        }
        double eta = externalIndex / internalIndex;
        /* sqrt of a long double argument resolves to the x87 sqrtl */
        double theta2 = (double)sqrtl(1.0L - (eta * eta) * (1.0L - (theta1 * theta1)));
        double rs = (externalIndex * theta1 - internalIndex * theta2) / (externalIndex * theta1 + internalIndex * theta2);
        double rp = (internalIndex * theta1 - externalIndex * theta2) / (internalIndex * theta1 + externalIndex * theta2);
        double reflectance = (rs * rs + rp * rp);
        if (unif(d) < reflectance + b->reflection) return add(dir, muls(normal, (double)(theta1 * 2.0L)));
        return add(muls(add(dir, muls(normal, theta1)), eta), muls(normal, -theta2));
    }
    /* Material::bounce; measured materials scatter like the base class too */
    return random_in_hemisphere(d, normal);
}

/* ---- measured BRDF: MatusikBRDF.h --------------------------------------------------- */

#define BRDF_RES_THETA_H 90
#define BRDF_RES_THETA_D 90
#define BRDF_RES_PHI_D 360
#define BRDF_N (BRDF_RES_THETA_H * BRDF_RES_THETA_D * BRDF_RES_PHI_D / 2)

static void brdf_normalize(double* v) { /* MatusikBRDF.h:124-130 */
    double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] = v[0] / len;
    v[1] = v[1] / len;
    v[2] = v[2] / len;
}

static void brdf_rotate(const double* vector, const double* axis, double angle, double* out) { /* :133-156 */
    double temp;
    double cross[3];
    double cos_ang = cos(angle);
    double sin_ang = sin(angle);
    out[0] = vector[0] * cos_ang;
    out[1] = vector[1] * cos_ang;
    out[2] = vector[2] * cos_ang;
    temp = axis[0] * vector[0] + axis[1] * vector[1] + axis[2] * vector[2];
    temp = temp * (1.0 - cos_ang);
    out[0] += axis[0] * temp;
    out[1] += axis[1] * temp;
    out[2] += axis[2] * temp;
    cross[0] = axis[1] * vector[2] - axis[2] * vector[1]; /* cross_product(axis, vector) :117-122 */
    cross[1] = axis[2] * vector[0] - axis[0] * vector[2];
    cross[2] = axis[0] * vector[1] - axis[1] * vector[0];
    out[0] += cross[0] * sin_ang;
    out[1] += cross[1] * sin_ang;
    out[2] += cross[2] * sin_ang;
}

/* A NaN angle converts to INT_MIN on x86 and is then clamped to 0 by every index function;
 * make that explicit so the restatement does not rely on an out-of-range conversion. */
static int to_int(double v) {
    if (!(v == v)) return 0 - 2147483647 - 1;
    if (v >= 2147483648.0 || v <= -2147483649.0) return 0 - 2147483647 - 1;
    return (int)v;
}

static int theta_half_index(double theta_half) { /* :211-223 */
    if (theta_half <= 0.0) return 0;
    double theta_half_deg = ((theta_half / (M_PI / 2.0)) * BRDF_RES_THETA_H);
    double temp = theta_half_deg * BRDF_RES_THETA_H;
    temp = sqrt(temp);
    int ret_val = to_int(temp);
    if (ret_val < 0) ret_val = 0;
    if (ret_val >= BRDF_RES_THETA_H) ret_val = BRDF_RES_THETA_H - 1;
    return ret_val;
}

static int theta_diff_index(double theta_diff) { /* :228-237 */
    int tmp = to_int(theta_diff / (M_PI * 0.5) * BRDF_RES_THETA_D);
    if (tmp < 0) return 0;
    if (tmp < BRDF_RES_THETA_D - 1) return tmp;
    return BRDF_RES_THETA_D - 1;
}

static int phi_diff_index(double phi_diff) { /* :241-257 */
    if (phi_diff < 0.0) phi_diff += M_PI;
    int tmp = to_int(phi_diff / M_PI * BRDF_RES_PHI_D / 2);
    if (tmp < 0) return 0;
    if (tmp < BRDF_RES_PHI_D / 2 - 1) return tmp;
    return BRDF_RES_PHI_D / 2 - 1;
}

/* std_coords_to_half_diff_coords :159-205 + the index expression of lookup_brdf_val :63-66 */
static int brdf_index(double theta_in, double fi_in, double theta_out, double fi_out) {
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
    double out[3] = {out_vec_x, out_vec_y, out_vec_z};
    brdf_normalize(out);
    double half_x = (in_vec_x + out_vec_x) / 2.0f;
    double half_y = (in_vec_y + out_vec_y) / 2.0f;
    double half_z = (in_vec_z + out_vec_z) / 2.0f;
    double half[3] = {half_x, half_y, half_z};
    brdf_normalize(half);
    double theta_half = acos(half[2]);
    double fi_half = atan2(half[1], half[0]);
    double bi_normal[3] = {0.0, 1.0, 0.0};
    double normal[3] = {0.0, 0.0, 1.0};
    double temp[3];
    double diff[3];
    brdf_rotate(in, normal, -fi_half, temp);
    brdf_rotate(temp, bi_normal, -theta_half, diff);
    double theta_diff = acos(diff[2]);
    double fi_diff = atan2(diff[1], diff[0]);
    return phi_diff_index(fi_diff) + theta_diff_index(theta_diff) * BRDF_RES_PHI_D / 2 +
           theta_half_index(theta_half) * BRDF_RES_PHI_D / 2 * BRDF_RES_THETA_D;
}

/* lookup_brdf_val :50-80 */
static void brdf_lookup(const double* table, double theta_in, double fi_in, double theta_out, double fi_out,
                        double* r, double* g, double* b, int* ind_out) {
    int ind = brdf_index(theta_in, fi_in, theta_out, fi_out);
    if (ind_out) *ind_out = ind;
    double red_val = table[ind] * (1.0 / 1500.0);
    double green_val = table[ind + BRDF_N] * (1.15 / 1500.0);
    double blue_val = table[ind + 2 * BRDF_N] * (1.66 / 1500.0);
    if (red_val < 0.0 || green_val < 0.0 || blue_val < 0.0) {
        red_val = 0.0;
        green_val = 0.0;
        blue_val = 0.0;
    }
    *r = red_val;
    *g = green_val;
    *b = blue_val;
}

/* vector_to_spherical: raytrace3.cpp:725-728 (pow(x,2) is folded to x*x by g++) */
static void vector_to_spherical(v3 in, double* theta, double* phi) {
    *theta = atan(in.y / in.x);
    *phi = acos(in.z / sqrt(in.x * in.x + in.y * in.y + in.z * in.z));
}

/* ---- estimators ---------------------------------------------------------------------- */

typedef struct hitrec {
    v3 point; /* the OFFSET point: next_event.cpp:518-526 */
    int body;
} hitrec;

/* One trace() level: main.cpp:367-403 / next_event.cpp:491-528.  Returns the body hit or -1. */
static int trace_segment(const por_scene* s, draws* d, v3* o, v3* dir, v3* point_out) {
    double mint;
    int hit = nearest(s, *o, *dir, -1, &mint);
    if (hit < 0) return -1;
    const ptor_body* b = &s->bodies[hit];
    v3 point = add(*o, muls(*dir, mint));
    v3 normal = shape_normal(b, point);
    v3 direction = bounce(b, d, *dir, normal);
    if (dot(direction, *dir) > 0.0f) {
        point = add(*o, muls(*dir, (double)(mint * 1.0000001L)));
    } else {
        point = add(*o, muls(*dir, (double)(mint * 0.9999999L)));
    }
    *point_out = point;
    *o = point;
    *dir = direction;
    return hit;
}

/* Renderer::trace of main.cpp:367-405 unrolled: the recursion returns
 * trace(next)*color + emission, i.e. a Horner evaluation from the deepest hit outwards. */
static v3 radiance_pt(const por_scene* s, draws* d, v3 o, v3 dir) {
    int bodies[POR_MAX_DEPTH];
    int n = 0;
    v3 p;
    while (n < POR_MAX_DEPTH) {
        int hit = trace_segment(s, d, &o, &dir, &p);
        if (hit < 0) break;
        bodies[n++] = hit;
    }
    v3 L = V(0.0L, 0.0L, 0.0L);
    for (int k = n - 1; k >= 0; k--) {
        const ptor_body* b = &s->bodies[bodies[k]];
        if (s->flags & POR_F_RAYTRACE3) L = add(L, V(0.0L, 0.0L, 0.0L)); /* + direct_color raytrace3.cpp:840 */
        L = add(mul(L, ld3(b->color)), ld3(b->emission));
    }
    return L;
}

#define POR_MAX_LIGHTS 65536 /* the reference's list is a std::vector; more than this many is reported, not cut */
typedef struct lightlist {
    int idx[POR_MAX_LIGHTS];
    int count;
} lightlist;

/* Renderer ctor next_event.cpp:362-367 */
static lightlist find_lights(const por_scene* s) {
    lightlist l;
    l.count = 0;
    for (int i = 0; i < s->body_count; i++) {
        const double* e = s->bodies[i].emission;
        if ((e[0] >= 1.0L) || (e[1] >= 1.0L) || (e[2] >= 1.0L)) {
            if (l.count >= POR_MAX_LIGHTS) {
                fprintf(stderr, "oracle: more than %d lights\n", POR_MAX_LIGHTS);
                abort();
            }
            l.idx[l.count++] = i;
        }
    }
    return l;
}

/* Pixel body of next_event.cpp:384-470.  `recs`/`nrec_out` (optional) expose the gather path. */
static int* g_brdf_ind_out; /* diagnostics: table index of each lit measured vertex (see por_trace_brdf_indices) */

static v3 radiance_nee(const por_scene* s, const cam* c, const lightlist* lights, draws* d, double x, double y,
                       hitrec* recs_out, int* nrec_out, int* visible_out) {
    v3 o = c->origin;
    v3 dir = cam_ray_dir(c, x, y);
    hitrec recs[POR_MAX_DEPTH];
    int n = 0;
    while (n < POR_MAX_DEPTH) {
        v3 p;
        int hit = trace_segment(s, d, &o, &dir, &p);
        if (hit < 0) break;
        recs[n].point = p;
        recs[n].body = hit;
        n++;
    }
    /* Body* light = lights[unifRand(lights.size() - 1)];  :390 */
    long li = unif_long(d, (long)lights->count - 1);
    int light = lights->idx[li];
    const ptor_body* lb = &s->bodies[light];
    /* light->shape->center through a Shape* reads the base member, never assigned: (0,0,0) */
    v3 lcenter = (s->flags & POR_F_LIGHT_TRUE_CENTER) ? ld3(lb->center) : V(0.0L, 0.0L, 0.0L);
    v3 lterm = mul(ld3(lb->emission), ld3(lb->color));
    v3 final_color = V(0.0L, 0.0L, 0.0L);
    int vis = 0;
    for (int gi = n - 1; gi >= 0; gi--) {
        v3 P = recs[gi].point;
        v3 bdir = normalize(sub(P, lcenter));
        double mint;
        int hit = nearest(s, lcenter, bdir, light, &mint);
        if (hit != recs[gi].body) continue;
        vis |= 1 << gi;
        const ptor_body* hb = &s->bodies[hit];
        v3 next_point = (gi >= 1) ? recs[gi - 1].point : cam_point(c, x, y);
        v3 nprime = normalize(sub(next_point, P));
        if (hb->material == PTOR_MEASURED) {
            /* Defined by this build (the reference never executes its BRDF lookups,
             * raytrace3.cpp:555,646-676): the RGB of the table replaces the scalar flux. */
            double ti, pi_, to, po, r, g, b;
            vector_to_spherical(bdir, &ti, &pi_);
            vector_to_spherical(nprime, &to, &po);
            int ind;
            brdf_lookup(s->brdf_tables[hb->brdf_id], ti, pi_, to, po, &r, &g, &b, &ind);
            if (g_brdf_ind_out) g_brdf_ind_out[gi] = ind;
            final_color = add(mul(add(final_color, mul(lterm, V(r, g, b))), ld3(hb->color)), ld3(hb->emission));
        } else {
            v3 light_brdf = bounce(hb, d, bdir, nprime);
            double light_flux = fabs(dot(light_brdf, shape_normal(hb, P)));
            final_color = add(mul(add(final_color, muls(lterm, light_flux)), ld3(hb->color)), ld3(hb->emission));
        }
    }
    if (recs_out) memcpy(recs_out, recs, sizeof(hitrec) * (size_t)n);
    if (nrec_out) *nrec_out = n;
    if (visible_out) *visible_out = vis;
    return final_color;
}

/* Pixel body of bidirectional.cpp:392-494: the gather path as in next_event.cpp, a light path
 * ("shoot") started at the chosen light, and one connection per (gather vertex, shoot vertex) pair,
 * both loops running from the deepest vertex to the first.
 *
 * flux_lookup (bidirectional.cpp:377-382) is a double[5][4] indexed by [shoot_hits.size()][shoot_i]
 * (:484).  A light path of five vertices, which is every light path of a closed scene, reads row 5,
 * past the end of the array: the reference's value there is whatever its stack holds.  The factor
 * only ever scales emission*colour of the chosen light (:486), which is (0,0,0) for every light of
 * the reference's scene (:684-735), so any FINITE value gives the same pixel; frames of the
 * reference built at -O0 and at -O3 agree bit for bit (tests/golden/bidir_ub_check.json).
 * kFluxRow5 stands for that value here; the product refuses scenes in which it could matter. */
static const double kFluxLookup[5][4] = {{0, 0, 0, 0},
                                         {1.0L, 0, 0, 0},
                                         {0.5L, 0.5L, 0, 0},
                                         {0.5L, 0.25L, 0.25L, 0},
                                         {0.5L, 0.25L, 0.125L, 0.125L}};
static const double kFluxRow5 = 1.0;

static v3 radiance_bidir(const por_scene* s, const cam* c, const lightlist* lights, draws* d, double x, double y,
                         int* nrec_out, int* visible_out) {
    v3 o = c->origin;
    v3 dir = cam_ray_dir(c, x, y);
    hitrec gather[POR_MAX_DEPTH], shoot[POR_MAX_DEPTH];
    int n = 0, ns = 0;
    while (n < POR_MAX_DEPTH) { /* trace(ray, 0, &gather_hits) :394 */
        v3 p;
        int hit = trace_segment(s, d, &o, &dir, &p);
        if (hit < 0) break;
        gather[n].point = p;
        gather[n].body = hit;
        n++;
    }
    long li = unif_long(d, (long)lights->count - 1); /* :398 */
    int light = lights->idx[li];
    const ptor_body* lb = &s->bodies[light];
    /* light->shape->center through a Shape* is the never-assigned base member (0,0,0); getNormal
     * is virtual and uses the shape's own centre (:402-404) */
    v3 lcenter = (s->flags & POR_F_LIGHT_TRUE_CENTER) ? ld3(lb->center) : V(0.0L, 0.0L, 0.0L);
    o = lcenter;
    dir = random_in_hemisphere(d, shape_normal(lb, lcenter));
    while (ns < POR_MAX_DEPTH) { /* trace(ray, 0, &shoot_hits) :406 */
        v3 p;
        int hit = trace_segment(s, d, &o, &dir, &p);
        if (hit < 0) break;
        shoot[ns].point = p;
        shoot[ns].body = hit;
        ns++;
    }
    v3 lterm = mul(ld3(lb->emission), ld3(lb->color));
    v3 final_color = V(0.0L, 0.0L, 0.0L);
    int vis = 0;
    for (int gi = n - 1; gi >= 0; gi--) {
        for (int si = ns - 1; si >= 0; si--) {
            v3 G = gather[gi].point, S = shoot[si].point;
            v3 bdir = normalize(sub(G, S)); /* bi_ray :415-417 */
            double mint;
            int hit = nearest(s, S, bdir, light, &mint); /* :419-433 */
            if (hit != gather[gi].body) continue;        /* :435-437 */
            vis |= 1 << (gi * POR_MAX_DEPTH + si);
            const ptor_body* hb = &s->bodies[hit];
            v3 next_point = (gi >= 1) ? gather[gi - 1].point : cam_point(c, x, y); /* :449-454 */
            v3 gather_brdf = bounce(hb, d, bdir, normalize(sub(next_point, G)));   /* :463-465 */
            double gather_flux = fabs(dot(gather_brdf, shape_normal(hb, G)));      /* :467 */
            const ptor_body* sb = &s->bodies[shoot[si].body];
            v3 shoot_brdf = bounce(sb, d, bdir, normalize(sub(G, S)));             /* :470-472 */
            double shoot_flux = fabs(dot(shoot_brdf, shape_normal(sb, S)));        /* :474 */
            double lookup = ns < 5 ? kFluxLookup[ns][si] : kFluxRow5;
            double flux = lookup * gather_flux * shoot_flux;                       /* :484 */
            final_color = add(mul(add(final_color, muls(lterm, flux)), ld3(hb->color)), ld3(hb->emission)); /* :486-487 */
        }
    }
    if (nrec_out) *nrec_out = n | (ns << 8);
    if (visible_out) *visible_out = vis;
    return final_color;
}

/* ---- frame loop: Renderer::iterate main.cpp:349-365 / next_event.cpp:374-473 -------- */

static void stream(draws* d, const por_scene* s, uint32_t it, uint32_t dom, uint32_t idx) {
    ptor_rng_init(&d->rng, s->seed, it, dom, idx);
}

/* One iteration.  In PIXEL mode the y jitter, each row's x jitter and each pixel use their own
 * streams; in the other modes every draw comes from `d` in program order, as in the reference. */
static void iterate(const por_scene* s, const cam* c, const lightlist* lights, draws* d, uint32_t it, double* buffer) {
    int i = 0;
    int W = s->width, H = s->height;
    uint32_t row = 0;
    draws dj = *d;
    dj.count = 0;
    if (d->mode == PTOR_MODE_PIXEL) stream(&dj, s, it, PTOR_DOM_FRAME, 0);
    draws* jy = (d->mode == PTOR_MODE_PIXEL) ? &dj : d;
    for (double y = unif(jy) / (double)H, ystep = (double)(1.0L / (double)H); y < 0.99999L; y += ystep) {
        if (d->mode == PTOR_MODE_PIXEL) stream(&dj, s, it, PTOR_DOM_ROW, row);
        row++;
        draws* jx = (d->mode == PTOR_MODE_PIXEL) ? &dj : d;
        for (double x = unif(jx) / (double)W, xstep = (double)(1.0L / (double)W); x < 0.99999L; x += xstep) {
            if (d->mode == PTOR_MODE_PIXEL) stream(d, s, it, PTOR_DOM_PIXEL, (uint32_t)i);
            v3 col;
            if (s->estimator == PTOR_EST_NEE)
                col = radiance_nee(s, c, lights, d, x, y, NULL, NULL, NULL);
            else if (s->estimator == PTOR_EST_BIDIR)
                col = radiance_bidir(s, c, lights, d, x, y, NULL, NULL);
            else
                col = radiance_pt(s, d, c->origin, cam_ray_dir(c, x, y));
            buffer[3 * i + 0] += col.x;
            buffer[3 * i + 1] += col.y;
            buffer[3 * i + 2] += col.z;
            i++;
        }
    }
    if (d->mode == PTOR_MODE_PIXEL) d->count += dj.count; /* informational only */
}

/* Renderer::ray_iterate(start_row, row_count) light_tracing.cpp:774-790: the rows' pixels on the
 * grid, i = start_row * width onwards; the radiance comes from this scene's estimator (the
 * reference's ray_trace gathers a photon map, which is outside this restatement).  PIXEL mode only:
 * every pixel has its own stream. */
static void ray_iterate(const por_scene* s, const cam* c, const lightlist* lights, draws* d, uint32_t it, int start_row,
                        int row_count, double* buffer) {
    int i = start_row * s->width;
    for (int y = start_row; y < (start_row + row_count); y++) {
        for (int x = 0; x < s->width; x++) {
            double fx = (double)x / (double)s->width, fy = (double)y / (double)s->height;
            stream(d, s, it, PTOR_DOM_PIXEL, (uint32_t)i);
            v3 col;
            if (s->estimator == PTOR_EST_NEE)
                col = radiance_nee(s, c, lights, d, fx, fy, NULL, NULL, NULL);
            else
                col = radiance_pt(s, d, c->origin, cam_ray_dir(c, fx, fy));
            buffer[3 * i + 0] += col.x;
            buffer[3 * i + 1] += col.y;
            buffer[3 * i + 2] += col.z;
            i++;
        }
    }
}

/* ======================= exported test API ============================================ */

/* accum[W*H*3] += n_it passes of ray_iterate over rows [start_row, start_row + row_count). */
void por_render_rows(const por_scene* s, int start_row, int row_count, int first_it, int n_it, double* accum) {
    cam c = make_cam(&s->camera);
    lightlist lights = find_lights(s);
    draws d;
    memset(&d, 0, sizeof d);
    d.mode = PTOR_MODE_PIXEL;
    for (int it = first_it; it < first_it + n_it; it++) ray_iterate(s, &c, &lights, &d, (uint32_t)it, start_row, row_count, accum);
}

/* The rows of worker i of thread_count: light_tracing.cpp:1441-1451. */
void por_row_split(int height, int thread_count, int i, int* worker_start_row, int* worker_rows_out) {
    int row_group_size = height / thread_count;
    int worker_rows = row_group_size;
    int worker_start = i * row_group_size;
    if (i == thread_count - 1) worker_rows = height - (i * row_group_size);
    *worker_start_row = worker_start;
    *worker_rows_out = worker_rows;
}

/* accum[W*H*3] += n_it iterations [first_it, first_it+n_it).  Returns the number of draws. */
long long por_render(const por_scene* s, int mode, int first_it, int n_it, const int* script, int script_len,
                     double* accum) {
    cam c = make_cam(&s->camera);
    lightlist lights = find_lights(s);
    draws d;
    memset(&d, 0, sizeof d);
    d.mode = mode;
    d.script = script;
    d.script_len = script_len;
    if (mode == PTOR_MODE_SEQ) ptor_rng_init(&d.rng, s->seed, 0, PTOR_DOM_SEQ, 0);
    for (int it = first_it; it < first_it + n_it; it++) iterate(s, &c, &lights, &d, (uint32_t)it, accum);
    return d.count;
}

/* Frame geometry of one iteration in PIXEL mode: ys[H], xs[H*W], row_len[H]; returns the
 * number of rows.  Entries beyond a row's length / the row count are left untouched. */
int por_frame_layout(const por_scene* s, int it, double* ys, double* xs, int* row_len) {
    draws dj;
    memset(&dj, 0, sizeof dj);
    dj.mode = PTOR_MODE_PIXEL;
    int W = s->width, H = s->height;
    int row = 0;
    stream(&dj, s, (uint32_t)it, PTOR_DOM_FRAME, 0);
    for (double y = unif(&dj) / (double)H, ystep = (double)(1.0L / (double)H); y < 0.99999L; y += ystep) {
        stream(&dj, s, (uint32_t)it, PTOR_DOM_ROW, (uint32_t)row);
        int col = 0;
        for (double x = unif(&dj) / (double)W, xstep = (double)(1.0L / (double)W); x < 0.99999L; x += xstep) {
            xs[(size_t)row * W + col] = x;
            col++;
        }
        ys[row] = y;
        row_len[row] = col;
        row++;
    }
    return row;
}

/* t_out[n][body_count] of every body + nearest index/t with body `skip` excluded. */
void por_intersect(const por_scene* s, const double* rays6, int n, int skip, double* t_all, int* idx, double* t) {
    for (int i = 0; i < n; i++) {
        v3 o = ld3(rays6 + 6 * i), dir = ld3(rays6 + 6 * i + 3);
        if (t_all)
            for (int b = 0; b < s->body_count; b++) t_all[(size_t)i * s->body_count + b] = intersect(&s->bodies[b], o, dir);
        double mint;
        int hit = nearest(s, o, dir, skip, &mint);
        if (idx) idx[i] = hit;
        if (t) t[i] = mint;
    }
}

void por_get_rays(const por_scene* s, const double* xy, int n, double* rays6) {
    cam c = make_cam(&s->camera);
    for (int i = 0; i < n; i++) {
        v3 dir = cam_ray_dir(&c, xy[2 * i], xy[2 * i + 1]);
        rays6[6 * i + 0] = c.origin.x;
        rays6[6 * i + 1] = c.origin.y;
        rays6[6 * i + 2] = c.origin.z;
        rays6[6 * i + 3] = dir.x;
        rays6[6 * i + 4] = dir.y;
        rays6[6 * i + 5] = dir.z;
    }
}

/* Plane ctor normalisation (main.cpp:176-179): stored = normalize(given). */
void por_plane_normal(const double* in3, double* out3) {
    v3 n = normalize(ld3(in3));
    out3[0] = n.x;
    out3[1] = n.y;
    out3[2] = n.z;
}

/* Rectangle ctor normal (raytrace3.cpp:260) from the twelve corner coordinates. */
void por_rect_normal(const double* corners12, double* out3) {
    v3 n = rect_normal(corners12);
    out3[0] = n.x;
    out3[1] = n.y;
    out3[2] = n.z;
}

int por_bounce(const por_scene* s, int body, const double* ray6, const double* normal3, const int* script,
               int script_len, double* out3) {
    draws d;
    memset(&d, 0, sizeof d);
    d.mode = PTOR_MODE_SCRIPT;
    d.script = script;
    d.script_len = script_len;
    v3 r = bounce(&s->bodies[body], &d, ld3(ray6 + 3), ld3(normal3));
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
    return (int)d.count;
}

/* Per-path records in PIXEL mode for n explicit (iteration, pixel index, x, y) samples:
 * rgb[n][3], nrec[n], bodies[n][5], points[n][5][3], visible[n] (bit gi set when vertex gi was
 * lit; NEE only), draws_used[n]. */
void por_trace_paths(const por_scene* s, const int* it_idx2, const double* xy, int n, double* rgb, int* nrec,
                     int* bodies, double* points, int* visible, int* draws_used) {
    cam c = make_cam(&s->camera);
    lightlist lights = find_lights(s);
    for (int k = 0; k < n; k++) {
        draws d;
        memset(&d, 0, sizeof d);
        d.mode = PTOR_MODE_PIXEL;
        stream(&d, s, (uint32_t)it_idx2[2 * k], PTOR_DOM_PIXEL, (uint32_t)it_idx2[2 * k + 1]);
        hitrec recs[POR_MAX_DEPTH];
        int nr = 0, vis = 0;
        v3 col;
        if (s->estimator == PTOR_EST_NEE) {
            col = radiance_nee(s, &c, &lights, &d, xy[2 * k], xy[2 * k + 1], recs, &nr, &vis);
        } else if (s->estimator == PTOR_EST_BIDIR) {
            /* no records: nrec = gather count | shoot count << 8, visible bit gi * 5 + si per connection */
            int counts = 0;
            col = radiance_bidir(s, &c, &lights, &d, xy[2 * k], xy[2 * k + 1], &counts, &vis);
            nr = counts;
        } else {
            v3 o = c.origin, dir = cam_ray_dir(&c, xy[2 * k], xy[2 * k + 1]);
            draws d2 = d;
            col = radiance_pt(s, &d, o, dir);
            while (nr < POR_MAX_DEPTH) { /* replay for the records */
                v3 p;
                int hit = trace_segment(s, &d2, &o, &dir, &p);
                if (hit < 0) break;
                recs[nr].point = p;
                recs[nr].body = hit;
                nr++;
            }
        }
        rgb[3 * k] = col.x;
        rgb[3 * k + 1] = col.y;
        rgb[3 * k + 2] = col.z;
        if (nrec) nrec[k] = nr;
        for (int j = 0; j < POR_MAX_DEPTH; j++) {
            if (bodies) bodies[5 * k + j] = j < nr ? recs[j].body : -1;
            if (points) {
                points[15 * k + 3 * j + 0] = j < nr ? recs[j].point.x : 0.0;
                points[15 * k + 3 * j + 1] = j < nr ? recs[j].point.y : 0.0;
                points[15 * k + 3 * j + 2] = j < nr ? recs[j].point.z : 0.0;
            }
        }
        if (visible) visible[k] = vis;
        if (draws_used) draws_used[k] = (int)d.count;
    }
}

/* The reference's extended-precision expressions, evaluated with real x87 long double. */
void por_x87(int kind, const double* a, const double* b, int n, double* out) {
    for (int i = 0; i < n; i++) {
        switch (kind) {
            case 0: out[i] = (double)(1.0L / a[i]); break;                                        /* main.cpp:93 */
            case 1: out[i] = (double)(((int)a[i] / (double)2147483647) * 2.0L - 1.0L); break;      /* main.cpp:108 */
            case 2: out[i] = (double)(a[i] * 1.0000001L); break;                                  /* main.cpp:396 */
            case 3: out[i] = (double)(a[i] * 0.9999999L); break;                                  /* main.cpp:400 */
            default: {                                                                             /* main.cpp:292 */
                double eta = a[i], theta1 = b[i];
                out[i] = (double)sqrtl(1.0L - (eta * eta) * (1.0L - (theta1 * theta1)));
            }
        }
    }
}

/* k values of a stream (contract check against the product's generator). */
void por_rng_fill(uint64_t seed, uint32_t it, uint32_t dom, uint32_t idx, int n, int32_t* k_out, double* u_out) {
    ptor_rng r;
    ptor_rng_init(&r, seed, it, dom, idx);
    for (int i = 0; i < n; i++) {
        int32_t k = ptor_rng_next(&r);
        if (k_out) k_out[i] = k;
        if (u_out) u_out[i] = k / (double)2147483647;
    }
}

/* MatusikBRDF lookups on an in-memory table (3*1458000 doubles). */
/* Diagnostics (single-threaded): the table index of every lit measured vertex of n explicit samples,
 * ind[n][5], -1 where vertex gi is not a lit measured one. */
void por_trace_brdf_indices(const por_scene* s, const int* it_idx2, const double* xy, int n, int* ind) {
    cam c = make_cam(&s->camera);
    lightlist lights = find_lights(s);
    for (int k = 0; k < n; k++) {
        draws d;
        memset(&d, 0, sizeof d);
        d.mode = PTOR_MODE_PIXEL;
        stream(&d, s, (uint32_t)it_idx2[2 * k], PTOR_DOM_PIXEL, (uint32_t)it_idx2[2 * k + 1]);
        for (int j = 0; j < POR_MAX_DEPTH; j++) ind[5 * k + j] = -1;
        g_brdf_ind_out = ind + 5 * k;
        (void)radiance_nee(s, &c, &lights, &d, xy[2 * k], xy[2 * k + 1], NULL, NULL, NULL);
        g_brdf_ind_out = NULL;
    }
}

void por_brdf_lookup(const double* table, const double* angles4, int n, double* rgb, int* ind) {
    for (int i = 0; i < n; i++) {
        const double* a = angles4 + 4 * i;
        brdf_lookup(table, a[0], a[1], a[2], a[3], &rgb[3 * i], &rgb[3 * i + 1], &rgb[3 * i + 2], ind ? &ind[i] : NULL);
    }
}

void por_vector_to_spherical(const double* v3in, int n, double* theta_phi) {
    for (int i = 0; i < n; i++) vector_to_spherical(ld3(v3in + 3 * i), &theta_phi[2 * i], &theta_phi[2 * i + 1]);
}

/* save(): main.cpp:408-434 / next_event.cpp:533-554.  Writes the PPM text into `out`
 * (capacity cap); returns the byte count needed. */
long por_save_ppm(const double* buffer, int samples, int width, int height, char* out, long cap) {
    long n = 0;
    char line[96];
    int len = snprintf(line, sizeof line, "P3\n%d %d\n255\n", width, height);
    if (n + len <= cap) memcpy(out + n, line, (size_t)len);
    n += len;
    const double* pixel = buffer;
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            int r = (int)((255.0L * pixel[0]) / (double)samples);
            int g = (int)((255.0L * pixel[1]) / (double)samples);
            int b = (int)((255.0L * pixel[2]) / (double)samples);
            if (r > 255) r = 255;
            if (g > 255) g = 255;
            if (b > 255) b = 255;
            len = snprintf(line, sizeof line, "%d %d %d\n", r, g, b);
            if (n + len <= cap) memcpy(out + n, line, (size_t)len);
            n += len;
            pixel += 3;
        }
    }
    return n;
}

/* ---- CPU baseline timing: independent single-thread shards -------------------------- */

typedef struct shard_job {
    const por_scene* s;
    int first_it, n_it;
    double* buf;
} shard_job;

static void* shard_main(void* p) {
    shard_job* j = (shard_job*)p;
    por_render(j->s, PTOR_MODE_PIXEL, j->first_it, j->n_it, NULL, 0, j->buf);
    return NULL;
}

/* Render `iters_per_thread` iterations on each of `threads` threads (disjoint iteration ranges,
 * private buffers).  Returns wall seconds; checksum = sum of all buffers. */
double por_time_render(const por_scene* s, int iters_per_thread, int threads, double* checksum) {
    size_t n = (size_t)s->width * s->height * 3;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    shard_job* jobs = (shard_job*)malloc(sizeof(shard_job) * (size_t)threads);
    for (int k = 0; k < threads; k++) {
        jobs[k].s = s;
        jobs[k].first_it = k * iters_per_thread;
        jobs[k].n_it = iters_per_thread;
        jobs[k].buf = (double*)calloc(n, sizeof(double));
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < threads; k++) pthread_create(&th[k], NULL, shard_main, &jobs[k]);
    for (int k = 0; k < threads; k++) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double tot = 0.0;
    for (int k = 0; k < threads; k++) {
        for (size_t i = 0; i < n; i++) tot += jobs[k].buf[i];
        free(jobs[k].buf);
    }
    if (checksum) *checksum = tot;
    free(th);
    free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
