// Host-side parts of the C ABI that need no GPU: the reference's hard-coded scenes as data
// and the byte-compatible PPM writer.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/ptgpu.h"

namespace {

struct Builder {
    pt_body* out;
    int cap;
    int n = 0;
    bool overflow = false;

    pt_body* next() {
        if (n >= cap) {
            overflow = true;
            return nullptr;
        }
        pt_body* b = &out[n++];
        memset(b, 0, sizeof *b);
        b->brdf_id = -1;
        return b;
    }
    static void set3(double* d, double x, double y, double z) {
        d[0] = x;
        d[1] = y;
        d[2] = z;
    }
    pt_body* sphere(double cx, double cy, double cz, double r) {
        pt_body* b = next();
        if (!b) return nullptr;
        b->shape = PT_SPHERE;
        set3(b->center, cx, cy, cz);
        b->radius = r;
        return b;
    }
    // Rectangle(p1, p2, p3, p4) raytrace3.cpp:254
    pt_body* rectangle(const double (&c)[12]) {
        pt_body* b = next();
        if (!b) return nullptr;
        b->shape = PT_RECTANGLE;
        memcpy(b->corners, c, sizeof c);
        return b;
    }
    // Rectangle(p1, p2, p3, p4) light_tracing.cpp:484
    pt_body* rectangle_mt(const double (&c)[12]) {
        pt_body* b = rectangle(c);
        if (b) b->shape = PT_RECTANGLE_MT;
        return b;
    }
    pt_body* plane(double cx, double cy, double cz, double nx, double ny, double nz) {
        pt_body* b = next();
        if (!b) return nullptr;
        b->shape = PT_PLANE;
        set3(b->center, cx, cy, cz);
        set3(b->normal, nx, ny, nz);
        return b;
    }
};

void diffuse(pt_body* b, double r, double g, double bl) {
    if (!b) return;
    b->material = PT_DIFFUSE;
    Builder::set3(b->color, r, g, bl);
}
void emitter(pt_body* b, double r, double g, double bl) {
    if (!b) return;
    b->material = PT_DIFFUSE;
    Builder::set3(b->color, 0.0, 0.0, 0.0);
    Builder::set3(b->emission, r, g, bl);
}
void chrome(pt_body* b, double r, double g, double bl) {
    if (!b) return;
    b->material = PT_CHROME;
    Builder::set3(b->color, r, g, bl);
}
void glass(pt_body* b, double r, double g, double bl, double ior, double refl) {
    if (!b) return;
    b->material = PT_GLASS;
    Builder::set3(b->color, r, g, bl);
    b->ior = ior;
    b->reflection = refl;
}

// V3(x,y,z).normalize() as the reference evaluates it (main.cpp:92-102): the reciprocal of the
// length is formed in long double, narrowed, then multiplied in.
void ref_normalize(double x, double y, double z, double* o) {
    double s = std::sqrt(x * x + y * y + z * z);
    double inv = (double)(1.0L / (long double)s);
    o[0] = x * inv;
    o[1] = y * inv;
    o[2] = z * inv;
}

// Camera of both programs: main.cpp:476-481 == next_event.cpp:608-613.
void reference_camera(pt_camera* c) {
    Builder::set3(c->origin, 0.0, -0.5, 0.0);
    Builder::set3(c->topleft, -1.3, 1.0, 1.0);
    Builder::set3(c->topright, 1.3, 1.0, 1.0);
    Builder::set3(c->bottomleft, -1.3, 1.0, -1.0);
}

// Bodies 0..13 shared by both programs; the few constants that differ are parameters.
// The two "divider" planes are declared `Material x = Glass(...)` in the reference
// (main.cpp:538,596), which slices them to plain diffuse materials of the Glass colour, and their
// normals are normalised once by the caller and once more by the Plane constructor.
void common_bodies(Builder& B, bool nee) {
    glass(B.sphere(1.0, 2.0, 0.0, 0.5), 1.0, 1.0, 1.0, 1.5, 0.1);    // glass        main.cpp:487 / next_event.cpp:619
    chrome(B.sphere(-1.1, 2.8, 0.0, 0.5), 0.8, 0.8, 0.8);            // chrome       :490 / :622
    if (nee) {
        glass(B.sphere(-0.2, 1.6, -0.4, 0.1), 0.0, 0.8, 0.0, 1.0, 0.2);       // green   next_event.cpp:626
        glass(B.sphere(0.0, 1.11716, -0.4, 0.1), 0.8, 0.0, 0.0, 1.0, 0.2);    // red     :630
        glass(B.sphere(0.2, 1.6, -0.4, 0.1), 0.0, 0.0, 0.8, 1.0, 0.2);        // blue    :634
        glass(B.sphere(0.0, 1.4, -0.28284, 0.1), 1.0, 1.0, 1.0, 1.5, 0.1);    // pyramid :638
    } else {
        glass(B.sphere(-0.15, 1.55, -0.4, 0.1), 0.0, 0.8, 0.0, 1.0, 0.2);     // green   main.cpp:494
        glass(B.sphere(0.0, 1.187867, -0.4, 0.1), 0.8, 0.0, 0.0, 1.0, 0.2);   // red     :498
        glass(B.sphere(0.15, 1.55, -0.4, 0.1), 0.0, 0.0, 0.8, 1.0, 0.2);      // blue    :502
        glass(B.sphere(0.0, 1.4, -0.28, 0.1), 1.0, 1.0, 1.0, 1.5, 0.1);       // pyramid :506
    }
    diffuse(B.plane(0.0, 3.5, -0.5, 0.0, 0.0, 1.0), 0.9, 0.9, 0.9);   // floor  main.cpp:513 / next_event.cpp:645
    diffuse(B.plane(0.0, 4.5, 0.0, 0.0, -1.0, 0.0), 0.9, 0.9, 0.9);   // back   :517 / :649
    diffuse(B.plane(-1.9, 0.0, 0.0, 1.0, 0.0, 0.0), 0.9, 0.5, 0.5);   // left   :520 / :652
    diffuse(B.plane(1.9, 0.0, 0.0, -1.0, 0.0, 0.0), 0.5, 0.5, 0.9);   // right  :523 / :655
    emitter(B.plane(0.0, 0.0, 2.5, 0.0, 0.0, -1.0), 1.6, 1.47, 1.29); // top light :527 / :659
    diffuse(B.plane(0.0, -2.5, 0.0, 0.0, 1.0, 0.0), 0.9, 0.9, 0.9);   // front  :533 / :665
    double n[3];
    ref_normalize(1.0, -1.0, -1.0, n);
    if (nee)
        diffuse(B.plane(-1.4, 4.0, 2.0, n[0], n[1], n[2]), 0.0, 0.8, 0.0);  // top-left divider next_event.cpp:669
    else
        diffuse(B.plane(-1.8, 4.4, 2.4, n[0], n[1], n[2]), 0.0, 0.8, 0.0);  // main.cpp:537
    ref_normalize(-1.0, -1.0, -1.0, n);
    diffuse(B.plane(1.8, 4.4, 2.4, n[0], n[1], n[2]), 0.8, 0.0, 0.0);       // top-right divider :594 / :682
}

// raytrace3.cpp:929-985: two balls, six rectangular walls (the "front" wall repeats the back wall's
// plane), two sphere lights and a rectangular ceiling light sharing one light material.
void scene_raytrace3(Builder& B) {
    glass(B.sphere(1.0, 2.0, 0.0, 0.5), 1.0, 1.0, 1.0, 1.5, 0.1);  // :931-934
    chrome(B.sphere(-1.1, 2.8, 0.0, 0.5), 0.8, 0.8, 0.8);          // :936-939
    diffuse(B.rectangle({-1.9, 4.5, -0.5, 1.9, 4.5, -0.5, 1.9, -2.5, -0.5, -1.9, -2.5, -0.5}), 0.9, 0.9, 0.9);  // floor :941-944
    diffuse(B.rectangle({-1.9, 4.5, 2.5, 1.9, 4.5, 2.5, 1.9, 4.5, -0.5, -1.9, 4.5, -0.5}), 0.9, 0.9, 0.9);      // back  :945-948
    diffuse(B.rectangle({-1.9, -2.5, 2.5, -1.9, 4.5, 2.5, -1.9, 4.5, -0.5, -1.9, -2.5, -0.5}), 0.9, 0.5, 0.5);  // left  :949-952
    diffuse(B.rectangle({1.9, 4.5, 2.5, 1.9, -2.5, 2.5, 1.9, -2.5, -0.5, 1.9, 4.5, -0.5}), 0.5, 0.5, 0.9);      // right :953-956
    diffuse(B.rectangle({1.9, 4.5, 2.5, -1.9, 4.5, 2.5, -1.9, -2.5, 2.5, 1.9, -2.5, 2.5}), 0.9, 0.9, 0.9);      // top   :957-960
    diffuse(B.rectangle({1.9, 4.5, 2.5, -1.9, 4.5, 2.5, -1.9, 4.5, -0.5, 1.9, 4.5, -0.5}), 0.9, 0.9, 0.9);      // front :961-964
    auto light = [](pt_body* b) {  // Material(V3(0.2, 0.187, 0.169), V3(2.0, 1.87, 1.69)) :966
        if (!b) return;
        b->material = PT_DIFFUSE;
        Builder::set3(b->color, 0.2, 0.187, 0.169);
        Builder::set3(b->emission, 2.0, 1.87, 1.69);
    };
    light(B.sphere(-0.63333, 4.5, 2.1, 0.1));  // :968-970
    light(B.sphere(0.63333, 4.5, 2.1, 0.1));   // :972-974
    light(B.rectangle({1.4, 3.5, 2.5, -1.4, 3.5, 2.5, -1.4, -2.5, 2.5, 1.4, -2.5, 2.5}));  // top light :976-979
}

// light_tracing.cpp:1237-1415, array order :1408-1413: six balls, the room as six Moller-Trumbore
// rectangles, a rectangular ceiling light, 18 ball lights sharing one material (colour = emission,
// :1337) and the "screen" rectangle in front of the camera.
void scene_light_room32(Builder& B) {
    auto mat3 = [](pt_body* b, double cr, double cg, double cb, double er, double eg, double eb) {  // Material(color, emission, power)
        if (!b) return;
        b->material = PT_DIFFUSE;
        Builder::set3(b->color, cr, cg, cb);
        Builder::set3(b->emission, er, eg, eb);
    };
    glass(B.sphere(1.0, 2.0, 0.0, 0.5), 1.0, 1.0, 1.0, 1.5, 0.1);       // glass   :1251-1253
    chrome(B.sphere(-1.1, 2.8, 0.0, 0.5), 0.8, 0.8, 0.8);               // chrome  :1254-1256
    glass(B.sphere(-0.15, 1.55, -0.4, 0.1), 0.0, 0.8, 0.0, 1.0, 0.2);   // green   :1258-1260
    glass(B.sphere(0.0, 1.187867, -0.4, 0.1), 0.8, 0.0, 0.0, 1.0, 0.2); // red     :1262-1264
    glass(B.sphere(0.15, 1.55, -0.4, 0.1), 0.0, 0.0, 0.8, 1.0, 0.2);    // blue    :1266-1268
    glass(B.sphere(0.0, 1.4, -0.28, 0.1), 1.0, 1.0, 1.0, 1.5, 0.1);     // pyramid :1270-1272
    diffuse(B.rectangle_mt({-1.9, 4.5, -0.5, 1.9, 4.5, -0.5, 1.9, -2.5, -0.5, -1.9, -2.5, -0.5}), 0.9, 0.9, 0.9);  // floor :1274-1276
    diffuse(B.rectangle_mt({-1.9, 4.5, 2.5, 1.9, 4.5, 2.5, 1.9, 4.5, -0.5, -1.9, 4.5, -0.5}), 0.9, 0.9, 0.9);      // back  :1277-1279
    diffuse(B.rectangle_mt({-1.9, -2.5, 2.5, -1.9, 4.5, 2.5, -1.9, 4.5, -0.5, -1.9, -2.5, -0.5}), 0.9, 0.5, 0.5);  // left  :1280-1282
    diffuse(B.rectangle_mt({1.9, 4.5, 2.5, 1.9, -2.5, 2.5, 1.9, -2.5, -0.5, 1.9, 4.5, -0.5}), 0.5, 0.5, 0.9);      // right :1283-1285
    diffuse(B.rectangle_mt({1.9, 4.5, 2.5, -1.9, 4.5, 2.5, -1.9, -2.5, 2.5, 1.9, -2.5, 2.5}), 0.9, 0.9, 0.9);      // top   :1286-1288
    mat3(B.rectangle_mt({1.4, 3.5, 2.4, -1.4, 3.5, 2.4, -1.4, -2.5, 2.4, 1.4, -2.5, 2.4}), 0.0, 0.0, 0.0, 0.3, 0.335, 0.245);  // top light :1309-1312
    diffuse(B.rectangle_mt({1.9, -2.5, 2.5, -1.9, -2.5, 2.5, -1.9, -2.5, -0.5, 1.9, -2.5, -0.5}), 0.9, 0.9, 0.9);  // front :1290-1292
    auto ball = [&](double x, double y) { mat3(B.sphere(x, y, 2.1, 0.1), 0.3, 0.47, 0.39, 0.3, 0.47, 0.39); };  // :1337
    ball(1.9, 3.625);    // right  :1348
    ball(-1.9, 3.625);   // left   :1352
    ball(-0.63333, 4.5); // back   :1381
    ball(0.63333, 4.5);  // back2  :1385
    const double ys[6] = {2.75, 1.875, 1.0, 0.125, -0.75, -1.625};  // right2..7 / left2..7 :1356-1379
    for (double y : ys) {
        ball(1.9, y);
        ball(-1.9, y);
    }
    ball(-0.63333, -2.5);  // front  :1388
    ball(0.63333, -2.5);   // front2 :1391
    mat3(B.rectangle_mt({-0.4, 0.1, 0.4, 0.4, 0.1, 0.4, 0.4, 0.1, -0.4, -0.4, 0.1, -0.4}), 0.0, 0.0, 0.0, 0.1, 0.1, 0.1);  // screen :1403-1405
}

// next_event.cpp:605-689, array order :687
void scene_nee15(Builder& B) {
    common_bodies(B, true);
    emitter(B.sphere(1.9, 3.0, 2.1, 0.1), 1.9, 1.77, 1.59);  // right light :673
}

// main.cpp:473-607, array order :600-604
void scene_main32(Builder& B) {
    common_bodies(B, false);
    const double e[3] = {2.0, 1.87, 1.69};
    emitter(B.sphere(1.9, 3.625, 2.1, 0.1), e[0], e[1], e[2]);      // right      :541
    emitter(B.sphere(-1.9, 3.625, 2.1, 0.1), e[0], e[1], e[2]);     // left       :545
    emitter(B.sphere(-0.63333, 4.5, 2.1, 0.1), e[0], e[1], e[2]);   // back       :574
    emitter(B.sphere(0.63333, 4.5, 2.1, 0.1), e[0], e[1], e[2]);    // back2      :578
    const double ys[6] = {2.75, 1.875, 1.0, 0.125, -0.75, -1.625};  // right2..7 / left2..7 :549-572
    for (double y : ys) {
        emitter(B.sphere(1.9, y, 2.1, 0.1), e[0], e[1], e[2]);
        emitter(B.sphere(-1.9, y, 2.1, 0.1), e[0], e[1], e[2]);
    }
    emitter(B.sphere(-0.63333, -2.5, 2.1, 0.1), e[0], e[1], e[2]);  // front      :582
    emitter(B.sphere(0.63333, -2.5, 2.1, 0.1), e[0], e[1], e[2]);   // front2     :586
}

// bidirectional.cpp:629-762, array order :756-761: next_event's bodies with the top-left divider where
// main.cpp has it (:693) and main.cpp's 18 sphere lights in main.cpp's order (:697-745).
void scene_bidir32(Builder& B) {
    common_bodies(B, true);
    if (B.overflow) return;
    Builder::set3(B.out[12].center, -1.8, 4.4, 2.4);  // top-left divider bidirectional.cpp:693
    const double e[3] = {2.0, 1.87, 1.69};
    emitter(B.sphere(1.9, 3.625, 2.1, 0.1), e[0], e[1], e[2]);      // right  :697
    emitter(B.sphere(-1.9, 3.625, 2.1, 0.1), e[0], e[1], e[2]);     // left   :701
    emitter(B.sphere(-0.63333, 4.5, 2.1, 0.1), e[0], e[1], e[2]);   // back   :730
    emitter(B.sphere(0.63333, 4.5, 2.1, 0.1), e[0], e[1], e[2]);    // back2  :734
    const double ys[6] = {2.75, 1.875, 1.0, 0.125, -0.75, -1.625};  // right2..7 / left2..7 :705-728
    for (double y : ys) {
        emitter(B.sphere(1.9, y, 2.1, 0.1), e[0], e[1], e[2]);
        emitter(B.sphere(-1.9, y, 2.1, 0.1), e[0], e[1], e[2]);
    }
    emitter(B.sphere(-0.63333, -2.5, 2.1, 0.1), e[0], e[1], e[2]);  // front  :738
    emitter(B.sphere(0.63333, -2.5, 2.1, 0.1), e[0], e[1], e[2]);   // front2 :742
}

// bipath2.cpp:605-725, array order :722-724: the common bodies with next_event's small spheres,
// main's top-left divider position, a dimmer top light that still counts as a light
// (emission.x == 1.0 >= 1.0, next_event.cpp:364) and nine sphere lights — the array lists a tenth,
// left4, but body_count is 23 (:721), so it is never seen: ten lights in all, which exercises the
// light pick of next_event.cpp:390 beyond two lights.
void scene_bipath23(Builder& B) {
    common_bodies(B, true);
    if (B.overflow) return;
    Builder::set3(B.out[10].emission, 1.0, 0.87, 0.69);  // top light   bipath2.cpp:659
    Builder::set3(B.out[12].center, -1.8, 4.4, 2.4);     // top-left divider :669
    const double e[3] = {2.0, 1.87, 1.69};
    emitter(B.sphere(1.9, 3.0, 2.1, 0.1), e[0], e[1], e[2]);        // right  :673
    emitter(B.sphere(-1.9, 3.0, 2.1, 0.1), e[0], e[1], e[2]);       // left   :677
    emitter(B.sphere(-0.63333, 4.5, 2.1, 0.1), e[0], e[1], e[2]);   // back   :705
    emitter(B.sphere(0.63333, 4.5, 2.1, 0.1), e[0], e[1], e[2]);    // back2  :709
    const double ys[3] = {3.75, 2.25, 1.5};                         // right2..4 / left2..4 :681-703
    for (int k = 0; k < 3; ++k) {
        emitter(B.sphere(1.9, ys[k], 2.1, 0.1), e[0], e[1], e[2]);
        if (k < 2) emitter(B.sphere(-1.9, ys[k], 2.1, 0.1), e[0], e[1], e[2]);
    }
}

// 64-bit LCG used only to lay out the synthetic 10k-primitive scene.
struct Lcg {
    uint64_t s;
    double next() {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        return (double)(s >> 11) * (1.0 / 9007199254740992.0);
    }
    double range(double a, double b) { return a + (b - a) * next(); }
};

void scene_synthetic(Builder& B, int extra) {
    scene_nee15(B);
    Lcg g = {20250211ULL};
    for (int i = 0; i < extra; ++i) {
        double r = g.range(0.02, 0.06);
        double cx = g.range(-1.8, 1.8), cy = g.range(0.8, 4.4), cz = g.range(-0.45, 2.3);
        double cr = g.range(0.2, 0.9), cg = g.range(0.2, 0.9), cb = g.range(0.2, 0.9);
        pt_body* b = B.sphere(cx, cy, cz, r);
        switch (i % 3) {
            case 0: diffuse(b, cr, cg, cb); break;
            case 1: glass(b, cr, cg, cb, 1.5, 0.1); break;
            default: chrome(b, cr, cg, cb); break;
        }
    }
}

thread_local std::string g_host_err;

}  // namespace

extern "C" {

// What the Shape constructors do to their arguments: Plane stores normalize(inormal)
// (main.cpp:176-179, with V3::normalize's long double reciprocal, main.cpp:92-102); Sphere stores
// its arguments unchanged (main.cpp:204-208).
int pt_body_construct(pt_body* b) {
    if (!b) return PT_ERR_INVALID;
    if (b->shape == PT_PLANE) {
        double n[3];
        ref_normalize(b->normal[0], b->normal[1], b->normal[2], n);
        memcpy(b->normal, n, sizeof n);
    }
    if (b->shape == PT_RECTANGLE || b->shape == PT_RECTANGLE_MT) {
        // this->normal = p3.sub(p1).cross(p2.sub(p1)).normalize()  raytrace3.cpp:260, cross :128-133;
        // light_tracing.cpp:494, cross :146-151
        const double* c = b->corners;
        const double ax = c[6] - c[0], ay = c[7] - c[1], az = c[8] - c[2];
        const double bx = c[3] - c[0], by = c[4] - c[1], bz = c[5] - c[2];
        double n[3];
        ref_normalize((ay * bz) - (az * by), (az * bx) - (ax * bz), (ax * by) - (ay * bx), n);
        memcpy(b->normal, n, sizeof n);
    }
    return PT_OK;
}

int pt_preset_scene(const char* name, int width, int height, pt_scene_desc* desc, pt_body* bodies, int cap) {
    if (!name || !desc || !bodies) return PT_ERR_INVALID;
    memset(desc, 0, sizeof *desc);
    desc->width = width;
    desc->height = height;
    reference_camera(&desc->camera);
    desc->seed = 1;
    Builder B = {bodies, cap};
    std::string s(name);
    if (s == "nee15") {
        desc->estimator = PT_EST_NEE;
        scene_nee15(B);
    } else if (s == "main32") {
        desc->estimator = PT_EST_PT;
        scene_main32(B);
    } else if (s == "raytrace3") {
        desc->estimator = PT_EST_PT;
        desc->flags = PT_F_RAYTRACE3;
        scene_raytrace3(B);
    } else if (s == "light_room32") {  // main.cpp's estimator on light_tracing.cpp's scene, sampled as ray_iterate does
        desc->estimator = PT_EST_PT;
        desc->flags = PT_F_PIXEL_GRID;
        scene_light_room32(B);
    } else if (s == "pt15") {  // main.cpp's estimator on the 15-body / 2-light scene (bipath.cpp:556-558)
        desc->estimator = PT_EST_PT;
        scene_nee15(B);
    } else if (s == "nee15_lit") {
        // nee15 with light colour 1 instead of the reference's 0, so that the direct term
        // (Le*Lc)*flux of next_event.cpp:462 is live (other programs of the reference light their
        // scenes this way, e.g. raytrace3.cpp:967)
        desc->estimator = PT_EST_NEE;
        scene_nee15(B);
        if (!B.overflow) {
            Builder::set3(bodies[10].color, 1.0, 1.0, 1.0);
            Builder::set3(bodies[14].color, 1.0, 1.0, 1.0);
        }
    } else if (s == "nee15_measured") {
        // nee15 with the glass ball, the chrome ball, the floor and the back wall switched to
        // measured materials (tables 0..3); the two lights get colour 1 so that the direct
        // term next_event.cpp:462 is not multiplied by zero.
        desc->estimator = PT_EST_NEE;
        scene_nee15(B);
        if (!B.overflow) {
            const int ids[4] = {0, 1, 6, 7};
            for (int k = 0; k < 4; ++k) {
                bodies[ids[k]].material = PT_MEASURED;
                bodies[ids[k]].brdf_id = k;
            }
            Builder::set3(bodies[10].color, 1.0, 1.0, 1.0);
            Builder::set3(bodies[14].color, 1.0, 1.0, 1.0);
        }
    } else if (s == "nee32") {  // next_event's estimator on the 32-body scene (bidirectional.cpp:755-760 lists the same bodies)
        desc->estimator = PT_EST_NEE;
        scene_main32(B);
    } else if (s == "bidir32") {  // bidirectional.cpp as shipped
        desc->estimator = PT_EST_BIDIR;
        scene_bidir32(B);
    } else if (s == "bipath23") {  // next_event's estimator on bipath2.cpp's 23-body scene
        desc->estimator = PT_EST_NEE;
        scene_bipath23(B);
    } else if (s == "synthetic10k") {
        desc->estimator = PT_EST_NEE;
        scene_synthetic(B, 9985);
    } else if (s == "synthetic1k") {
        desc->estimator = PT_EST_NEE;
        scene_synthetic(B, 985);
    } else {
        return PT_ERR_INVALID;
    }
    if (B.overflow) return PT_ERR_NOMEM;
    desc->bodies = bodies;
    desc->body_count = B.n;
    return PT_OK;
}

// The rows main() gives each worker thread: light_tracing.cpp:1441-1451.
int pt_row_split(int height, int workers, int worker, int* start_row, int* row_count) {
    if (height < 0 || workers < 1 || worker < 0 || worker >= workers || !start_row || !row_count) return PT_ERR_INVALID;
    const int row_group_size = height / workers;
    *start_row = worker * row_group_size;
    *row_count = worker == workers - 1 ? height - worker * row_group_size : row_group_size;
    return PT_OK;
}

// save(): main.cpp:408-434 / next_event.cpp:533-554.  int(255.0L * v / (double)samples) is an
// x87 extended-precision expression truncated toward zero; values above 255 are clamped, nothing
// else is.  Output is the reference's ASCII P3 text, byte for byte.
int pt_save_ppm(const double* buffer, int samples, int width, int height, const char* path) {
    if (!buffer || !path || width < 0 || height < 0) return PT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return PT_ERR_IO;
    fprintf(f, "P3\n%d %d\n255\n", width, height);
    const double* px = buffer;
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            int r = (int)((255.0L * px[0]) / (double)samples);
            int g = (int)((255.0L * px[1]) / (double)samples);
            int b = (int)((255.0L * px[2]) / (double)samples);
            if (r > 255) r = 255;
            if (g > 255) g = 255;
            if (b > 255) b = 255;
            fprintf(f, "%d %d %d\n", r, g, b);
            px += 3;
        }
    }
    if (fclose(f) != 0) return PT_ERR_IO;
    return PT_OK;
}

}  // extern "C"
