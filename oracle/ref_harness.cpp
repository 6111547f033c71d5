// TEST INFRASTRUCTURE ONLY — harness around the reference's own sources.
//
// The reference programs (main.cpp, next_event.cpp, raytrace3.cpp, bidirectional.cpp) need Boost only
// for their thread glue: four #include lines, one global mutex, the lock in worker() and the
// thread spawn / join / delete statements of main().  The image has no Boost, and nothing stands in
// for it: oracle/Makefile writes, at build time and outside the repository's history, a copy of each
// program from which exactly the lines that mention `boost` or `threads` are left out
// (grep -v -E 'boost|threads'), cut in two at `int main(`.  Every remaining line is the reference's,
// in the reference's order; what is left out is the threading, which this build replaces anyway.
// The two parts are textually included here into one namespace per program, with `main` renamed.
// Between them a macro turns the one call of save() inside main() into ptor::scene_hook(&scene):
// that is how the scene each program builds on the stack of its main()
// (/root/reference/main.cpp:473-607, next_event.cpp:605-689, ...) is borrowed, so scene constants are
// never restated here.  All randomness comes from the linker-wrapped rand() in ref_rt.cpp.
//
// Build recipe: oracle/Makefile (outputs only under oracle/_ref/).
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <functional>
#include <queue>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "ptor_rt.h"
#include "ptor_types.h"

#ifndef PTOR_REF_MAIN_A
#error "define PTOR_REF_<MAIN|NEE|RT3|BIDIR>_<A|B> (the two parts of each program, see oracle/Makefile) and PTOR_REF_BRDF"
#endif

// inside main() only: save(buffer, iterations, width, height) -> the harness's scene hook
#define PTOR_SAVE_AS_HOOK(b, s, w, h) ptor::scene_hook((void*)&scene)

namespace ref_pt {
#include PTOR_REF_MAIN_A
#define save PTOR_SAVE_AS_HOOK
#define main ref_main
#include PTOR_REF_MAIN_B
#undef main
#undef save
}  // namespace ref_pt

namespace ref_nee {
#include PTOR_REF_NEE_A
#define save PTOR_SAVE_AS_HOOK
#define main ref_main
#include PTOR_REF_NEE_B
#undef main
#undef save
}  // namespace ref_nee

// raytrace3.cpp: main.cpp's estimator with a hit threshold of 1e-7 and a third shape, Rectangle
// (raytrace3.cpp:244-330, :761-851).  It includes MatusikBRDF.h itself, hence the guard reset below.
// Built with -fno-access-control: Rectangle's corner points are private members.
namespace ref_rt3 {
#include PTOR_REF_RT3_A
#define save PTOR_SAVE_AS_HOOK
#define main ref_main
#include PTOR_REF_RT3_B
#undef main
#undef save
}  // namespace ref_rt3

// bidirectional.cpp: next_event.cpp's gather path plus a light sub-path and the gather x shoot
// connection loop (bidirectional.cpp:374-497), on its own 32-body scene (:683-762).
namespace ref_bidir {
#include PTOR_REF_BIDIR_A
#define save PTOR_SAVE_AS_HOOK
#define main ref_main
#include PTOR_REF_BIDIR_B
#undef main
#undef save
}  // namespace ref_bidir

// light_tracing.cpp up to `struct Scene`: its vector, ray, shape and material classes, of which the
// Moller-Trumbore Rectangle (light_tracing.cpp:477-575) is the one no other program has.  The rest of
// that file needs kdtree.h, which is absent.
namespace ref_lt {
#include PTOR_REF_LT_HEAD
}  // namespace ref_lt

#undef _MATUSIKBRDFREADER_H_
namespace ref_brdf {
#include PTOR_REF_BRDF
}

namespace {

struct PT {
    enum { prog = 0 };
    typedef ref_pt::V3 V3;
    typedef ref_pt::Ray Ray;
    typedef ref_pt::Scene Scene;
    typedef ref_pt::Renderer Renderer;
    typedef ref_pt::Body Body;
    typedef ref_pt::Shape Shape;
    typedef ref_pt::Plane Plane;
    typedef ref_pt::Sphere Sphere;
    typedef ref_pt::Material Material;
    typedef ref_pt::Chrome Chrome;
    typedef ref_pt::Glass Glass;
    typedef ref_pt::Camera Camera;
    static int run_main(int argc, const char** argv) { return ref_pt::ref_main(argc, argv); }
    static void save(V3* b, int s, int w, int h) { ref_pt::save(b, s, w, h); }
};

struct NEE {
    enum { prog = 1 };
    typedef ref_nee::V3 V3;
    typedef ref_nee::Ray Ray;
    typedef ref_nee::Scene Scene;
    typedef ref_nee::Renderer Renderer;
    typedef ref_nee::Body Body;
    typedef ref_nee::Shape Shape;
    typedef ref_nee::Plane Plane;
    typedef ref_nee::Sphere Sphere;
    typedef ref_nee::Material Material;
    typedef ref_nee::Chrome Chrome;
    typedef ref_nee::Glass Glass;
    typedef ref_nee::Camera Camera;
    static int run_main(int argc, const char** argv) { return ref_nee::ref_main(argc, argv); }
    static void save(V3* b, int s, int w, int h) { ref_nee::save(b, s, w, h); }
};

struct RT3 {
    enum { prog = 2 };
    typedef ref_rt3::V3 V3;
    typedef ref_rt3::Ray Ray;
    typedef ref_rt3::Scene Scene;
    typedef ref_rt3::Renderer Renderer;
    typedef ref_rt3::Body Body;
    typedef ref_rt3::Shape Shape;
    typedef ref_rt3::Plane Plane;
    typedef ref_rt3::Sphere Sphere;
    typedef ref_rt3::Rectangle Rectangle;
    typedef ref_rt3::Material Material;
    typedef ref_rt3::Chrome Chrome;
    typedef ref_rt3::Glass Glass;
    typedef ref_rt3::Camera Camera;
    static int run_main(int argc, const char** argv) { return ref_rt3::ref_main(argc, argv); }
    static void save(V3* b, int s, int w, int h) { ref_rt3::save(b, s, w, h); }
};

struct BIDIR {
    enum { prog = 3 };
    typedef ref_bidir::V3 V3;
    typedef ref_bidir::Ray Ray;
    typedef ref_bidir::Scene Scene;
    typedef ref_bidir::Renderer Renderer;
    typedef ref_bidir::Body Body;
    typedef ref_bidir::Shape Shape;
    typedef ref_bidir::Plane Plane;
    typedef ref_bidir::Sphere Sphere;
    typedef ref_bidir::Material Material;
    typedef ref_bidir::Chrome Chrome;
    typedef ref_bidir::Glass Glass;
    typedef ref_bidir::Camera Camera;
    static int run_main(int argc, const char** argv) { return ref_bidir::ref_main(argc, argv); }
    static void save(V3* b, int s, int w, int h) { ref_bidir::save(b, s, w, h); }
};

// main.cpp / next_event.cpp keep `Body** objects` + body_count, raytrace3.cpp a vector<Body*>
// (and never sets body_count).
template <class B>
int body_count_of(B** const&, int body_count) { return body_count; }
template <class B>
int body_count_of(const std::vector<B*>& v, int) { return (int)v.size(); }
template <class B>
void set_objects(B**& dst, std::vector<B*>& v) { dst = v.data(); }
template <class B>
void set_objects(std::vector<B*>& dst, std::vector<B*>& v) { dst = v; }

// The third shape exists only in raytrace3.cpp.
template <class T>
bool describe_rectangle(typename T::Shape*, ptor_body&) { return false; }
template <>
bool describe_rectangle<RT3>(RT3::Shape* sh, ptor_body& o) {
    RT3::Rectangle* r = dynamic_cast<RT3::Rectangle*>(sh);
    if (!r) return false;
    o.shape = PTOR_RECTANGLE;
    const RT3::V3 p[4] = {r->p1, r->p2, r->p3, r->p4};
    for (int k = 0; k < 4; k++) {
        o.corners[3 * k] = p[k].x;
        o.corners[3 * k + 1] = p[k].y;
        o.corners[3 * k + 2] = p[k].z;
    }
    o.normal[0] = r->normal.x;
    o.normal[1] = r->normal.y;
    o.normal[2] = r->normal.z;
    return true;
}
template <class T>
typename T::Shape* make_rectangle(const ptor_body&) { return nullptr; }
template <>
RT3::Shape* make_rectangle<RT3>(const ptor_body& b) {
    const double* c = b.corners;
    return new RT3::Rectangle(RT3::V3(c[0], c[1], c[2]), RT3::V3(c[3], c[4], c[5]), RT3::V3(c[6], c[7], c[8]),
                              RT3::V3(c[9], c[10], c[11]));
}

std::function<void(void*)> g_session;
std::string g_scratch = "/tmp";

void session_hook(void* scene) {
    if (g_session) g_session(scene);
}

struct CoutSilencer {
    std::streambuf* old;
    std::ostringstream sink;
    CoutSilencer() { old = std::cout.rdbuf(sink.rdbuf()); }
    ~CoutSilencer() { std::cout.rdbuf(old); }
};

struct ChdirScratch {
    char old[4096];
    bool ok;
    ChdirScratch() {
        ok = getcwd(old, sizeof old) != 0;
        if (chdir(g_scratch.c_str()) != 0) ok = false;
    }
    ~ChdirScratch() {
        if (ok && chdir(old) != 0) {}
    }
};

// Run `fn(Scene*)` while the reference's main() of program T is alive with its scene built.
template <class T>
void with_scene(const std::function<void(typename T::Scene*)>& fn) {
    static bool registered = false;
    if (!registered) {
        registered = true;
        typedef typename T::Ray (*getray_fp)(typename T::Camera*, double, double);
        typedef typename T::V3 (*iadd_fp)(typename T::V3*, typename T::V3);
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wpmf-conversions"
        void* a = (void*)(getray_fp)(&T::Camera::getRay);
        void* b = (void*)(iadd_fp)(&T::V3::iadd);
#pragma GCC diagnostic pop
        ptor_rt_register(T::prog, a, b);
    }
    CoutSilencer quiet;
    ChdirScratch cd;
    g_session = [&](void* s) { fn((typename T::Scene*)s); };
    ptor::scene_hook_target = &session_hook;
    const char* argv[] = {"ref", "2", "2", "1", "1"};
    ptor_rt_mode_seq(1, 0);  // main() draws nothing, but keep the wrap defined
    T::run_main(5, argv);
    ptor::scene_hook_target = 0;
    g_session = nullptr;
}

template <class V>
void put3(double* d, const V& v) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}

template <class T>
int scene_bodies(ptor_body* out, int cap, ptor_camera* cam) {
    int count = 0;
    with_scene<T>([&](typename T::Scene* s) {
        count = body_count_of(s->objects, s->body_count);
        put3(cam->origin, s->camera.origin);
        put3(cam->topleft, s->camera.topleft);
        put3(cam->topright, s->camera.topright);
        put3(cam->bottomleft, s->camera.bottomleft);
        for (int i = 0; i < count && i < cap; i++) {
            typename T::Body* b = s->objects[i];
            ptor_body o;
            memset(&o, 0, sizeof o);
            o.brdf_id = -1;
            if (typename T::Plane* p = dynamic_cast<typename T::Plane*>(b->shape)) {
                o.shape = PTOR_PLANE;
                put3(o.center, p->center);
                put3(o.normal, p->normal);
            } else if (typename T::Sphere* sp = dynamic_cast<typename T::Sphere*>(b->shape)) {
                o.shape = PTOR_SPHERE;
                put3(o.center, sp->center);
                o.radius = sp->radius;
            } else if (!describe_rectangle<T>(b->shape, o)) {
                o.shape = -1;
            }
            if (typename T::Glass* gl = dynamic_cast<typename T::Glass*>(b->material)) {
                o.material = PTOR_GLASS;
                o.ior = gl->ior;
                o.reflection = gl->reflection;
            } else if (dynamic_cast<typename T::Chrome*>(b->material)) {
                o.material = PTOR_CHROME;
            } else {
                o.material = PTOR_DIFFUSE;
            }
            put3(o.color, b->material->color);
            put3(o.emission, b->material->emission);
            out[i] = o;
        }
    });
    return count;
}

template <class T>
typename T::Ray make_ray(const double* r) {
    return typename T::Ray(typename T::V3(r[0], r[1], r[2]), typename T::V3(r[3], r[4], r[5]));
}

template <class T>
int intersect_all(const double* rays6, int n, double* t_out) {
    int count = 0;
    with_scene<T>([&](typename T::Scene* s) {
        count = body_count_of(s->objects, s->body_count);
        for (int i = 0; i < n; i++) {
            typename T::Ray ray = make_ray<T>(rays6 + 6 * i);
            for (int b = 0; b < count; b++) t_out[(size_t)i * count + b] = s->objects[b]->shape->intersect(ray);
        }
    });
    return count;
}

template <class T>
void set_mode(int mode, uint64_t seed, const int* script, int script_len) {
    if (mode == PTOR_MODE_SCRIPT)
        ptor_rt_mode_script(script, script_len);
    else if (mode == PTOR_MODE_PIXEL)
        ptor_rt_mode_pixel(seed);
    else
        ptor_rt_mode_seq(seed, 0);
}

template <class T>
long long render(int W, int H, int first_it, int n_it, int mode, uint64_t seed, const int* script,
                 int script_len, double* buffer_out) {
    long long draws = 0;
    with_scene<T>([&](typename T::Scene* s) {
        typename T::Scene sc = *s;
        sc.width = W;
        sc.height = H;
        typename T::Renderer r(sc);
        set_mode<T>(mode, seed, script, script_len);
        for (int it = first_it; it < first_it + n_it; it++) {
            ptor_rt_begin_iteration((uint32_t)it);
            r.iterate();
        }
        draws = ptor_rt_draws();
        for (int i = 0; i < W * H; i++) put3(buffer_out + 3 * (size_t)i, r.buffer[i]);
    });
    return draws;
}

// Frames of the reference's Renderer::iterate() on a caller-supplied body list: the bodies are
// built with the reference's own constructors from their constructor arguments (Plane normalises
// its normal, main.cpp:176-179), replace the scene's object array, and the reference Renderer is
// constructed from that scene (which rebuilds the light list, next_event.cpp:362-367).
template <class T>
long long render_custom(const ptor_body* bodies, int n, int W, int H, int first_it, int n_it, int mode, uint64_t seed,
                        double* buffer_out) {
    long long draws = -1;
    with_scene<T>([&](typename T::Scene* s) {
        typedef typename T::V3 V3;
        std::vector<typename T::Shape*> shapes;
        std::vector<typename T::Material*> mats;
        std::vector<typename T::Body*> objs;
        for (int i = 0; i < n; i++) {
            const ptor_body& b = bodies[i];
            typename T::Shape* sh;
            if (b.shape == PTOR_RECTANGLE)
                sh = make_rectangle<T>(b);
            else if (b.shape == PTOR_PLANE)
                sh = new typename T::Plane(V3(b.center[0], b.center[1], b.center[2]), V3(b.normal[0], b.normal[1], b.normal[2]));
            else
                sh = new typename T::Sphere(V3(b.center[0], b.center[1], b.center[2]), b.radius);
            V3 col(b.color[0], b.color[1], b.color[2]), em(b.emission[0], b.emission[1], b.emission[2]);
            typename T::Material* m;
            if (b.material == PTOR_GLASS) {
                m = new typename T::Glass(col, b.ior, b.reflection);
                m->emission = em;
            } else if (b.material == PTOR_CHROME) {
                m = new typename T::Chrome(col);
                m->emission = em;
            } else {
                m = new typename T::Material(col, em);
            }
            shapes.push_back(sh);
            mats.push_back(m);
            objs.push_back(new typename T::Body(sh, m));
        }
        typename T::Scene sc = *s;
        sc.width = W;
        sc.height = H;
        sc.body_count = n;
        set_objects(sc.objects, objs);
        {
            typename T::Renderer r(sc);
            set_mode<T>(mode, seed, 0, 0);
            for (int it = first_it; it < first_it + n_it; it++) {
                ptor_rt_begin_iteration((uint32_t)it);
                r.iterate();
            }
            draws = ptor_rt_draws();
            for (int i = 0; i < W * H; i++) put3(buffer_out + 3 * (size_t)i, r.buffer[i]);
        }
        for (auto* o : objs) delete o;
        for (auto* m : mats) delete m;
        for (auto* sh : shapes) delete sh;
    });
    return draws;
}

}  // namespace

extern "C" {

// ptor_render on a caller-supplied body list (constructor arguments, see render_custom).
long long ptor_render_custom(int prog, const ptor_body* bodies, int n, int W, int H, int first_it, int n_it, int mode,
                             uint64_t seed, double* buffer_out) {
    if (prog == 2) return render_custom<RT3>(bodies, n, W, H, first_it, n_it, mode, seed, buffer_out);
    for (int i = 0; i < n; i++)
        if (bodies[i].shape == PTOR_RECTANGLE) return -1;  // only raytrace3.cpp has the shape
    if (prog == 3) return render_custom<BIDIR>(bodies, n, W, H, first_it, n_it, mode, seed, buffer_out);
    return prog == 0 ? render_custom<PT>(bodies, n, W, H, first_it, n_it, mode, seed, buffer_out)
                     : render_custom<NEE>(bodies, n, W, H, first_it, n_it, mode, seed, buffer_out);
}

int ptor_set_scratch(const char* dir) {
    g_scratch = dir;
    return 0;
}

// prog: 0 = main.cpp, 1 = next_event.cpp, 2 = raytrace3.cpp, 3 = bidirectional.cpp.  Returns body count.
int ptor_scene_bodies(int prog, ptor_body* out, int cap, ptor_camera* cam) {
    if (prog == 2) return scene_bodies<RT3>(out, cap, cam);
    if (prog == 3) return scene_bodies<BIDIR>(out, cap, cam);
    return prog == 0 ? scene_bodies<PT>(out, cap, cam) : scene_bodies<NEE>(out, cap, cam);
}

// t_out[n][body_count]: Shape::intersect of every body, in array order. Returns body count.
int ptor_intersect_all(int prog, const double* rays6, int n, double* t_out) {
    if (prog == 2) return intersect_all<RT3>(rays6, n, t_out);
    return prog == 0 ? intersect_all<PT>(rays6, n, t_out) : intersect_all<NEE>(rays6, n, t_out);
}

// Nearest hit as decided by the reference's own scan loop (next_event.cpp:496-507): trace()
// entered at depth 4 scans once, records (offset point, body) and returns at depth 5.
// idx[i] = -1 on a miss.  point3 receives the recorded (offset) point.  The bounce it performs
// consumes scripted draws (all RAND_MAX/2).
int ptor_nearest_nee(const double* rays6, int n, int* idx, double* point3) {
    with_scene<NEE>([&](NEE::Scene* s) {
        NEE::Scene sc = *s;
        sc.width = 1;
        sc.height = 1;
        NEE::Renderer r(sc);
        int half[3] = {RAND_MAX / 2, RAND_MAX / 3, RAND_MAX / 5};
        for (int i = 0; i < n; i++) {
            ptor_rt_mode_script(half, 3);
            std::vector<ref_nee::HitRecord> hits;
            r.trace(make_ray<NEE>(rays6 + 6 * i), 4, &hits);
            idx[i] = -1;
            if (!hits.empty()) {
                for (int b = 0; b < sc.body_count; b++)
                    if (sc.objects[b] == hits[0].target) idx[i] = b;
                put3(point3 + 3 * i, hits[0].point);
            }
        }
    });
    return 0;
}

// Camera::getRay (main.cpp:154-157) for n (x,y) pairs -> rays6.
int ptor_get_rays(int prog, const double* xy, int n, double* rays6) {
    auto body = [&](auto* s) {
        for (int i = 0; i < n; i++) {
            auto ray = s->camera.getRay(xy[2 * i], xy[2 * i + 1]);
            put3(rays6 + 6 * i, ray.origin);
            put3(rays6 + 6 * i + 3, ray.direction);
        }
    };
    if (prog == 0)
        with_scene<PT>([&](PT::Scene* s) { body(s); });
    else
        with_scene<NEE>([&](NEE::Scene* s) { body(s); });
    return 0;
}

// material->bounce(ray, normal) of body `body` with scripted draws. Returns draws consumed.
int ptor_bounce(int prog, int body, const double* ray6, const double* normal3, const int* script,
                int script_len, double* out3) {
    long long draws = 0;
    auto run = [&](auto* s, auto tag) {
        typedef decltype(tag) T;
        ptor_rt_mode_script(script, script_len);
        typename T::V3 d = s->objects[body]->material->bounce(
            make_ray<T>(ray6), typename T::V3(normal3[0], normal3[1], normal3[2]));
        put3(out3, d);
        draws = ptor_rt_draws();
    };
    if (prog == 0)
        with_scene<PT>([&](PT::Scene* s) { run(s, PT()); });
    else
        with_scene<NEE>([&](NEE::Scene* s) { run(s, NEE()); });
    return (int)draws;
}

// Full frames through the reference's own Renderer::iterate(). buffer_out[W*H*3] receives the
// Renderer's private buffer after n_it iterations. Returns the number of rand() calls.
long long ptor_render(int prog, int W, int H, int first_it, int n_it, int mode, uint64_t seed,
                      const int* script, int script_len, double* buffer_out) {
    if (prog == 2) return render<RT3>(W, H, first_it, n_it, mode, seed, script, script_len, buffer_out);
    if (prog == 3) return render<BIDIR>(W, H, first_it, n_it, mode, seed, script, script_len, buffer_out);
    return prog == 0 ? render<PT>(W, H, first_it, n_it, mode, seed, script, script_len, buffer_out)
                     : render<NEE>(W, H, first_it, n_it, mode, seed, script, script_len, buffer_out);
}

// Renderer::trace of main.cpp (main.cpp:367-405) for n rays, each with its own PIXEL stream
// (seed, iteration, index i). rgb_out[n][3].
int ptor_trace_pt(const double* rays6, int n, uint64_t seed, int iteration, double* rgb_out) {
    with_scene<PT>([&](PT::Scene* s) {
        PT::Scene sc = *s;
        sc.width = 1;
        sc.height = 1;
        PT::Renderer r(sc);
        for (int i = 0; i < n; i++) {
            ptor_rt_mode_stream(seed, (uint32_t)iteration, 0 /* PIXEL domain */, (uint32_t)i);
            put3(rgb_out + 3 * i, r.trace(make_ray<PT>(rays6 + 6 * i), 0));
        }
    });
    return 0;
}

// The reference's save() (main.cpp:408-434 / next_event.cpp:533-554) on a caller buffer.
// The file is written into the scratch directory under the reference's own name;
// name_out receives that name.
int ptor_save(int prog, const double* buffer, int samples, int W, int H, char* name_out, int cap) {
    CoutSilencer quiet;
    ChdirScratch cd;
    if (prog == 0) {
        std::vector<ref_pt::V3> b(W * H);
        for (int i = 0; i < W * H; i++) b[i] = ref_pt::V3(buffer[3 * i], buffer[3 * i + 1], buffer[3 * i + 2]);
        ref_pt::save(b.data(), samples, W, H);
        snprintf(name_out, cap, "output_%d.ppm", samples);
    } else {
        std::vector<ref_nee::V3> b(W * H);
        for (int i = 0; i < W * H; i++) b[i] = ref_nee::V3(buffer[3 * i], buffer[3 * i + 1], buffer[3 * i + 2]);
        ref_nee::save(b.data(), samples, W, H);
        snprintf(name_out, cap, "output.ppm");
    }
    return 0;
}

// MatusikBRDF::lookup_brdf_val (MatusikBRDF.h:50-80) on a table file in the reference format.
int ptor_brdf_lookup(const char* file, const double* angles4, int n, double* rgb_out) {
    ref_brdf::MatusikBRDF brdf;
    if (!brdf.read_brdf(std::string(file))) return -1;
    for (int i = 0; i < n; i++) {
        const double* a = angles4 + 4 * i;
        brdf.lookup_brdf_val(a[0], a[1], a[2], a[3], rgb_out[3 * i], rgb_out[3 * i + 1], rgb_out[3 * i + 2]);
    }
    return 0;
}

// light_tracing.cpp's Rectangle(p1, p2, p3, p4) (:484-495) built from corners12, its stored normal, and
// Rectangle::intersect (:503-575) for n rays: t_out[n].
int ptor_lt_rectangle(const double* corners12, const double* rays6, int n, double* t_out, double* normal3_out) {
    const double* c = corners12;
    ref_lt::Rectangle r(ref_lt::V3(c[0], c[1], c[2]), ref_lt::V3(c[3], c[4], c[5]), ref_lt::V3(c[6], c[7], c[8]),
                        ref_lt::V3(c[9], c[10], c[11]));
    if (normal3_out) put3(normal3_out, r.normal);
    for (int i = 0; i < n; i++) {
        const double* q = rays6 + 6 * i;
        t_out[i] = r.intersect(ref_lt::Ray(ref_lt::V3(q[0], q[1], q[2]), ref_lt::V3(q[3], q[4], q[5])));
    }
    return 0;
}

// CPU baseline: `threads` independent workers, each with its own Renderer and its own
// sequential rand() stream (what per-thread rand state gives on the author's platform),
// each running `iters_per_thread` full frames of the reference's iterate(). Returns seconds.
double ptor_time_iterate(int prog, int W, int H, int iters_per_thread, int threads, uint64_t seed,
                         double* checksum) {
    double secs = 0.0;
    auto run = [&](auto* s, auto tag) {
        typedef decltype(tag) T;
        typename T::Scene sc = *s;
        sc.width = W;
        sc.height = H;
        std::vector<double> sums(threads, 0.0);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int k = 0; k < threads; k++) {
            pool.emplace_back([&, k]() {
                typename T::Renderer r(sc);
                ptor_rt_mode_seq(seed, (uint32_t)k);
                for (int it = 0; it < iters_per_thread; it++) r.iterate();
                double acc = 0.0;
                for (int i = 0; i < W * H; i++) acc += r.buffer[i].x + r.buffer[i].y + r.buffer[i].z;
                sums[k] = acc;
            });
        }
        for (auto& th : pool) th.join();
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double tot = 0.0;
        for (double v : sums) tot += v;
        if (checksum) *checksum = tot;
    };
    if (prog == 0)
        with_scene<PT>([&](PT::Scene* s) { run(s, PT()); });
    else
        with_scene<NEE>([&](NEE::Scene* s) { run(s, NEE()); });
    return secs;
}

}  // extern "C"
