// TEST INFRASTRUCTURE ONLY — compile check of the reference-side binding a maintainer would add:
// the reference's main.cpp is included unmodified (main renamed), and the replacement body of
// worker() (main.cpp:436-458) that calls the C ABI is compiled against the reference's own classes
// and include/ptgpu.h.  Syntax and types only (g++ -fsyntax-only, `make -C oracle binding-check`);
// nothing is linked or run.
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>

#include "shim/ptor_boost_shim.hpp"

#define main ref_main
#include PTOR_REF_MAIN
#undef main

#include "ptgpu.h"

static pt_body to_pt(Body* b) {  // Body = Shape* + Material* (main.cpp:309-318)
    pt_body o = {};
    o.brdf_id = -1;
    if (Plane* p = dynamic_cast<Plane*>(b->shape)) {
        o.shape = PT_PLANE;
        memcpy(o.center, &p->center, 24);
        memcpy(o.normal, &p->normal, 24);
    } else {
        Sphere* s = (Sphere*)b->shape;
        o.shape = PT_SPHERE;
        memcpy(o.center, &s->center, 24);
        o.radius = s->radius;
    }
    if (Glass* g = dynamic_cast<Glass*>(b->material)) {
        o.material = PT_GLASS;
        o.ior = g->ior;
        o.reflection = g->reflection;
    } else {
        o.material = dynamic_cast<Chrome*>(b->material) ? PT_CHROME : PT_DIFFUSE;
    }
    memcpy(o.color, &b->material->color, 24);
    memcpy(o.emission, &b->material->emission, 24);
    return o;
}

void gpu_worker(int worker_num, int iterations, Scene* scene, V3* buffer) {
    std::vector<pt_body> bodies;
    for (int i = 0; i < scene->body_count; i++) bodies.push_back(to_pt(scene->objects[i]));
    pt_scene_desc d = {};
    d.width = scene->width;
    d.height = scene->height;
    static_assert(sizeof(Camera) >= sizeof(pt_camera), "origin, topleft, topright, bottomleft lead the class");
    memcpy(&d.camera, &scene->camera, sizeof(pt_camera));
    d.bodies = bodies.data();
    d.body_count = (int)bodies.size();
    d.estimator = PT_EST_PT;
    d.seed = (uint64_t)time(NULL);
    pt_scene* s;
    if (pt_scene_create(&d, &s) != PT_OK) {
        fprintf(stderr, "%s\n", pt_last_error());
        return;
    }
    std::vector<double> priv((size_t)scene->width * scene->height * 3, 0.0);
    pt_render(s, worker_num * iterations, iterations, priv.data());
    pt_scene_destroy(s);
    boost::mutex::scoped_lock lock(buffer_mutex);
    for (size_t i = 0; i < priv.size() / 3; i++) {
        buffer[i].x += priv[3 * i];
        buffer[i].y += priv[3 * i + 1];
        buffer[i].z += priv[3 * i + 2];
    }
}
static_assert(sizeof(V3) == 3 * sizeof(double), "V3* buffer is the double[W*H*3] accumulator of the ABI");
