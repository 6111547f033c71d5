// Drop-in executables for four of the reference's programs:
//   main        <width> <height> <iterations> <thread count>   (main.cpp:460-471, 642-656)
//   next_event  <width> <height> <iterations> <thread count>   (next_event.cpp:587-598, 701-712)
//   bidirectional <width> <height> <iterations> <thread count> (bidirectional.cpp:611-623, 776-785:
//               next_event's command line, worker split, save() and file name; its own scene and estimator)
//   raytrace3   <width> <height> <iterations> <thread count>   (raytrace3.cpp:906-918, 1021-1036;
//               main's command line, worker split and save(), its own scene and hit threshold)
// Same positional arguments, same hard-coded scenes, same output file names and PPM bytes; the
// per-pixel estimator runs on the GPU through the ptgpu C ABI.  Built four times from this file
// (-DPT_PROG_NEE selects next_event, -DPT_PROG_NEE -DPT_PROG_BIDIR bidirectional, -DPT_PROG_RT3 raytrace3).
//
// <thread count> keeps the reference's arithmetic: each of the T workers renders
// iterations/T full frames into a private buffer and the buffers are summed (main.cpp:436-458);
// here worker k renders iterations [k*n, (k+1)*n) and the sum runs in worker order.  save()
// still divides by the requested total.
//
// Extra options, after the four positionals: --seed N (default: time, like srand(time(NULL))),
// --gpu ID, --gpus N (render every worker's iterations on N GPUs of this node: interleaved
// destination tiles per GPU and one RCCL gather, through libptnode.so, which is loaded only then;
// the frame is the same bit for bit), --out PATH, --literal-height (next_event only: height = argv[1] as
// next_event.cpp:596 literally does), --quiet, and --progressive: the output loop the reference
// keeps commented out (main.cpp:614-640): rounds of min(10, max(remaining / T, 1)) iterations per
// worker, the image saved after every round with the iterations done so far as sample count
// (main: output_<done>.ppm, next_event: output.ppm rewritten), "Iterations between output: n" on
// stdout.
#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/ptgpu.h"
#include "../../include/ptgpu_node.h"

namespace {

// The node library sits next to libptgpu.so, one directory above this executable.
struct NodeApi {
    void* so = nullptr;
    decltype(&pt_node_create) create = nullptr;
    decltype(&pt_node_destroy) destroy = nullptr;
    decltype(&pt_node_render) render = nullptr;
    decltype(&pt_node_last_error) last_error = nullptr;
    bool load() {
        char exe[4096];
        ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
        if (n <= 0) return false;
        exe[n] = 0;
        std::string dir(exe);
        dir = dir.substr(0, dir.find_last_of('/'));  // .../bin
        dir = dir.substr(0, dir.find_last_of('/'));  // package directory
        so = dlopen((dir + "/libptnode.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!so) {
            fprintf(stderr, "ptgpu: cannot load libptnode.so: %s\n", dlerror());
            return false;
        }
        create = (decltype(create))dlsym(so, "pt_node_create");
        destroy = (decltype(destroy))dlsym(so, "pt_node_destroy");
        render = (decltype(render))dlsym(so, "pt_node_render");
        last_error = (decltype(last_error))dlsym(so, "pt_node_last_error");
        return create && destroy && render && last_error;
    }
};

}  // namespace

int main(int argc, const char* argv[]) {
    if (argc < 5) {
        printf("%s <width> <height> <iterations> <thread count>\n", argv[0]);
        return 0;
    }
    int width = (int)strtol(argv[1], NULL, 10);
    int height = (int)strtol(argv[2], NULL, 10);
    int iterations = (int)strtol(argv[3], NULL, 10);
    int thread_count = (int)strtol(argv[4], NULL, 10);
    uint64_t seed = (uint64_t)time(NULL);
    int gpu = 0, gpus = 0;
    bool quiet = false, progressive = false;
    std::string out;
    for (int i = 5; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--seed" && i + 1 < argc) seed = strtoull(argv[++i], NULL, 10);
        else if (a == "--gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--quiet") quiet = true;
        else if (a == "--progressive") progressive = true;
#if defined(PT_PROG_NEE) && !defined(PT_PROG_BIDIR)
        else if (a == "--literal-height") height = width;
#endif
        else {
            fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        }
    }
    if (thread_count < 1) thread_count = 1;
    const bool user_out = !out.empty();
    (void)user_out;
#ifdef PT_PROG_NEE
#ifdef PT_PROG_BIDIR
    const char* preset = "bidir32";
#else
    const char* preset = "nee15";
#endif
    int per_worker = iterations / thread_count;                          // next_event.cpp:703, bidirectional.cpp:778
    if (out.empty()) out = "output.ppm";                                 // next_event.cpp:535
#else
#ifdef PT_PROG_RT3
    const char* preset = "raytrace3";
#else
    const char* preset = "main32";
#endif
    int per_worker = (int)((double)iterations / (double)thread_count);   // main.cpp:644 (double -> int parameter)
    if (out.empty()) out = "output_" + std::to_string(iterations) + ".ppm";  // main.cpp:411-412
#endif
    std::vector<pt_body> bodies(64);
    pt_scene_desc desc;
    if (pt_preset_scene(preset, width, height, &desc, bodies.data(), (int)bodies.size()) != PT_OK) {
        fprintf(stderr, "cannot build scene\n");
        return 1;
    }
    desc.seed = seed;
    if (pt_set_device(gpu) != PT_OK) {
        fprintf(stderr, "ptgpu: %s\n", pt_last_error());
        return 1;
    }
    pt_scene* scene = NULL;
    pt_node* node = NULL;
    NodeApi api;
    if (gpus >= 1) {
        if (!api.load()) return 1;
        std::vector<int> devices(gpus);
        for (int d = 0; d < gpus; ++d) devices[d] = gpu + d;
        if (api.create(&desc, devices.data(), gpus, &node) != PT_OK) {
            fprintf(stderr, "ptgpu: %s\n", api.last_error());
            return 1;
        }
    } else if (pt_scene_create(&desc, &scene) != PT_OK) {
        fprintf(stderr, "ptgpu: %s\n", pt_last_error());
        return 1;
    }
    // one render call of either form: adds iterations [first, first + count) into dst
    auto render = [&](int first, int count, double* dst) -> bool {
        if (node) {
            if (api.render(node, first, count, dst) == PT_OK) return true;
            fprintf(stderr, "ptgpu: %s\n", api.last_error());
            return false;
        }
        if (pt_render(scene, first, count, dst) == PT_OK) return true;
        fprintf(stderr, "ptgpu: %s\n", pt_last_error());
        return false;
    };
    size_t n = (size_t)width * height * 3;
    std::vector<double> buffer(n, 0.0), priv;
    auto t0 = std::chrono::steady_clock::now();
    double frames = 0.0;
    if (progressive) {
        // main.cpp:614-640: every worker renders `iteration_step` frames into a private buffer that
        // is then added to the shared one; the image is saved after each round.
        int iterations_remaining = iterations, next_iteration = 0;
        while (iterations_remaining > 0) {
            int iteration_step = (int)((double)iterations_remaining / (double)thread_count);
            if (iteration_step < 1) iteration_step = 1;
            if (iteration_step > 10) iteration_step = 10;
            printf("Iterations between output: %d\n", iteration_step);
            for (int k = 0; k < thread_count; ++k) {
                priv.assign(n, 0.0);
                if (!render(next_iteration, iteration_step, priv.data())) return 1;
                for (size_t i = 0; i < n; ++i) buffer[i] += priv[i];
                next_iteration += iteration_step;
                iterations_remaining -= iteration_step;
                frames += iteration_step;
            }
            const int done = iterations - iterations_remaining;
#ifdef PT_PROG_NEE
            std::string name = out;
#else
            std::string name = user_out ? out + "." + std::to_string(done) : "output_" + std::to_string(done) + ".ppm";
            printf("saving to file \"%s\"\n", name.c_str());
#endif
            if (pt_save_ppm(buffer.data(), done, width, height, name.c_str()) != PT_OK) {
                fprintf(stderr, "cannot write %s\n", name.c_str());
                return 1;
            }
        }
    } else {
        for (int k = 0; k < thread_count; ++k) {
            double* dst = buffer.data();
            if (thread_count > 1) {
                priv.assign(n, 0.0);
                dst = priv.data();
            }
            if (!render(k * per_worker, per_worker, dst)) return 1;
            if (thread_count > 1)
                for (size_t i = 0; i < n; ++i) buffer[i] += priv[i];
        }
        frames = (double)per_worker * thread_count;
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!quiet) {
        double paths = (double)width * height * frames;
        fprintf(stderr, "ptgpu: %.0f paths in %.3f s = %.3e paths/s (seed %llu)\n", paths, secs, paths / secs,
                (unsigned long long)seed);
    }
#ifndef PT_PROG_NEE
    printf("saving to file \"%s\"\n", out.c_str());  // main.cpp:413
#endif
    int rc = pt_save_ppm(buffer.data(), iterations, width, height, out.c_str());
    if (node) api.destroy(node);
    pt_scene_destroy(scene);
    if (rc != PT_OK) {
        fprintf(stderr, "cannot write %s\n", out.c_str());
        return 1;
    }
    return 0;
}
