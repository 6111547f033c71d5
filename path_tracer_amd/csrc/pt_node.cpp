// ptnode — one process, several GPUs of a node: the reference's worker threads (main.cpp:436-458,
// 642-654: N threads, each with a private frame, summed under a mutex) as N devices, each rendering
// its interleaved 32x8 destination tiles of the same iterations, and one RCCL gather of the owned
// tiles to the first device.  See include/ptgpu_node.h.  Host code only: every kernel it runs is
// launched through the ptgpu C ABI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptgpu.h"
#include "../../include/ptgpu_node.h"

namespace {

thread_local std::string g_node_err;

int fail(int code, const std::string& msg) {
    g_node_err = msg;
    return code;
}

#define NODE_HIP(expr)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NODE_NCCL(expr)                                                                          \
    do {                                                                                         \
        ncclResult_t e_ = (expr);                                                                \
        if (e_ != ncclSuccess) return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(e_)); \
    } while (0)
#define NODE_PT(expr)                                       \
    do {                                                    \
        int rc_ = (expr);                                   \
        if (rc_ != PT_OK) return fail(rc_, pt_last_error()); \
    } while (0)

constexpr int kTileW = 32, kTileH = 8, kTileDoubles = kTileW * kTileH * 3;

}  // namespace

struct pt_node {
    int n = 0;
    int width = 0, height = 0;
    int tiles = 0;  // tiles per device in the exchange layout
    std::vector<int> device;
    std::vector<pt_scene*> scene;
    std::vector<hipStream_t> stream;
    std::vector<double*> accum, packed;
    double* gathered = nullptr;  // on device[0]: n * tiles * 768 doubles
    double* staging = nullptr;   // pinned host copy of the frame: one memcpy from the caller, then DMA to every device
    std::vector<ncclComm_t> comm;
};

extern "C" {

const char* pt_node_last_error(void) { return g_node_err.c_str(); }

int pt_node_destroy(pt_node* nd) {
    if (!nd) return PT_OK;
    for (int d = 0; d < (int)nd->device.size(); ++d) {
        (void)hipSetDevice(nd->device[d]);
        if (d < (int)nd->comm.size() && nd->comm[d]) (void)ncclCommDestroy(nd->comm[d]);
        if (d < (int)nd->scene.size() && nd->scene[d]) (void)pt_scene_destroy(nd->scene[d]);
        if (d < (int)nd->accum.size() && nd->accum[d]) (void)hipFree(nd->accum[d]);
        if (d < (int)nd->packed.size() && nd->packed[d]) (void)hipFree(nd->packed[d]);
        if (d == 0 && nd->gathered) (void)hipFree(nd->gathered);
        if (d < (int)nd->stream.size() && nd->stream[d]) (void)hipStreamDestroy(nd->stream[d]);
    }
    if (nd->staging) (void)hipHostFree(nd->staging);
    delete nd;
    return PT_OK;
}

int pt_node_create(const pt_scene_desc* desc, const int* devices, int n_devices, pt_node** out) {
    if (!desc || !out || n_devices < 1) return fail(PT_ERR_INVALID, "pt_node_create: bad argument");
    int available = 0;
    NODE_HIP(hipGetDeviceCount(&available));
    int entry_device = 0;  // the caller's current device, restored before returning
    NODE_HIP(hipGetDevice(&entry_device));
    pt_node* nd = new (std::nothrow) pt_node();
    if (!nd) return fail(PT_ERR_NOMEM, "pt_node_create: out of host memory");
    nd->n = n_devices;
    nd->width = desc->width;
    nd->height = desc->height;
    for (int d = 0; d < n_devices; ++d) {
        const int id = devices ? devices[d] : d;
        if (id < 0 || id >= available) {
            delete nd;
            return fail(PT_ERR_INVALID, "pt_node_create: device " + std::to_string(id) + " of " + std::to_string(available) +
                                            " visible does not exist");
        }
        for (int e = 0; e < d; ++e)
            if (nd->device[e] == id) {
                delete nd;
                return fail(PT_ERR_INVALID, "pt_node_create: device listed twice");
            }
        nd->device.push_back(id);
    }
    nd->scene.assign(n_devices, nullptr);
    nd->stream.assign(n_devices, nullptr);
    nd->accum.assign(n_devices, nullptr);
    nd->packed.assign(n_devices, nullptr);
    nd->comm.assign(n_devices, nullptr);
    auto restore = [&]() {
        (void)pt_set_device(entry_device);  // also the library's thread-local device for later pt_scene_create calls
        (void)hipSetDevice(entry_device);
    };
    auto bail = [&](int rc) {
        const std::string keep = g_node_err;
        pt_node_destroy(nd);
        restore();
        g_node_err = keep;
        return rc;
    };
    const size_t frame_bytes = (size_t)desc->width * desc->height * 3 * sizeof(double);
    for (int d = 0; d < n_devices; ++d) {
        int rc = pt_set_device(nd->device[d]);
        if (rc == PT_OK) rc = pt_scene_create(desc, &nd->scene[d]);
        if (rc == PT_OK) rc = pt_set_partition(nd->scene[d], d, n_devices, kTileW, kTileH);
        if (rc != PT_OK) return bail(fail(rc, pt_last_error()));
        if (d == 0) nd->tiles = pt_tiles_per_rank(nd->scene[0], n_devices);
        hipError_t e = hipStreamCreateWithFlags(&nd->stream[d], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void**)&nd->accum[d], frame_bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&nd->packed[d], (size_t)nd->tiles * kTileDoubles * sizeof(double));
        if (e == hipSuccess && d == 0)
            e = hipMalloc((void**)&nd->gathered, (size_t)n_devices * nd->tiles * kTileDoubles * sizeof(double));
        if (e != hipSuccess) return bail(fail(PT_ERR_NOMEM, std::string("pt_node_create: ") + hipGetErrorString(e)));
    }
    hipError_t he = hipHostMalloc((void**)&nd->staging, frame_bytes, hipHostMallocPortable);
    if (he != hipSuccess) return bail(fail(PT_ERR_NOMEM, std::string("pt_node_create: hipHostMalloc: ") + hipGetErrorString(he)));
    // one communicator per device, all in this process
    ncclResult_t ne = ncclCommInitAll(nd->comm.data(), n_devices, nd->device.data());
    if (ne != ncclSuccess) return bail(fail(PT_ERR_DEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(ne)));
    restore();
    *out = nd;
    return PT_OK;
}

int pt_node_brdf_upload(pt_node* nd, int brdf_id, const double* table, size_t n_doubles) {
    if (!nd) return fail(PT_ERR_INVALID, "pt_node_brdf_upload: null node");
    for (int d = 0; d < nd->n; ++d) NODE_PT(pt_brdf_upload(nd->scene[d], brdf_id, table, n_doubles));
    return PT_OK;
}

int pt_node_render(pt_node* nd, int first_iteration, int n_iterations, double* accum_host) {
    if (!nd || !accum_host) return fail(PT_ERR_INVALID, "pt_node_render: null argument");
    const size_t frame_bytes = (size_t)nd->width * nd->height * 3 * sizeof(double);
    const size_t count = (size_t)nd->tiles * kTileDoubles;
    int entry_device = 0;
    NODE_HIP(hipGetDevice(&entry_device));
    // Whatever happens, nothing of this call is left in flight when it returns: the caller owns
    // accum_host again, and the next render starts from settled accumulators.
    auto finish = [&](int rc) {
        std::string keep = g_node_err;
        for (int d = 0; d < nd->n; ++d) {
            hipError_t e = hipSetDevice(nd->device[d]);
            if (e == hipSuccess) e = hipStreamSynchronize(nd->stream[d]);
            if (e != hipSuccess && rc == PT_OK) {  // the first failure is the one reported
                rc = PT_ERR_DEVICE;
                keep = std::string("pt_node_render: device ") + std::to_string(nd->device[d]) + ": " + hipGetErrorString(e);
            }
        }
        (void)hipSetDevice(entry_device);
        g_node_err = keep;
        return rc;
    };
#define NODE_TRY_HIP(expr)                                                                                         \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return finish(fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_))); \
    } while (0)
#define NODE_TRY_PT(expr)                                             \
    do {                                                              \
        int rc_ = (expr);                                             \
        if (rc_ != PT_OK) return finish(fail(rc_, pt_last_error())); \
    } while (0)
    // Every device starts from the caller's accumulator and adds its own tiles' samples in
    // iteration order, exactly as a single device would; the other pixels are never read back.
    memcpy(nd->staging, accum_host, frame_bytes);
    for (int d = 0; d < nd->n; ++d) {
        NODE_TRY_HIP(hipSetDevice(nd->device[d]));
        NODE_TRY_HIP(hipMemcpyAsync(nd->accum[d], nd->staging, frame_bytes, hipMemcpyHostToDevice, nd->stream[d]));
        NODE_TRY_PT(pt_render_device(nd->scene[d], first_iteration, n_iterations, nd->accum[d], nd->stream[d]));
        NODE_TRY_PT(pt_tiles_pack(nd->scene[d], nd->accum[d], nd->packed[d], nd->tiles, nd->stream[d]));
    }
    ncclResult_t ne = ncclGroupStart();
    for (int d = 0; d < nd->n && ne == ncclSuccess; ++d)
        ne = ncclGather(nd->packed[d], nd->gathered, count, ncclDouble, 0, nd->comm[d], nd->stream[d]);
    const ncclResult_t ge = ncclGroupEnd();  // always closes the group that was opened
    if (ne == ncclSuccess) ne = ge;
    if (ne != ncclSuccess) return finish(fail(PT_ERR_DEVICE, std::string("ncclGather: ") + ncclGetErrorString(ne)));
    NODE_TRY_HIP(hipSetDevice(nd->device[0]));
    NODE_TRY_PT(pt_tiles_unpack(nd->scene[0], nd->n, nd->tiles, nd->gathered, nd->accum[0], nd->stream[0]));
    NODE_TRY_HIP(hipMemcpyAsync(nd->staging, nd->accum[0], frame_bytes, hipMemcpyDeviceToHost, nd->stream[0]));
    const int rc = finish(PT_OK);
    if (rc == PT_OK) memcpy(accum_host, nd->staging, frame_bytes);
    return rc;
#undef NODE_TRY_HIP
#undef NODE_TRY_PT
}

}  // extern "C"
