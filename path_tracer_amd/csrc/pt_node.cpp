// ptnode — one process, several GPUs of a node: the reference's worker threads (main.cpp:436-458,
// 642-654: N threads, each with a private frame, summed under a mutex) as N devices, each rendering
// its interleaved 32x8 destination tiles of the same iterations, and one RCCL gather of the owned
// tiles to the first device.  See include/ptgpu_node.h.  Host code only: every kernel it runs is
// launched through the ptgpu C ABI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

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
    delete nd;
    return PT_OK;
}

int pt_node_create(const pt_scene_desc* desc, const int* devices, int n_devices, pt_node** out) {
    if (!desc || !out || n_devices < 1) return fail(PT_ERR_INVALID, "pt_node_create: bad argument");
    int available = 0;
    NODE_HIP(hipGetDeviceCount(&available));
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
    auto bail = [&](int rc) {
        const std::string keep = g_node_err;
        pt_node_destroy(nd);
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
    // one communicator per device, all in this process
    ncclResult_t ne = ncclCommInitAll(nd->comm.data(), n_devices, nd->device.data());
    if (ne != ncclSuccess) return bail(fail(PT_ERR_DEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(ne)));
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
    // Every device starts from the caller's accumulator and adds its own tiles' samples in
    // iteration order, exactly as a single device would; the other pixels are never read back.
    for (int d = 0; d < nd->n; ++d) {
        NODE_HIP(hipSetDevice(nd->device[d]));
        NODE_HIP(hipMemcpyAsync(nd->accum[d], accum_host, frame_bytes, hipMemcpyHostToDevice, nd->stream[d]));
        NODE_PT(pt_render_device(nd->scene[d], first_iteration, n_iterations, nd->accum[d], nd->stream[d]));
        NODE_PT(pt_tiles_pack(nd->scene[d], nd->accum[d], nd->packed[d], nd->tiles, nd->stream[d]));
    }
    NODE_NCCL(ncclGroupStart());
    for (int d = 0; d < nd->n; ++d) {
        ncclResult_t e = ncclGather(nd->packed[d], nd->gathered, count, ncclDouble, 0, nd->comm[d], nd->stream[d]);
        if (e != ncclSuccess) {
            (void)ncclGroupEnd();
            return fail(PT_ERR_DEVICE, std::string("ncclGather: ") + ncclGetErrorString(e));
        }
    }
    NODE_NCCL(ncclGroupEnd());
    NODE_HIP(hipSetDevice(nd->device[0]));
    NODE_PT(pt_tiles_unpack(nd->scene[0], nd->n, nd->tiles, nd->gathered, nd->accum[0], nd->stream[0]));
    NODE_HIP(hipMemcpyAsync(accum_host, nd->accum[0], frame_bytes, hipMemcpyDeviceToHost, nd->stream[0]));
    for (int d = 0; d < nd->n; ++d) {
        NODE_HIP(hipSetDevice(nd->device[d]));
        NODE_HIP(hipStreamSynchronize(nd->stream[d]));
    }
    return PT_OK;
}

}  // extern "C"
