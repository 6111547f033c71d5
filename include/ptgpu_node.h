/* ptnode — several GPUs of one node driven from one process, on top of the ptgpu C ABI.
 *
 * Replaces the thread glue of the reference programs (worker(), main.cpp:436-458; spawn/join,
 * main.cpp:642-654 and next_event.cpp:701-708): where the reference starts N threads that each
 * render iterations/N whole frames into a private buffer and sums the buffers under a mutex, a
 * node renders the SAME iterations on N devices, device k taking the 32x8-pixel destination tiles
 * whose row-major index is congruent to k modulo N, and gathers the owned tiles onto the first
 * device with one RCCL collective (ncclGather of FP64 tiles over xGMI).  Tiles are disjoint and a
 * pixel's samples are added in iteration order on its one device, so the frame is bit-identical to
 * pt_render() on a single GPU for any N.
 *
 * Lives in libptnode.so (links RCCL); libptgpu.so itself has no RCCL dependency.
 * Returns PT_OK or a negative pt_status; pt_node_last_error() holds the message.
 */
#ifndef PTGPU_NODE_H
#define PTGPU_NODE_H

#include "ptgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_node pt_node; /* opaque: one pt_scene, stream and RCCL communicator per device */

/* `Renderer(Scene)` on each of n_devices GPUs (main.cpp:437: every worker constructs its own
 * Renderer).  devices = HIP device indices, or NULL for 0..n_devices-1. */
int pt_node_create(const pt_scene_desc* desc, const int* devices, int n_devices, pt_node** out);
int pt_node_destroy(pt_node* node);

/* pt_brdf_upload on every device of the node. */
int pt_node_brdf_upload(pt_node* node, int brdf_id, const double* table, size_t n_doubles);

/* pt_render across the node: adds iterations [first_iteration, first_iteration + n_iterations)
 * into accum_host (W*H*3 doubles, the reference's V3* buffer).  Synchronous. */
int pt_node_render(pt_node* node, int first_iteration, int n_iterations, double* accum_host);

const char* pt_node_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_NODE_H */
