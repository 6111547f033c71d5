/* ptgpu — C ABI of the MI355X path-tracing / next-event-estimation estimator.
 *
 * Drop-in boundary for the per-pixel radiance estimator of Melraidin/path_tracer.  The
 * reference has no FFI of its own: its seam is the `Renderer` class inside each program
 * (main.cpp:330-406, next_event.cpp:347-531) plus the process command line.  Each entry point
 * below names the reference interface it replaces.  Plain pointers and sizes only; every
 * function returns PT_OK (0) or a negative pt_status and never calls exit().
 *
 * Memory: `accum` buffers are W*H*3 doubles, row-major, top row first, RGB interleaved —
 * the layout of the reference's `V3* buffer` (main.cpp:333,339,609).  Renders ADD into them,
 * like Renderer::iterate() (main.cpp:362).
 *
 * A handle is not thread-safe; use one handle per host thread and per GPU.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument */
    PT_ERR_DEVICE = -2,    /* HIP call failed */
    PT_ERR_NOMEM = -3,
    PT_ERR_IO = -4,
    PT_ERR_UNSUPPORTED = -5
} pt_status;

enum { PT_SPHERE = 0, PT_PLANE = 1, PT_RECTANGLE = 2,             /* Sphere main.cpp:198, Plane main.cpp:171,
                                                                      Rectangle raytrace3.cpp:244 (PT_EST_PT only) */
       PT_RECTANGLE_MT = 3 };                                      /* Rectangle light_tracing.cpp:477-575: two
                                                                      Moller-Trumbore triangle tests (PT_EST_PT only) */
enum { PT_DIFFUSE = 0, PT_CHROME = 1, PT_GLASS = 2, PT_MEASURED = 3 }; /* Material :230, Chrome :257, Glass :269,
                                                                      MatusikBRDF.h:37 */
enum { PT_EST_PT = 0, PT_EST_NEE = 1,                              /* main.cpp trace() / next_event.cpp iterate() */
       PT_EST_BIDIR = 2 };                                         /* bidirectional.cpp iterate() :374-497, see below */
/* PT_EST_BIDIR: the gather path of next_event.cpp, a light path of up to five vertices started at the
 * chosen light (bidirectional.cpp:398-406), and one connection per (gather vertex, light-path vertex)
 * pair, deepest first in both (:410-488).  The reference scales a connection's direct term by
 * flux_lookup[shoot_hits.size()][shoot_i], a double[5][4] (:377-382, :484): a light path of five
 * vertices, which is every light path of a closed scene, reads past the end of that array.  The value
 * only multiplies emission*colour of the chosen light, which is (0,0,0) for every light of the
 * reference's scene, and then any finite value gives the same pixel.  pt_scene_create therefore accepts
 * PT_EST_BIDIR only for scenes in which that holds for every light (and no body is measured, every
 * glass has ior >= 1, all numbers are finite and moderate): PT_ERR_UNSUPPORTED otherwise.  Fused
 * pipeline only (at most 127 bodies), spheres and planes only. */

/* Scene flags */
#define PT_F_LIGHT_TRUE_CENTER 1u /* NEE shadow rays start at the light's real centre.  Default (flag
                                     clear) reproduces next_event.cpp:167,406-407, where
                                     light->shape->center reads the never-assigned base member (0,0,0). */

#define PT_F_RAYTRACE3 2u /* PT_EST_PT as raytrace3.cpp runs it (shoot_trace = false, raytrace3.cpp:555): a hit
                             needs t > 0.0000001L (:776) instead of t > 0, and a level returns
                             color * (trace(next) + V3()) + emission (:805,840). */

#define PT_F_PIXEL_GRID 4u /* Samples sit on the pixel grid, as Renderer::ray_iterate takes them
                              (light_tracing.cpp:774-790): pixel (col, row) looks through
                              getRay((double)col / (double)width, (double)row / (double)height), every row is
                              full and no jitter is drawn.  Default (flag clear): the jittered running sums of
                              Renderer::iterate (main.cpp:352-357). */

/* One Body (main.cpp:309-318): a Shape plus a Material, by value.
 * `normal` is the Plane constructor ARGUMENT (main.cpp:176-179); the library normalises it the way
 * the constructor does.  `radius` is the Sphere constructor argument (main.cpp:204-208). */
typedef struct pt_body {
    int32_t shape;    /* PT_SPHERE | PT_PLANE */
    int32_t material; /* PT_DIFFUSE | PT_CHROME | PT_GLASS | PT_MEASURED */
    double center[3];
    double normal[3];
    double radius;
    double color[3];
    double emission[3];
    double ior;        /* Glass only */
    double reflection; /* Glass only */
    int32_t brdf_id;   /* PT_MEASURED only: table slot 0..PT_MAX_BRDF-1, else -1 */
    int32_t reserved;
    double corners[12]; /* PT_RECTANGLE, PT_RECTANGLE_MT: the constructor arguments p1, p2, p3, p4
                           (raytrace3.cpp:254, light_tracing.cpp:484); `normal` is then an output of
                           pt_body_construct, center and radius are unused */
} pt_body;

/* Camera(origin, topleft, topright, bottomleft): main.cpp:144-152. */
typedef struct pt_camera {
    double origin[3];
    double topleft[3];
    double topright[3];
    double bottomleft[3];
} pt_camera;

/* Scene (main.cpp:321-327) + what Renderer(Scene) needs. Copied at create time. */
typedef struct pt_scene_desc {
    int32_t width;
    int32_t height;
    pt_camera camera;
    const pt_body* bodies;
    int32_t body_count;
    int32_t estimator; /* PT_EST_PT | PT_EST_NEE | PT_EST_BIDIR */
    uint32_t flags;
    uint32_t reserved;
    uint64_t seed; /* replaces srand(time(NULL)) main.cpp:461 */
} pt_scene_desc;

typedef struct pt_scene pt_scene; /* opaque: Renderer + device state */

#define PT_MAX_BRDF 8
#define PT_BRDF_TABLE_DOUBLES (3 * 90 * 90 * 180) /* MatusikBRDF.h:24-26,102 */
#define PT_MAX_DEPTH 5                            /* trace(): n > 4 returns, main.cpp:368 */

/* Select the HIP device used by handles created afterwards on this thread (default 0). */
int pt_set_device(int device);

/* Replaces `Renderer(Scene)` (main.cpp:336-343; light list next_event.cpp:362-367). */
int pt_scene_create(const pt_scene_desc* desc, pt_scene** out);
int pt_scene_destroy(pt_scene* scene);

/* Apply the Shape constructors to one body in place: a Plane stores normalize(normal)
 * (main.cpp:176-179); a Sphere is unchanged; a Rectangle of either kind stores
 * normalize((p3-p1) x (p2-p1)) in `normal` (raytrace3.cpp:260, light_tracing.cpp:494).
 * pt_scene_create does this to its own copy. Host only. */
int pt_body_construct(pt_body* body);

/* The bodies as stored after construction (plane normals normalised). Returns the count. */
int pt_scene_bodies(const pt_scene* scene, pt_body* out, int cap);

/* Replaces MatusikBRDF::read_brdf (MatusikBRDF.h:84-108): upload one table (3*1458000 doubles,
 * channel-planar R,G,B) into HBM, or read it from a file in the reference's format. */
int pt_brdf_upload(pt_scene* scene, int brdf_id, const double* table, size_t n_doubles);
int pt_brdf_load_file(pt_scene* scene, int brdf_id, const char* path);

/* Multi-GPU sharding: this handle renders only the destination tiles (tile_w x tile_h pixels)
 * whose number % world == rank; other pixels of `accum` are left untouched.  Tiles are numbered
 * along the tile rows, every row starting one column further right than the one above: number
 * ty * tiles_x + c is the tile at column (c + ty) % tiles_x of tile row ty (tiles_x = ceil(width /
 * tile_w)), so that a rank's tiles do not line up in columns of the picture when tiles_x is a
 * multiple of world.  Default: rank 0 of 1.  The reference's counterpart is worker()'s per-thread split
 * (main.cpp:436-458, 642-645). */
int pt_set_partition(pt_scene* scene, int rank, int world, int tile_w, int tile_h);

/* Row-range sharding, the work split of the light-tracing family: Renderer::ray_iterate(start_row,
 * row_count) fills buffer[start_row * width ...] row by row (light_tracing.cpp:774-790).  After
 * pt_set_rows the handle renders only destination pixels of rows [start_row, start_row + row_count);
 * it combines with pt_set_partition (a pixel must satisfy both).  Default: every row.
 *   pt_row_split: the rows main() hands worker `worker` of `workers` (light_tracing.cpp:1441-1451):
 *   height / workers rows each, the last worker also takes the remainder. */
int pt_set_rows(pt_scene* scene, int start_row, int row_count);
int pt_row_split(int height, int workers, int worker, int* start_row, int* row_count);

/* The exchange step of a multi-GPU render, in place of the mutex-guarded sum of worker()
 * (main.cpp:444-457): destination tiles are disjoint, so the ranks gather each other's owned tiles
 * instead of summing whole frames.  Exchange layout of a partition: its owned tiles in order, a tile
 * as 256 pixels row-major within the tile, a pixel as three doubles; pixels of edge tiles outside
 * the image are zero.
 *   pt_tiles_per_rank: ceil(tile count / world), the tile capacity every rank's packed buffer needs.
 *   pt_tiles_pack:     accum_dev (W*H*3) -> packed_dev (tiles_capacity*768 doubles) for this handle's
 *                      partition; tiles beyond the owned count are zero-filled.
 *   pt_tiles_unpack:   gathered_dev (world * tiles_per_rank * 768 doubles, rank-major: the result of
 *                      an all-gather of the packed buffers) -> every pixel of accum_dev.
 * Both go through the handle's row range as well (pt_set_rows): pixels outside it are packed as zero
 * and left untouched on unpack.
 * Device pointers; the kernels are queued on `stream`. */
int pt_tiles_per_rank(const pt_scene* scene, int world);
int pt_tiles_pack(pt_scene* scene, const double* accum_dev, double* packed_dev, int tiles_capacity, void* stream);
int pt_tiles_unpack(pt_scene* scene, int world, int tiles_per_rank, const double* gathered_dev, double* accum_dev,
                    void* stream);

/* Two device pipelines compute the same estimator with identical results.  FUSED: one kernel per
 * batch, path state in registers/LDS, body list read through the scalar cache (up to 127 bodies).
 * WAVEFRONT: one kernel per stage over compacted queues of live paths, path state in HBM, body
 * list as a bounding hierarchy read through the scalar cache (any body count).  AUTO picks FUSED
 * when it applies.
 * No counterpart in the reference (its trace() is one recursive function, main.cpp:367-405). */
enum { PT_PIPELINE_AUTO = 0, PT_PIPELINE_FUSED = 1, PT_PIPELINE_WAVEFRONT = 2 };
int pt_set_pipeline(pt_scene* scene, int pipeline);

/* Replaces `for (i < iterations) renderer.iterate()` (main.cpp:439-442, 349-365): adds
 * iterations [first_iteration, first_iteration + n_iterations) into accum (host memory).
 * Samples of one pixel are added in iteration order, so the result does not depend on how a
 * render is split into calls. */
int pt_render(pt_scene* scene, int first_iteration, int n_iterations, double* accum_host);

/* Same, with `accum_dev` in device memory and work queued on `stream` (a hipStream_t, or NULL).
 * Returns without synchronising; when the work queued on `stream` has completed, so has the render
 * (the library's two internal side streams are joined back into `stream`).  A scene owns one set of
 * work buffers: queue successive renders of one scene on the same stream, or synchronise between
 * them; renders of different scenes are independent. */
int pt_render_device(pt_scene* scene, int first_iteration, int n_iterations, double* accum_dev, void* stream);

/* pt_render_device for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same
 * arguments and the same bits in accum_dev as pt_render_device; the frame tables and the batch's
 * samples live in a fixed device array of PT_NOALLOC_SCRATCH_BYTES that the library holds on every
 * GPU it runs on (its address is resolved once per device, by the first such call there), every
 * kernel is queued on `stream`, and the call makes no hipMalloc/hipFree and creates no stream or
 * event.  Limits: the fused pipeline only (at most 127 bodies, PT_ERR_UNSUPPORTED otherwise), and
 * one iteration of the partition's tables and samples must fit the scratch (about 26 bytes per
 * pixel: a 2 560 000-pixel frame; larger frames return PT_ERR_UNSUPPORTED).  All calls on one GPU
 * share that GPU's array, from whatever scene: two such calls that overlap in time on one GPU race;
 * queue them on one stream or synchronise between them.  pt_get_stats does not cover these calls. */
#define PT_NOALLOC_SCRATCH_BYTES (64u << 20)
int pt_render_device_noalloc(pt_scene* scene, int first_iteration, int n_iterations, double* accum_dev, void* stream);

/* Nearest-hit scan (main.cpp:372-383) over a ray batch: rays6[n][6] = origin, direction.
 * `skip` is a body index excluded from the scan (the light in next_event.cpp:415-419) or -1.
 * idx[i] = -1 and t[i] = DBL_MAX on a miss.  Host pointers. */
int pt_intersect_batch(pt_scene* scene, const double* rays6, int n, int skip, int32_t* idx, double* t);

/* Shape::intersect of every body for every ray (main.cpp:181-191, 210-222): t_all[n][body_count]. */
int pt_intersect_all(pt_scene* scene, const double* rays6, int n, double* t_all);

/* Diagnostics for parity tests: trace n explicit samples (iteration, pixel index, x, y) and return
 * radiance rgb[n][3], gather-path length nrec[n], bodies[n][5], offset points[n][5][3], a bit mask
 * of lit vertices visible[n] (NEE) and the number of random draws used[n]. Any output may be NULL. */
int pt_trace_paths(pt_scene* scene, const int32_t* it_idx2, const double* xy, int n, double* rgb, int32_t* nrec,
                   int32_t* bodies, double* points, int32_t* visible, int32_t* used);

/* Diagnostics for the measured-material term: for the same n explicit samples as pt_trace_paths, the
 * table index (MatusikBRDF.h:63-66) of every lit vertex that lies on a PT_MEASURED body, ind[n][5],
 * -1 elsewhere.  Next-event scenes with a measured body only. */
int pt_trace_brdf_indices(pt_scene* scene, const int32_t* it_idx2, const double* xy, int n, int32_t* ind);

/* Frame geometry of one iteration (main.cpp:352-357): ys[H], xs[H*W], row_len[H]; *nrows rows. */
int pt_frame_layout(pt_scene* scene, int iteration, double* ys, double* xs, int32_t* row_len, int32_t* nrows);

/* The random-number contract, evaluated on the device: n draws k (and u = k / 2147483647.0,
 * main.cpp:19-22) of stream (seed, iteration, domain, index). */
int pt_rng_fill(uint64_t seed, uint32_t iteration, uint32_t domain, uint32_t index, int n, int32_t* k, double* u);

/* The double-arithmetic emulation of the reference's x87 extended-precision expressions, evaluated
 * on the host (on_device = 0) or on the GPU.  kind 0: (double)(1.0L / a)  main.cpp:93;
 * 1: (double)(u*2.0L - 1.0L) with u = k/2147483647.0, k = (int)a  main.cpp:108;
 * 2: (double)(a*1.0000001L), 3: (double)(a*0.9999999L)  main.cpp:396,400;
 * 4: Glass theta2 with eta = a, theta1 = b  main.cpp:292. */
int pt_x87_eval(int kind, int on_device, const double* a, const double* b, int n, double* out);

/* MatusikBRDF::lookup_brdf_val (MatusikBRDF.h:50-80) on an uploaded table: angles4[n][4] =
 * theta_in, phi_in, theta_out, phi_out -> rgb[n][3], ind[n] (table index). */
int pt_brdf_lookup(pt_scene* scene, int brdf_id, const double* angles4, int n, double* rgb, int32_t* ind);

/* The lookup as the estimator makes it for a lit vertex of a measured body: vec6[n][6] = incoming
 * direction, outgoing direction; each goes through vector_to_spherical (raytrace3.cpp:725-728) and
 * the four angles through lookup_brdf_val (MatusikBRDF.h:50-80), the composition sketched in
 * raytrace3.cpp:646-676.  The device computes the table index from the vectors without
 * trigonometric calls; it equals the index of pt_brdf_lookup on the angles except within a few ulps
 * of a bin edge.  The scene must contain a PT_MEASURED body. */
int pt_brdf_lookup_vectors(pt_scene* scene, int brdf_id, const double* vec6, int n, double* rgb, int32_t* ind);

/* Replaces save() (main.cpp:408-434 / next_event.cpp:533-554): ASCII P3, byte-compatible. */
int pt_save_ppm(const double* buffer, int samples, int width, int height, const char* path);

/* The reference's hard-coded scenes as data: "nee15" (next_event.cpp:605-689), "main32"
 * (main.cpp:473-607), "pt15" (main.cpp's estimator on the 15-body scene), "nee15_lit" (nee15 with light
 * colour 1 so the direct term is non-zero), "nee15_measured" (nee15_lit with bodies 0,1,6,7 measured),
 * "nee32" (next_event's estimator on the 32-body scene), "bidir32" (bidirectional.cpp's estimator on its own
 * 32-body, 19-light scene, bidirectional.cpp:629-762), "bipath23" (next_event's estimator on the 23-body, 10-light scene of bipath2.cpp:605-725) and
 * "synthetic10k" / "synthetic1k" (nee15 room + 9985 / 985 extra spheres), and "light_room32" (main.cpp's estimator
 * on the 32-body scene of light_tracing.cpp:1237-1415: six balls, eight Moller-Trumbore rectangles, 18 ball
 * lights; its camera; PT_F_PIXEL_GRID).  Fills desc (desc->bodies = bodies). */
int pt_preset_scene(const char* name, int width, int height, pt_scene_desc* desc, pt_body* bodies, int cap);

/* Statistics of the last pt_render* call on this handle. */
typedef struct pt_stats {
    double paths;          /* (pixel, iteration) samples: in-image destination pixels of this handle's partition
                              times iterations */
    double kernel_ms;      /* device time of the estimator kernels (HIP events) */
    double launches;       /* estimator kernel launches */
} pt_stats;
int pt_get_stats(const pt_scene* scene, pt_stats* out);

const char* pt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
