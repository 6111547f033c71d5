// ptgpu — kernels and C ABI (see include/ptgpu.h).  gfx950 only.
//
// Render pipeline for one batch of B iterations:
//   k_layout_*     : per iteration, the jitter tables of Renderer::iterate (main.cpp:352-357):
//                    y per row, x at every 8th pixel of a row (running sums, exactly as the
//                    reference's loops accumulate them; k_trace continues the sum from the nearest
//                    checkpoint), row lengths and the running-index row starts.
//   k_trace<EST>   : one thread per (destination pixel, iteration): camera ray, <=5 segments of
//                    nearest-hit scan + bounce, then (NEE) light pick, <=5 shadow scans and the
//                    backward colour recurrence.  Writes one FP64 RGB sample.
//   k_accumulate   : per destination pixel, adds the batch's samples in iteration order into
//                    the caller's FP64 buffer (single writer, no atomics, deterministic).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptgpu.h"
#include "pt_device.hpp"
#include "pt_wavefront.hpp"

using namespace ptd;

namespace {

thread_local std::string g_err;
thread_local int g_device = 0;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// V3::normalize on the host as the reference evaluates it (main.cpp:92-102): the reciprocal of the
// length is formed in long double and narrowed before the three multiplications.
void pt_normalize3(double x, double y, double z, double* o) {
    const double len = std::sqrt(x * x + y * y + z * z);
    const double inv = (double)(1.0L / (long double)len);
    o[0] = x * inv;
    o[1] = y * inv;
    o[2] = z * inv;
}

constexpr int kBlock = 256;
// Threads of an estimator block: a whole tile, or one wave of it for the path tracer, whose blocks
// hold no LDS records: a wave that has finished then frees its slot without waiting for the other
// three of its tile (main32 9.73e9 -> 9.94e9 paths/s).  The next-event kernels measured 1-6 % slower
// with 128- or 64-thread blocks and keep the tile.
#ifndef PT_TRACE_BLOCK_PT
#define PT_TRACE_BLOCK_PT 64
#endif
constexpr int trace_block(int est) { return est == 0 /* kEstPT */ ? PT_TRACE_BLOCK_PT : kBlock; }
constexpr int kXcdsLog2 = 3;  // eight accelerator dies; the blocks of a grid go to them in turn
constexpr int kXStride = 8;  // the x running sum of a row is stored at every 8th column

struct Partition {
    int rank, world;
    int tile_w, tile_h;  // tile_w * tile_h == 256, so both are powers of two
    int tile_w_log2;
    int group_log2;  // k_trace: 2^group_log2 consecutive blocks share a tile (set per launch, see launch_trace)
    int tiles_x, tiles_y;
    int owned_tiles;
    int row_lo, row_hi;  // destination rows [row_lo, row_hi) of pt_set_rows (0, height by default)
};

// Destination slot of thread `tid` in owned tile `q`, or -1 outside the image or the row range;
// *row_out (optional) receives the destination row.
// Tile numbers run along the tile rows, and every tile row starts one column further right than the
// row above it (number ty * tiles_x + c sits at column (c + ty) % tiles_x).  Without that shift a frame
// whose tile rows hold a multiple of `world` tiles gives a rank the same columns of every row (at
// 1024x768 over eight ranks: one 32-pixel stripe in every 256), and a rank whose stripes cross the
// expensive part of the picture holds the others up.
PTD_INLINE int slot_of(const Partition& P, int q, int tid, int W, int H, int* row_out = nullptr) {
    int tile = q * P.world + P.rank;
    int ty = tile / P.tiles_x;
    int tx = (tile - ty * P.tiles_x + ty) % P.tiles_x;
    int px = tx * P.tile_w + (tid & (P.tile_w - 1));
    int py = ty * P.tile_h + (tid >> P.tile_w_log2);
    if (row_out) *row_out = py;
    if (px >= W || py >= H || py < P.row_lo || py >= P.row_hi) return -1;
    return py * W + px;
}

// ---- frame layout -------------------------------------------------------------------------
// The jitter tables of a batch of iterations, in three small kernels on one stream.  Their blocks
// are one wave (y and x chains) or four (prefix sums) of few registers and next to no LDS, so that
// they find room on a CU beside the estimator blocks of the neighbouring batch, which leave one
// wave slot per SIMD free.
//   k_layout_y    : thread per iteration: the y chain (y0 = u/H, then y += ystep while y < 0.99999L,
//                   main.cpp:352-354); the number of rows it yields goes to nrows[b].
//   k_layout_rows : thread per (iteration, row): the row's x chain (x0 = u/W, then x += xstep while
//                   x < 0.99999L, main.cpp:355-357), stored at every 8th column, checkpoint-major
//                   (xs[b][c/8][r]: the 64 rows of a wave store 512 contiguous bytes); the row length
//                   goes to row_start[b][r+1].
//   k_layout_scan : block per iteration: row lengths -> running-index row starts, npix[b].
// xstep/ystep are the doubles the reference initialises from 1.0L/(double)W and 1.0L/(double)H.
// GRID (PT_F_PIXEL_GRID, Renderer::ray_iterate light_tracing.cpp:785-790): no jitter, every row full,
// y of row r = (double)r / (double)H; k_trace forms x = (double)c / (double)W itself.
constexpr int kLayoutWave = 64;
constexpr int kScanBlock = 256;

__global__ void __launch_bounds__(kLayoutWave) k_layout_y(uint64_t seed, int H, int first_it, int B, double ystep,
                                                          double* __restrict__ ys, int32_t* __restrict__ nrows, int grid) {
    const int b = blockIdx.x * kLayoutWave + threadIdx.x;
    if (b >= B) return;
    double* ys_b = ys + (size_t)b * H;
    if (grid) {
        for (int r = 0; r < H; ++r) ys_b[r] = (double)r / (double)H;
        nrows[b] = H;
        return;
    }
    Rng j = rng_init(seed, (uint32_t)(first_it + b), kDomFrame, 0);
    double y = k_to_unit(rng_next(j)) / (double)H;
    int r = 0;
    while (y < 0.99999 && r < H) {
        ys_b[r] = y;
        r++;
        y += ystep;
    }
    nrows[b] = r;
}

__global__ void __launch_bounds__(kLayoutWave) k_layout_rows(uint64_t seed, int W, int H, int first_it, double xstep,
                                                             double* __restrict__ xs, const int32_t* __restrict__ nrows,
                                                             int32_t* __restrict__ row_start, int grid) {
    const int r = blockIdx.x * kLayoutWave + threadIdx.x;
    const int b = blockIdx.y;
    if (r >= H) return;
    int len = 0;
    if (grid) {
        len = W;
    } else if (r < nrows[b]) {
        const int CK = (W + kXStride - 1) / kXStride;
        double* col = xs + (size_t)b * CK * H + r;
        Rng j = rng_init(seed, (uint32_t)(first_it + b), kDomRow, (uint32_t)r);
        double x = k_to_unit(rng_next(j)) / (double)W;
        while (x < 0.99999 && len < W) {
            if ((len & (kXStride - 1)) == 0) col[(size_t)(len / kXStride) * H] = x;
            len++;
            x += xstep;
        }
    }
    row_start[(size_t)b * (H + 1) + r + 1] = len;  // the row length, until k_layout_scan has run
}

__global__ void __launch_bounds__(kScanBlock) k_layout_scan(int H, int32_t* __restrict__ row_start, int32_t* __restrict__ npix) {
    __shared__ int s_part[kScanBlock];
    const int b = blockIdx.x, tid = threadIdx.x;
    int32_t* rs_b = row_start + (size_t)b * (H + 1);
    const int chunk = (H + kScanBlock - 1) / kScanBlock;
    const int r0 = min(H, tid * chunk), r1 = min(H, r0 + chunk);
    int local = 0;
    for (int r = r0; r < r1; ++r) local += rs_b[r + 1];
    s_part[tid] = local;
    __syncthreads();
    // inclusive Hillis-Steele scan over the per-thread totals
    for (int off = 1; off < kScanBlock; off <<= 1) {
        int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - local;  // exclusive prefix
    for (int r = r0; r < r1; ++r) {
        run += rs_b[r + 1];
        rs_b[r + 1] = run;
    }
    if (tid == 0) rs_b[0] = 0;
    if (tid == kScanBlock - 1) npix[b] = s_part[tid];
}

// Queues the three layout kernels of iterations [first_it, first_it + B) on `st`.
void launch_frame_layout(hipStream_t st, uint64_t seed, int W, int H, int first_it, int B, double xstep, double ystep, double* xs,
                         double* ys, int32_t* row_start, int32_t* npix, int grid) {
    hipLaunchKernelGGL(k_layout_y, dim3((B + kLayoutWave - 1) / kLayoutWave), dim3(kLayoutWave), 0, st, seed, H, first_it, B, ystep,
                       ys, npix, grid);
    hipLaunchKernelGGL(k_layout_rows, dim3((H + kLayoutWave - 1) / kLayoutWave, B), dim3(kLayoutWave), 0, st, seed, W, H, first_it,
                       xstep, xs, npix, row_start, grid);
    hipLaunchKernelGGL(k_layout_scan, dim3(B), dim3(kScanBlock), 0, st, H, row_start, npix);
}

// ---- k_trace -----------------------------------------------------------------------------
// __launch_bounds__(256, 5): five 256-thread blocks per CU (5 waves per SIMD).  Their LDS records
// (5 x 32 KB) just fit the 160 KiB of a CU and the register file allows 96 VGPRs per lane; the
// estimator is bound by VALU issue, and the fifth wave buys ~8 %.
// COMPACT selects trace_path_nee_compact (NEE scenes whose direct term is dead everywhere): the
// shadow tasks of a wave are dealt to its lanes through 64 two-byte LDS slots per wave.  That form
// keeps four positions per path in LDS (26 368 B per block) and fits 80 VGPRs: six blocks per CU.
// The bidirectional kernel deals its connections through the same kind of slots (trace_path_bidir) and
// keeps all five positions: five blocks.  The path-tracing kernel keeps no positions and fits 64
// VGPRs: eight waves per SIMD, in one-wave blocks (trace_block).
template <int EST, bool MEASURED, bool COMPACT = false>
__global__ void __launch_bounds__(trace_block(EST), EST == kEstPT ? 8 : COMPACT ? 6 : 5) k_trace(DScene S, Partition P, int first_it, int n_it, const double* __restrict__ xs,
                                                  const double* __restrict__ ys, const int32_t* __restrict__ row_start,
                                                  const int32_t* __restrict__ npix, double xstep,
                                                  double* __restrict__ samples, size_t sample_stride) {
    // path tracing keeps no vertex positions; the compact next-event form keeps the deepest vertex's in
    // registers (see trace_path_nee_compact), which lets a sixth block share the CU's LDS
    constexpr int kTraceBlock = trace_block(EST), kTraceParts = kBlock / kTraceBlock;
    __shared__ double s_pts[EST == kEstPT ? 1 : (COMPACT ? kMaxDepth - 1 : kMaxDepth) * 3 * kTraceBlock];
    __shared__ int8_t s_bod[kMaxDepth * kTraceBlock];
    __shared__ uint16_t s_slots[COMPACT || EST == kEstBidir ? kTraceBlock : 1];
    const int tid = threadIdx.x;
    // blockIdx.x = (owned tile << gl) + g, iteration b = (blockIdx.y << gl) + g: see launch_trace
    const int gl = P.group_log2;
    const int b = (blockIdx.y << gl) + (blockIdx.x & ((1 << gl) - 1));
    if (b >= n_it) return;
    const int qp = blockIdx.x >> gl;
    const int q = qp / kTraceParts;
    const int ttid = (qp % kTraceParts) * kTraceBlock + tid;  // thread's place in the tile
    const int W = S.width, H = S.height;
    int dest_row;
    const int slot = slot_of(P, q, ttid, W, H, &dest_row);
    const size_t j = (size_t)q * kBlock + ttid;
    double* out = samples + (size_t)b * 3 * sample_stride + j;
    V3 rgb = mk(0.0, 0.0, 0.0);
#ifdef PT_PROFILE_SECTIONS
    if (tid < kProfWaves * kProfSlots) s_prof[tid] = 0;
    __syncthreads();
    {
    PT_PROF(kProfTotal);
#endif
    const bool valid = slot >= 0 && slot < npix[b];
    double x = 0.0, y = 0.0;
    Draws dr;
    dr.rng.s0 = dr.rng.s1 = dr.rng.s2 = dr.rng.s3 = 0;
    dr.count = 0;
    if (valid) {
#ifdef PT_PROFILE_SECTIONS
        long long t_setup = __builtin_amdgcn_s_memtime();
#endif
        const int32_t* rs = row_start + (size_t)b * (H + 1);
        int r = dest_row;  // == slot / W; short rows before it can only move the source row down
        while (slot >= rs[r + 1]) r++;
        int c = slot - rs[r];
        // x of column c: the row's running sum x += xstep (main.cpp:355-357), continued from the
        // stored value at the previous multiple of 8
        const int CK = (W + kXStride - 1) / kXStride;
        if (S.flags & 4u) {
            x = (double)c / (double)W;  // light_tracing.cpp:787
        } else {
            x = xs[((size_t)b * CK + c / kXStride) * H + r];
            for (int i = c & (kXStride - 1); i > 0; --i) x += xstep;
        }
        y = ys[(size_t)b * H + r];
        dr.rng = rng_init(S.seed, (uint32_t)(first_it + b), kDomPixel, (uint32_t)slot);
#ifdef PT_PROFILE_SECTIONS
        const unsigned long long act_setup = __builtin_amdgcn_ballot_w64(true);
        if ((tid & 63) == __builtin_ctzll(act_setup)) {
            const long long dt = (long long)__builtin_amdgcn_s_memtime() - t_setup;
            s_prof[(tid >> 6) * kProfSlots + kProfSetup] += dt;
            s_prof[(tid >> 6) * kProfSlots + kProfN + kProfSetup] += dt * (long long)__builtin_popcountll(act_setup);
        }
#endif
    }
    Records rec;
    rec.pts = s_pts;
    rec.bod = s_bod;
    rec.stride = kTraceBlock;
    rec.tid = tid;
    if (COMPACT) {
        rgb = trace_path_nee_compact(S, dr, valid, x, y, rec, s_slots + (tid & ~63));
    } else if (EST == kEstBidir) {
        rgb = trace_path_bidir(S, dr, valid, x, y, rec, s_slots + (tid & ~63));
    } else if (valid) {
        PathResult R = trace_path<EST, MEASURED, true>(S, dr, x, y, rec);
        rgb = R.rgb;
    }
#ifdef PT_PROFILE_SECTIONS
    }
    __syncthreads();
    if (tid < kProfWaves * kProfSlots) atomicAdd(&g_prof[tid % kProfSlots], (unsigned long long)s_prof[tid]);
#endif
    rgb = canonical_nan(rgb);
    out[0] = rgb.x;
    out[sample_stride] = rgb.y;
    out[2 * sample_stride] = rgb.z;
}

// ---- wavefront pipeline: first stage (the other stages are in pt_wavefront.hpp) ---------------
template <bool KEYS>
__global__ void __launch_bounds__(ptw::kExtBlock, PT_BOUNCE_WAVES) k_wf_primary(DScene S, Partition P, int first_it, int n_it,
                                                                const double* __restrict__ xs, const double* __restrict__ ys,
                                                                const int32_t* __restrict__ row_start,
                                                                const int32_t* __restrict__ npix, double xstep, ptw::WfState W,
                                                                size_t J) {
    __shared__ uint4 s_tile[64 * 4];
    ptw::wf_primary_body<KEYS>(S, P, [](const Partition& Pp, int q, int tid, int Wd, int H) { return slot_of(Pp, q, tid, Wd, H); },
                               first_it, xs, ys, row_start, npix, xstep, kXStride, W, J, n_it, s_tile);
}

// ---- k_accumulate ------------------------------------------------------------------------
// buffer[i] += sample for every iteration whose frame reaches running index i
// (main.cpp:362), in iteration order.
// A block walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The accumulation of a batch runs on its
// own stream beside the estimator kernel of the next batch, and what it takes from that kernel is
// nearly its own duration when it runs flat out (without it the 15-body scene renders 3.5 % faster):
// with one block per tile it pulls 5 TB/s for 0.65 ms and the estimator's waves wait on their table
// loads and sample stores meanwhile.  512 blocks, two per compute unit, spread the same traffic over
// a longer time: C2 8.78e9 -> 8.91e9, C4 8.77e9 -> 8.90e9, main32 9.98e9 -> 1.01e10 paths/s (64 blocks
// make the accumulation longer than the batch it trails).
constexpr int kAccumulateBlocks = 512;
__global__ void __launch_bounds__(kBlock) k_accumulate(Partition P, int W, int H, int B,
                                                       const int32_t* __restrict__ npix,
                                                       const double* __restrict__ samples, size_t sample_stride,
                                                       double* __restrict__ accum) {
    const int tid = threadIdx.x;
    for (int q = blockIdx.x; q < P.owned_tiles; q += gridDim.x) {
        const int slot = slot_of(P, q, tid, W, H);
        if (slot < 0) continue;
        const size_t j = (size_t)q * kBlock + tid;
        double* a = accum + (size_t)slot * 3;
        double ax = a[0], ay = a[1], az = a[2];
        for (int b = 0; b < B; ++b) {
            if (slot < npix[b]) {
                const double* s = samples + (size_t)b * 3 * sample_stride + j;
                ax += s[0];
                ay += s[sample_stride];
                az += s[2 * sample_stride];
            }
        }
        a[0] = ax;
        a[1] = ay;
        a[2] = az;
    }
}

// ---- tile exchange -----------------------------------------------------------------------
// The exchange layout of a partition: its owned tiles one after the other, a tile as 256 pixels in
// row-major order within the tile, a pixel as its three FP64 channels: packed[(q*256 + tid)*3 + c].
// Pixels of an edge tile outside the image are zero.  blockIdx.y selects the source rank of a
// gathered buffer (rank-major, tiles_per_rank tiles each).
__global__ void __launch_bounds__(kBlock) k_tiles_pack(Partition P, int W, int H, int tiles_capacity,
                                                       const double* __restrict__ accum, double* __restrict__ packed) {
    const int q = blockIdx.x, tid = threadIdx.x;
    const int slot = q < P.owned_tiles ? slot_of(P, q, tid, W, H) : -1;
    double* o = packed + ((size_t)q * kBlock + tid) * 3;
    const double* a = accum + (size_t)max(slot, 0) * 3;
    o[0] = slot >= 0 ? a[0] : 0.0;
    o[1] = slot >= 0 ? a[1] : 0.0;
    o[2] = slot >= 0 ? a[2] : 0.0;
}
__global__ void __launch_bounds__(kBlock) k_tiles_unpack(Partition P, int W, int H, int tiles_per_rank,
                                                         const double* __restrict__ gathered, double* __restrict__ accum) {
    const int q = blockIdx.x, tid = threadIdx.x;
    P.rank = blockIdx.y;
    const int total = P.tiles_x * P.tiles_y;
    if (q * P.world + P.rank >= total) return;
    const int slot = slot_of(P, q, tid, W, H);
    if (slot < 0) return;
    const double* g = gathered + (((size_t)P.rank * tiles_per_rank + q) * kBlock + tid) * 3;
    double* a = accum + (size_t)slot * 3;
    a[0] = g[0];
    a[1] = g[1];
    a[2] = g[2];
}

// ---- test / diagnostic kernels ---------------------------------------------------------------
__global__ void k_intersect_batch(DScene S, const double* __restrict__ rays6, int n, int skip,
                                  int32_t* __restrict__ idx, double* __restrict__ t) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o = ld3(rays6 + 6 * (size_t)i), d = ld3(rays6 + 6 * (size_t)i + 3);
    double mint;
    int hit = nearest_hit<true>(S, o, d, skip, mint);
    idx[i] = hit;
    t[i] = mint;
}

__global__ void k_intersect_all(DScene S, const double* __restrict__ rays6, int n, double* __restrict__ t_all) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o = ld3(rays6 + 6 * (size_t)i), d = ld3(rays6 + 6 * (size_t)i + 3);
    for (int b = 0; b < S.n_bodies; ++b) t_all[(size_t)i * S.n_bodies + b] = intersect_body(S, S.geom[b], o, d);
}

template <int EST, bool MEASURED>
__global__ void __launch_bounds__(kBlock) k_trace_paths(DScene S, const int32_t* __restrict__ it_idx2,
                                                        const double* __restrict__ xy, int n, double* rgb,
                                                        int32_t* nrec, int32_t* bodies, double* points,
                                                        int32_t* visible, int32_t* used, int32_t* brdf_ind = nullptr) {
    __shared__ double s_pts[kMaxDepth * 3 * kBlock];
    __shared__ int8_t s_bod[kMaxDepth * kBlock];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Draws dr;
    dr.rng = rng_init(S.seed, (uint32_t)it_idx2[2 * i], kDomPixel, (uint32_t)it_idx2[2 * i + 1]);
    dr.count = 0;
    Records rec;
    rec.pts = s_pts;
    rec.bod = s_bod;
    rec.stride = kBlock;
    rec.tid = threadIdx.x;
    int ind5[kMaxDepth] = {-1, -1, -1, -1, -1};
    PathResult R = trace_path<EST, MEASURED, false>(S, dr, xy[2 * i], xy[2 * i + 1], rec, MEASURED && brdf_ind ? ind5 : nullptr);
    if (MEASURED && brdf_ind)
        for (int k = 0; k < kMaxDepth; ++k) brdf_ind[5 * i + k] = ind5[k];
    if (rgb) {
        const V3 c = canonical_nan(R.rgb);
        rgb[3 * i] = c.x;
        rgb[3 * i + 1] = c.y;
        rgb[3 * i + 2] = c.z;
    }
    if (nrec) nrec[i] = R.nrec;
    for (int k = 0; k < kMaxDepth; ++k) {
        bool have = k < R.nrec;
        V3 p = have ? rec.point(k) : mk(0.0, 0.0, 0.0);
        if (bodies) bodies[5 * i + k] = have ? rec.body(k) : -1;
        if (points) {
            points[15 * i + 3 * k] = p.x;
            points[15 * i + 3 * k + 1] = p.y;
            points[15 * i + 3 * k + 2] = p.z;
        }
    }
    if (visible) visible[i] = R.visible;
    if (used) used[i] = dr.count;
}

__global__ void k_rng_fill(uint64_t seed, uint32_t it, uint32_t dom, uint32_t idx, int n, int32_t* k, double* u) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng r = rng_init(seed, it, dom, idx);
    for (int i = 0; i < n; ++i) {
        int32_t v = rng_next(r);
        k[i] = v;
        u[i] = k_to_unit(v);
    }
}

// kind: 0 recip, 1 2u-1 (a = k), 2 *1.0000001L, 3 *0.9999999L, 4 glass theta2(eta = a, theta1 = b)
PTX_INLINE double x87_eval(int kind, double a, double b) {
    switch (kind) {
        case 0: return ptx::x87_recip(a);
        case 1: return ptx::x87_2u_minus_1((double)(int32_t)a / 2147483647.0, (int32_t)a);
        case 2: return ptx::x87_mul_1p0000001(a);
        case 3: return ptx::x87_mul_0p9999999(a);
        default: return ptx::x87_glass_theta2(a, b);
    }
}

__global__ void k_x87_eval(int kind, const double* __restrict__ a, const double* __restrict__ b, int n,
                           double* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x87_eval(kind, a[i], b[i]);
}

__global__ void k_brdf_lookup(const double* __restrict__ table, const double* __restrict__ angles4, int n,
                              double* __restrict__ rgb, int32_t* __restrict__ ind) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* a = angles4 + 4 * (size_t)i;
    int id;
    V3 f = brdf_lookup(table, a[0], a[1], a[2], a[3], &id);
    rgb[3 * i] = f.x;
    rgb[3 * i + 1] = f.y;
    rgb[3 * i + 2] = f.z;
    ind[i] = id;
}

__global__ void k_brdf_lookup_vectors(const double* __restrict__ table, const double* __restrict__ bins,
                                      const double* __restrict__ vec6, int n, double* __restrict__ rgb,
                                      int32_t* __restrict__ ind) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 a = ld3(vec6 + 6 * (size_t)i), b = ld3(vec6 + 6 * (size_t)i + 3);
    ind[i] = brdf_index_of_vectors(bins, a, b);
    const V3 f = measured_term(table, bins, a, b);
    rgb[3 * i] = f.x;
    rgb[3 * i + 1] = f.y;
    rgb[3 * i + 2] = f.z;
}

// ---- host state ------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return PT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) return fail(PT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        bytes = need;
        return PT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Device allocation local to one diagnostic call: freed on every return path.
template <class T>
struct Tmp {
    T* p = nullptr;
    Tmp() = default;
    Tmp(const Tmp&) = delete;
    Tmp& operator=(const Tmp&) = delete;
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, sizeof(T) * std::max<size_t>(n, 1)); }
    operator T*() const { return p; }
};

// ---- fixed device scratch of pt_render_device_noalloc ----------------------------------------
// One array per GPU the library's code object is loaded on, held until the process ends.  In host
// code the array's name is only a handle: the address a kernel can use comes from
// hipGetSymbolAddress, asked once per device (with that device current) and kept here.
constexpr size_t kScratchBytes = (size_t)PT_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

// Address of `scratch` on `device`, which must be the current device.
int scratch_address(int device, void** out) {
    if (device < 0 || device >= kMaxDevices) return fail(PT_ERR_INVALID, "device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(PT_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return PT_OK;
}

}  // namespace

struct pt_scene {
    int device = 0;
    void* scratch_ptr = nullptr;  // this device's copy of the file-scope scratch array
    int width = 0, height = 0, estimator = 0;
    uint32_t flags = 0;
    uint64_t seed = 0;
    pt_camera camera;
    std::vector<pt_body> bodies;  // as stored (normals normalised)
    std::vector<int32_t> lights;
    Partition part;
    DScene ds;
    DevBuf d_geom, d_mat, d_lights, d_brdf[PT_MAX_BRDF];
    DevBuf d_grp_members, d_grp_bounds, d_grp_supers, d_grp_halves, d_grp_half4, d_grp_half_order, d_grp_exact;
    DevBuf d_scan_planes, d_scan_spheres, d_scan_screen, d_scan_screen0, d_brdf_bins, d_scan_rects, d_scan_rects_mt;
    // frame tables and samples are double-buffered: the layout of batch k+1 and the accumulation of
    // batch k-1 overlap the estimator kernel of batch k on two side streams
    DevBuf d_xs[2], d_ys[2], d_rowstart[2], d_npix[2], d_samples[2], d_accum;
    hipStream_t st_layout = nullptr, st_accum = nullptr;
    hipEvent_t ev_entry = nullptr, ev_layout[2] = {nullptr, nullptr}, ev_trace[2] = {nullptr, nullptr},
               ev_acc[2] = {nullptr, nullptr};
    double xstep = 0, ystep = 0;
    int batch = 1;        // iterations per fused batch, at most
    int batch_alloc = 0;  // iterations per batch the fused buffers are sized for (grows with the calls)
    std::vector<hipEvent_t> ev;  // start/stop pairs of the estimator launches of the last render
    size_t ev_used = 0;
    double last_paths = 0;
    bool wf_sort = true;        // wavefront pipeline: reorder rays by key before every scan
    bool has_measured = false;  // any PT_MEASURED body: selects the kernel variant with the table lookup
    bool has_rectangles = false;
    int pipeline = PT_PIPELINE_AUTO;
    int wf_batch = 1;  // iterations per wavefront batch, at most
    int wf_batch_alloc = 0;  // iterations per batch the wavefront and frame buffers are sized for (grows with the calls)
    DevBuf d_wf;       // one allocation carved into the wavefront state arrays
    DevBuf d_wf_sort;  // the ray sort: sorted keys (unused afterwards) and the library's temporary storage
    size_t wf_sort_tmp_bytes = 0;
    // Queue counts of an earlier batch, copied to pinned host memory without waiting for them, and the
    // capacity they refer to: a hint for how much of a queue is worth sorting (see render_wavefront).
    int32_t* wf_counts_host = nullptr;
    int wf_counts_cap = 0;
    ptw::WfState wf;
};

namespace {

constexpr int kMaxBatchIterations = 1024;

int setup_partition(pt_scene* s, int rank, int world, int tw, int th) {
    if (world < 1 || rank < 0 || rank >= world) return fail(PT_ERR_INVALID, "partition: rank/world out of range");
    if (tw < 1 || th < 1 || tw * th != kBlock) return fail(PT_ERR_INVALID, "partition: tile_w*tile_h must be 256");
    Partition& P = s->part;
    P.rank = rank;
    P.world = world;
    P.tile_w = tw;
    P.tile_h = th;
    P.tile_w_log2 = 0;
    while ((1 << P.tile_w_log2) < tw) P.tile_w_log2++;
    P.tiles_x = (s->width + tw - 1) / tw;
    P.tiles_y = (s->height + th - 1) / th;
    int total = P.tiles_x * P.tiles_y;
    P.owned_tiles = total > rank ? (total - rank + world - 1) / world : 0;
    // batch: about 2^27 paths per launch, whatever share of the frame the partition owns (a rank that
    // owns an eighth of a 1024x768 frame takes 1 024 iterations at a time: with the launches as long as a
    // single GPU's, the ends of launches, where the chip runs half empty, weigh no more than there).
    // The frame layout and the accumulation of a batch run beside the estimator of its neighbours on
    // the little that kernel leaves free of a CU (six blocks of 80 registers fill it), so their share
    // per path must be small: at 3840x2160 a batch of two iterations (2^24 paths) made the accumulator
    // traffic per path as large as the samples' and the estimator waited for it (7.0e9 paths/s; 2^25:
    // 7.8e9, 2^26: 8.2e9, 2^27: 8.5e9).  The samples of a batch take 3.2 GB at most, its tables 0.8 GB.
    // PTGPU_BATCH_LOG2 overrides (development).
    size_t per_it = (size_t)std::max(1, P.owned_tiles) * kBlock;
    const char* lg = getenv("PTGPU_BATCH_LOG2");
    const int log2_paths = lg ? std::min(30, std::max(16, atoi(lg))) : 27;
    size_t b = ((size_t)1 << log2_paths) / per_it;
    s->batch = (int)std::min<size_t>(kMaxBatchIterations, std::max<size_t>(1, b));
    P.group_log2 = 0;
    // whole groups of iterations (see launch_trace), so that only the last batch of a call launches blocks with nothing to do
    if (s->batch >= (1 << kXcdsLog2)) s->batch &= ~((1 << kXcdsLog2) - 1);
    s->batch_alloc = 0;  // sized again by the next call
    // the wavefront pipeline keeps 364 B of state per path in HBM: batches of 2^26 paths (24 GB, with
    // the samples 28 GB of the 288) — the more paths, the more coherent the waves the ray sort
    // forms (2^24: 6.85e8, 2^25: 7.09e8, 2^26: 7.19e8, 2^27: 7.28e8 paths/s on the 10 000-body scene)
    size_t wb = ((size_t)1 << 26) / per_it;
    s->wf_batch = (int)std::min<size_t>(128, std::max<size_t>(1, wb));
    s->wf_batch_alloc = 0;  // sized again by the next call
    return PT_OK;
}

constexpr int kMaxBodiesFused = 127;  // body indices are kept as int8 in the fused kernel's LDS records

// Which pipeline pt_render_device runs (see pt_set_pipeline).
bool uses_wavefront(const pt_scene* s) {
    return s->pipeline == PT_PIPELINE_WAVEFRONT || (s->pipeline == PT_PIPELINE_AUTO && (int)s->bodies.size() > kMaxBodiesFused);
}

int ensure_render_buffers(pt_scene* s) {
    const size_t W = s->width, H = s->height, B = (size_t)std::max(1, uses_wavefront(s) ? s->wf_batch_alloc : s->batch_alloc);
    int rc;
    size_t J = (size_t)std::max(1, s->part.owned_tiles) * kBlock;
    for (int k = 0; k < 2; ++k) {
        if ((rc = s->d_xs[k].ensure(B * H * ((W + kXStride - 1) / kXStride) * sizeof(double)))) return rc;
        if ((rc = s->d_ys[k].ensure(B * H * sizeof(double)))) return rc;
        if ((rc = s->d_rowstart[k].ensure(B * (H + 1) * sizeof(int32_t)))) return rc;
        if ((rc = s->d_npix[k].ensure(B * sizeof(int32_t)))) return rc;
        if ((rc = s->d_samples[k].ensure(B * 3 * J * sizeof(double)))) return rc;
    }
    if (!s->st_layout) {
        // Default priority on purpose: with side streams of another priority a render on a stream of the
        // caller's followed by an RCCL collective in the same process never completed (the process has four
        // hardware queues), and the layout kernels find their room beside the estimator without it.
        HIP_TRY(hipStreamCreateWithFlags(&s->st_layout, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s->st_accum, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_entry, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&s->ev_layout[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&s->ev_trace[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&s->ev_acc[k], hipEventDisableTiming));
        }
    }
    return PT_OK;
}


// Builds and uploads everything the kernels read about a scene: body and material records, the
// compact scan lists with their slabs, the grouped layout of the large-scene scan, the rectangle list, the
// scene-wide flags.  One step per method, in the order of run().
struct SceneUpload {
    pt_scene* s;
    DScene& D;
    const int n;
    std::vector<DGeom> geom;
    std::vector<DMat> mat;
    std::vector<DRect> rects;
    std::vector<DRectMT> rects_mt;
    std::vector<DGeom> scan_planes, scan_spheres;
    std::vector<DScreen2> scan_screen, scan_screen0;
    std::vector<char> in_slab;
    float c1max = 0.0f, r2max = 0.0f;
    double hit_eps = 0.0;
    int n_planes = 0, n_spheres = 0;

    explicit SceneUpload(pt_scene* scene)
        : s(scene), D(scene->ds), n((int)scene->bodies.size()), geom(n), mat(n), in_slab(n, 0) {
        memset(&D, 0, sizeof D);
    }

    // DGeom / DMat / DRect of every body, the hit threshold.
    void body_records() {
        // t > 0.0000001L (raytrace3.cpp:776) for a double t: t > the largest double not above the literal
        hit_eps = 0.0;
        if (s->flags & PT_F_RAYTRACE3) {
            const long double lit = 0.0000001L;
            hit_eps = (double)lit;
            if ((long double)hit_eps > lit) hit_eps = std::nextafter(hit_eps, 0.0);
        }
        for (int i = 0; i < n; ++i) {
            const pt_body& b = s->bodies[i];
            DGeom g;
            memset(&g, 0, sizeof g);
            g.cx = b.center[0];
            g.cy = b.center[1];
            g.cz = b.center[2];
            g.shape = b.shape;
            if (b.shape == PT_RECTANGLE) {
                // Rectangle ctor raytrace3.cpp:254-265; pt_body_construct has stored the normal
                g.shape = kRect;
                g.nx = b.normal[0];
                g.ny = b.normal[1];
                g.nz = b.normal[2];
                g.pad[0] = (int32_t)rects.size();
                DRect r;
                memset(&r, 0, sizeof r);
                const double* c = b.corners;
                double v1[3], v3[3];
                pt_normalize3(c[3] - c[0], c[4] - c[1], c[5] - c[2], v1);
                pt_normalize3(c[9] - c[6], c[10] - c[7], c[11] - c[8], v3);
                for (int k = 0; k < 3; ++k) {
                    r.p1[k] = c[k];
                    r.p3[k] = c[6 + k];
                    r.v1[k] = v1[k];
                    r.v3[k] = v3[k];
                    r.n[k] = b.normal[k];
                }
                r.body = i;
                rects.push_back(r);
            } else if (b.shape == PT_RECTANGLE_MT) {
                // Rectangle ctor light_tracing.cpp:484-495; pt_body_construct has stored the normal
                g.shape = kRectMT;
                g.nx = b.normal[0];
                g.ny = b.normal[1];
                g.nz = b.normal[2];
                g.pad[0] = (int32_t)rects_mt.size();
                DRectMT r;
                memset(&r, 0, sizeof r);
                const double* c = b.corners;
                for (int k = 0; k < 3; ++k) {
                    r.p1[k] = c[k];
                    r.e1[k] = c[3 + k] - c[k];
                    r.e2[k] = c[6 + k] - c[k];
                    r.e3[k] = c[9 + k] - c[k];
                }
                r.body = i;
                rects_mt.push_back(r);
            } else if (b.shape == PT_PLANE) {
                g.nx = b.normal[0];
                g.ny = b.normal[1];
                g.nz = b.normal[2];
                // axis-aligned unit normals take the one-division form of the scan
                auto unit = [](double v) { return v == 1.0 || v == -1.0; };
                if (unit(g.nx) && g.ny == 0.0 && g.nz == 0.0) g.shape = kPlaneX;
                if (g.nx == 0.0 && unit(g.ny) && g.nz == 0.0) g.shape = kPlaneY;
                if (g.nx == 0.0 && g.ny == 0.0 && unit(g.nz)) g.shape = kPlaneZ;
            } else {
                g.nx = b.radius * b.radius;  // Sphere ctor radius2 main.cpp:207
                g.fcx = (float)g.cx;
                g.fcy = (float)g.cy;
                g.fcz = (float)g.cz;
                g.fr2 = (float)g.nx;
                c1max = std::max(c1max, std::fabs(g.fcx) + std::fabs(g.fcy) + std::fabs(g.fcz));
                r2max = std::max(r2max, g.fr2);
            }
            geom[i] = g;
            DMat m;
            memset(&m, 0, sizeof m);
            m.colr = b.color[0];
            m.colg = b.color[1];
            m.colb = b.color[2];
            m.emr = b.emission[0];
            m.emg = b.emission[1];
            m.emb = b.emission[2];
            m.ior = b.ior;
            m.refl = b.reflection;
            m.eta = 1.0 / b.ior;
            m.material = b.material;
            m.brdf_id = b.brdf_id;
            m.emits = (b.emission[0] != 0.0) || (b.emission[1] != 0.0) || (b.emission[2] != 0.0);
            mat[i] = m;
        }
    }

    // A slab joins, per axis, the two axis-aligned planes with the smallest and the largest
    // coordinate, provided the bound slab_big is usable.
    void find_slabs() {
        // An axis without a slab holds an empty interval and no bodies: scan_slab on it — which a scan could
        // reach through a stale loop guard, see scan_lists — forms infinite quotients that win nothing.
        for (int axis = 0; axis < 3; ++axis) {
            D.slab_lo[axis] = INFINITY;
            D.slab_hi[axis] = -INFINITY;
            D.slab_big[axis] = 0.0;
            D.slab_ilo[axis] = D.slab_ihi[axis] = -1;
        }
        for (int axis = 0; axis < 3 && hit_eps == 0.0; ++axis) {  // the slab form assumes the threshold t > 0
            int lo = -1, hi = -1;
            auto coord = [&](int i) { return axis == 0 ? geom[i].cx : axis == 1 ? geom[i].cy : geom[i].cz; };
            for (int i = 0; i < n; ++i) {
                if (geom[i].shape != kPlaneX + axis) continue;
                if (lo < 0 || coord(i) < coord(lo)) lo = i;
                if (hi < 0 || coord(i) > coord(hi)) hi = i;
            }
            if (lo < 0 || !(coord(lo) < coord(hi))) continue;
            // |o_k| < big  =>  |c - o_k| < 2^40 (c_hi - c_lo) for both planes
            const double big = std::ldexp(coord(hi) - coord(lo), 40) - std::max(std::fabs(coord(lo)), std::fabs(coord(hi)));
            if (!(big > 0.0) || !std::isfinite(big)) continue;
            D.slab_big[axis] = big;
            D.slab_mask |= 1 << axis;
            D.slab_lo[axis] = coord(lo);
            D.slab_hi[axis] = coord(hi);
            D.slab_ilo[axis] = lo;
            D.slab_ihi[axis] = hi;
            in_slab[lo] = in_slab[hi] = 1;
        }
    }

    // Compact lists of the fused scan (body index in pad[0]), each in ascending body order: the
    // planes outside the slabs and the spheres; an odd sphere count is padded with a record whose
    // screen never passes (r2 = -inf).
    void scan_lists() {
        for (int i = 0; i < n; ++i) {
            DGeom g = geom[i];
            g.pad[0] = i;
            if (g.shape == kSphere) {
                scan_spheres.push_back(g);
            } else if (g.shape != kRect && g.shape != kRectMT) {
                if (!in_slab[i]) scan_planes.push_back(g);
            }
        }
        // Neither list is ever empty: an empty one gets records that can never be hit, so that the scan
        // loops always have a first iteration.  The compiler turns `for (k < n)` into a guard and a
        // count-down loop, and under register pressure it keeps a wave-uniform boolean such as that guard
        // in a vector register and reads it back as a lane mask of the lanes active at that moment; a
        // later scan run by other lanes of the wave (lanes that had left the first loop early) then sees
        // "n > 0" and enters the loop with a count of zero.  Seen in the bidirectional kernel, whose light
        // path can be walked by lanes whose gather path ended at once, on a scene without planes.  The
        // same goes for the slab bits of DScene::slab_mask, hence the harmless defaults in find_slabs.
        if (scan_planes.empty()) {
            DGeom g;
            memset(&g, 0, sizeof g);
            g.shape = kPlane;  // normal (0,0,0): n.d = 0 lies inside the parallel cut-off of Plane::intersect, never a hit
            g.pad[0] = -1;
            scan_planes.push_back(g);
        }
        n_planes = (int)scan_planes.size();
        while (scan_spheres.empty() || (scan_spheres.size() & 1)) {
            DGeom g;
            memset(&g, 0, sizeof g);
            g.nx = -1.0;  // radius^2 < 0: the exact test cannot pass either
            g.fr2 = -INFINITY;
            g.shape = kSphere;
            g.pad[0] = -1;
            scan_spheres.push_back(g);
        }
        n_spheres = (int)scan_spheres.size();
        // the origin-dependent term of every record for the origin (0,0,0) (see origin0_term)
        auto put_term = [](DGeom& g, double v) {
            uint64_t u;
            memcpy(&u, &v, 8);
            g.pad[1] = (int32_t)(uint32_t)u;
            g.pad[2] = (int32_t)(uint32_t)(u >> 32);
        };
        for (DGeom& g : scan_planes) put_term(g, g.nx * (g.cx - 0.0) + g.ny * (g.cy - 0.0) + g.nz * (g.cz - 0.0));
        for (DGeom& g : scan_spheres) {
            const double dx = 0.0 - g.cx, dy = 0.0 - g.cy, dz = 0.0 - g.cz;
            put_term(g, (dx * dx + dy * dy + dz * dz) - g.nx);
        }
        scan_screen.assign(scan_spheres.size() / 2, DScreen2());
    scan_screen0.assign(scan_spheres.size() / 2, DScreen2());
        for (size_t k = 0; k < scan_spheres.size(); ++k) {
            const DGeom& g = scan_spheres[k];
            DScreen2& r = scan_screen[k / 2];
            r.cx[k & 1] = g.fcx;
            r.cy[k & 1] = g.fcy;
            r.cz[k & 1] = g.fcz;
            r.r2[k & 1] = g.fr2;
            DScreen2& r0 = scan_screen0[k / 2];
            r0.cx[k & 1] = g.fcx;
            r0.cy[k & 1] = g.fcy;
            r0.cz[k & 1] = g.fcz;
            // float c-term of the float centre, one rounding (the screen's error bound allows several)
            r0.r2[k & 1] = (float)(((double)g.fcx * g.fcx + (double)g.fcy * g.fcy + (double)g.fcz * g.fcz) - (double)g.fr2);
        }
    }

    // Grouped layout of the large-scene scan (wavefront pipeline): the spheres in groups of
    // kGroupBodies, ordered along a Morton curve so that a group is spatially compact.  Per group a
    // bounding sphere in float (centre, R^2 rounded outwards): a wave examines the members only when
    // some lane's ray might enter the bound.  Member records are the float (centre, r^2) pairs of
    // the per-body screen; slots past the end can never pass (r^2 = -inf).  grp_exact holds, per member
    // slot, the FP64 record and the body index.  (Planes are scanned from the plane list, as in the fused kernel.)
    int sphere_groups() {
        std::vector<int32_t> order;
        {
            std::vector<int> sph;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int i = 0; i < n; ++i) {
                if (geom[i].shape == kSphere) {
                    sph.push_back(i);
                    const double c[3] = {geom[i].cx, geom[i].cy, geom[i].cz};
                    for (int k = 0; k < 3; ++k) {
                        lo[k] = std::min(lo[k], c[k]);
                        hi[k] = std::max(hi[k], c[k]);
                    }
                }
            }
            auto morton = [&](int i) {
                const double c[3] = {geom[i].cx, geom[i].cy, geom[i].cz};
                uint64_t code = 0;
                uint32_t q[3];
                for (int k = 0; k < 3; ++k) {
                    double u = hi[k] > lo[k] ? (c[k] - lo[k]) / (hi[k] - lo[k]) : 0.0;
                    if (!(u >= 0.0)) u = 0.0;  // also NaN
                    if (u > 1.0) u = 1.0;
                    q[k] = (uint32_t)(u * 1023.0);
                }
                for (int bit = 9; bit >= 0; --bit)
                    for (int k = 0; k < 3; ++k) code = (code << 1) | ((q[k] >> bit) & 1u);
                return code;
            };
            std::vector<std::pair<uint64_t, int>> keyed;
            for (int i : sph) keyed.push_back({morton(i), i});
            std::sort(keyed.begin(), keyed.end());
            for (auto& kv : keyed) order.push_back(kv.second);
            // Within the curve's order, spatial median splits: a range is cut across the longest side
            // of its box of centres into two parts whose sizes are multiples of a super-group (then of a
            // group), so that every group of kGroupBodies consecutive spheres and every super-group of
            // kSuperGroups consecutive groups is a compact cell of a kd-partition rather than a stretch
            // of the curve, which jumps across the scene at octant borders.
            if (!getenv("PTGPU_MORTON_GROUPS")) {
                auto coord = [&](int i, int axis) { return axis == 0 ? geom[i].cx : axis == 1 ? geom[i].cy : geom[i].cz; };
                struct Range { size_t a, b; };
                std::vector<Range> todo;
                todo.push_back({0, order.size()});
                while (!todo.empty()) {
                    const Range r = todo.back();
                    todo.pop_back();
                    const size_t cnt = r.b - r.a;
                    if (cnt <= (size_t)kGroupBodies) continue;
                    const size_t half_section = (size_t)ptw::kSectionGroups * kGroupBodies / 2, super = (size_t)kGroupBodies * kSuperGroups;
                    const size_t unit = cnt > half_section ? half_section : cnt > super ? super : (size_t)kGroupBodies;
                    const size_t units = (cnt + unit - 1) / unit;
                    const size_t left = ((units + 1) / 2) * unit;  // < cnt because units >= 2
                    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
                    for (size_t k = r.a; k < r.b; ++k)
                        for (int ax = 0; ax < 3; ++ax) {
                            const double v = coord(order[k], ax);
                            if (v < blo[ax]) blo[ax] = v;
                            if (v > bhi[ax]) bhi[ax] = v;
                        }
                    int axis = 0;
                    for (int ax = 1; ax < 3; ++ax)
                        if (bhi[ax] - blo[ax] > bhi[axis] - blo[axis]) axis = ax;
                    std::nth_element(order.begin() + r.a, order.begin() + r.a + left, order.begin() + r.b,
                                     [&](int x, int y) {
                                         const double cx = coord(x, axis), cy = coord(y, axis);
                                         return cx < cy || (cx == cy && x < y);  // NaN compares false: any order, still a permutation
                                     });
                    todo.push_back({r.a, r.a + left});
                    todo.push_back({r.a + left, r.b});
                }
            }
            // whole pairs of super-groups (kSuperGroups groups each)
            while (order.empty() || order.size() % (2 * kSuperGroups * kGroupBodies)) order.push_back(-1);
            const int n_groups = (int)order.size() / kGroupBodies;
            std::vector<DScreen2> members(order.size() / 2), bounds(n_groups / 2);
            for (size_t k = 0; k < order.size(); ++k) {
                DScreen2& r = members[k / 2];
                const int i = order[k];
                const bool sphere = i >= 0 && geom[i].shape == kSphere;
                // non-spheres are always examined (an all-NaN record never screens out), padding never
                const float other = i >= 0 ? std::nanf("") : 0.0f;
                r.cx[k & 1] = sphere ? geom[i].fcx : other;
                r.cy[k & 1] = sphere ? geom[i].fcy : other;
                r.cz[k & 1] = sphere ? geom[i].fcz : other;
                r.r2[k & 1] = sphere ? geom[i].fr2 : (i >= 0 ? std::nanf("") : -INFINITY);
            }
            // Bounding sphere (float centre, radius rounded outwards) of the spheres in member slots
            // [first, first + count), measured from the spheres themselves around the centre actually
            // stored; nothing there: radius -inf (parked below).
            auto bound_of_slots = [&](size_t first, size_t count, DScreen2& r, int side) {
                double c[3] = {0.0, 0.0, 0.0};
                int cnt = 0;
                const size_t last = std::min(order.size(), first + count);
                for (size_t k = first; k < last; ++k) {
                    const int i = order[k];
                    if (i < 0) continue;
                    c[0] += geom[i].cx;
                    c[1] += geom[i].cy;
                    c[2] += geom[i].cz;
                    cnt++;
                }
                float fc[3] = {0.0f, 0.0f, 0.0f};
                float fr2 = -INFINITY;
                if (cnt > 0) {
                    // centre: from the centroid towards the smallest enclosing ball (Badoiu-Clarkson steps
                    // towards the farthest sphere); any centre is valid, a better one gives a smaller radius
                    double ctr[3] = {c[0] / cnt, c[1] / cnt, c[2] / cnt};
                    for (int it = 1; it <= 64 && cnt > 1; ++it) {
                        double far = -1.0, fv[3] = {0.0, 0.0, 0.0};
                        for (size_t k = first; k < last; ++k) {
                            const int i = order[k];
                            if (i < 0) continue;
                            const double dx = geom[i].cx - ctr[0], dy = geom[i].cy - ctr[1], dz = geom[i].cz - ctr[2];
                            const double dist = std::sqrt(dx * dx + dy * dy + dz * dz), reach = dist + std::sqrt(geom[i].nx);
                            if (reach > far && dist > 0.0) {
                                far = reach;
                                const double pull = reach / dist;  // towards the far side of that sphere
                                fv[0] = dx * pull;
                                fv[1] = dy * pull;
                                fv[2] = dz * pull;
                            }
                        }
                        if (!(far > 0.0) || !std::isfinite(far)) break;
                        for (int k = 0; k < 3; ++k) ctr[k] += fv[k] / (it + 1.0);
                    }
                    for (int k = 0; k < 3; ++k) fc[k] = (float)ctr[k];
                    if (!std::isfinite(fc[0]) || !std::isfinite(fc[1]) || !std::isfinite(fc[2]))
                        for (int k = 0; k < 3; ++k) fc[k] = (float)(c[k] / cnt);
                    double R = 0.0;
                    for (size_t k = first; k < last; ++k) {
                        const int i = order[k];
                        if (i < 0) continue;
                        const double dx = geom[i].cx - fc[0], dy = geom[i].cy - fc[1], dz = geom[i].cz - fc[2];
                        R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt(geom[i].nx));
                    }
                    // the record keeps R, enlarged by the 2.5e-4 the distance cut of group_possible2_masks asks for
                    fr2 = std::nextafter((float)(R * (1.0 + 2.5e-4)), INFINITY);
                    if (!(fr2 >= 0.0f)) fr2 = INFINITY;  // NaN: always enter
                    c1max = std::max(c1max, std::fabs(fc[0]) + std::fabs(fc[1]) + std::fabs(fc[2]));
                    if (std::isfinite(fr2)) r2max = std::max(r2max, fr2 * fr2 * 1.00001f);
                }
                r.cx[side] = fc[0];
                r.cy[side] = fc[1];
                r.cz[side] = fc[2];
                r.r2[side] = fr2;
            };
            for (int g = 0; g < n_groups; ++g) bound_of_slots((size_t)g * kGroupBodies, (size_t)kGroupBodies, bounds[g / 2], g & 1);
            // second level: kSuperGroups consecutive groups; third level: the two halves of a section of
            // kSectionGroups groups
            const int n_super = n_groups / kSuperGroups;
            std::vector<DScreen2> supers(n_super / 2);
            for (int sg = 0; sg < n_super; ++sg)
                bound_of_slots((size_t)sg * kSuperGroups * kGroupBodies, (size_t)kSuperGroups * kGroupBodies, supers[sg / 2], sg & 1);
            const size_t section_slots = (size_t)ptw::kSectionGroups * kGroupBodies;
            const int n_sections = (int)((order.size() + section_slots - 1) / section_slots);
            std::vector<DScreen2> halves((size_t)n_sections);
            for (int t = 0; t < n_sections; ++t)
                for (int h = 0; h < 2; ++h) bound_of_slots((size_t)t * section_slots + (size_t)h * (section_slots / 2), section_slots / 2, halves[t], h);
            // An empty group or super-group (radius -inf above) must never be entered.  With the record
            // holding R, not R^2, that is said by a centre far outside any scene and R = 0: for
            // |d| <= 1 the discriminant is about -|D|^2 (1 - cos^2) - ... <= -2e36, far below every error
            // bound.  (Rays with a longer direction examine every group anyway; the member slots of an
            // empty group can never pass.)
            auto park_empty = [](std::vector<DScreen2>& recs) {
                for (DScreen2& r : recs)
                    for (int h = 0; h < 2; ++h)
                        if (!(r.r2[h] >= 0.0f)) {
                            r.cx[h] = r.cy[h] = r.cz[h] = 1e18f;
                            r.r2[h] = 0.0f;
                        }
            };
            park_empty(bounds);
            park_empty(supers);
            park_empty(halves);
            int rcg;
            if ((rcg = s->d_grp_supers.ensure(sizeof(DScreen2) * supers.size()))) return rcg;
            HIP_TRY(hipMemcpy(s->d_grp_supers.p, supers.data(), sizeof(DScreen2) * supers.size(), hipMemcpyHostToDevice));
            D.grp_supers = (const DScreen2*)s->d_grp_supers.p;
            if ((rcg = s->d_grp_halves.ensure(sizeof(DScreen2) * halves.size()))) return rcg;
            HIP_TRY(hipMemcpy(s->d_grp_halves.p, halves.data(), sizeof(DScreen2) * halves.size(), hipMemcpyHostToDevice));
            D.grp_halves = (const DScreen2*)s->d_grp_halves.p;
            const size_t n_halves = 2 * halves.size();
            std::vector<float> half4(4 * n_halves);
            for (size_t h = 0; h < n_halves; ++h) {
                const DScreen2& r = halves[h / 2];
                half4[4 * h + 0] = r.cx[h & 1];
                half4[4 * h + 1] = r.cy[h & 1];
                half4[4 * h + 2] = r.cz[h & 1];
                half4[4 * h + 3] = r.r2[h & 1];
            }
            if ((rcg = s->d_grp_half4.ensure(sizeof(float) * half4.size()))) return rcg;
            HIP_TRY(hipMemcpy(s->d_grp_half4.p, half4.data(), sizeof(float) * half4.size(), hipMemcpyHostToDevice));
            D.grp_half4 = (const float*)s->d_grp_half4.p;
            D.grp_half_order = nullptr;
            if (n_halves <= 64 && !getenv("PTGPU_INDEX_ORDER")) {  // (the variable: index order, for tests of that path)
                // per cell of the sort box (see ray_key) the halves by distance from the cell's centre
                const int C = ptw::kOrderCells;
                std::vector<uint32_t> table((size_t)C * C * C * 32, 0u);
                std::vector<std::pair<float, int>> byd(n_halves);
                for (int cz = 0; cz < C; ++cz)
                    for (int cy = 0; cy < C; ++cy)
                        for (int cx = 0; cx < C; ++cx) {
                            const int cc[3] = {cx, cy, cz};
                            float p[3];
                            for (int k = 0; k < 3; ++k) {
                                const bool any = hi[k] > lo[k];
                                p[k] = any ? (float)(lo[k] + (cc[k] + 0.5) * (hi[k] - lo[k]) / C) : 0.0f;
                            }
                            for (size_t h = 0; h < n_halves; ++h) {
                                const float dx = p[0] - half4[4 * h], dy = p[1] - half4[4 * h + 1], dz = p[2] - half4[4 * h + 2];
                                byd[h] = {std::sqrt(dx * dx + dy * dy + dz * dz) - half4[4 * h + 3], (int)h};
                            }
                            std::sort(byd.begin(), byd.end());
                            uint32_t* row = table.data() + ((size_t)(cz * C + cy) * C + cx) * 32;
                            for (size_t k = 0; k < n_halves; k += 2) row[k / 2] = (uint32_t)byd[k].second | ((uint32_t)byd[k + 1].second << 8);
                        }
                if ((rcg = s->d_grp_half_order.ensure(sizeof(uint32_t) * table.size()))) return rcg;
                HIP_TRY(hipMemcpy(s->d_grp_half_order.p, table.data(), sizeof(uint32_t) * table.size(), hipMemcpyHostToDevice));
                D.grp_half_order = (const uint32_t*)s->d_grp_half_order.p;
            }
            if ((rcg = s->d_grp_members.ensure(sizeof(DScreen2) * members.size()))) return rcg;
            if ((rcg = s->d_grp_bounds.ensure(sizeof(DScreen2) * bounds.size()))) return rcg;
            std::vector<DSlotExact> exact(order.size());
            for (size_t k = 0; k < order.size(); ++k) {
                DSlotExact& e = exact[k];
                memset(&e, 0, sizeof e);
                e.body = order[k];
                if (order[k] >= 0) {
                    const DGeom& g = geom[order[k]];
                    e.cx = g.cx;
                    e.cy = g.cy;
                    e.cz = g.cz;
                    e.r2 = g.nx;
                }
            }
            if ((rcg = s->d_grp_exact.ensure(sizeof(DSlotExact) * exact.size()))) return rcg;
            HIP_TRY(hipMemcpy(s->d_grp_members.p, members.data(), sizeof(DScreen2) * members.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(s->d_grp_bounds.p, bounds.data(), sizeof(DScreen2) * bounds.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(s->d_grp_exact.p, exact.data(), sizeof(DSlotExact) * exact.size(), hipMemcpyHostToDevice));
            D.grp_members = (const DScreen2*)s->d_grp_members.p;
            D.grp_bounds = (const DScreen2*)s->d_grp_bounds.p;
            D.grp_exact = (const DSlotExact*)s->d_grp_exact.p;
            D.n_groups = n_groups;
            for (int k = 0; k < 3; ++k) {  // box of the sphere centres (the whole scene when there are none)
                const bool any = hi[k] > lo[k];
                D.sort_mid[k] = any ? (float)(0.5 * (lo[k] + hi[k])) : 0.0f;
                D.sort_inv[k] = any ? (float)(2.0 / (hi[k] - lo[k])) : 1.0f;
            }
        }
        return PT_OK;
    }

    // Device copies of the lists and the scalar fields of DScene.
    int upload() {
        if (scan_screen.empty()) scan_screen.resize(1);
        if (scan_screen0.empty()) scan_screen0.resize(1);
        int rc;
        if ((rc = s->d_scan_planes.ensure(sizeof(DGeom) * scan_planes.size()))) return rc;
        if ((rc = s->d_scan_spheres.ensure(sizeof(DGeom) * scan_spheres.size()))) return rc;
        if ((rc = s->d_scan_screen.ensure(sizeof(DScreen2) * scan_screen.size()))) return rc;
        if ((rc = s->d_scan_screen0.ensure(sizeof(DScreen2) * scan_screen0.size()))) return rc;
        HIP_TRY(hipMemcpy(s->d_scan_planes.p, scan_planes.data(), sizeof(DGeom) * scan_planes.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_scan_spheres.p, scan_spheres.data(), sizeof(DGeom) * scan_spheres.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_scan_screen.p, scan_screen.data(), sizeof(DScreen2) * scan_screen.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_scan_screen0.p, scan_screen0.data(), sizeof(DScreen2) * scan_screen0.size(), hipMemcpyHostToDevice));
        if ((rc = s->d_geom.ensure(sizeof(DGeom) * n))) return rc;
        if ((rc = s->d_mat.ensure(sizeof(DMat) * n))) return rc;
        if ((rc = s->d_lights.ensure(sizeof(int32_t) * std::max<size_t>(1, s->lights.size())))) return rc;
        HIP_TRY(hipMemcpy(s->d_geom.p, geom.data(), sizeof(DGeom) * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_mat.p, mat.data(), sizeof(DMat) * n, hipMemcpyHostToDevice));
        if (!s->lights.empty())
            HIP_TRY(hipMemcpy(s->d_lights.p, s->lights.data(), sizeof(int32_t) * s->lights.size(), hipMemcpyHostToDevice));
        D.geom = (const DGeom*)s->d_geom.p;
        D.mat = (const DMat*)s->d_mat.p;
        D.lights = (const int32_t*)s->d_lights.p;
        D.scan_planes = (const DGeom*)s->d_scan_planes.p;
        D.scan_spheres = (const DGeom*)s->d_scan_spheres.p;
        D.scan_screen = (const DScreen2*)s->d_scan_screen.p;
        D.scan_screen0 = (const DScreen2*)s->d_scan_screen0.p;
        D.n_planes = n_planes;
        D.n_spheres = n_spheres;
        D.hit_eps = hit_eps;
        D.n_rects = (int)rects.size();
        if (!rects.empty()) {
            int rcr = s->d_scan_rects.ensure(sizeof(DRect) * rects.size());
            if (rcr) return rcr;
            HIP_TRY(hipMemcpy(s->d_scan_rects.p, rects.data(), sizeof(DRect) * rects.size(), hipMemcpyHostToDevice));
            D.scan_rects = (const DRect*)s->d_scan_rects.p;
        }
        D.n_rects_mt = (int)rects_mt.size();
        if (!rects_mt.empty()) {
            int rcr = s->d_scan_rects_mt.ensure(sizeof(DRectMT) * rects_mt.size());
            if (rcr) return rcr;
            HIP_TRY(hipMemcpy(s->d_scan_rects_mt.p, rects_mt.data(), sizeof(DRectMT) * rects_mt.size(), hipMemcpyHostToDevice));
            D.scan_rects_mt = (const DRectMT*)s->d_scan_rects_mt.p;
        }
        {
            // det > -EPSILON && det < EPSILON, EPSILON = 0.000001L (light_tracing.cpp:20,542): for a double
            // det that is |det| < e with e the smallest double not below the literal
            const long double lit = 0.000001L;
            double e = (double)lit;
            if ((long double)e < lit) e = std::nextafter(e, INFINITY);
            D.mt_eps = e;
        }
        if (s->has_measured) {
            // Bin edges of the table indices (MatusikBRDF.h:211-257) as thresholds, descending:
            // theta_half bin k starts at (k/90)^2 pi/2 and theta_diff bin k at k pi/180 (cosines),
            // fi_diff bin k at k pi/180 (cotangents).  Entry 0 is never read; the padding is never reached.
            std::vector<double> bins(kBinsTotal, -INFINITY);
            const double half_pi = M_PI / 2.0;
            for (int k = 1; k < 90; ++k) {
                bins[kBinsHalf + k] = std::cos(((double)k * k / 8100.0) * half_pi);
                bins[kBinsDiff + k] = std::cos((double)k * half_pi / 90.0);
            }
            for (int k = 1; k < 180; ++k) bins[kBinsPhi + k] = std::cos(k * M_PI / 180.0) / std::sin(k * M_PI / 180.0);
            int rcb = s->d_brdf_bins.ensure(sizeof(double) * bins.size());
            if (rcb) return rcb;
            HIP_TRY(hipMemcpy(s->d_brdf_bins.p, bins.data(), sizeof(double) * bins.size(), hipMemcpyHostToDevice));
            D.brdf_bins = (const double*)s->d_brdf_bins.p;
        }
        D.n_bodies = n;
        D.sphere_c1max = c1max * 1.0001f;
        D.sphere_r2max = r2max * 1.0001f;
        D.n_lights = (int)s->lights.size();
        D.width = s->width;
        D.height = s->height;
        D.estimator = s->estimator;
        D.flags = s->flags;
        return PT_OK;
    }

    // The direct term of next-event estimation is dead for the whole scene when emission*colour of
    // every light is exactly zero and the flux it would be scaled by is finite for every material
    // (no measured tables, glass ior >= 1, finite colours); light indices must fit 7 bits.
    // flux of the direct term can only be finite-by-construction in a scene of finite, moderate numbers
    void classify() {
        bool moderate = true;
        for (int i = 0; i < n; ++i) {
            const pt_body& b = s->bodies[i];
            auto ok = [](double v) { return std::isfinite(v) && std::fabs(v) < 1e100; };
            for (int k = 0; k < 3; ++k)
                moderate = moderate && ok(b.color[k]) && ok(b.emission[k]) && ok(b.center[k]) && ok(b.normal[k]);
            moderate = moderate && ok(b.radius);
        }
        for (int k = 0; k < 3; ++k)
            moderate = moderate && std::isfinite(s->camera.origin[k]) && std::isfinite(s->camera.topleft[k]) &&
                       std::isfinite(s->camera.topright[k]) && std::isfinite(s->camera.bottomleft[k]);
        D.flux_finite = moderate ? 1 : 0;
        bool dead = (s->estimator == PT_EST_NEE ? s->lights.size() >= 2 : s->estimator == PT_EST_BIDIR) && moderate;
        for (int i = 0; i < n && dead; ++i) {
            const pt_body& b = s->bodies[i];
            if (b.material == PT_MEASURED) dead = false;
            if (b.material == PT_GLASS && !(b.ior >= 1.0)) dead = false;
        }
        for (int li : s->lights)
            for (int k = 0; k < 3; ++k) dead = dead && (s->bodies[li].emission[k] * s->bodies[li].color[k] == 0.0);
        D.nee_emission_only = dead ? 1 : 0;
        // Absorbing emitters (see trace_path_nee_compact): emission, colour exactly (0, 0, 0).  A lit vertex
        // on one sets the radiance to its emission whatever lies deeper, provided the deeper radiance is
        // finite: colours and emissions up to 1e30 keep every product of five levels far from overflow.
        bool small = dead;
        for (int i = 0; i < n && small; ++i)
            for (int k = 0; k < 3; ++k)
                small = small && std::fabs(s->bodies[i].color[k]) <= 1e30 && std::fabs(s->bodies[i].emission[k]) <= 1e30;
        for (int i = 0; i < n; ++i) {
            const pt_body& b = s->bodies[i];
            mat[i].absorber = small && mat[i].emits && b.color[0] == 0.0 && b.color[1] == 0.0 && b.color[2] == 0.0;
        }
        D.seed = s->seed;
        for (int k = 0; k < 3; ++k) {
            D.cam_o[k] = s->camera.origin[k];
            D.cam_tl[k] = s->camera.topleft[k];
            D.cam_xd[k] = s->camera.topright[k] - s->camera.topleft[k];
            D.cam_yd[k] = s->camera.bottomleft[k] - s->camera.topleft[k];
        }
    }

    int run() {
        body_records();
        find_slabs();
        scan_lists();
        int rc = sphere_groups();
        if (rc == PT_OK) classify();  // before the upload: it marks the absorbing emitters in the material records
        if (rc == PT_OK) rc = upload();
        return rc;
    }
};

int upload_scene(pt_scene* s) { return SceneUpload(s).run(); }

int ensure_wavefront_buffers(pt_scene* s) {
    const size_t J = (size_t)std::max(1, s->part.owned_tiles) * kBlock;
    const size_t cap = J * (size_t)s->wf_batch_alloc;
    if (s->wf.cap == cap && s->d_wf.p && s->d_wf_sort.p && s->wf_counts_host) return PT_OK;
    s->wf.cap = 0;  // not usable until every allocation and query below has succeeded
    // per path: a ray record, kMaxDepth vertex records; doubles: xy 2; 32-bit words: rng 4, nrec 1, light 1,
    // need 1, queue 11, sorted 1, keys 1 + 5 = 25; plus the 16 queue counters
    size_t bytes = cap * (sizeof(ptw::RayRec) + kMaxDepth * sizeof(ptw::VertexRec) + 2 * sizeof(double) + 25 * sizeof(int32_t)) +
                   16 * sizeof(int32_t);
    int rc = s->d_wf.ensure(bytes);
    if (rc) return rc;
    ptw::WfState& W = s->wf;
    W.rays = (ptw::RayRec*)s->d_wf.p;  // 256-byte aligned, like every allocation
    W.verts = (ptw::VertexRec*)(W.rays + cap);
    W.xy = (double*)(W.verts + cap * kMaxDepth);
    W.rng = (uint4*)(W.xy + 2 * cap);  // 16-byte aligned: cap is a multiple of 256
    int32_t* w = (int32_t*)(W.rng + cap);
    W.nrec = w;
    W.light = W.nrec + cap;
    W.need = W.light + cap;
    W.queue = W.need + cap;
    W.qcount = W.queue + 11 * cap;
    W.sorted = W.qcount + 16;
    W.keys = (uint32_t*)(W.sorted + cap);
    W.skeys = W.keys + cap;
    // the ray sort (rocprim radix sort of 16-bit keys with the path ids as values)
    size_t tmp = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, W.keys, W.keys, W.queue, W.sorted, cap, 0u, ptw::kSortKeyBits, (hipStream_t)0));
    s->wf_sort_tmp_bytes = tmp;
    if ((rc = s->d_wf_sort.ensure(cap * sizeof(uint32_t) + ((tmp + 255) & ~(size_t)255) + 256))) return rc;
    if (!s->wf_counts_host) {
        HIP_TRY(hipHostMalloc((void**)&s->wf_counts_host, 16 * sizeof(int32_t), hipHostMallocDefault));
        for (int k = 0; k < 16; ++k) s->wf_counts_host[k] = -1;
    }
    s->wf_counts_cap = 0;  // no counts for this capacity yet
    W.cap = cap;
    return PT_OK;
}

// One batch through the wavefront pipeline (see pt_wavefront.hpp).
int render_wavefront(pt_scene* s, hipStream_t st, int it0, int B, const double* xs, const double* ys, const int32_t* rs,
                     const int32_t* npix, double* samples, size_t J) {
    using namespace ptw;
    const WfState W = s->wf;
    const Partition P = s->part;
    const int nblk = (int)((J * (size_t)B + kWfBlock - 1) / kWfBlock);
    const int nblk_push = (int)((J * (size_t)B + kPushBlock - 1) / kPushBlock);
    const int nblk_ext = (int)((J * (size_t)B + kExtBlock - 1) / kExtBlock);
    HIP_TRY(hipMemsetAsync(W.qcount, 0, 16 * sizeof(int32_t), st));
    const int cap = (int)(J * (size_t)B);  // paths of this batch = positions of every depth's gather entries
    // The library sort takes its size from the host.  The gather entries of a depth are one per position
    // (dead ones carry the end key), so their sort covers the batch.  Of a shadow queue, whose count is
    // known on the device only, the first n entries are sorted, n from the count the same queue had in an
    // earlier batch (plus a margin; everything while no count has arrived): a scan reads entry i from the
    // sorted list when i < n and from the queue itself otherwise, so any n gives the same results and a
    // poor guess only costs coherence.
    auto sort_size = [&](int mode, int qi) -> int {
        if (const char* f = getenv("PTGPU_SORT_FRACTION"))  // tests: a fixed part of the capacity, whatever the counts
            return (int)std::min<double>((double)cap, std::max(0.0, atof(f)) * (double)cap);
        if (mode == kGather) return cap;
        const int before = s->wf_counts_host ? s->wf_counts_host[qi] : -1;  // may change under us; any value will do
        if (before < 0 || s->wf_counts_cap <= 0) return cap;
        const double frac = (double)before / (double)s->wf_counts_cap;
        return (int)std::min<double>((double)cap, frac * 1.03 * (double)cap + 4096.0);
    };
    // A sort the library refuses (it reports bad arguments or too little temporary storage through its
    // return value only) leaves n_sorted at 0: the scan then reads the queue in its own order, same results.
    auto sorted_list = [&](int mode, int qi, int gi, int* n_sorted) -> const int32_t* {
        *n_sorted = 0;
        if (!s->wf_sort) return W.sorted;
        const int n = sort_size(mode, qi);
        const uint32_t* keys_in = mode == kShadow ? W.skeys + (size_t)gi * W.cap : W.keys;
        uint32_t* keys_out = (uint32_t*)s->d_wf_sort.p;
        void* tmp = (char*)s->d_wf_sort.p + ((W.cap * sizeof(uint32_t) + 255) & ~(size_t)255);
        size_t tmp_bytes = s->wf_sort_tmp_bytes;
        const unsigned bits = mode == kShadow ? kShadowKeyBits : kSortKeyBits;
        size_t need = 0;  // the library picks its algorithm by size: ask what this call needs
        if (rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, W.queue + (size_t)qi * W.cap, W.sorted, (size_t)n, 0u, bits,
                                      st) != hipSuccess || need > tmp_bytes)
            return W.sorted;
        if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, W.queue + (size_t)qi * W.cap, W.sorted, (size_t)n, 0u, bits,
                                      st) != hipSuccess) {
            (void)hipGetLastError();
            return W.sorted;
        }
        *n_sorted = n;
        return W.sorted;
    };
    // depth 0: the camera rays, in pixel order; every stage writes the sort keys of the next depth's
    // entries (its own threads' positions) when that depth will be sorted
    if (s->wf_sort)
        hipLaunchKernelGGL(k_wf_primary<true>, dim3(P.owned_tiles * 4, B), dim3(kExtBlock), 0, st, s->ds, P, it0, B, xs, ys, rs, npix, s->xstep, W, J);
    else
        hipLaunchKernelGGL(k_wf_primary<false>, dim3(P.owned_tiles * 4, B), dim3(kExtBlock), 0, st, s->ds, P, it0, B, xs, ys, rs, npix, s->xstep, W, J);
    for (int depth = 1; depth < kMaxDepth; ++depth) {
        int n_sorted = 0;
        const int32_t* list = sorted_list(kGather, depth, 0, &n_sorted);
        if (s->wf_sort && depth + 1 < kMaxDepth)
            hipLaunchKernelGGL(k_wf_bounce<true>, dim3(nblk_ext), dim3(kExtBlock), 0, st, s->ds, W, depth, list, n_sorted, cap);
        else
            hipLaunchKernelGGL(k_wf_bounce<false>, dim3(nblk_ext), dim3(kExtBlock), 0, st, s->ds, W, depth, list, n_sorted, cap);
    }
    if (s->estimator == PT_EST_NEE) {
        if (s->wf_sort) {
            HIP_TRY(hipMemsetAsync(W.skeys, 0xFF, (size_t)kMaxDepth * W.cap * sizeof(uint32_t), st));
            hipLaunchKernelGGL(k_wf_light<true>, dim3(nblk_push), dim3(kPushBlock), 0, st, s->ds, W);
        } else {
            hipLaunchKernelGGL(k_wf_light<false>, dim3(nblk_push), dim3(kPushBlock), 0, st, s->ds, W);
        }
        for (int gi = kMaxDepth - 1; gi >= 0; --gi) {
            int n_sorted = 0;
            const int32_t* list = sorted_list(kShadow, kShadowQueue0 + gi, gi, &n_sorted);
            hipLaunchKernelGGL(k_wf_shadow, dim3(nblk_ext), dim3(kExtBlock), 0, st, s->ds, W, kShadowQueue0 + gi, gi, list, n_sorted);
        }
        if (s->has_measured)
            hipLaunchKernelGGL(k_wf_nee_finish<true>, dim3(nblk), dim3(kWfBlock), 0, st, s->ds, W, samples, J);
        else
            hipLaunchKernelGGL(k_wf_nee_finish<false>, dim3(nblk), dim3(kWfBlock), 0, st, s->ds, W, samples, J);
    } else {
        hipLaunchKernelGGL(k_wf_finish, dim3(nblk), dim3(kWfBlock), 0, st, s->ds, W, samples, J);
    }
    HIP_TRY(hipGetLastError());
    if (s->wf_counts_host && (s->wf_counts_cap == 0 || s->wf_counts_cap == cap)) {  // (a shorter last batch is no measure)
        HIP_TRY(hipMemcpyAsync(s->wf_counts_host, W.qcount, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        s->wf_counts_cap = cap;
    }
    if (getenv("PTGPU_PRINT_QCOUNTS")) {  // development: the queue counts of this batch
        int32_t qc[16];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(qc, W.qcount, sizeof qc, hipMemcpyDeviceToHost));
        fprintf(stderr, "batch of %d paths: gather", cap);
        for (int k = 0; k <= kMaxDepth; ++k) fprintf(stderr, " %d", qc[k]);
        fprintf(stderr, "; shadow");
        for (int k = 0; k < kMaxDepth; ++k) fprintf(stderr, " %d", qc[kShadowQueue0 + k]);
        fprintf(stderr, "\n");
    }
    return PT_OK;
}

// Queues the fused estimator kernel of the scene's estimator and material mix for one batch of B
// iterations.  The hardware hands the blocks of a grid to the eight dies in turn (block i to die
// i % 8) and strictly in order, so a die whose blocks are dearer than the others' holds all of them
// up; with one block per (tile, iteration) and a tile row of 8k tiles every die would own the same
// columns of the picture (measured at 1024x768: 4 % lower average occupancy than at 3840x2160, the
// same instructions per path).  The grid therefore keeps eight consecutive blocks on ONE tile, in
// eight different iterations: the dies work through statistically identical shares in step
// (1024x768: +5.7 % for the 15-body scene, +5.6 % for the 32-body one, +5.3 % with the direct term
// live; 3840x2160 unchanged).  Blocks past the last iteration of a batch that is no multiple of eight
// return at once; each still waits its turn for a block slot, so batches are whole groups
// (setup_partition) and a batch shorter than eight iterations uses smaller groups.
void launch_trace(const pt_scene* s, hipStream_t st, const Partition& P0, int it0, int B, const double* xs, const double* ys,
                  const int32_t* rs, const int32_t* npix, double* samples, size_t J) {
    Partition P = P0;
    P.group_log2 = kXcdsLog2;
    while (P.group_log2 > 0 && (1 << P.group_log2) > B) P.group_log2--;
    const int G = 1 << P.group_log2;
    const dim3 grid_tile((unsigned)P.owned_tiles * G, (unsigned)((B + G - 1) / G));
    const dim3 grid_pt(grid_tile.x * (kBlock / trace_block(kEstPT)), grid_tile.y);
#define PT_LAUNCH_TRACE(EST_, ...) \
    hipLaunchKernelGGL((k_trace<EST_, __VA_ARGS__>), EST_ == kEstPT ? grid_pt : grid_tile, dim3(trace_block(EST_)), 0, st, s->ds, P, it0, B, xs, ys, rs, npix, s->xstep, samples, J)
    if (s->estimator == PT_EST_BIDIR)
        PT_LAUNCH_TRACE(kEstBidir, false);
    else if (s->estimator == PT_EST_NEE && s->ds.nee_emission_only)
        PT_LAUNCH_TRACE(kEstNEE, false, true);
    else if (s->estimator == PT_EST_NEE && s->has_measured)
        PT_LAUNCH_TRACE(kEstNEE, true);
    else if (s->estimator == PT_EST_NEE)
        PT_LAUNCH_TRACE(kEstNEE, false);
    else
        PT_LAUNCH_TRACE(kEstPT, false);
#undef PT_LAUNCH_TRACE
}

int check_measured_tables(const pt_scene* s) {
    for (const pt_body& b : s->bodies)
        if (b.material == PT_MEASURED) {
            if (b.brdf_id < 0 || b.brdf_id >= PT_MAX_BRDF) return fail(PT_ERR_INVALID, "measured body: brdf_id out of range");
            if (!s->d_brdf[b.brdf_id].p) return fail(PT_ERR_INVALID, "measured body: BRDF table not uploaded");
        }
    return PT_OK;
}

}  // namespace

extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }

int pt_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    g_device = device;
    return PT_OK;
}

int pt_scene_create(const pt_scene_desc* desc, pt_scene** out) {
    if (!desc || !out) return fail(PT_ERR_INVALID, "pt_scene_create: null argument");
    if (desc->width < 1 || desc->height < 1) return fail(PT_ERR_INVALID, "pt_scene_create: width/height must be >= 1");
    if ((int64_t)desc->width * desc->height > (int64_t)1 << 28) return fail(PT_ERR_INVALID, "pt_scene_create: frame too large");
    if (desc->body_count < 1 || !desc->bodies) return fail(PT_ERR_INVALID, "pt_scene_create: no bodies");
    if (desc->estimator != PT_EST_PT && desc->estimator != PT_EST_NEE && desc->estimator != PT_EST_BIDIR)
        return fail(PT_ERR_INVALID, "pt_scene_create: estimator");
    pt_scene* s = new (std::nothrow) pt_scene();
    if (!s) return fail(PT_ERR_NOMEM, "pt_scene_create: out of host memory");
    s->device = g_device;
    hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) {
        delete s;
        return fail(PT_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    }
    s->width = desc->width;
    s->height = desc->height;
    s->estimator = desc->estimator;
    s->flags = desc->flags;
    s->seed = desc->seed;
    s->camera = desc->camera;
    s->bodies.assign(desc->bodies, desc->bodies + desc->body_count);
    for (size_t i = 0; i < s->bodies.size(); ++i) {
        pt_body& b = s->bodies[i];
        if (b.shape != PT_SPHERE && b.shape != PT_PLANE && b.shape != PT_RECTANGLE && b.shape != PT_RECTANGLE_MT) {
            delete s;
            return fail(PT_ERR_INVALID, "pt_scene_create: unknown shape");
        }
        if (b.shape == PT_RECTANGLE || b.shape == PT_RECTANGLE_MT) {
            s->has_rectangles = true;
            if (s->estimator != PT_EST_PT) {
                delete s;
                return fail(PT_ERR_UNSUPPORTED, "pt_scene_create: rectangles exist only under plain path tracing (PT_EST_PT)");
            }
        }
        if (b.material < PT_DIFFUSE || b.material > PT_MEASURED) {
            delete s;
            return fail(PT_ERR_INVALID, "pt_scene_create: unknown material");
        }
        pt_body_construct(&b);
        if (b.material == PT_MEASURED) s->has_measured = true;
        // Renderer ctor light list next_event.cpp:362-367
        if (b.emission[0] >= 1.0 || b.emission[1] >= 1.0 || b.emission[2] >= 1.0) s->lights.push_back((int32_t)i);
    }
    if ((s->flags & PT_F_RAYTRACE3) && s->estimator != PT_EST_PT) {
        delete s;
        return fail(PT_ERR_UNSUPPORTED, "pt_scene_create: PT_F_RAYTRACE3 applies to PT_EST_PT");
    }
    if (s->estimator != PT_EST_PT && s->lights.empty()) {
        delete s;
        return fail(PT_ERR_INVALID, "pt_scene_create: next-event and bidirectional estimation need a body with emission >= 1");
    }
    if (s->estimator == PT_EST_BIDIR && (int)s->bodies.size() > kMaxBodiesFused) {
        delete s;
        return fail(PT_ERR_UNSUPPORTED, "pt_scene_create: the bidirectional estimator runs the fused pipeline, at most 127 bodies");
    }
    // Loop steps of Renderer::iterate: double xstep = 1.0L / (double)width (main.cpp:352,355)
    s->xstep = (double)(1.0L / (long double)(double)s->width);
    s->ystep = (double)(1.0L / (long double)(double)s->height);
    s->part.row_lo = 0;
    s->part.row_hi = s->height;
    int rc = setup_partition(s, 0, 1, 32, 8);
    if (rc == PT_OK) rc = upload_scene(s);
    if (rc == PT_OK && s->estimator == PT_EST_BIDIR && !s->ds.nee_emission_only)
        rc = fail(PT_ERR_UNSUPPORTED,
                  "pt_scene_create: the bidirectional estimator is defined only where emission*colour of every light is zero "
                  "(the reference scales that term by a value read past the end of flux_lookup, bidirectional.cpp:377-382,484), "
                  "with no measured body, every glass ior >= 1 and finite, moderate numbers");
    if (rc != PT_OK) {
        pt_scene_destroy(s);
        return rc;
    }
    *out = s;
    return PT_OK;
}

int pt_scene_destroy(pt_scene* s) {
    if (!s) return PT_OK;
    (void)hipSetDevice(s->device);
    s->d_geom.release();
    s->d_mat.release();
    s->d_grp_members.release();
    s->d_grp_bounds.release();
    s->d_grp_supers.release();
    s->d_grp_halves.release();
    s->d_grp_half4.release();
    s->d_grp_half_order.release();
    s->d_grp_exact.release();
    s->d_scan_planes.release();
    s->d_scan_spheres.release();
    s->d_scan_screen.release();
    s->d_scan_screen0.release();
    s->d_brdf_bins.release();
    s->d_scan_rects.release();
    s->d_scan_rects_mt.release();
    s->d_lights.release();
    for (auto& b : s->d_brdf) b.release();
    for (int k = 0; k < 2; ++k) {
        s->d_xs[k].release();
        s->d_ys[k].release();
        s->d_rowstart[k].release();
        s->d_npix[k].release();
        s->d_samples[k].release();
        if (s->ev_layout[k]) (void)hipEventDestroy(s->ev_layout[k]);
        if (s->ev_trace[k]) (void)hipEventDestroy(s->ev_trace[k]);
        if (s->ev_acc[k]) (void)hipEventDestroy(s->ev_acc[k]);
    }
    if (s->ev_entry) (void)hipEventDestroy(s->ev_entry);
    if (s->st_layout) (void)hipStreamDestroy(s->st_layout);
    if (s->st_accum) (void)hipStreamDestroy(s->st_accum);
    s->d_accum.release();
    s->d_wf.release();
    s->d_wf_sort.release();
    if (s->wf_counts_host) (void)hipHostFree(s->wf_counts_host);
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    delete s;
    return PT_OK;
}

int pt_scene_bodies(const pt_scene* s, pt_body* out, int cap) {
    if (!s || (!out && cap > 0)) return fail(PT_ERR_INVALID, "pt_scene_bodies: null argument");
    int n = (int)s->bodies.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = s->bodies[i];
    return n;
}

int pt_brdf_upload(pt_scene* s, int id, const double* table, size_t n) {
    if (!s || !table) return fail(PT_ERR_INVALID, "pt_brdf_upload: null argument");
    if (id < 0 || id >= PT_MAX_BRDF) return fail(PT_ERR_INVALID, "pt_brdf_upload: id out of range");
    if (n != (size_t)PT_BRDF_TABLE_DOUBLES) return fail(PT_ERR_INVALID, "pt_brdf_upload: table must hold 3*90*90*180 doubles");
    HIP_TRY(hipSetDevice(s->device));
    int rc = s->d_brdf[id].ensure(n * sizeof(double));
    if (rc) return rc;
    // The file keeps three planes (red, green, blue; MatusikBRDF.h:68-72).  On the device the three
    // values of an index are adjacent, so that a lookup touches one or two 64-byte sectors, not three,
    // and carry the channel scales of the lookup already.
    std::vector<double> rgb(n);
    const size_t N = n / 3;
    for (size_t i = 0; i < N; ++i) ptd::brdf_entry(table[i], table[N + i], table[2 * N + i], &rgb[3 * i]);
    HIP_TRY(hipMemcpy(s->d_brdf[id].p, rgb.data(), n * sizeof(double), hipMemcpyHostToDevice));
    s->ds.brdf[id] = (const double*)s->d_brdf[id].p;
    return PT_OK;
}

int pt_brdf_load_file(pt_scene* s, int id, const char* path) {
    if (!s || !path) return fail(PT_ERR_INVALID, "pt_brdf_load_file: null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(PT_ERR_IO, std::string("cannot open ") + path);
    int32_t dims[3];
    if (fread(dims, sizeof(int32_t), 3, f) != 3) {
        fclose(f);
        return fail(PT_ERR_IO, "short read of BRDF header");
    }
    long long n = (long long)dims[0] * dims[1] * dims[2];
    if (n != 90 * 90 * 180) {  // MatusikBRDF.h:92-100
        fclose(f);
        return fail(PT_ERR_IO, "Dimensions don't match");
    }
    std::vector<double> t((size_t)PT_BRDF_TABLE_DOUBLES);
    size_t got = fread(t.data(), sizeof(double), t.size(), f);
    fclose(f);
    if (got != t.size()) return fail(PT_ERR_IO, "short read of BRDF table");
    return pt_brdf_upload(s, id, t.data(), t.size());
}

int pt_set_partition(pt_scene* s, int rank, int world, int tile_w, int tile_h) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    return setup_partition(s, rank, world, tile_w, tile_h);
}

int pt_set_rows(pt_scene* s, int start_row, int row_count) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (start_row < 0 || row_count < 0 || (int64_t)start_row + row_count > s->height)
        return fail(PT_ERR_INVALID, "pt_set_rows: rows outside the frame");
    s->part.row_lo = start_row;
    s->part.row_hi = start_row + row_count;
    return PT_OK;
}

int pt_tiles_per_rank(const pt_scene* s, int world) {
    if (!s || world < 1) return fail(PT_ERR_INVALID, "pt_tiles_per_rank: bad argument");
    const int total = s->part.tiles_x * s->part.tiles_y;
    return (total + world - 1) / world;
}

int pt_tiles_pack(pt_scene* s, const double* accum_dev, double* packed_dev, int tiles_capacity, void* stream) {
    if (!s || !accum_dev || !packed_dev) return fail(PT_ERR_INVALID, "pt_tiles_pack: null argument");
    if (tiles_capacity < s->part.owned_tiles) return fail(PT_ERR_INVALID, "pt_tiles_pack: capacity below the owned tile count");
    if (tiles_capacity == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipLaunchKernelGGL(k_tiles_pack, dim3(tiles_capacity), dim3(kBlock), 0, (hipStream_t)stream, s->part, s->width, s->height,
                       tiles_capacity, accum_dev, packed_dev);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_tiles_unpack(pt_scene* s, int world, int tiles_per_rank, const double* gathered_dev, double* accum_dev, void* stream) {
    if (!s || !gathered_dev || !accum_dev || world < 1) return fail(PT_ERR_INVALID, "pt_tiles_unpack: bad argument");
    Partition P = s->part;
    if (world != P.world) {  // the layout of `world` ranks with this handle's tile shape
        P.world = world;
        P.rank = 0;
    }
    const int total = P.tiles_x * P.tiles_y;
    if (tiles_per_rank < (total + world - 1) / world) return fail(PT_ERR_INVALID, "pt_tiles_unpack: tiles_per_rank too small");
    HIP_TRY(hipSetDevice(s->device));
    hipLaunchKernelGGL(k_tiles_unpack, dim3((total + world - 1) / world, world), dim3(kBlock), 0, (hipStream_t)stream, P,
                       s->width, s->height, tiles_per_rank, gathered_dev, accum_dev);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_set_pipeline(pt_scene* s, int pipeline) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (pipeline != PT_PIPELINE_AUTO && pipeline != PT_PIPELINE_FUSED && pipeline != PT_PIPELINE_WAVEFRONT)
        return fail(PT_ERR_INVALID, "pt_set_pipeline: unknown pipeline");
    s->pipeline = pipeline;
    return PT_OK;
}

int pt_render_device(pt_scene* s, int first_iteration, int n_iterations, double* accum_dev, void* stream_) {
    if (!s || !accum_dev) return fail(PT_ERR_INVALID, "pt_render_device: null argument");
    if (n_iterations < 0 || first_iteration < 0) return fail(PT_ERR_INVALID, "pt_render_device: negative iteration range");
    HIP_TRY(hipSetDevice(s->device));
    int rc = check_measured_tables(s);
    if (rc) return rc;
    // a call of few iterations does not need the state of a full batch (23 GB for large scenes)
    if (uses_wavefront(s))
        s->wf_batch_alloc = std::max(s->wf_batch_alloc, std::min(s->wf_batch, std::max(1, n_iterations)));
    else
        s->batch_alloc = std::max(s->batch_alloc, std::min(s->batch, std::max(1, n_iterations)));
    if ((rc = ensure_render_buffers(s))) return rc;
    hipStream_t st = (hipStream_t)stream_;
    const Partition P = s->part;
    s->ev_used = 0;
    s->last_paths = 0;
    if (P.owned_tiles == 0 || n_iterations == 0) return PT_OK;
    const int W = s->width, H = s->height;
    const size_t J = (size_t)P.owned_tiles * kBlock;
    const bool use_wavefront = uses_wavefront(s);
    if (!use_wavefront && (int)s->bodies.size() > kMaxBodiesFused)
        return fail(PT_ERR_UNSUPPORTED, "the fused pipeline handles at most 127 bodies");
    if (use_wavefront && s->estimator == PT_EST_BIDIR)
        return fail(PT_ERR_UNSUPPORTED, "the bidirectional estimator runs the fused pipeline only");
    const int batch = use_wavefront ? s->wf_batch_alloc : s->batch_alloc;
    if (use_wavefront && (rc = ensure_wavefront_buffers(s))) return rc;
    const int nbatch = (n_iterations + batch - 1) / batch;
    // Three streams: the caller's stream runs the estimator kernels back to back; the frame layout
    // of the next batch runs ahead on st_layout, the accumulation of the previous batch trails on
    // st_accum.  Batch k uses buffer set k & 1.  Everything is ordered after the work already
    // queued on the caller's stream, and the caller's stream ends up waiting for the last accumulation.
    HIP_TRY(hipEventRecord(s->ev_entry, st));
    HIP_TRY(hipStreamWaitEvent(s->st_layout, s->ev_entry, 0));
    HIP_TRY(hipStreamWaitEvent(s->st_accum, s->ev_entry, 0));
    auto launch_layout = [&](int k) -> int {
        const int q = k & 1;
        const int B = std::min(batch, n_iterations - k * batch);
        if (k >= 2) {
            // batch k-2 has read these tables: its estimator kernel and its accumulation (npix)
            HIP_TRY(hipStreamWaitEvent(s->st_layout, s->ev_trace[q], 0));
            HIP_TRY(hipStreamWaitEvent(s->st_layout, s->ev_acc[q], 0));
        }
        launch_frame_layout(s->st_layout, s->seed, W, H, first_iteration + k * batch, B, s->xstep, s->ystep, (double*)s->d_xs[q].p,
                            (double*)s->d_ys[q].p, (int32_t*)s->d_rowstart[q].p, (int32_t*)s->d_npix[q].p,
                            (int)(s->flags & PT_F_PIXEL_GRID));
        HIP_TRY(hipEventRecord(s->ev_layout[q], s->st_layout));
        return PT_OK;
    };
    if ((rc = launch_layout(0))) return rc;
    for (int k = 0; k < nbatch; ++k) {
        const int q = k & 1;
        const int B = std::min(batch, n_iterations - k * batch);
        const int it0 = first_iteration + k * batch;
        const double* xs = (const double*)s->d_xs[q].p;
        const double* ys = (const double*)s->d_ys[q].p;
        const int32_t* rs = (const int32_t*)s->d_rowstart[q].p;
        const int32_t* npix = (const int32_t*)s->d_npix[q].p;
        double* samples = (double*)s->d_samples[q].p;
        if (k + 1 < nbatch && (rc = launch_layout(k + 1))) return rc;
        HIP_TRY(hipStreamWaitEvent(st, s->ev_layout[q], 0));
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(st, s->ev_acc[q], 0));  // batch k-2's samples have been consumed
        if (s->ev.size() < s->ev_used + 2) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            s->ev.push_back(a);
            s->ev.push_back(b);
        }
        HIP_TRY(hipEventRecord(s->ev[s->ev_used], st));
        if (use_wavefront) {
            int rcw = render_wavefront(s, st, it0, B, xs, ys, rs, npix, samples, J);
            if (rcw) return rcw;
        } else
            launch_trace(s, st, P, it0, B, xs, ys, rs, npix, samples, J);
        HIP_TRY(hipEventRecord(s->ev[s->ev_used + 1], st));
        s->ev_used += 2;
        HIP_TRY(hipEventRecord(s->ev_trace[q], st));
        HIP_TRY(hipStreamWaitEvent(s->st_accum, s->ev_trace[q], 0));
        // beside the estimator of the next batch: few blocks that walk the tiles (see kAccumulateBlocks)
        hipLaunchKernelGGL(k_accumulate, dim3(std::min(P.owned_tiles, kAccumulateBlocks)), dim3(kBlock), 0, s->st_accum, P, W, H, B, npix,
                           samples, J, accum_dev);
        HIP_TRY(hipEventRecord(s->ev_acc[q], s->st_accum));
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamWaitEvent(st, s->ev_acc[(nbatch - 1) & 1], 0));
    // destination pixels of this partition inside the image, times iterations (a short frame leaves up
    // to W + H of its last slots without a sample; those few are counted all the same)
    double owned_pixels = 0;
    for (int q = 0; q < P.owned_tiles; ++q) {
        const int tile = q * P.world + P.rank;
        const int ty = tile / P.tiles_x;
        const int x0 = ((tile - ty * P.tiles_x + ty) % P.tiles_x) * P.tile_w, y0 = ty * P.tile_h;  // as slot_of places the tile
        const int y1 = std::min(std::min(y0 + P.tile_h, H), P.row_hi), y00 = std::max(y0, P.row_lo);
        owned_pixels += (double)std::max(0, std::min(P.tile_w, W - x0)) * (double)std::max(0, y1 - y00);
    }
    s->last_paths = (double)n_iterations * owned_pixels;
    return PT_OK;
}

int pt_render(pt_scene* s, int first_iteration, int n_iterations, double* accum_host) {
    if (!s || !accum_host) return fail(PT_ERR_INVALID, "pt_render: null argument");
    HIP_TRY(hipSetDevice(s->device));
    size_t bytes = (size_t)s->width * s->height * 3 * sizeof(double);
    int rc = s->d_accum.ensure(bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(s->d_accum.p, accum_host, bytes, hipMemcpyHostToDevice));
    rc = pt_render_device(s, first_iteration, n_iterations, (double*)s->d_accum.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(accum_host, s->d_accum.p, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

// The fused pipeline with every work buffer carved out of the device's fixed scratch array and every
// kernel queued on the caller's stream, batch after batch: no hipMalloc / hipFree, no stream or
// event creation.  Nothing in the scratch has to start at zero: the layout kernels write every table
// entry that is read afterwards and k_trace writes every sample slot.
int pt_render_device_noalloc(pt_scene* s, int first_iteration, int n_iterations, double* accum_dev, void* stream_) {
    if (!s || !accum_dev) return fail(PT_ERR_INVALID, "pt_render_device_noalloc: null argument");
    if (n_iterations < 0 || first_iteration < 0) return fail(PT_ERR_INVALID, "pt_render_device_noalloc: negative iteration range");
    HIP_TRY(hipSetDevice(s->device));
    int rc = check_measured_tables(s);
    if (rc) return rc;
    if ((int)s->bodies.size() > kMaxBodiesFused || s->pipeline == PT_PIPELINE_WAVEFRONT)
        return fail(PT_ERR_UNSUPPORTED, "pt_render_device_noalloc: fused pipeline only (at most 127 bodies)");
    // this device's copy of the scratch array: asked for once per device, at the first call that needs it
    // (scenes that never come here do not depend on it), with the scene's device current
    if (!s->scratch_ptr && (rc = scratch_address(s->device, &s->scratch_ptr))) return rc;
    hipStream_t st = (hipStream_t)stream_;
    const Partition P = s->part;
    if (P.owned_tiles == 0 || n_iterations == 0) return PT_OK;
    const int W = s->width, H = s->height;
    const size_t J = (size_t)P.owned_tiles * kBlock;
    const size_t CK = (size_t)(W + kXStride - 1) / kXStride;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // bytes of one iteration's tables and samples, each array 256-byte aligned for any batch size
    const size_t per_it = (size_t)H * CK * 8 + (size_t)H * 8 + (size_t)(H + 1) * 4 + 4 + 3 * J * 8;
    const size_t slack = 5 * 256;
    if (per_it + slack > kScratchBytes)
        return fail(PT_ERR_UNSUPPORTED, "pt_render_device_noalloc: one iteration of this frame needs " + std::to_string(per_it + slack) +
                                            " bytes of work buffers, the fixed scratch holds " + std::to_string(kScratchBytes));
    int batch = (int)std::min<size_t>((size_t)std::max(1, s->batch), (kScratchBytes - slack) / per_it);
    if (batch >= (1 << kXcdsLog2)) batch &= ~((1 << kXcdsLog2) - 1);  // whole groups, see launch_trace
    unsigned char* base = (unsigned char*)s->scratch_ptr;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        void* p = base + off;
        off += up(bytes);
        return p;
    };
    double* xs = (double*)carve((size_t)batch * H * CK * 8);
    double* ys = (double*)carve((size_t)batch * H * 8);
    int32_t* rs = (int32_t*)carve((size_t)batch * (H + 1) * 4);
    int32_t* npix = (int32_t*)carve((size_t)batch * 4);
    double* samples = (double*)carve((size_t)batch * 3 * J * 8);
    if (off > kScratchBytes) return fail(PT_ERR_UNSUPPORTED, "pt_render_device_noalloc: work buffers exceed the fixed scratch");
    for (int it0 = first_iteration, left = n_iterations; left > 0; it0 += batch, left -= batch) {
        const int B = std::min(batch, left);
        launch_frame_layout(st, s->seed, W, H, it0, B, s->xstep, s->ystep, xs, ys, rs, npix, (int)(s->flags & PT_F_PIXEL_GRID));
        launch_trace(s, st, P, it0, B, xs, ys, rs, npix, samples, J);
        hipLaunchKernelGGL(k_accumulate, dim3(P.owned_tiles), dim3(kBlock), 0, st, P, W, H, B, npix, samples, J, accum_dev);
        HIP_TRY(hipGetLastError());
    }
    return PT_OK;
}

int pt_get_stats(const pt_scene* s, pt_stats* out) {
    if (!s || !out) return fail(PT_ERR_INVALID, "pt_get_stats: null argument");
    HIP_TRY(hipSetDevice(s->device));
    double ms = 0;
    for (size_t i = 0; i + 1 < s->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(s->ev[i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, s->ev[i], s->ev[i + 1]));
        ms += t;
    }
    out->paths = s->last_paths;
    out->kernel_ms = ms;
    out->launches = (double)(s->ev_used / 2);
    return PT_OK;
}

int pt_intersect_batch(pt_scene* s, const double* rays6, int n, int skip, int32_t* idx, double* t) {
    if (!s || !rays6 || !idx || !t || n < 0) return fail(PT_ERR_INVALID, "pt_intersect_batch: bad argument");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    Tmp<double> d_r, d_t;
    Tmp<int32_t> d_i;
    HIP_TRY(d_r.alloc((size_t)6 * n));
    HIP_TRY(d_t.alloc((size_t)n));
    HIP_TRY(d_i.alloc((size_t)n));
    HIP_TRY(hipMemcpy(d_r, rays6, sizeof(double) * 6 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_intersect_batch, dim3((n + 255) / 256), dim3(256), 0, 0, s->ds, d_r, n, skip, d_i, d_t);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(idx, d_i, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, d_t, sizeof(double) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_intersect_all(pt_scene* s, const double* rays6, int n, double* t_all) {
    if (!s || !rays6 || !t_all || n < 0) return fail(PT_ERR_INVALID, "pt_intersect_all: bad argument");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    const size_t nb = s->bodies.size();
    Tmp<double> d_r, d_t;
    HIP_TRY(d_r.alloc((size_t)6 * n));
    HIP_TRY(d_t.alloc((size_t)n * nb));
    HIP_TRY(hipMemcpy(d_r, rays6, sizeof(double) * 6 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_intersect_all, dim3((n + 255) / 256), dim3(256), 0, 0, s->ds, d_r, n, d_t);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(t_all, d_t, sizeof(double) * n * nb, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_trace_paths(pt_scene* s, const int32_t* it_idx2, const double* xy, int n, double* rgb, int32_t* nrec,
                   int32_t* bodies, double* points, int32_t* visible, int32_t* used) {
    if (!s || !it_idx2 || !xy || n < 0) return fail(PT_ERR_INVALID, "pt_trace_paths: bad argument");
    if ((int)s->bodies.size() > kMaxBodiesFused)
        return fail(PT_ERR_UNSUPPORTED, "pt_trace_paths: per-path diagnostics run the fused estimator, at most 127 bodies");
    if (s->estimator == PT_EST_BIDIR)
        return fail(PT_ERR_UNSUPPORTED, "pt_trace_paths: no per-path diagnostics for the bidirectional estimator");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc = check_measured_tables(s);
    if (rc) return rc;
    Tmp<int32_t> d_ii, d_nrec, d_bod, d_vis, d_used;
    Tmp<double> d_xy, d_rgb, d_pts;
    HIP_TRY(d_ii.alloc((size_t)2 * n));
    HIP_TRY(d_xy.alloc((size_t)2 * n));
    HIP_TRY(d_rgb.alloc((size_t)3 * n));
    HIP_TRY(d_nrec.alloc((size_t)n));
    HIP_TRY(d_bod.alloc((size_t)5 * n));
    HIP_TRY(d_pts.alloc((size_t)15 * n));
    HIP_TRY(d_vis.alloc((size_t)n));
    HIP_TRY(d_used.alloc((size_t)n));
    HIP_TRY(hipMemcpy(d_ii, it_idx2, sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_xy, xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    dim3 grid((n + kBlock - 1) / kBlock);
    if (s->estimator == PT_EST_NEE && s->has_measured)
        hipLaunchKernelGGL((k_trace_paths<kEstNEE, true>), grid, dim3(kBlock), 0, 0, s->ds, d_ii, d_xy, n, d_rgb, d_nrec,
                           d_bod, d_pts, d_vis, d_used);
    else if (s->estimator == PT_EST_NEE)
        hipLaunchKernelGGL((k_trace_paths<kEstNEE, false>), grid, dim3(kBlock), 0, 0, s->ds, d_ii, d_xy, n, d_rgb, d_nrec,
                           d_bod, d_pts, d_vis, d_used);
    else
        hipLaunchKernelGGL((k_trace_paths<kEstPT, false>), grid, dim3(kBlock), 0, 0, s->ds, d_ii, d_xy, n, d_rgb, d_nrec,
                           d_bod, d_pts, d_vis, d_used);
    HIP_TRY(hipGetLastError());
    if (rgb) HIP_TRY(hipMemcpy(rgb, d_rgb, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    if (nrec) HIP_TRY(hipMemcpy(nrec, d_nrec, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (bodies) HIP_TRY(hipMemcpy(bodies, d_bod, sizeof(int32_t) * 5 * n, hipMemcpyDeviceToHost));
    if (points) HIP_TRY(hipMemcpy(points, d_pts, sizeof(double) * 15 * n, hipMemcpyDeviceToHost));
    if (visible) HIP_TRY(hipMemcpy(visible, d_vis, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (used) HIP_TRY(hipMemcpy(used, d_used, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_trace_brdf_indices(pt_scene* s, const int32_t* it_idx2, const double* xy, int n, int32_t* ind) {
    if (!s || !it_idx2 || !xy || !ind || n < 0) return fail(PT_ERR_INVALID, "pt_trace_brdf_indices: bad argument");
    if (s->estimator != PT_EST_NEE || !s->has_measured)
        return fail(PT_ERR_INVALID, "pt_trace_brdf_indices: the scene has no measured body under next-event estimation");
    if ((int)s->bodies.size() > kMaxBodiesFused)
        return fail(PT_ERR_UNSUPPORTED, "pt_trace_brdf_indices: per-path diagnostics run the fused estimator, at most 127 bodies");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc = check_measured_tables(s);
    if (rc) return rc;
    Tmp<int32_t> d_ii, d_ind;
    Tmp<double> d_xy;
    HIP_TRY(d_ii.alloc((size_t)2 * n));
    HIP_TRY(d_xy.alloc((size_t)2 * n));
    HIP_TRY(d_ind.alloc((size_t)5 * n));
    HIP_TRY(hipMemcpy(d_ii, it_idx2, sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_xy, xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_trace_paths<kEstNEE, true>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, s->ds, d_ii, d_xy, n,
                       (double*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, (int32_t*)nullptr,
                       (int32_t*)nullptr, (int32_t*)d_ind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(ind, d_ind, sizeof(int32_t) * 5 * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_frame_layout(pt_scene* s, int iteration, double* ys, double* xs, int32_t* row_len, int32_t* nrows) {
    if (!s || !ys || !xs || !row_len || !nrows) return fail(PT_ERR_INVALID, "pt_frame_layout: null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());  // table set 0 is shared with renders that may still be in flight
    int rc = ensure_render_buffers(s);
    if (rc) return rc;
    const int W = s->width, H = s->height;
    const int CK = (W + kXStride - 1) / kXStride;
    launch_frame_layout(nullptr, s->seed, W, H, iteration, 1, s->xstep, s->ystep, (double*)s->d_xs[0].p, (double*)s->d_ys[0].p,
                        (int32_t*)s->d_rowstart[0].p, (int32_t*)s->d_npix[0].p, (int)(s->flags & PT_F_PIXEL_GRID));
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> rs(H + 1);
    std::vector<double> ck((size_t)H * CK);
    HIP_TRY(hipMemcpy(ck.data(), s->d_xs[0].p, sizeof(double) * ck.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ys, s->d_ys[0].p, sizeof(double) * (size_t)H, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rs.data(), s->d_rowstart[0].p, sizeof(int32_t) * (H + 1), hipMemcpyDeviceToHost));
    int nr = 0;
    for (int r = 0; r < H; ++r) {
        row_len[r] = rs[r + 1] - rs[r];
        if (row_len[r] > 0) nr = r + 1;
        // expand the checkpoints the way k_trace does: continue the running sum (IEEE double adds)
        for (int c = 0; c < row_len[r]; ++c) {
            volatile double x = ck[(size_t)(c / kXStride) * H + r];
            for (int i = c & (kXStride - 1); i > 0; --i) x = x + s->xstep;
            if (s->flags & PT_F_PIXEL_GRID) x = (double)c / (double)W;
            xs[(size_t)r * W + c] = x;
        }
    }
    *nrows = nr;
    return PT_OK;
}

int pt_rng_fill(uint64_t seed, uint32_t iteration, uint32_t domain, uint32_t index, int n, int32_t* k, double* u) {
    if (n < 0 || !k || !u) return fail(PT_ERR_INVALID, "pt_rng_fill: bad argument");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(g_device));
    Tmp<int32_t> d_k;
    Tmp<double> d_u;
    HIP_TRY(d_k.alloc((size_t)n));
    HIP_TRY(d_u.alloc((size_t)n));
    hipLaunchKernelGGL(k_rng_fill, dim3(1), dim3(64), 0, 0, seed, iteration, domain, index, n, d_k, d_u);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(k, d_k, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(u, d_u, sizeof(double) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_x87_eval(int kind, int on_device, const double* a, const double* b, int n, double* out) {
    if (!a || !b || !out || n < 0 || kind < 0 || kind > 4) return fail(PT_ERR_INVALID, "pt_x87_eval: bad argument");
    if (!on_device) {
        for (int i = 0; i < n; ++i) out[i] = x87_eval(kind, a[i], b[i]);
        return PT_OK;
    }
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(g_device));
    Tmp<double> d_a, d_b, d_o;
    HIP_TRY(d_a.alloc((size_t)n));
    HIP_TRY(d_b.alloc((size_t)n));
    HIP_TRY(d_o.alloc((size_t)n));
    HIP_TRY(hipMemcpy(d_a, a, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_b, b, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_x87_eval, dim3((n + 255) / 256), dim3(256), 0, 0, kind, d_a, d_b, n, d_o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_brdf_lookup(pt_scene* s, int id, const double* angles4, int n, double* rgb, int32_t* ind) {
    if (!s || !angles4 || !rgb || !ind || n < 0) return fail(PT_ERR_INVALID, "pt_brdf_lookup: bad argument");
    if (id < 0 || id >= PT_MAX_BRDF || !s->d_brdf[id].p) return fail(PT_ERR_INVALID, "pt_brdf_lookup: table not uploaded");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    Tmp<double> d_a, d_rgb;
    Tmp<int32_t> d_ind;
    HIP_TRY(d_a.alloc((size_t)4 * n));
    HIP_TRY(d_rgb.alloc((size_t)3 * n));
    HIP_TRY(d_ind.alloc((size_t)n));
    HIP_TRY(hipMemcpy(d_a, angles4, sizeof(double) * 4 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_brdf_lookup, dim3((n + 255) / 256), dim3(256), 0, 0, (const double*)s->d_brdf[id].p, d_a, n, d_rgb,
                       d_ind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rgb, d_rgb, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ind, d_ind, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_brdf_lookup_vectors(pt_scene* s, int id, const double* vec6, int n, double* rgb, int32_t* ind) {
    if (!s || !vec6 || !rgb || !ind || n < 0) return fail(PT_ERR_INVALID, "pt_brdf_lookup_vectors: bad argument");
    if (id < 0 || id >= PT_MAX_BRDF || !s->d_brdf[id].p) return fail(PT_ERR_INVALID, "pt_brdf_lookup_vectors: table not uploaded");
    if (!s->ds.brdf_bins) return fail(PT_ERR_INVALID, "pt_brdf_lookup_vectors: the scene has no measured body");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    Tmp<double> d_v, d_rgb;
    Tmp<int32_t> d_ind;
    HIP_TRY(d_v.alloc((size_t)6 * n));
    HIP_TRY(d_rgb.alloc((size_t)3 * n));
    HIP_TRY(d_ind.alloc((size_t)n));
    HIP_TRY(hipMemcpy(d_v, vec6, sizeof(double) * 6 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_brdf_lookup_vectors, dim3((n + 255) / 256), dim3(256), 0, 0, (const double*)s->d_brdf[id].p,
                       s->ds.brdf_bins, (const double*)d_v, n, (double*)d_rgb, (int32_t*)d_ind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rgb, d_rgb, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ind, d_ind, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // extern "C"

#ifdef PT_SCAN_STATS
// Development only: counters of the grouped scan since the last call (and reset).
extern "C" int pt_scan_stats(unsigned long long* out) {
    unsigned long long h[8] = {};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ptw::g_scan_stats), sizeof h) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out[k] = h[k];
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ptw::g_scan_stats), z, sizeof z) != hipSuccess) return -1;
    return 8;
}
#endif

#ifdef PT_PROFILE_SECTIONS
// Development only: per section, cycles summed over waves and cycles times active lanes, since the
// last call (and reset): out[k] and out[kProfN + k].  Returns kProfN.
extern "C" int pt_prof_read(unsigned long long* out, int n) {
    unsigned long long h[ptd::kProfSlots] = {};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ptd::g_prof), sizeof h) != hipSuccess) return -1;
    for (int k = 0; k < n && k < ptd::kProfSlots; ++k) out[k] = h[k];
    unsigned long long z[ptd::kProfSlots] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_prof), z, sizeof z) != hipSuccess) return -1;
    return ptd::kProfN;
}
#endif
