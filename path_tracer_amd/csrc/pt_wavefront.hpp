// Wavefront form of the estimator for scenes too large for the fused kernel: path state lives in
// HBM (a 64-byte record with the ray and the random stream, and vertex records, per path; columns for
// the rest), each stage is its own kernel, and the nearest-hit scan walks a bounding hierarchy of the
// spheres with scalar loads.
//
//   k_wf_primary         camera rays and per-path RNG streams, then as k_wf_bounce at depth 0
//   k_wf_bounce          nearest hit of the path's ray (main.cpp:372-383) and, in the same kernel, point /
//                        normal / bounce / offset / record (main.cpp:391-403), the path's entry for the
//                        next depth and its sort key
//   k_wf_light           light pick                                (next_event.cpp:390), shadow queues and their sort keys
//   k_wf_shadow          nearest hit of the shadow ray, light excluded (next_event.cpp:405-426)
//   k_wf_nee_finish      visibility tests + colour recurrence     (next_event.cpp:428-463), the sample
//   k_wf_finish          plain path tracing: Horner, the sample
// Between two depths the host queues one radix sort of (key, path id) pairs, and one per shadow queue.
//
// The gather entries of depth d+1 are written at the positions the threads of depth d hold: the
// path id of a survivor with its key, or -1 with the end key for a path that left the scene (or a
// slot without a sample), so the list needs no counter and no atomic; the sort moves the dead entries
// behind the live ones, where whole waves of them return at once.  k_wf_light fills one compacted
// shadow-task queue per vertex, one scan each; the colour recurrence then walks a path's vertices
// deepest first.  The arithmetic of every stage is the same device code the fused kernel uses, so
// both pipelines give identical bits.
#pragma once
#include "pt_device.hpp"

namespace ptw {

using namespace ptd;

constexpr int kWfBlock = 256;

// The scans visit the paths in ray-key order, i.e. at scattered path ids: what a scan reads and
// writes per path is kept together so that it costs one cache line, not one per component (as
// structure-of-arrays columns the gather scans fetched 600 B per ray from HBM).
struct alignas(64) RayRec {  // the path's current ray and its random stream
    double ox, oy, oz, dx, dy, dz;
    uint32_t s0, s1, s2, s3;
};
struct alignas(32) VertexRec {  // a recorded vertex: the offset point and its body
    double x, y, z;
    int32_t body;
    int32_t shadow;  // the path's light (set by k_wf_light) until the vertex's shadow scan stores the body it hit
};

// Path state is written by one stage and read by a later kernel after gigabytes of other traffic:
// the stores say so (non-temporal), which leaves the caches to the scene records (+1.5 %).
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
#define PT_NT_STORE(v, ptr) __builtin_nontemporal_store(u32x4_nt{(v).x, (v).y, (v).z, (v).w}, reinterpret_cast<u32x4_nt*>(ptr))
#define PT_NT1(v, ptr) __builtin_nontemporal_store((v), (ptr))

// A wave that owns one record per lane at arbitrary path ids moves the records four lanes to a
// record (two for the vertex records), 16 bytes a lane: every load or store instruction then covers
// whole records instead of a sixteen-byte piece of 64 different ones, which is what the memory
// pipeline is built for (the per-lane form made the shading stage twice as slow).  The records pass
// through `tile`, 64 x 64 bytes of LDS owned by the wave.  Every lane of the wave must call; lanes
// with on == false move nothing (and get zeros).
constexpr uint32_t kNoPath = 0xffffffffu;
PTD_INLINE uint32_t wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
PTD_INLINE RayRec wave_load_rays(const RayRec* __restrict__ rays, uint32_t p, bool on, uint4* tile) {
    const uint32_t lane = wave_lane(), piece = lane & 3u;
    const uint32_t mine = on ? p : kNoPath;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t ps = (uint32_t)__shfl((int)mine, (int)(r * 16u + (lane >> 2)));
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ps != kNoPath) v = reinterpret_cast<const uint4*>(rays + ps)[piece];
        tile[r * 64u + lane] = v;  // piece `piece` of the record of lane r * 16 + lane / 4
    }
    __builtin_amdgcn_wave_barrier();
    RayRec out;
    uint4* o4 = reinterpret_cast<uint4*>(&out);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) o4[k] = tile[lane * 4u + k];
    __builtin_amdgcn_wave_barrier();
    return out;
}
// PIECES: how many 16-byte pieces of each record are written (3: origin and direction only).
template <int PIECES = 4>
PTD_INLINE void wave_store_rays(RayRec* __restrict__ rays, uint32_t p, bool on, const RayRec& rec, uint4* tile) {
    const uint32_t lane = wave_lane(), piece = lane & 3u;
    const uint32_t mine = on ? p : kNoPath;
    const uint4* i4 = reinterpret_cast<const uint4*>(&rec);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) tile[lane * 4u + k] = i4[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t ps = (uint32_t)__shfl((int)mine, (int)(r * 16u + (lane >> 2)));
        const uint4 v = tile[r * 64u + lane];
        if (ps != kNoPath && (int)piece < PIECES) PT_NT_STORE(v, reinterpret_cast<uint4*>(rays + ps) + piece);
    }
    __builtin_amdgcn_wave_barrier();
}
// Vertex records: element `index` of `verts` per lane, two lanes to a record.
PTD_INLINE void wave_store_verts(VertexRec* __restrict__ verts, uint32_t index, bool on, const VertexRec& rec, uint4* tile) {
    const uint32_t lane = wave_lane(), piece = lane & 1u;
    const uint32_t mine = on ? index : kNoPath;
    const uint4* i4 = reinterpret_cast<const uint4*>(&rec);
    tile[lane * 2u] = i4[0];
    tile[lane * 2u + 1u] = i4[1];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t r = 0; r < 2; ++r) {
        const uint32_t is = (uint32_t)__shfl((int)mine, (int)(r * 32u + (lane >> 1)));
        const uint4 v = tile[r * 64u + lane];
        if (is != kNoPath) PT_NT_STORE(v, reinterpret_cast<uint4*>(verts + is) + piece);
    }
    __builtin_amdgcn_wave_barrier();
}

struct WfState {
    RayRec* rays;      // [cap]
    VertexRec* verts;  // [kMaxDepth][cap]
    double* xy;      // [2][cap]  image-plane coordinates of the sample
    uint4* rng;      // [cap]     the stream after the light pick (k_wf_light -> k_wf_nee_finish)
    int32_t* nrec;   // [cap]
    int32_t* light;  // [cap]
    int32_t* need;   // [cap]     1 + deepest emitting vertex (scenes with a dead direct term), else 0
    int32_t* queue;  // [11][cap] 0: every path id; 1..4: the gather entries of that depth (path id or -1 by position); 6..10: shadow tasks of vertex 0..4
    int32_t* qcount; // [16]
    int32_t* sorted; // [cap]     a queue reordered by ray key (see ray_key)
    uint32_t* keys;  // [cap]     sort keys of the next depth's gather entries, by position
    uint32_t* skeys; // [5][cap]  sort keys of the shadow-task queues
    size_t cap;
};

// Append `value` for every thread with `pred` set: one atomic per block (the waves' counts meet in
// LDS).  Atomics on one counter retire at about one per 9 ns, which is what a push costs: the
// shading stages therefore run blocks of kPushBlock threads, a quarter of the atomics of 256-thread
// blocks.  Every thread of the block must call it; `slots` is blockDim.x / 64 + 1 words of LDS.  The
// order of a queue is arbitrary; no result depends on it.  Returns the position of the thread's entry
// (-1 without one).
constexpr int kPushBlock = 1024;
PTD_INLINE int queue_push(int32_t* q, int32_t* count, bool pred, int value, int32_t* slots) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (int)(blockDim.x >> 6);
    __syncthreads();  // `slots` may still be read from the previous push
    if (lane == 0) slots[wave] = (int)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < nwaves; ++w) {
            const int c = slots[w];
            slots[w] = total;  // exclusive prefix inside the block
            total += c;
        }
        slots[nwaves] = total ? atomicAdd(count, total) : 0;
    }
    __syncthreads();
    const int pos = slots[nwaves] + slots[wave] + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (pred) PT_NT1(value, q + pos);
    return pred ? pos : -1;
}

PTD_INLINE Rng rng_of(const uint4& v) {
    Rng r;
    r.s0 = v.x, r.s1 = v.y, r.s2 = v.z, r.s3 = v.w;
    return r;
}
PTD_INLINE V3 load3(const double* a, size_t cap, size_t p, int first) {
    return mk(a[(size_t)first * cap + p], a[(size_t)(first + 1) * cap + p], a[(size_t)(first + 2) * cap + p]);
}
PTD_INLINE void store3(double* a, size_t cap, size_t p, int first, V3 v) {
    a[(size_t)first * cap + p] = v.x;
    a[(size_t)(first + 1) * cap + p] = v.y;
    a[(size_t)(first + 2) * cap + p] = v.z;
}

PTD_INLINE uint32_t grid_cell(float u, int cells) {  // u in [-1, 1] -> 0..cells-1, NaN -> 0
    int c = (int)((u + 1.0f) * (0.5f * (float)cells));
    c = c < cells - 1 ? c : cells - 1;
    return (uint32_t)(c > 0 ? c : 0);
}

// Nearest-hit scan of large scenes.  Planes are scanned first, exactly as in the fused kernel.
// The spheres are arranged in spatially compact groups of kGroupBodies (built at upload by median
// splits), kSuperGroups groups to a super-group, kSectionGroups groups to a section whose two
// halves have bounds of their own.  Every level's float screening records (16 B per bound or
// member, two to a record) are the same for all lanes of a wave, so they are read with scalar
// loads straight from global memory (the whole structure is some 18 B per sphere and stays in L2):
// no staging, no barriers, waves progress independently.  Per pair of bounds a wave runs the
// packed bounding-sphere screen, first on the halves of the sections (nearest first, from a table
// per origin cell), then on the super-groups of a half some lane might reach, then on their groups; only the groups some lane's ray might
// enter have their members screened, four per step; a lane whose screen of a member is
// inconclusive notes the member's slot in a four-entry list of its own, and whenever some lane's
// list is full every lane evaluates its noted members in FP64 from the full records (per-lane
// loads).  So the FP64 work is per lane, not per wave: a wave of 64 unrelated rays meets some 200
// members, a lane three or four.  The bound screen is valid for |d| <= 1 — a hit by the
// reference's formula b < 0, b^2 > |D|^2 - r^2 then implies |o - c_B|_Q < R in the norm
// |v|_Q^2 = |v|^2 - (v.d)^2, and an origin outside the bound has the bound's centre ahead — so
// lanes with a longer direction examine every group.  Results are merged with the explicit rule
// (t < mint) || (t == mint && index > hit), which makes them independent of the order.
constexpr int kSectionGroups = 1024 / kGroupBodies;
constexpr int kOrderCells = 8;  // cells per axis of the visiting-order table (DScene::grp_half_order)

// Development counters of the grouped scan (-DPT_SCAN_STATS): per wave, the screens it runs.
#ifdef PT_SCAN_STATS
__device__ unsigned long long g_scan_stats[8];  // waves, super pairs entered, group pairs tested, groups entered, member steps with work, evaluations, lanes-evaluating
#define SCAN_STAT(k, v)                                                                  \
    do {                                                                                 \
        if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) \
            atomicAdd(&g_scan_stats[k], (unsigned long long)(v));                        \
    } while (0)
#else
#define SCAN_STAT(k, v) \
    do {                \
    } while (0)
#endif

// The members a lane has noted for exact evaluation (slots of the grouped layout).
struct Pending {
    uint32_t slot[4];
    uint32_t n;
};
#ifndef PT_EVAL_BATCH
#define PT_EVAL_BATCH 2
#endif
// The records of PT_EVAL_BATCH noted members are requested before the first is used.
PTD_INLINE void pending_eval(const DScene& S, Pending& pd, V3 o, V3 d, int skip, int& hit, double& mint) {
    const DSlotExact* __restrict__ exact = S.grp_exact;
#pragma unroll
    for (uint32_t j0 = 0; j0 < 4; j0 += PT_EVAL_BATCH) {
        if (__builtin_amdgcn_ballot_w64(pd.n > j0) == 0) break;
        DGeom s[PT_EVAL_BATCH];
        int body[PT_EVAL_BATCH];
#pragma unroll
        for (uint32_t k = 0; k < PT_EVAL_BATCH; ++k) {
            body[k] = -1;
            if (pd.n > j0 + k) {
                const DSlotExact* e = exact + pd.slot[j0 + k];
                s[k].cx = e->cx;
                s[k].cy = e->cy;
                s[k].cz = e->cz;
                s[k].nx = e->r2;
                body[k] = e->body;  // -1: a padding slot, reachable only by a NaN ray
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < PT_EVAL_BATCH; ++k) {
            if (k > 0 && __builtin_amdgcn_ballot_w64(pd.n > j0 + k) == 0) break;
            SCAN_STAT(5, 1);
            if (body[k] >= 0) {
                double t;
                const bool ok = scan_sphere_exact(s[k], o, d, t);
                merge_hit(ok, t, body[k], skip, S.hit_eps, hit, mint);
            }
        }
    }
    pd.n = 0;
}

// Called by whole waves with a ray in every participating lane (idle lanes of a partial wave do
// not enter).
PTD_INLINE int nearest_hit_grouped(const DScene& S, V3 o, V3 d, int skip, double& tmin) {
    const RayF rf = make_rayf(o, d, S.sphere_c1max, S.sphere_r2max);
    const float fox = rf.ox, foy = rf.oy, foz = rf.oz, fdx = rf.dx, fdy = rf.dy, fdz = rf.dz;
    const float e_disc = rf.e_disc, e_b = rf.e_b;
    const float dlen2 = fdx * fdx + fdy * fdy + fdz * fdz;
    const lanemask long_dir = PTD_MASK(!(dlen2 <= 1.000001f));
    // tau = |d| mint of the distance cut (group_possible2_masks) with its margin; the cut's proof needs
    // |d| <= 1 of the FP64 direction itself, and lanes outside the screen's validity (|d| > 1, NaN) never cut
    const bool short_dir = (d.x * d.x + d.y * d.y + d.z * d.z) <= 1.0 + 1e-12;
    const float cut_len = (short_dir && dlen2 <= 1.000001f) ? __builtin_sqrtf(dlen2) * 1.0004f : INFINITY;
    auto cut_of = [&](double m) { return cut_len * ((float)m * 1.00001f); };
    int hit = -1;
    double mint = 1.7976931348623157e308;  // DBL_MAX
    SCAN_STAT(0, 1);
    scan_all_planes<false>(S, S.scan_planes, o, d, skip, hit, mint);
    const DRect* __restrict__ rects = S.scan_rects;  // raytrace3.cpp's third shape, a short list
    const int n_rects = S.n_rects;
    for (int k = 0; k < n_rects; ++k) {
        double t;
        const bool ok = scan_rect(rects[k], o, d, t);
        merge_hit(ok, t, rects[k].body, skip, S.hit_eps, hit, mint);
    }
    const DRectMT* __restrict__ rects_mt = S.scan_rects_mt;  // light_tracing.cpp's rectangle, likewise
    const int n_rects_mt = S.n_rects_mt;
    for (int k = 0; k < n_rects_mt; ++k) {
        double t;
        const bool ok = scan_rect_mt(rects_mt[k], o, d, S.mt_eps, t);
        merge_hit(ok, t, rects_mt[k].body, skip, S.hit_eps, hit, mint);
    }
    Pending pd;
    pd.slot[0] = pd.slot[1] = pd.slot[2] = pd.slot[3] = 0u;
    pd.n = 0u;
    float tau = cut_of(mint);  // DBL_MAX converts to +inf: no cut before the first hit
    const float* __restrict__ half4 = S.grp_half4;
    const DScreen2* __restrict__ supers = S.grp_supers;
    const DScreen2* __restrict__ bounds = S.grp_bounds;
    const DScreen2* __restrict__ members = S.grp_members;
    const int n_groups = S.n_groups;  // a multiple of 2 kSuperGroups
    constexpr int kHalfPairs = kSectionGroups / kSuperGroups / 4;  // super-group pairs per half
    const int n_super_pairs = n_groups / kSuperGroups / 2;
    const int n_halves = 2 * ((n_groups + kSectionGroups - 1) / kSectionGroups);
    // The halves are visited nearest first (by the cell the wave's first ray starts in): once a near
    // hit is known the distance cut disposes of the far ones at this level.  Any order gives the same
    // result.
    const uint32_t* __restrict__ order = S.grp_half_order;
    if (order) {
        const uint32_t cx = grid_cell((fox - S.sort_mid[0]) * S.sort_inv[0], kOrderCells);
        const uint32_t cy = grid_cell((foy - S.sort_mid[1]) * S.sort_inv[1], kOrderCells);
        const uint32_t cz = grid_cell((foz - S.sort_mid[2]) * S.sort_inv[2], kOrderCells);
        order += (size_t)__builtin_amdgcn_readfirstlane((int)((cz * kOrderCells + cy) * kOrderCells + cx)) * 32;
    }
#pragma unroll 1
    for (int k = 0; k < n_halves; k += 2) {
        int h2[2] = {k, k + 1};
        if (order) {
            const uint32_t pk = order[k / 2];
            h2[0] = (int)(pk & 0xffu);
            h2[1] = (int)(pk >> 8);
        }
        const float4 ra = *reinterpret_cast<const float4*>(half4 + 4 * h2[0]);
        const float4 rb = *reinterpret_cast<const float4*>(half4 + 4 * h2[1]);
        DScreen2 hrec;
        hrec.cx = f32x2{ra.x, rb.x};
        hrec.cy = f32x2{ra.y, rb.y};
        hrec.cz = f32x2{ra.z, rb.z};
        hrec.r2 = f32x2{ra.w, rb.w};
        lanemask half[2];
        group_possible2_masks(fox, foy, foz, fdx, fdy, fdz, e_disc, e_b, tau, hrec, half[0], half[1]);
        if ((half[0] | half[1] | long_dir) == 0) continue;
#pragma unroll 1
        for (int w = 0; w < 2; ++w) {
            if ((half[w] | long_dir) == 0) continue;
            const int spg0 = h2[w] * kHalfPairs;  // first super-group pair of the half
            const int n_pairs = min(kHalfPairs, n_super_pairs - spg0);  // <= 0: the scene ends before this half
            for (int sp = 0; sp < n_pairs; ++sp) {
                const int spg = spg0 + sp;  // super-group pair of the scene
                lanemask reach[2];
                group_possible2_masks(fox, foy, foz, fdx, fdy, fdz, e_disc, e_b, tau, supers[spg], reach[0], reach[1]);
                if (((reach[0] | reach[1] | long_dir)) == 0) continue;
                SCAN_STAT(1, 1);
#pragma unroll 1
                for (int gq = 0; gq < kSuperGroups; ++gq) {  // the 2 x kSuperGroups groups of the pair, two at a time
                    const int sside = gq / (kSuperGroups / 2);
                    if ((reach[sside] | long_dir) == 0) continue;
                    const int gp = spg * kSuperGroups + gq;  // group pair of the scene
                    lanemask enter[2];
                    group_possible2_masks(fox, foy, foz, fdx, fdy, fdz, e_disc, e_b, tau, bounds[gp], enter[0], enter[1]);
                    SCAN_STAT(2, 1);
                    if (((enter[0] | enter[1] | long_dir)) == 0) continue;
#pragma unroll 1
                    for (int side = 0; side < 2; ++side) {
                        if ((enter[side] | long_dir) == 0) continue;
                        SCAN_STAT(3, 1);
                        const int g = 2 * gp + side;
#pragma unroll 1
                        for (int q = 0; q < kGroupBodies / 4; ++q) {
                            const DScreen2 sa = members[(size_t)g * (kGroupBodies / 2) + 2 * q];
                            const DScreen2 sb = members[(size_t)g * (kGroupBodies / 2) + 2 * q + 1];
                            lanemask need[4];
                            sphere_possible2_masks(fox, foy, foz, fdx, fdy, fdz, e_disc, e_b, sa, need[0], need[1]);
                            sphere_possible2_masks(fox, foy, foz, fdx, fdy, fdz, e_disc, e_b, sb, need[2], need[3]);
                            if ((need[0] | need[1] | need[2] | need[3]) == 0) continue;
                            SCAN_STAT(4, 1);
#pragma unroll
                            for (int h = 0; h < 4; ++h) {
                                if (need[h] == 0) continue;
                                const uint32_t slot = (uint32_t)(g * kGroupBodies + 4 * q + h);
                                // the lanes of need[h] append `slot` to their lists
#pragma unroll
                                for (uint32_t j = 0; j < 4; ++j)
                                    pd.slot[j] = sel32(need[h] & PTD_MASK(pd.n == j), slot, pd.slot[j]);
                                pd.n += sel32(need[h], 1u, 0u);
                                if (PTD_MASK(pd.n == 4u) != 0) {
                                    pending_eval(S, pd, o, d, skip, hit, mint);
                                    tau = cut_of(mint);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    pending_eval(S, pd, o, d, skip, hit, mint);
    tmin = mint;
    return hit;
}

// ---- kernels ---------------------------------------------------------------------------------

enum { kGather = 0, kShadow = 1 };

// Rays are reordered before a scan so that the 64 rays of a wave start near one another and
// point the same way: a wave then enters few of the body groups.  The key is a cell of the
// direction (cube face, 4x4) and a cell of the origin (16^3 over the box of the sphere centres), 20
// bits; shadow rays, which share their origin, use a fine direction cell (cube face, 64x64), 16
// bits.  One radix sort of the keys per scan (rocprim, path ids as values; stable, so a bin keeps
// the queue's order, which is pixel order: the more paths a batch holds, the closer together on the
// screen the 64 rays of a wave).  The order of a queue is invisible in the results.  (Measured on
// the 10 000-body scene with batches of 2^26 paths: 4x4 / 8^3 7.24e8, 4x4 / 12^3 7.31e8, 4x4 / 16^3
// 7.45e8, 4x4 / 32^3 7.47e8, 8x8 / 32^3 7.23e8, 2x2 / 32^3 7.37e8 paths/s; sorting on the upper 8 or 12
// bits of a 16-bit key alone: 3.9e8 and 4.9e8 against 6.8e8.)
#ifndef PT_KEY_D
#define PT_KEY_D 4
#endif
#ifndef PT_KEY_O
#define PT_KEY_O 16
#endif
#ifndef PT_KEY_BITS
#define PT_KEY_BITS 20
#endif
constexpr int kKeyDir = PT_KEY_D, kKeyOrg = PT_KEY_O;
template <int MODE>
PTD_INLINE uint32_t ray_key(const DScene& S, V3 o, V3 d) {
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const float ax = __builtin_fabsf(dx), ay = __builtin_fabsf(dy), az = __builtin_fabsf(dz);
    uint32_t face;
    float u, v, m;
    if (ax >= ay && ax >= az) {
        face = dx < 0.0f ? 1u : 0u;
        m = ax;
        u = dy;
        v = dz;
    } else if (ay >= az) {
        face = dy < 0.0f ? 3u : 2u;
        m = ay;
        u = dx;
        v = dz;
    } else {
        face = dz < 0.0f ? 5u : 4u;
        m = az;
        u = dx;
        v = dy;
    }
    const float inv = m > 0.0f ? 1.0f / m : 0.0f;
    if (MODE == kShadow) return (face * 64u + grid_cell(v * inv, 64)) * 64u + grid_cell(u * inv, 64);  // < 24576
    const uint32_t dir = (face * kKeyDir + grid_cell(v * inv, kKeyDir)) * kKeyDir + grid_cell(u * inv, kKeyDir);
    const uint32_t cx = grid_cell(((float)o.x - S.sort_mid[0]) * S.sort_inv[0], kKeyOrg);
    const uint32_t cy = grid_cell(((float)o.y - S.sort_mid[1]) * S.sort_inv[1], kKeyOrg);
    const uint32_t cz = grid_cell(((float)o.z - S.sort_mid[2]) * S.sort_inv[2], kKeyOrg);
    return dir * (kKeyOrg * kKeyOrg * kKeyOrg) + ((cz * kKeyOrg + cy) * kKeyOrg + cx);
}

// The stage that fills a queue stores every entry's key at its queue position (k_wf_shade,
// k_wf_light); the key buffers are first filled with ones, so entries past the queue's count carry
// the largest key of the sorted bits and sort to the end (the library sort takes its size from the
// host, which does not know the count).
constexpr uint32_t kSortKeyBits = PT_KEY_BITS;  // gather rays
constexpr uint32_t kShadowKeyBits = 16;         // shadow rays: 6 * 64 * 64 cells
static_assert(6ull * PT_KEY_D * PT_KEY_D * PT_KEY_O * PT_KEY_O * PT_KEY_O < (1ull << PT_KEY_BITS) - 1, "gather keys must stay below the end key");
static_assert(6ull * 64 * 64 < (1ull << kShadowKeyBits) - 1, "shadow keys must stay below the end key");

constexpr int kExtBlock = 64;
constexpr uint32_t kEndKey = 0xffffffffu;  // sorts behind every ray key

// After the scan of a gather ray: shade the hit (main.cpp:391-403), record the vertex, store the
// path's new ray and stream, and write position i of the next depth's entries.  `live`: the lane holds
// a path (rec is its record, hit / t the result of its scan); every lane of the wave must call.
// A path's record count goes to W.nrec once, where the path ends.
template <bool KEYS, bool FIRST>
PTD_INLINE void wf_shade_and_store(const DScene& S, const WfState& W, int depth, size_t i, uint32_t p, bool live, RayRec rec,
                                   int hit, double t, V3 eye_point, uint4* tile) {
    bool alive = false;
    VertexRec v;
    v.x = v.y = v.z = 0.0, v.body = -1, v.shadow = -1;
    uint32_t key = kEndKey;
    if (live) {
        if (hit >= 0) {
            alive = true;
            V3 o = mk(rec.ox, rec.oy, rec.oz), d = mk(rec.dx, rec.dy, rec.dz);
            Draws dr;
            dr.rng.s0 = rec.s0, dr.rng.s1 = rec.s1, dr.rng.s2 = rec.s2, dr.rng.s3 = rec.s3;
            dr.count = 0;
            const DMat& m = S.mat[hit];
            // the vertex before this one: the ray origin, or the camera-plane point for the first
            const V3 prev = depth > 0 ? o : eye_point;
            V3 point = shade_hit(S.geom[hit], m, dr, o, d, t);
            if (m.emits | ((m.material != kDiffuse) & same_point(point, prev))) W.need[p] = depth + 1;  // see trace_path_nee_compact
            v.x = point.x, v.y = point.y, v.z = point.z, v.body = hit;
            rec.ox = o.x, rec.oy = o.y, rec.oz = o.z, rec.dx = d.x, rec.dy = d.y, rec.dz = d.z;
            rec.s0 = dr.rng.s0, rec.s1 = dr.rng.s1, rec.s2 = dr.rng.s2, rec.s3 = dr.rng.s3;
            if (KEYS) key = ray_key<kGather>(S, o, d);
        }
        if (!alive || depth == kMaxDepth - 1) PT_NT1(alive ? depth + 1 : depth, W.nrec + p);
    }
    // (the first stage stores the record of a camera ray that hits nothing too: the light pick reads its stream)
    wave_store_rays<4>(W.rays, p, FIRST ? live : alive, rec, tile);
    wave_store_verts(W.verts + (size_t)depth * W.cap, p, alive, v, tile);
    if (depth + 1 < kMaxDepth && i < W.cap) {
        PT_NT1(alive ? (int32_t)p : -1, W.queue + (size_t)(depth + 1) * W.cap + i);
        if (KEYS) PT_NT1(key, W.keys + i);
    }
}

// Depth >= 1: one path per thread over the n_entries positions of the depth's entries, the first
// n_sorted of them in the order of `sorted`, the rest in their own.  One wave to a block: nothing is
// shared between waves, and a block of one ends as soon as its own rays are done.
// Six waves per SIMD: the compiler then spills nine of the kernel's 89 vector registers, in the shading
// part, which a path passes once; the scan, whose scalar loads want the waves, keeps its registers
// (9.48e8 -> 9.58e8 paths/s on the 10 000-body scene; seven waves spill 17 and lose: 9.38e8).
#ifndef PT_BOUNCE_WAVES
#define PT_BOUNCE_WAVES 6
#endif
#ifndef PT_SHADOW_WAVES
#define PT_SHADOW_WAVES 6
#endif
template <bool KEYS>
__global__ void __launch_bounds__(kExtBlock, PT_BOUNCE_WAVES) k_wf_bounce(DScene S, WfState W, int depth, const int32_t* __restrict__ sorted, int n_sorted,
                                                         int n_entries) {
    const size_t i = (size_t)blockIdx.x * kExtBlock + threadIdx.x;
    if ((size_t)blockIdx.x * kExtBlock >= (size_t)n_entries) return;  // whole waves only: the record moves below are wave-wide
    const bool inside = i < (size_t)n_entries;                        // (n_entries is a multiple of 256 in every caller)
    int32_t id = -1;
    if (inside) id = i < (size_t)n_sorted ? sorted[i] : W.queue[(size_t)depth * W.cap + i];
    const bool live = id >= 0;
    const uint32_t p = live ? (uint32_t)id : 0u;
    __shared__ uint4 s_tile[64 * 4];
    RayRec rec = wave_load_rays(W.rays, p, live, s_tile);
    int hit = -1;
    double t = 0.0;
    if (live) hit = nearest_hit_grouped(S, mk(rec.ox, rec.oy, rec.oz), mk(rec.dx, rec.dy, rec.dz), -1, t);
    wf_shade_and_store<KEYS, false>(S, W, depth, i, p, live, rec, hit, t, mk(0.0, 0.0, 0.0), s_tile);
}

// The shadow ray towards vertex gi of every task of queue qi (next_event.cpp:405-407): from the light's
// origin, the light excluded from the scan; the vertex record receives the body it hit.
__global__ void __launch_bounds__(kExtBlock, PT_SHADOW_WAVES) k_wf_shadow(DScene S, WfState W, int qi, int gi, const int32_t* __restrict__ sorted, int n_sorted) {
    const int count = W.qcount[qi];
    const int i = blockIdx.x * kExtBlock + threadIdx.x;
    if (i >= count) return;
    const size_t p = (size_t)(i < n_sorted ? sorted[i] : W.queue[(size_t)qi * W.cap + i]);
    VertexRec* vr = W.verts + (size_t)gi * W.cap + p;
    const VertexRec v = *vr;
    const int light = v.shadow;
    const V3 lc = light_origin(S, light);
    const V3 d = normalize(sub(mk(v.x, v.y, v.z), lc));
    double mint;
    vr->shadow = nearest_hit_grouped(S, lc, d, light, mint);
}

// The first stage: camera ray and per-path RNG stream, then the scan and the shading of depth 0 as
// k_wf_bounce does them (the rays leave the camera in pixel order, which is as coherent as a sort
// would make them).  One wave to a block; grid (4 * owned tiles, iterations of the batch); path id
// p = b * J + j with j = tile * 256 + place in the tile, and position p of the entries.  Queue 0 is
// every path id of the batch in order, the few without a sample (outside the image or the row range,
// past a short frame's end) included: they get no records and no shadow tasks, every later stage
// passes them by, and their sample slot, which the accumulation never reads, is written as zero.
template <bool KEYS, class PartitionT, class SlotFn>
__device__ inline void wf_primary_body(const DScene& S, const PartitionT& P, SlotFn slot_of_fn, int first_it,
                                       const double* __restrict__ xs, const double* __restrict__ ys,
                                       const int32_t* __restrict__ row_start, const int32_t* __restrict__ npix,
                                       double xstep, int xstride, WfState W, size_t J, int n_it, uint4* tile) {
    const int lane = threadIdx.x;  // blocks of one wave
    const int b = blockIdx.y;
    const int q = blockIdx.x >> 2, ttid = (int)(blockIdx.x & 3u) * 64 + lane;
    const int Wd = S.width, H = S.height;
    const int slot = slot_of_fn(P, q, ttid, Wd, H);
    const size_t p = (size_t)b * J + (size_t)q * kWfBlock + ttid;
    const bool valid = slot >= 0 && slot < npix[b];
    RayRec rr;
    rr.ox = rr.oy = rr.oz = rr.dx = rr.dy = rr.dz = 0.0;
    rr.s0 = rr.s1 = rr.s2 = rr.s3 = 0u;
    double x = 0.0, y = 0.0, t = 0.0;
    int hit = -1;
    V3 eye_point = mk(0.0, 0.0, 0.0);
    if (valid) {
        const int32_t* rs = row_start + (size_t)b * (H + 1);
        int r = slot / Wd;
        while (slot >= rs[r + 1]) r++;
        int c = slot - rs[r];
        const int CK = (Wd + xstride - 1) / xstride;
        if (S.flags & 4u) {
            x = (double)c / (double)Wd;  // light_tracing.cpp:787
        } else {
            x = xs[((size_t)b * CK + c / xstride) * H + r];
            for (int i = c % xstride; i > 0; --i) x += xstep;
        }
        y = ys[(size_t)b * H + r];
        V3 o = ld3(S.cam_o);
        eye_point = cam_point(S, x, y);
        V3 d = normalize(sub(eye_point, o));  // Camera::getRay main.cpp:154-157
        rr.ox = o.x, rr.oy = o.y, rr.oz = o.z, rr.dx = d.x, rr.dy = d.y, rr.dz = d.z;
        const Rng rng = rng_init(S.seed, (uint32_t)(first_it + b), kDomPixel, (uint32_t)slot);
        rr.s0 = rng.s0, rr.s1 = rng.s1, rr.s2 = rng.s2, rr.s3 = rng.s3;
        hit = nearest_hit_grouped(S, o, d, -1, t);
    }
    PT_NT1(x, W.xy + p);
    PT_NT1(y, W.xy + W.cap + p);
    PT_NT1(0, W.need + p);
    PT_NT1((int32_t)p, W.queue + p);
    if (!valid) {
        PT_NT1(0, W.nrec + p);
        __builtin_nontemporal_store(u32x4_nt{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4_nt*>(W.rays + p) + 3);  // the stream k_wf_light reads
    }
    wf_shade_and_store<KEYS, true>(S, W, 0, p, (uint32_t)p, valid, rr, hit, t, eye_point, tile);
    if (blockIdx.x == 0 && b == 0 && lane == 0) W.qcount[0] = (int32_t)(J * (size_t)n_it);
}

constexpr int kShadowQueue0 = kMaxDepth + 1;  // queue index of the shadow tasks of vertex 0

// Light pick for every generated path (queue 0), next_event.cpp:390, and the shadow-task queues.
// In a scene whose direct term is dead (DScene::nee_emission_only) the colour recurrence starts at
// zero and stays there until the deepest emitting vertex, so only vertices 0..need-1 get a shadow
// ray; otherwise every recorded vertex does.
// KEYS: the sort key of every shadow ray is stored at its queue position (W.skeys was filled with the
// end key), so the shadow scans' sorts need no pass over the vertices.
template <bool KEYS>
__global__ void __launch_bounds__(kPushBlock) k_wf_light(DScene S, WfState W) {
    const int count = W.qcount[0];
    const int i = blockIdx.x * kPushBlock + threadIdx.x;
    if ((int)(blockIdx.x * kPushBlock) >= count) return;
    const bool active = i < count;
    size_t p = 0;
    int tasks = 0;
    int light = 0;
    if (active) {
        p = (size_t)W.queue[i];
        Draws dr;
        dr.rng = rng_of(reinterpret_cast<const uint4*>(W.rays + p)[3]);  // the stream as the path's last bounce left it
        dr.count = 0;
        light = pick_light(S, dr);
        W.light[p] = light;
        W.rng[p] = make_uint4(dr.rng.s0, dr.rng.s1, dr.rng.s2, dr.rng.s3);
        tasks = S.nee_emission_only ? W.need[p] : W.nrec[p];
    }
    __shared__ int32_t s_push[kPushBlock / 64 + 1];
    for (int g = 0; g < kMaxDepth; ++g) {
        const bool task = tasks > g;
        uint32_t key = 0;
        if (task) {
            VertexRec* v = W.verts + (size_t)g * W.cap + p;
            v->shadow = light;  // what the shadow scan reads
            if (KEYS) {
                const V3 lc = light_origin(S, light);
                key = ray_key<kShadow>(S, lc, normalize(sub(mk(v->x, v->y, v->z), lc)));  // the ray of k_wf_shadow
            }
        }
        const int pos = queue_push(W.queue + (size_t)(kShadowQueue0 + g) * W.cap, W.qcount + kShadowQueue0 + g, task, (int)p, s_push);
        if (KEYS && task) PT_NT1(key, W.skeys + (size_t)g * W.cap + pos);
    }
}

// The colour recurrence of a path over its lit vertices, deepest first (next_event.cpp:428-463), once
// all its shadow scans are done, and the sample.  One pass over the paths instead of one per vertex:
// a per-vertex stage reads the RNG, colour and light columns in full for a queue that holds a third
// of the paths.
template <bool MEASURED>
__global__ void __launch_bounds__(kWfBlock, 4) k_wf_nee_finish(DScene S, WfState W, double* __restrict__ samples, size_t J) {
    const int count = W.qcount[0];
    const int i = blockIdx.x * kWfBlock + threadIdx.x;
    if (i >= count) return;
    const size_t p = (size_t)W.queue[i];
    const int tasks = S.nee_emission_only ? W.need[p] : W.nrec[p];  // the vertices k_wf_light queued shadow rays for
    V3 L = mk(0.0, 0.0, 0.0);
    if (tasks > 0) {
        const int light = W.light[p];
        const V3 lc = light_origin(S, light);
        Draws dr;
        dr.rng = rng_of(W.rng[p]);
        dr.count = 0;
#pragma unroll 1
        for (int gi = tasks - 1; gi >= 0; --gi) {
            const VertexRec v = W.verts[(size_t)gi * W.cap + p];
            const int hit = v.shadow;  // what the shadow scan of this vertex hit
            if (hit != v.body) continue;  // occluded: vertex skipped, next_event.cpp:428-430
            const V3 Pt = mk(v.x, v.y, v.z);
            const V3 bd = normalize(sub(Pt, lc));
            V3 next_point;
            if (gi >= 1) {
                const VertexRec* q = W.verts + (size_t)(gi - 1) * W.cap + p;
                next_point = mk(q->x, q->y, q->z);
            } else {
                next_point = cam_point(S, W.xy[p], W.xy[W.cap + p]);  // camera.getPoint(x, y) next_event.cpp:446
            }
            L = nee_vertex<MEASURED, true>(S, S.geom[hit], S.mat[hit], dr, light, Pt, bd, next_point, L);
        }
    }
    const size_t b = p / J, j = p % J;
    double* out = samples + b * 3 * J + j;
    L = canonical_nan(L);
    out[0] = L.x;
    out[J] = L.y;
    out[2 * J] = L.z;
}

// Plain path tracing: the sample of every generated path, path id p = b * J + j -> samples[b][c][j].
__global__ void __launch_bounds__(kWfBlock) k_wf_finish(DScene S, WfState W, double* __restrict__ samples, size_t J) {
    const int count = W.qcount[0];
    const int i = blockIdx.x * kWfBlock + threadIdx.x;
    if (i >= count) return;
    const size_t p = (size_t)W.queue[i];
    // trace(next)*color + emission, deepest hit first (main.cpp:404)
    V3 L = mk(0.0, 0.0, 0.0);
    for (int k = W.nrec[p] - 1; k >= 0; --k) {
        DMat m = S.mat[W.verts[(size_t)k * W.cap + p].body];
        if (S.flags & 2u) L = add(L, mk(0.0, 0.0, 0.0));  // raytrace3.cpp:840, see trace_path
        L = add(mul(L, mk(m.colr, m.colg, m.colb)), mk(m.emr, m.emg, m.emb));
    }
    const size_t b = p / J, j = p % J;
    double* out = samples + b * 3 * J + j;
    L = canonical_nan(L);
    out[0] = L.x;
    out[J] = L.y;
    out[2 * J] = L.z;
}

}  // namespace ptw
