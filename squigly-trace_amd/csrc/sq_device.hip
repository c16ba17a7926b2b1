// sq_device.hip — gfx950 kernels and the render half of the C-ABI (include/squigly_hip.h).
//
// What runs on the GPU (all hand-written for CDNA4, wave64; per-ray primitives in sq_scene.h):
//   camera-ray generation            src/Lib.hs:107-114                      sq_primary
//   RNG + bounce (scatter / mirror)  src/Lib.hs:133-134,155-198              sq_gen_bounce1, sq_shade1
//   BIH traversal + Moller-Trumbore  src/BIH.hs:101-141, Geometry.hs:117-177 sq_trace_rays (dominant kernel)
//   emissive shade, radiance fold    src/Lib.hs:135-137                      sq_shade1, sq_accumulate
//   ordered per-pixel accumulation   src/Lib.hs:85-88                        sq_accumulate
//   atan tonemap                     src/Lib.hs:93-104                       sq_accumulate
//
// Pipeline ("wavefront" form of renderPixel):  every sample of a pixel shoots the same primary ray
// (src/Lib.hs:81-87), so it is traced once per pixel and the pixels that hit are compacted (wave ballot +
// prefix rank).  Then, per batch of samples, every sample owns one slot: its first bounce ray is generated
// into the slot, traced by a persistent kernel (which compacts live slots on the fly and lets a lane pull
// its next ray as soon as the previous one finishes), shaded, replaced in place by the second bounce ray,
// traced again, and folded into a per-sample radiance; radiances are summed per pixel in sample order.
// Every value is computed by the same fp32 expression tree as the reference.
// The one-lane-per-pixel kernel sq_render_pixels serves raycast mode and is a cross-check variant.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/squigly_hip.h"
#include "../../include/squigly_host.h"
#include "sq_error.h"
#include "sq_scene.h"

using sq::f3;
using namespace sqd;

#define SQ_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return sq_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int kBlock = 256;        // per-pixel / per-sample kernels
constexpr int kTraceBlock = 512;   // persistent trace kernel, streaming form: 8 waves share one LDS copy of the top of the tree
#ifndef SQ_RESIDENT_BLOCK
#define SQ_RESIDENT_BLOCK 1024       // diagnostic builds: 768 / 512 = three / two waves per SIMD (how the frame time follows occupancy)
#endif
constexpr int kResidentBlock = SQ_RESIDENT_BLOCK; // persistent trace kernel, resident form: one workgroup per CU owns the whole scene in LDS
constexpr int kOneshotRowBlock = 2; // rows per block when a one-shot call shards a frame over devices (balance: squigly-trace_amd/dist.py)
// Slots a wave reserves from the queue per atomic (multiple of 256 for the resident form).  Every reservation stalls
// the wave for the atomic's round trip and then for the flag loads of the scan, and a sparse chunk (8 % of the slots
// are live at the second bounce level) serves only part of the idle lanes: 64 -> 128 -> 256 -> 512 slots measured
// 4340 -> 5110 -> 5520 -> 5605 Msamples/s on the headline frame (768: 5520).  The per-wave list of live slots holds
// one byte per entry, an index within its 256-slot block (512 B of LDS per wave).
// The streaming form keeps 128: its rays are 5-10x longer, and on a frame of a few spp a wave that sits on 256 slots
// at the end of the queue makes the tail longer than the stalls it saves (-7 % at 1920x1080@16).
constexpr int kChunkResident = 512, kChunkStreaming = 128;
using LiveT = uint8_t;             // a live slot's index within its chunk
#ifndef SQ_POOL_KW
#define SQ_POOL_KW 1
// Instruction-arbitration priority of a trace wave (s_setprio): bits 1..0 in its return / branch steps, bits 3..2 in its leaf scan and pair
// windows (and the store / refill that follows them).  4 = priority 1 in the windows, 0 in the steps: with the flat steps the headline
// frame takes 52.4-52.5 ms instead of 53.3-53.4 (levels 1, 2 and 3 alike; raising the STEPS instead costs 0.5 ms), one rank's share at
// 8 ranks 8.09 -> 7.98 ms, the streaming form +-0 (profiles/r03zz5_setprio_flat.txt, r03zz6_setprio_stream.txt).  Round 2 had measured
// -0.5 % for the same setting on the kernels of its time and left it off.  -DSQ_SETPRIO=0 builds without it.
#ifndef SQ_SETPRIO
#define SQ_SETPRIO 4
#endif
#endif
constexpr int kStatSlots = 32;            // sq_get_stats
#ifndef SQ_STAGE_BATCHED
#define SQ_STAGE_BATCHED 1
#endif
constexpr int kPoolWindows = SQ_POOL_KW;   // pooled trace kernel: pair windows a wave works on at a time
#ifndef SQ_POOL_TPL
#define SQ_POOL_TPL 2
#endif
// Pooled trace kernel: triangles a lane tests per window.  Two for the resident form (one owner lookup and one set of pulls
// serve two tests: 83.0 -> 80.5 ms on the headline frame, same run); one for the streaming form, whose loads want the
// registers (the 1M-triangle scene loses 14 % with two).
#ifndef SQ_POOL_TPL_STREAM
#define SQ_POOL_TPL_STREAM 2
#endif
constexpr int kPoolTrisResident = SQ_POOL_TPL, kPoolTrisStreaming = SQ_POOL_TPL_STREAM;

// ----------------------------------------------------------------------------------------------
// Frame description shared by the kernels
// ----------------------------------------------------------------------------------------------
struct Frame {
    float cam_pos[3]; float cam_rot[9];
    int32_t samples, w, h, cast;
    int32_t row_block, shard, n_shards, local_rows;
    int32_t tile_rows, tiles_x;   // primary rays are enumerated in tiles of tile_rows x (64 / tile_rows) pixels (primary_tile), tiles_x per tile row
    float* out_avg; uint8_t* out_rgb;
    int32_t diag;             // option "coresidency": the per-sample kernels and the trace kernel count who runs beside whom (sq_get_stats 24..27)
};
__device__ __forceinline__ void pixel_coords(const Frame& F, int pix, int& y, int& x) {   // 32-bit: cheap div/mod
    const int j = pix / F.h;
    x = pix - j * F.h;
    const int blk = j / F.row_block;
    y = (blk * F.n_shards + F.shard) * F.row_block + (j - blk * F.row_block);
}
// The q-th primary ray of a shard, q in [0, primary_padded(F)): a wave takes a TILE of tile_rows x (64 / tile_rows) neighbouring pixels
// instead of 64 pixels of one row, so that the 64 rays of a one-ray-per-lane walk -- which runs the UNION of their paths -- stay together
// in both image directions (tile_rows adjacent local rows are adjacent image rows: it divides row_block, or the shard is the whole image).
// Returns the pixel's row-major local index (what px_pixel holds and everything downstream uses), or -1 for the padding of edge tiles.
__device__ __forceinline__ long long primary_tile(const Frame& F, long long q) {
    const int lane = (int)(q & 63), tw = 64 / F.tile_rows;
    const long long tile = q >> 6;
    const long long ty = tile / F.tiles_x; const int tx = (int)(tile - ty * F.tiles_x);
    const int jj = lane / tw, xx = lane - jj * tw;
    const long long j = ty * F.tile_rows + jj; const int x = tx * tw + xx;
    return (j < F.local_rows && x < F.h) ? j * F.h + x : -1;
}
__host__ __device__ __forceinline__ long long primary_padded(const Frame& F) {
    return (long long)((F.local_rows + F.tile_rows - 1) / F.tile_rows) * F.tiles_x * 64;
}
__device__ __forceinline__ void pixel_coords(const Frame& F, long long pix, int& y, int& x) {
    const int j = (int)(pix / F.h);
    x = (int)(pix - (long long)j * F.h);
    const int blk = j / F.row_block;
    y = (blk * F.n_shards + F.shard) * F.row_block + (j - blk * F.row_block);
}

// ----------------------------------------------------------------------------------------------
// Variant 1: one lane per pixel, everything in one kernel (cross-check variant; also raycast mode)
// ----------------------------------------------------------------------------------------------
template <typename StackT>
__global__ void __launch_bounds__(kBlock) sq_render_pixels(const SceneView S, const Frame F) {
    extern __shared__ float4 lds_raw[];
    SQ_LDS StackT* stk = to_lds<StackT>(lds_raw) + threadIdx.x;
    const long long pix = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (pix >= (long long)F.local_rows * F.h) return;
    int y, x; pixel_coords(F, pix, y, x);
    const GlobalNodes N{ S.branches, S.cull_child, S.cull_child != nullptr };
    const f3 o0 = sq::mk(F.cam_pos[0], F.cam_pos[1], F.cam_pos[2]);
    const f3 d0 = primary_dir(F.cam_rot, F.w, F.h, y, x);
    const int n = F.samples;
    f3 sum = sq::mk(0, 0, 0);                                           // sum = foldl (+) 0
    const Hit h0 = trace_one(S, N, o0, d0, stk, kBlock);
    if (h0.tri >= 0) {
        const Surface s0 = surface_of(S, h0.tri);
        const f3 p0 = o0 + sq::scale(h0.t, d0);
        if (F.cast) {                                                   // raycast, src/Lib.hs:141-151
            const f3 light = sq::mk(0, 3, -1);
            const float dl = sq::norm(p0 - light);
            const Hit sh = trace_one(S, N, p0, light - p0, stk, kBlock);
            f3 c = sq::mk(0, 0, 0);
            if (!(sh.tri >= 0 && !(hit_dist(p0, light - p0, sh.t) > dl))) c = sq::scale(2 / dl, s0.surf);
            for (int k = 0; k < n; ++k) sum = sum + c;
        } else {
            const long long rix = (long long)n * ((long long)x + (long long)y * (long long)F.w);   // src/Lib.hs:85
#pragma unroll 1
            for (int k = 0; k < n; ++k) {                               // raytrace gen scene ray 0, src/Lib.hs:127-137
                uint32_t n0, n1, n2;
                sq::tfgen3(rix + k, n0, n1, n2);
                f3 L1 = sq::mk(0, 0, 0);
                const f3 d1 = bounce_dir(d0, s0, n0, n1);
                const Hit h1 = trace_one(S, N, p0, d1, stk, kBlock);
                if (h1.tri >= 0) {
                    const Surface s1 = surface_of(S, h1.tri);
                    const f3 p1 = p0 + sq::scale(h1.t, d1);
                    const f3 d2 = bounce_dir(d1, s1, n1, n2);
                    const Hit h2 = trace_one(S, N, p1, d2, stk, kBlock);
                    f3 L2 = sq::mk(0, 0, 0);
                    if (h2.tri >= 0) { const Surface s2 = surface_of(S, h2.tri); L2 = s2.surf * sq::mk(0, 0, 0) + s2.emit; }
                    L1 = s1.surf * L2 + s1.emit;
                }
                sum = sum + (s0.surf * L1 + s0.emit);
            }
        }
    }
    const f3 avg = sq::scale(1 / (float)n, sum);                        // src/Lib.hs:88
    if (F.out_avg) { float* o = F.out_avg + pix * 3; o[0] = avg.x; o[1] = avg.y; o[2] = avg.z; }
    if (F.out_rgb) tonemap(avg, F.out_rgb + pix * 3);
}

// ----------------------------------------------------------------------------------------------
// Variant 2 (default): wavefront pipeline
// ----------------------------------------------------------------------------------------------
struct Work {                 // device workspace of one frame (HBM)
    // per active pixel (a pixel whose primary ray hits), indexed by a in [0, *n_active)
    int32_t* n_active;        // device counter
    int32_t* px_pixel;        // local pixel index
    float*   px_t0;           // primary hit: t
    int32_t* px_tri0;         // primary hit: triangle
    float*   px_sum;          // running ordered sum of sample radiances, 3 floats
    float*   px_mt;           // hit of the pixel's MIRROR bounce ray (reflectRay has no random input, so every
    int32_t* px_mtri;         //   sample of the pixel that mirrors at depth 0 shoots this same ray): t, triangle (-1 = miss)
    // per sample slot sid = k_local * A + a : the ray queue is dense in sid, dead entries are flagged
    uint8_t* state;           // one byte per slot: kDone / kRay1 / kMirror / kRay2.  The trace kernel's refill scan, shade1 and
                              //   sq_accumulate look at this byte first and touch a slot's 32 other bytes only if they need them
    // 45 bytes per slot (round 2: 61).  A slot's two quads are reused as the sample moves on:
    //   org : ray origin.xyz while the ray waits in the queue; the trace kernel puts the ray's HIT into .xy (t bits, triangle)
    //         when it is done with it -- nothing reads an origin after that (ray 1 starts at the pixel's primary hit point, which
    //         sq_shade1 recomputes).  .w = n1 of the sample's generator until sq_shade1 has used it; a kMirror slot, which holds
    //         no ray of its own, keeps n2 in .z as well.
    //   dir : ray direction.xyz.  .w = n2 of the generator (first bounce level), then the triangle hit by ray 1 (second level).
    // The radiance of a finished sample keeps an array of its own: folded into the origin quad as well (33 bytes per slot) the
    // frame measured 0.4 ms slower -- sq_accumulate and sq_shade1 are bound by the bytes they move, and 16-byte records carry
    // 12 bytes of radiance (profiles/r03f_slots_ab.txt).
    float4*  org;
    float4*  dir;
    float*   rad;             // finished sample radiance, 3 floats
    int32_t* head[2];         // dequeue cursors of the persistent trace kernel, one per bounce level
    unsigned long long* stats;  // cumulative trace-kernel statistics (TraceArgs::stats)
    int64_t  slot_capacity;
};
// Slot states.  kRay1 / kRay2: the slot holds a bounce ray of depth 1 / 2 for the trace launch of that level;
// kMirror: the sample mirrors at depth 0 and shares the pixel's mirror ray (traced once per pixel); kDone: its radiance is in `rad`.
constexpr uint8_t kDone = 0, kRay1 = 1, kMirror = 2, kRay2 = 3;

// Diagnostic (option "coresidency", off by default; results unchanged): does a wave of a per-sample kernel run BESIDE the
// resident trace workgroups?  The trace kernel keeps a gauge of its live workgroups in stats[24]; it runs one workgroup per
// CU, so a per-sample wave that starts (stats[26]) or ends (stats[27]) while at least `full` of them are live shares its CU
// with one.  stats[25] counts the per-sample waves that looked.
constexpr int kDiagGauge = 24, kDiagWaves = 25, kDiagStartBeside = 26, kDiagEndBeside = 27;
__device__ __forceinline__ void diag_aux_wave(const Work& W, int diag, bool at_start) {
    if (!diag || (threadIdx.x & 63) != 0) return;
    const unsigned long long live = atomicAdd(&W.stats[kDiagGauge], 0ull);
    if (at_start) atomicAdd(&W.stats[kDiagWaves], 1ull);
    if (live >= (unsigned long long)diag) atomicAdd(&W.stats[at_start ? kDiagStartBeside : kDiagEndBeside], 1ull);
}

// Appends `want` lanes of this wave to a list with one atomic: wave ballot + prefix rank.
__device__ __forceinline__ int wave_append(int32_t* counter, bool want) {
    const unsigned long long m = sq_ballot(want);
    if (m == 0) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// Primary rays: trace once per pixel, compact the pixels that hit.
template <typename StackT>
__global__ void __launch_bounds__(kBlock) sq_primary(const SceneView S, const Frame F, const Work W) {
    extern __shared__ float4 lds_raw[];
    SQ_LDS StackT* stk = to_lds<StackT>(lds_raw) + threadIdx.x;
    const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long pix = q < primary_padded(F) ? primary_tile(F, q) : -1;
    const bool in = pix >= 0;
    Hit h0; h0.tri = -1; h0.t = 0;
    if (in) {
        int y, x; pixel_coords(F, pix, y, x);
        const GlobalNodes N{ S.branches, S.cull_child, S.cull_child != nullptr };
        h0 = trace_one(S, N, sq::mk(F.cam_pos[0], F.cam_pos[1], F.cam_pos[2]), primary_dir(F.cam_rot, F.w, F.h, y, x), stk, kBlock);
    }
    const int a = wave_append(W.n_active, in && h0.tri >= 0);
    if (a >= 0) {
        W.px_pixel[a] = (int32_t)pix; W.px_t0[a] = h0.t; W.px_tri0[a] = h0.tri;
        W.px_sum[3 * a] = 0.0f; W.px_sum[3 * a + 1] = 0.0f; W.px_sum[3 * a + 2] = 0.0f;
    }
}

struct Pixel0 { f3 p0, d0; Surface s0; int y, x; };
__device__ __forceinline__ Pixel0 load_pixel0(const SceneView& S, const Frame& F, const Work& W, int a) {
    Pixel0 P;
    pixel_coords(F, (int)W.px_pixel[a], P.y, P.x);
    P.d0 = primary_dir(F.cam_rot, F.w, F.h, P.y, P.x);
    P.p0 = sq::mk(F.cam_pos[0], F.cam_pos[1], F.cam_pos[2]) + sq::scale(W.px_t0[a], P.d0);   // intersectPoint, src/Geometry.hs:134
    P.s0 = surface_of(S, W.px_tri0[a]);
    return P;
}
// surfColor == 0 (an emitter such as data/scene.sq:13-15) makes `surfColor * raytrace ...` exactly +0
// whenever the nested radiance is finite and >= +0, so the nested rays need not be traced.  Only used
// when every material component is >= +0 (checked at upload), where that premise holds.
__device__ __forceinline__ bool absorbs(const SceneView& S, const Surface& s) {
    return S.nonneg_materials && s.surf.x == 0.0f && s.surf.y == 0.0f && s.surf.z == 0.0f;
}
__device__ __forceinline__ void store_rad(const Work& W, long long sid, f3 L) {
    W.rad[3 * sid] = L.x; W.rad[3 * sid + 1] = L.y; W.rad[3 * sid + 2] = L.z;
}
__device__ __forceinline__ int2 slot_hit(const float4& org) { return make_int2(__float_as_int(org.x), __float_as_int(org.y)); }   // what the trace kernel left in org.xy

// Depth-0 bounce of every sample of the batch: RNG, bounceRay, ray 1 into slot sid (src/Lib.hs:133-134).
// One thread per active pixel (blockIdx.y splits the samples of a pixel when a frame has few pixels): everything that is
// the same for every sample of a pixel -- the pixel's coordinates, primary direction, hit point, surface, seed base -- is
// computed once, not 256 times; consecutive threads still write consecutive slots (sid = k * A + a).
__global__ void __launch_bounds__(kBlock) sq_gen_bounce1(const SceneView S, const Frame F, const Work W, int k_base, int k_count) {
    const int A = *W.n_active;
    diag_aux_wave(W, F.diag, true);
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < A; a += gridDim.x * kBlock) {
        const Pixel0 P = load_pixel0(S, F, W, a);
        const bool absorbing = absorbs(S, P.s0);
        const f3 rad_absorbing = P.s0.surf * sq::mk(0, 0, 0) + P.s0.emit;
        const long long rix = (long long)F.samples * ((long long)P.x + (long long)P.y * (long long)F.w);   // src/Lib.hs:85
        for (int kl = blockIdx.y; kl < k_count; kl += gridDim.y) {
            const long long sid = (long long)kl * A + a;
            if (absorbing) {
                store_rad(W, sid, rad_absorbing);
                W.state[sid] = kDone;
                continue;
            }
            uint32_t n0, n1, n2;
#ifdef SQ_DIAG_COHERENT   // timing experiment only (WRONG image): every pixel draws the same numbers, so neighbouring rays are parallel
            sq::tfgen3((long long)(k_base + kl), n0, n1, n2);
#else
            sq::tfgen3(rix + (k_base + kl), n0, n1, n2);                // mkTFGen (rix + k), src/Lib.hs:86
#endif
            if (!scatters(P.s0, n0)) {                                  // mirror: traced once per pixel (sq_mirror1_*); the slot only carries n1, n2
                W.state[sid] = kMirror;
                *reinterpret_cast<float2*>(reinterpret_cast<float*>(W.org + sid) + 2) = make_float2(__uint_as_float(n2), __uint_as_float(n1));   // .z = n2, .w = n1
                continue;
            }
            const f3 d1 = scatter_dir(P.d0, P.s0, n0, n1);
            W.state[sid] = kRay1;
            W.org[sid] = make_float4(P.p0.x, P.p0.y, P.p0.z, __uint_as_float(n1));
            W.dir[sid] = make_float4(d1.x, d1.y, d1.z, __uint_as_float(n2));
        }
    }
    diag_aux_wave(W, F.diag, false);
}

// The depth-0 mirror ray of every active pixel, once per frame (slot a = active pixel a).
// `base`: first of the *n_active slots the mirror rays use (0 when they have a launch of their own, the spare region
// behind the sample slots when they ride at the head of the first bounce launch).
__global__ void __launch_bounds__(kBlock) sq_mirror1_gen(const SceneView S, const Frame F, const Work W, long long base) {
    const int A = *W.n_active;
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < A; a += gridDim.x * kBlock) {
        const Pixel0 P = load_pixel0(S, F, W, a);
        const long long sl = base + a;
        if (absorbs(S, P.s0)) { W.state[sl] = kDone; continue; }
        const f3 d1 = mirror_dir(P.d0, P.s0);
        W.state[sl] = kRay1;
        W.org[sl] = make_float4(P.p0.x, P.p0.y, P.p0.z, 0.0f);
        W.dir[sl] = make_float4(d1.x, d1.y, d1.z, 0.0f);
    }
}
__global__ void __launch_bounds__(kBlock) sq_mirror1_store(const Work W, long long base) {
    const int A = *W.n_active;
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < A; a += gridDim.x * kBlock) {
        const bool live = W.state[base + a] == kRay1;
        const int2 hit = slot_hit(W.org[base + a]);
        W.px_mt[a] = live ? __int_as_float(hit.x) : 0.0f;
        W.px_mtri[a] = live ? hit.y : -1;
    }
}

// Primary rays through the pooled trace kernel (option "primary_pooled"): one slot per pixel of this shard, one trace launch with
// k_count = 1, then the hits are compacted into the active-pixel list exactly as sq_primary does.  The one-ray-per-lane walk of
// sq_primary(_resident) lasts as long as its most expensive wave (64 neighbouring pixels on dense geometry: 0.5 ms on the headline
// scene however small the shard); the pooled leaf phase walks such a wave faster.  W.n_active[48] is the launch's queue length.
__global__ void __launch_bounds__(kBlock) sq_primary_gen(const Frame F, const Work W, long long total) {
    if (blockIdx.x == 0 && threadIdx.x == 0) W.n_active[48] = (int32_t)total;
    for (long long pix = (long long)blockIdx.x * kBlock + threadIdx.x; pix < total; pix += (long long)gridDim.x * kBlock) {
        int y, x; pixel_coords(F, pix, y, x);
        const f3 d = primary_dir(F.cam_rot, F.w, F.h, y, x);
        W.state[pix] = kRay1;
        W.org[pix] = make_float4(F.cam_pos[0], F.cam_pos[1], F.cam_pos[2], 0.0f);
        W.dir[pix] = make_float4(d.x, d.y, d.z, 0.0f);
    }
}
__global__ void __launch_bounds__(kBlock) sq_primary_store(const Work W, long long total) {
    for (long long base = (long long)blockIdx.x * kBlock; base < total; base += (long long)gridDim.x * kBlock) {   // whole waves stay together (ballot)
        const long long pix = base + threadIdx.x;
        const bool in = pix < total;
        const int2 hit = in ? slot_hit(W.org[pix]) : make_int2(0, -1);
        const int a = wave_append(W.n_active, in && hit.y >= 0);
        if (a >= 0) {
            W.px_pixel[a] = (int32_t)pix; W.px_t0[a] = __int_as_float(hit.x); W.px_tri0[a] = hit.y;
            W.px_sum[3 * a] = 0.0f; W.px_sum[3 * a + 1] = 0.0f; W.px_sum[3 * a + 2] = 0.0f;
        }
    }
}

// After ray 1: a miss finishes the sample; a hit either finishes it (absorbing surface) or puts ray 2 in the slot.
// One thread per active pixel, like sq_gen_bounce1: the primary surface, the hit point and the pixel's mirror ray and its hit
// are per-pixel values.  It waits on memory two thirds of its time, so a slot's state byte decides what else is read (nothing
// for a finished slot, the generator words for a mirrored one, ray and hit for a traced one), the state and generator words
// are requested two samples ahead and the rest one sample ahead.
__global__ void __launch_bounds__(kBlock) sq_shade1(const SceneView S, const Frame F, const Work W, int k_count) {
    const int A = *W.n_active;
    diag_aux_wave(W, F.diag, true);
    struct First { uint8_t st; float4 org; };     // org: .xy = the hit the trace kernel left (traced slots), .w = n1
    struct Second { float4 dir; };                // .w = n2.  Ray 1 starts at the pixel's primary hit point P.p0 (sq_gen_bounce1 stored that very value): no origin is re-read
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < A; a += gridDim.x * kBlock) {
        const Pixel0 P = load_pixel0(S, F, W, a);
        const Surface& s0 = P.s0;
        const f3 d1_mirror = mirror_dir(P.d0, P.s0);
        const int2 hit_mirror = make_int2(__float_as_int(W.px_mt[a]), W.px_mtri[a]);
        const int step = gridDim.y;
        auto first = [&](int kl) { First f; const long long sid = (long long)kl * A + a; f.st = W.state[sid]; f.org = W.org[sid]; return f; };
        auto second = [&](int kl, uint8_t st) {
            Second q{};
            if (st == kRay1) { const long long sid = (long long)kl * A + a; q.dir = W.dir[sid]; }
            return q;
        };
        int kl = blockIdx.y;
        First f1{}, f2{}; Second q1{};
        if (kl < k_count) { f1 = first(kl); q1 = second(kl, f1.st); }
        if (kl + step < k_count) f2 = first(kl + step);
        for (; kl < k_count; kl += step) {
            const First cur = f1; const Second q = q1;
            f1 = f2;
            if (kl + step < k_count) q1 = second(kl + step, f1.st);
            if (kl + 2 * step < k_count) f2 = first(kl + 2 * step);
            if (cur.st == kDone) continue;
            const long long sid = (long long)kl * A + a;
            const uint2 r = make_uint2(__float_as_uint(cur.org.w), __float_as_uint(cur.st == kMirror ? cur.org.z : q.dir.w));   // (n1, n2)
            f3 d1, p0;
            int2 hit;
            if (cur.st == kMirror) { d1 = d1_mirror; p0 = P.p0; hit = hit_mirror; }          // the pixel's mirror ray and its hit
            else { d1 = sq::mk(q.dir.x, q.dir.y, q.dir.z); p0 = P.p0; hit = slot_hit(cur.org); }
            const int tri1 = hit.y;
            if (tri1 < 0) {                                             // raytrace ... 1 = black
                store_rad(W, sid, s0.surf * sq::mk(0, 0, 0) + s0.emit);
                W.state[sid] = kDone;
                continue;
            }
            const Surface s1 = surface_of(S, tri1);
            if (absorbs(S, s1)) {
                const f3 L1 = s1.surf * sq::mk(0, 0, 0) + s1.emit;
                store_rad(W, sid, s0.surf * L1 + s0.emit);
                W.state[sid] = kDone;
                continue;
            }
            const f3 p1 = p0 + sq::scale(__int_as_float(hit.x), d1);
            const f3 d2 = bounce_dir(d1, s1, r.x, r.y);                 // gen advanced by one: x = u = p(n1), v = p(n2)
            // Ray 2 is the last one: all it contributes is L2 = s2*0 + e2, the emission of whatever it hits
            // (src/Lib.hs:129,135-137).  Whatever the traversal returns is a triangle that mollerTrumbore accepted
            // for this very ray, so if the SAME function rejects every emissive triangle, the result is a
            // non-emissive hit or a miss, and L2 is exactly (+0,+0,+0) either way (materials are finite).
            // Only rays that could reach an emitter are traced.
            if (S.n_emitters >= 0) {
                bool may_reach = false;
                for (int j = 0; j < S.n_emitters && !may_reach; ++j) {
                    const int et = S.emitters[j];
                    const float* tp = S.tris + 9 * (size_t)et;
                    float t_unused;
                    may_reach = moller_trumbore(p1, d2, sq::mk(tp[0], tp[1], tp[2]), sq::mk(tp[3], tp[4], tp[5]), sq::mk(tp[6], tp[7], tp[8]), t_unused);
                }
                if (!may_reach) {
                    const f3 L1 = s1.surf * sq::mk(0, 0, 0) + s1.emit;
                    store_rad(W, sid, s0.surf * L1 + s0.emit);
                    W.state[sid] = kDone;
                    continue;
                }
            }
            W.state[sid] = kRay2;
            W.org[sid] = make_float4(p1.x, p1.y, p1.z, 0.0f);
            W.dir[sid] = make_float4(d2.x, d2.y, d2.z, __int_as_float(tri1));
        }
    }
    diag_aux_wave(W, F.diag, false);
}

// After ray 2: L2 = s2*0 + e2 (or black), L1 = s1*L2 + e1, L0 = s0*L1 + e0   (src/Lib.hs:135-137).
// Not a kernel of its own: the 8 % of the slots that still hold a second bounce ray are folded where their radiance is
// consumed (sq_accumulate), which saves a pass over every slot's state and the round trip of their radiance through HBM.
__device__ __forceinline__ f3 shade2_radiance(const SceneView& S, const Work& W, long long sid, const Surface& s0) {
    const int tri1 = reinterpret_cast<const int*>(W.dir + sid)[3], tri2 = reinterpret_cast<const int*>(W.org + sid)[1];   // dir.w, and the hit's triangle in org.y: 4 bytes each
    f3 L2 = sq::mk(0, 0, 0);
    if (tri2 >= 0) { const Surface s2 = surface_of(S, tri2); L2 = s2.surf * sq::mk(0, 0, 0) + s2.emit; }
    const Surface s1 = surface_of(S, tri1);
    const f3 L1 = s1.surf * L2 + s1.emit;
    return s0.surf * L1 + s0.emit;
}

// sum outcomes, in sample order (src/Lib.hs:88); on the last batch: avg, tonemap, store.
// GROUPED: see the comment in the loop; two kernels because the grouped loop's registers cost the plain one a third of its waves.
template <bool GROUPED>
__global__ void __launch_bounds__(kBlock) sq_accumulate(const SceneView S, const Frame F, const Work W, int k_count, int last) {
    const int A = *W.n_active;
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < A; a += gridDim.x * kBlock) {
        f3 sum = sq::mk(W.px_sum[3 * a], W.px_sum[3 * a + 1], W.px_sum[3 * a + 2]);
        const Surface s0 = surface_of(S, W.px_tri0[a]);
        // The sum is the reference's left fold over the samples (src/Lib.hs:88): one dependent add per sample.  With fewer active
        // pixels than threads (one rank's share of a frame at 8 ranks) a thread is a chain of `samples` memory round trips and the
        // launch lasts as long as that chain: there (GROUPED, chosen by the host from the shard's pixel count) the states and radiances of 16 samples are requested together (a slot that
        // still holds a second bounce ray has no radiance yet; what is read there is not used) and then added in order -- 384 ->
        // 280 us on such a share.  A whole frame is bound by the bytes it moves and keeps the one-sample-ahead loop (0.83 ms;
        // the grouped loop takes 0.98 ... 1.58 ms there with groups of 2 ... 16: profiles/r03l_accumulate_groups.txt).
        if constexpr (GROUPED) {
            constexpr int kAccGroup = 16;
            for (int k0 = 0; k0 < k_count; k0 += kAccGroup) {
                uint8_t st[kAccGroup]; float rx[kAccGroup], ry[kAccGroup], rz[kAccGroup];
#pragma unroll
                for (int j = 0; j < kAccGroup; ++j) {
                    const int k = min(k0 + j, k_count - 1);             // past the end: the last sample again, dropped below
                    const long long sid = (long long)k * A + a;
                    st[j] = W.state[sid];
                    rx[j] = W.rad[3 * sid]; ry[j] = W.rad[3 * sid + 1]; rz[j] = W.rad[3 * sid + 2];
                }
#pragma unroll
                for (int j = 0; j < kAccGroup; ++j) {
                    if (k0 + j >= k_count) break;
                    const long long sid = (long long)(k0 + j) * A + a;
                    const f3 rad = (st[j] == kRay2) ? shade2_radiance(S, W, sid, s0) : sq::mk(rx[j], ry[j], rz[j]);
                    sum = sum + rad;
                }
            }
        } else {
            uint8_t st = k_count > 0 ? W.state[a] : kDone;             // the next slot's state is requested one sample ahead
            for (int k = 0; k < k_count; ++k) {
                const long long sid = (long long)k * A + a;
                const uint8_t cur = st;
                if (k + 1 < k_count) st = W.state[sid + A];
                const f3 rad = (cur == kRay2) ? shade2_radiance(S, W, sid, s0) : sq::mk(W.rad[3 * sid], W.rad[3 * sid + 1], W.rad[3 * sid + 2]);
                sum = sum + rad;
            }
        }
        if (!last) { W.px_sum[3 * a] = sum.x; W.px_sum[3 * a + 1] = sum.y; W.px_sum[3 * a + 2] = sum.z; continue; }
        const f3 avg = sq::scale(1 / (float)F.samples, sum);
        const long long pix = W.px_pixel[a];
        if (F.out_avg) { float* o = F.out_avg + pix * 3; o[0] = avg.x; o[1] = avg.y; o[2] = avg.z; }
        if (F.out_rgb) tonemap(avg, F.out_rgb + pix * 3);
    }
}

// The dominant kernel.  Persistent: each wave reserves a chunk of the (dense) ray queue with one
// atomic, compacts the chunk's live entries with ballots into a small LDS list, and each lane pulls
// its next ray from that list as soon as the previous one is finished, so a wave's lanes stay busy
// although ray lengths differ by 10x and although part of the queue is dead.  The first n_lds
// branches (breadth-first = the top of the tree) are staged in LDS once per workgroup; every lane
// keeps its frame stack in LDS (lane-minor layout, one word per frame, at most height-1 frames).
struct TraceArgs {
    float4* org; const float4* dir;          // a finished ray's hit (t bits, triangle) goes into org[slot].xy
    const uint8_t* state; int32_t want;      // a slot is in this launch's queue iff state[slot] == want (kRay1 / kRay2)
    long long front_base; int32_t front;     // front != 0: the queue starts with *n_active extra entries, the slots
                                             //   front_base + [0, *n_active) (the per-pixel mirror rays ride at the head of the
                                             //   first bounce launch instead of having a launch, and a ramp-down, of their own)
    const int32_t* n_active; int32_t k_count; int32_t* head;
    int32_t n_lds; int32_t stack_cap; int32_t straggler_lanes;
    int32_t chunk;               // slots per reservation: a multiple of 64, at most kChunkResident / kChunkStreaming
    int32_t guide_shift;         // towards the end of the queue a reservation shrinks to (slots left >> guide_shift), so that the
                                 //   launch does not end with a few waves still working through a full reservation (first bounce level
                                 //   only by default: on the sparse second-level queue it costs more round trips than it saves, option "guided")
    int32_t refill_min;          // pooled form: idle lanes a wave collects before it fetches new rays for them
    int32_t flush_min;           // pooled form: a trailing part-filled window of the pair pool is run at once from this many pairs on
    int32_t descend_extra, descend_lanes;   // pooled form: further branch steps per iteration for lanes that keep descending, and how many such lanes it takes
    int32_t diag;                // option "coresidency": keep the gauge of live workgroups in stats[24]
    int32_t prio;                // option "trace_prio": s_setprio level of the trace kernel's waves when they start (0 = leave it; the pooled
                                 //   kernel then sets its own levels per phase, SQ_SETPRIO, so it only lasts in pool = 0 launches), for the overlapped
                                 //   schedules: per-sample kernels that share a SIMD with a trace workgroup then only get the issue
                                 //   slots the trace waves leave free
    int32_t pixel_major;         // queue ORDER: 0 = slot order (sample-major: neighbouring pixels, one sample each), 1 = all samples
                                 //   of a pixel in a row, so that a wave's rays start from one surface point (slots stay where they are)
    unsigned long long* stats;   // [0] rays traced; PROFILE builds: [1] advance iterations (waves), [2] lanes unwinding,
                                 // [3] lanes descending, [4] leaf iterations (waves), [5] lanes testing a triangle,
                                 // [6] outer iterations (waves), [7] refill executions (waves), [8] lanes refilled
                                 // pooled form: [1] iterations (waves), [2] lanes unwinding, [3] lanes descending,
                                 // [4] pair windows (waves), [5] pairs tested, [6] hits folded, [7] refill executions, [8] lanes refilled
};
// LDS carve-up of the trace kernel (bytes, all 16-B aligned), shared by host and device.
struct TraceLds { uint32_t quads, refs, verts, trix, live, tab, stack, total; };
__host__ __device__ inline TraceLds trace_lds_layout(int n_lds, bool resident, int n_verts, int n_tris,
                                                     int block, int stack_cap, int stack_elem, bool pool) {
    auto al = [](uint32_t b) { return (b + 15u) & ~15u; };
    TraceLds L; uint32_t off = 0;
    L.verts = off; off += resident ? al((uint32_t)n_verts * 16u) : 0u;     // first: a vertex's LDS address is its byte offset (ResidentTris)
    L.quads = off; off += al((uint32_t)n_lds * (resident ? 32u : 48u));   // resident: 2 quads per branch; streaming: 3
    L.refs = off;  off += resident ? al((uint32_t)n_lds * 8u) : 0u;
    L.trix = off;  off += resident ? al((uint32_t)(n_tris + ResidentTris::kRunPad) * 8u) : 0u;  // + zero records: get_run may read past the last triangle
    L.live = off;  off += al((uint32_t)(block / 64) * (uint32_t)(resident ? kChunkResident : kChunkStreaming) * (uint32_t)sizeof(LiveT));
    L.tab = off;   off += pool ? al((uint32_t)block * kPoolWindows) : 0u;   // pooled form: one byte per lane and window in flight
    L.stack = off; off += al((uint32_t)block * (uint32_t)stack_cap * (uint32_t)stack_elem);
    L.total = off;
    return L;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
// Copies the resident form of the scene (sq_scene.h) into the workgroup's LDS and points N / G at it.  The caller syncs.
template <int BLOCK>
__device__ __forceinline__ void stage_resident_scene(const SceneView& S, int n_branches, char* lds, const TraceLds& L, ResidentNodes& N, ResidentTris& G) {
    SQ_LDS v4f* lquads = to_lds<v4f>(lds + L.quads);
    SQ_LDS v2i* lrefs = to_lds<v2i>(lds + L.refs);
    SQ_LDS v4f* lv = to_lds<v4f>(lds + L.verts);
    SQ_LDS v4us* lt = to_lds<v4us>(lds + L.trix);
    SQ_LDS v2f* lboxes = to_lds<v2f>(lds + L.quads + 16u * (uint32_t)n_branches);   // kTailLayout: 24-byte boxes behind the 16-byte tails
    if constexpr (ResidentNodes::kTailLayout) {
        for (int i = threadIdx.x; i < n_branches; i += BLOCK) {
            const uint32_t* r = S.rbranch + 10 * (size_t)i;
            lquads[i] = v4f{ __uint_as_float(r[3]), __uint_as_float(r[7]), __uint_as_float(r[8]), __uint_as_float(r[9]) };
            lboxes[3 * i] = v2f{ __uint_as_float(r[0]), __uint_as_float(r[1]) };
            lboxes[3 * i + 1] = v2f{ __uint_as_float(r[2]), __uint_as_float(r[4]) };
            lboxes[3 * i + 2] = v2f{ __uint_as_float(r[5]), __uint_as_float(r[6]) };
        }
    } else
    for (int i = threadIdx.x; i < n_branches; i += BLOCK) {
        const uint32_t* r = S.rbranch + 10 * (size_t)i;
        lquads[i] = v4f{ __uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]) };
        lquads[n_branches + i] = v4f{ __uint_as_float(r[4]), __uint_as_float(r[5]), __uint_as_float(r[6]), __uint_as_float(r[7]) };
        lrefs[i] = v2i{ (int)r[8], (int)r[9] };
    }
#if SQ_STAGE_BATCHED
    // Several loads in flight per thread before the first LDS store (the plain loops wait for every load before the next is issued):
    // a trace launch's staging is on the path of every launch, and a rank's share of a frame at 8 ranks has three launches in 8.5 ms.
    for (int base = threadIdx.x; base < S.n_verts; base += BLOCK * 4) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = base + k * BLOCK; if (i < S.n_verts) v[k] = S.verts4[i]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = base + k * BLOCK; if (i < S.n_verts) lv[i] = v4f{ v[k].x, v[k].y, v[k].z, v[k].w }; }
    }
    for (int base = threadIdx.x; base < S.n_tris; base += BLOCK * 8) {     // vertex indices become byte offsets into the vertex table
        ushort4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = base + k * BLOCK; if (i < S.n_tris) t[k] = S.trix[i]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = base + k * BLOCK; if (i < S.n_tris) lt[i] = v4us{ (unsigned short)(t[k].x * 16u), (unsigned short)(t[k].y * 16u), (unsigned short)(t[k].z * 16u), t[k].w }; }
    }
#else
    for (int i = threadIdx.x; i < S.n_verts; i += BLOCK) { const float4 v = S.verts4[i]; lv[i] = v4f{ v.x, v.y, v.z, v.w }; }
    for (int i = threadIdx.x; i < S.n_tris; i += BLOCK) {                  // vertex indices become byte offsets into the vertex table
        const ushort4 t = S.trix[i];
        lt[i] = v4us{ (unsigned short)(t.x * 16u), (unsigned short)(t.y * 16u), (unsigned short)(t.z * 16u), t.w };
    }
#endif
    if (threadIdx.x < ResidentTris::kRunPad) lt[S.n_tris + threadIdx.x] = v4us{ 0, 0, 0, 0 };
    N = ResidentNodes{ lquads, lquads + n_branches, lrefs, lboxes, S.cull_child16 != nullptr, S.cull_child16, S.rtail, S.incremental_ok != 0 };
    G = ResidentTris{ lt };
    if ((uintptr_t)lv != 0) __builtin_trap();                              // the kernels that use this have no static LDS: dynamic LDS starts at address 0
}

// Primary rays with the scene in LDS (the resident form): same per-ray code as sq_primary, but a branch or triangle costs an
// LDS read instead of an L2 round trip.  A primary ray is a chain of ~200 dependent reads, so on small frames -- one rank's
// share of a frame at 8 ranks -- the launch is as long as that chain: 0.58 ms from L2, 0.1-0.2 ms from LDS.
template <typename StackT>
__global__ void __launch_bounds__(kResidentBlock) sq_primary_resident(const SceneView S, const Frame F, const Work W, int stack_cap) {
    extern __shared__ float4 lds_raw[];
    char* lds = reinterpret_cast<char*>(lds_raw);
    const TraceLds L = trace_lds_layout(S.n_branches, true, S.n_verts, S.n_tris, kResidentBlock, stack_cap, (int)sizeof(StackT), false);
    SQ_LDS StackT* stk = to_lds<StackT>(lds + L.stack) + threadIdx.x;
    ResidentNodes N; ResidentTris G;
    stage_resident_scene<kResidentBlock>(S, S.n_branches, lds, L, N, G);
    __syncthreads();
    const long long total = primary_padded(F);
    for (long long base = (long long)blockIdx.x * kResidentBlock; base < total; base += (long long)gridDim.x * kResidentBlock) {
        const long long pix = primary_tile(F, base + threadIdx.x);          // (total is a multiple of 64: whole waves)
        const bool in = pix >= 0;
        Hit h0; h0.tri = -1; h0.t = 0;
        if (in) {
            int y, x; pixel_coords(F, pix, y, x);
            h0 = trace_one(S, N, G, S.rroot, sq::mk(F.cam_pos[0], F.cam_pos[1], F.cam_pos[2]), primary_dir(F.cam_rot, F.w, F.h, y, x), stk, kResidentBlock);
        }
        const int a = wave_append(W.n_active, in && h0.tri >= 0);
        if (a >= 0) {
            W.px_pixel[a] = (int32_t)pix; W.px_t0[a] = h0.t; W.px_tri0[a] = h0.tri;
            W.px_sum[3 * a] = 0.0f; W.px_sum[3 * a + 1] = 0.0f; W.px_sum[3 * a + 2] = 0.0f;
        }
    }
}

template <typename StackT, bool RESIDENT, int BLOCK, bool PROFILE, bool POOL>
__device__ __forceinline__ void trace_rays_body(const SceneView& S, const TraceArgs& A) {
    extern __shared__ float4 lds_raw[];
    char* lds = reinterpret_cast<char*>(lds_raw);
    const TraceLds L = trace_lds_layout(A.n_lds, RESIDENT, S.n_verts, S.n_tris, BLOCK, A.stack_cap, (int)sizeof(StackT), POOL);
    constexpr int kChunk = RESIDENT ? kChunkResident : kChunkStreaming;
    SQ_LDS LiveT* live = to_lds<LiveT>(lds + L.live) + (threadIdx.x >> 6) * kChunk;   // this wave's list
    SQ_LDS StackT* stk = to_lds<StackT>(lds + L.stack) + threadIdx.x;
    SQ_LDS v4f* lquads = to_lds<v4f>(lds + L.quads);
    using NodeSrc = typename std::conditional<RESIDENT, ResidentNodes, HybridNodes>::type;
    using TriSrc = typename std::conditional<RESIDENT, ResidentTris, GlobalTris>::type;
    NodeSrc N; TriSrc G;
    uint32_t root_ref;
    if constexpr (RESIDENT) {                                         // stage the whole scene (coalesced loads)
        stage_resident_scene<BLOCK>(S, A.n_lds, lds, L, N, G);
        root_ref = S.rroot;
    } else {
        for (int i = threadIdx.x; i < 3 * A.n_lds; i += BLOCK) { const float4 q = S.branches[i]; lquads[i] = v4f{ q.x, q.y, q.z, q.w }; }
#if SQ_STREAM_CULL16
        N = HybridNodes{ lquads, HybridNodes::kMerged ? S.branches_m : S.branches, (uint32_t)A.n_lds, S.cull_child != nullptr, S.cull_child16 };
#else
        N = HybridNodes{ lquads, S.branches, (uint32_t)A.n_lds, S.cull_child != nullptr, S.cull_child };
#endif
        G = GlobalTris{ S.tris, S.leaves, S.packed_leaves != 0, (size_t)S.n_tris * sizeof(DevTri) > ((size_t)4 << 20) };
        root_ref = S.root_ref;
    }
    __syncthreads();
    if (A.diag && threadIdx.x == 0) atomicAdd(&A.stats[kDiagGauge], 1ull);
    if (A.prio == 1) __builtin_amdgcn_s_setprio(1); else if (A.prio == 2) __builtin_amdgcn_s_setprio(2); else if (A.prio == 3) __builtin_amdgcn_s_setprio(3);
    const long long n_front = A.front ? (long long)(*A.n_active) : 0;
    const long long n = (long long)(*A.n_active) * A.k_count + n_front;          // queue positions; slot_of() maps them to slots
    const unsigned n_pix = (unsigned)(*A.n_active), kq = (unsigned)A.k_count;
    auto slot_of = [&](long long q) -> long long {
        if (q < n_front) return A.front_base + q;
        if (!A.pixel_major) return q - n_front;
        const unsigned r = (unsigned)(q - n_front), a = r / kq, k = r - a * kq;   // slot = sample * pixels + pixel (Work)
        return (long long)k * n_pix + a;
    };
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    long long chunk_base = 0;               // wave-uniform
    int my_chunk = A.chunk;                 // wave-uniform: size of this wave's next reservation
    int list_pos = 0, list_len = 0;         // wave-uniform
    bool exhausted = false;                 // wave-uniform
    long long my_ray = -1;
    unsigned long long n_traced = 0;        // wave-uniform
    unsigned long long pf_adv = 0, pf_leaf = 0, pf_outer = 0, pf_refill = 0;   // wave-uniform (PROFILE)
    unsigned int pl_unw = 0, pl_desc = 0, pl_tri = 0, pl_ref = 0;               // per lane (PROFILE)
    Trav T; T.mode = M_DONE; T.sp = 0; T.cur = 0; T.R.tri = -1; T.R.t = 0;
    T.o = T.d = T.df = sq::mk(0, 0, 0);
    int sub_end[3] = { 0, 0, 0 };           // wave-uniform: list positions where the chunk's 2nd, 3rd, 4th 256-slot block start
    auto refill = [&](unsigned long long m, bool idle) {             // m = ballot(idle), wave-uniform
        while (!exhausted && list_pos == list_len) {                    // reserve and compact the next chunk
            int base = 0;
            if (lane == 0) base = atomicAdd(A.head, my_chunk);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n) { exhausted = true; break; }
            const int this_chunk = my_chunk;
            {   // guided self-scheduling: the next reservation is 1/(4 x waves) of what is left, between 64 and A.chunk slots
                const long long left = n - ((long long)base + this_chunk);
                const long long want = A.guide_shift >= 62 ? (long long)A.chunk : ((left >> A.guide_shift) & ~63ll);
                my_chunk = (int)(want < 64 ? 64 : (want > A.chunk ? A.chunk : want));
            }
            chunk_base = base; list_pos = 0; list_len = 0;
            sub_end[0] = sub_end[1] = sub_end[2] = 0x7fffffff;          // blocks a short reservation does not reach
#pragma unroll
            for (int j = 0; j < kChunk / 64; ++j) {
                if (j * 64 >= this_chunk) break;
                const long long idx = chunk_base + j * 64 + lane;
                const bool alive = idx < n && A.state[slot_of(idx)] == (uint8_t)A.want;
                const unsigned long long am = sq_ballot(alive);
                if (alive) live[list_len + __popcll(am & lt_mask)] = (LiveT)((j & 3) * 64 + lane);   // index within its 256-slot block
                list_len += __popcll(am);
                if ((j & 3) == 3 && j / 4 < 3) sub_end[j / 4] = list_len;
            }
            n_traced += (unsigned long long)list_len;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (exhausted) return;
        const int rank = __popcll(m & lt_mask);
        const int avail = list_len - list_pos;
        if (PROFILE) ++pf_refill;
        if (idle && rank < avail) {
            if (PROFILE) ++pl_ref;
            const int p = list_pos + rank;
            int block = 0;
            if (kChunk > 256) block += p >= sub_end[0];
            if (kChunk > 512) block += p >= sub_end[1];
            if (kChunk > 768) block += p >= sub_end[2];
            my_ray = slot_of(chunk_base + block * 256 + live[p]);
            const float4 o = A.org[my_ray], d = A.dir[my_ray];
            trav_begin(T, S, root_ref, sq::mk(o.x, o.y, o.z), sq::mk(d.x, d.y, d.z));
        }
        list_pos += min(__popcll(m), avail);
    };
    if constexpr (POOL) {
        // Pooled form.  One iteration offers every lane one return (pop a frame), one branch step, and then tests the
        // triangles of ALL leaves the wave's rays have open as a pool of (ray, triangle) pairs: the pairs are numbered
        // by a prefix sum over the lanes' leaf sizes and pair p = 64*window + lane is tested by lane `lane`, whatever
        // ray owns it (the owner's origin, direction and triangle offset come over the DPP/bpermute network).  So a
        // leaf of 14 triangles beside leaves of 3 no longer holds 64 lanes for 14 rounds: the wave runs
        // ceil(sum / 64) full-width rounds.  Accepted hits are folded into the owning lane's R in pair order, which is
        // leaf order, with the very comparison of the one-lane leaf loop (minimumBy's rule, src/BIH.hs:105-109) --
        // the arithmetic of mollerTrumbore does not depend on the lane that runs it.
        constexpr int KW = kPoolWindows;
        constexpr bool kFlat = (RESIDENT ? !ResidentNodes::kBoxInRegisters && !ResidentNodes::kIncremental : (SQ_FLAT_STREAM != 0) && (SQ_STREAM_CULL16 != 0) && !HybridNodes::kBoxInRegisters)
                               && !PROFILE && (SQ_FLAT_STEPS != 0) && (SQ_DESCEND_PREFETCH == 0);
        bool flat_ok = false;               // wave-uniform (kFlat): every ray of the wave is safe and the scene has culling boxes
        SQ_LDS uint8_t* tab = to_lds<uint8_t>(lds + L.tab) + (threadIdx.x >> 6) * (64 * KW);   // this wave's window-head tables
        for (int k = 0; k < KW; ++k) tab[k * 64 + lane] = 0;
        const int lane_tag = lane * 4 + 1;  // a lane's mark in the head table: non-zero, grows with the lane, and is its ds_bpermute address
        int lf_first = 0, lf_cnt = 0;       // M_LEAFQ: the untested rest of this lane's open leaf
        bool carry = false;                 // wave-uniform: the previous iteration left queued pairs untested
        unsigned int pl_hit = 0, pl_hslow = 0;
        TravProf prof{};
        BranchPf pf; pf.idx = 0xffffffffu; pf.cb_ok = false;   // near-child prefetch (SQ_DESCEND_PREFETCH; unused and optimised away when off)
        // PROFILE: wave time per section of the loop (s_memtime ticks = shader cycles; the stamps themselves cost ~10 %)
        unsigned long long tsec[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = PROFILE ? __builtin_amdgcn_s_memtime() : 0;
        auto stamp = [&](int sec) { if (PROFILE) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tsec[sec] += now - tlast; tlast = now; } };
        for (;;) {
            if (PROFILE) ++pf_adv;
            const bool idle = (T.mode == M_DONE);
            const unsigned long long m = sq_ballot(idle);       // before the store's exec region: the compare's mask IS the ballot
            if (idle && my_ray >= 0) { *reinterpret_cast<int2*>(A.org + my_ray) = make_int2(__float_as_int(T.R.t), T.R.tri); my_ray = -1; }
            if (m) {
                if (__popcll(m) >= A.refill_min || m == ~0ull) {
                    refill(m, idle);
                    // the flat steps are for waves whose rays are all safe (a ray's flag only changes here) in scenes with culling boxes
                    if constexpr (kFlat) flat_ok = N.cull_on && sq_ballot(T.mode != M_DONE && !T.safe) == 0;
                }
                if (exhausted && m == ~0ull) break;
            }
            stamp(0);
#ifdef SQ_EXTRA_VALU   // timing experiment (results unchanged): N more independent integer VALU instructions per iteration of a wave
            { int x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
#pragma unroll
              for (int e = 0; e < SQ_EXTRA_VALU / 4; ++e) {
                  asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(lane)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(lane));
                  asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(lane)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(lane)); }
              asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3)); }
#endif
#if SQ_SETPRIO         // instruction-arbitration priority of a wave in its return / branch steps (results unchanged)
            __builtin_amdgcn_s_setprio(SQ_SETPRIO & 3);
#endif
            if (PROFILE) pl_unw += (T.mode == M_UNWIND);
            if constexpr (kFlat) {
                // the return step with three exec regions (the step, COMBINE, FAR) instead of six nested ones: DONE, the pop and the
                // kind of the popped frame are decided by selects, the FAR half is trav_unwind_far_flat
                if (T.mode == M_UNWIND) {
                    constexpr uint32_t flag = StackTraits<StackT>::flag;
                    const bool done = T.sp == 0;
                    T.sp -= done ? 0 : 1;
                    const uint32_t e = stk[T.sp * BLOCK];                       // (slot 0, unused, for a ray that is done)
                    const bool is_combine = !done && (e & flag) != 0, is_far = !done && (e & flag) == 0;
                    if (is_combine) trav_unwind_combine<TriSrc, StackT>(T, G, e);
                    if (is_far) trav_unwind_far_flat<NodeSrc, StackT>(T, N, stk, BLOCK, e);
                    T.mode = done ? M_DONE : T.mode;
                }
            } else
            if (T.mode == M_UNWIND) trav_unwind(T, N, G, stk, BLOCK, PROFILE ? &prof : nullptr);
            stamp(1);
            if (PROFILE) pl_desc += (T.mode == M_DESCEND);
            if constexpr (kFlat) {
                if (T.mode == M_DESCEND) { if (flat_ok) trav_descend_flat<NodeSrc, StackT>(T, N, stk, BLOCK); else trav_descend(T, N, stk, BLOCK, &pf); }
            } else
            if (T.mode == M_DESCEND) trav_descend(T, N, stk, BLOCK, &pf);
            // With the culling boxes a ray takes four branch steps per leaf it opens: lanes that are still descending take up
            // to `descend_extra` more steps in this iteration (while at least `descend_lanes` of them are), instead of paying a
            // whole iteration -- return step, leaf scan, windows -- per branch step.
            for (int x = 0; x < A.descend_extra; ++x) {
                if (__popcll(sq_ballot(T.mode == M_DESCEND)) < A.descend_lanes) break;
                if (PROFILE) pl_desc += (T.mode == M_DESCEND);
                if constexpr (kFlat) {
                    if (T.mode == M_DESCEND) { if (flat_ok) trav_descend_flat<NodeSrc, StackT>(T, N, stk, BLOCK); else trav_descend(T, N, stk, BLOCK, &pf); }
                } else
                if (T.mode == M_DESCEND) trav_descend(T, N, stk, BLOCK, &pf);
            }
            stamp(2);
#if SQ_SETPRIO         // ... and in its leaf scan and pair windows (bits 3..2)
            __builtin_amdgcn_s_setprio((SQ_SETPRIO >> 2) & 3);
#endif
            if constexpr (kFlat && RESIDENT) {                              // open the leaf (src/BIH.hs:105): Nothing so far -- as selects
                const bool open = T.mode == M_LEAF;                         // (a resident leaf reference decodes without a load)
                const int2 lf = G.leaf(T.cur);
                lf_first = open ? lf.x : lf_first; lf_cnt = open ? lf.y : lf_cnt; T.R.tri = open ? -1 : T.R.tri;
                T.mode = open ? (lf.y > 0 ? M_LEAFQ : M_UNWIND) : T.mode;
            } else
            if (T.mode == M_LEAF) {                                         // open the leaf (src/BIH.hs:105): Nothing so far
                const int2 lf = G.leaf(T.cur);
                lf_first = lf.x; lf_cnt = lf.y; T.R.tri = -1;
                T.mode = lf.y > 0 ? M_LEAFQ : M_UNWIND;
            }
            if constexpr ((RESIDENT ? kPoolTrisResident : kPoolTrisStreaming) >= 2) {
                // NT triangles per lane and window.  The pool is counted in UNITS of NT consecutive triangles of one leaf (the
                // last unit of a leaf may be part-filled), so that one owner lookup and one set of pulls serve NT triangle
                // tests: the pulls are the most expensive part of a window (ds_bpermute, 6 cycles per CU each), and the
                // lookup's integer VALU work comes next.
                constexpr int NT = RESIDENT ? kPoolTrisResident : kPoolTrisStreaming;
                const int c2 = (T.mode == M_LEAFQ) ? lf_cnt : 0;                // triangles left in this lane's open leaf
                if (sq_ballot(c2 > 0) == 0) continue;
                const int u = (c2 + NT - 1) / NT;                               // units
                const int incl = wave_scan_add(u);
                const int U = __builtin_amdgcn_readlane(incl, 63);
                const int start = incl - u;
                const int tb2 = lf_first - NT * start;                          // first triangle of unit q = NT q + tb2
                const int tri_end = lf_first + c2;                              // one past the leaf's last triangle
                int nwin = U >> 6;
                const int rem = U & 63;
                const bool others = sq_ballot(T.mode == M_UNWIND || T.mode == M_DESCEND) != 0;
                if (rem && (!others || carry || rem >= A.flush_min)) ++nwin;
                stamp(3);
                for (int w = 0; w < nwin; ++w) {
                    const int base = w << 6;
                    const int h = start - base;
                    if (u > 0 && h < 64 && incl > base) tab[h > 0 ? h : 0] = (uint8_t)lane_tag;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    int own = tab[lane];
                    tab[lane] = 0;
                    own = wave_scan_max(own);
                    const int q = base + lane;
                    const bool act = q < U;
                    int tri0 = NT * q + lane_pull(tb2, own);
                    const int end_o = lane_pull(tri_end, own);
                    const f3 po = sq::mk(lane_pull(T.o.x, own), lane_pull(T.o.y, own), lane_pull(T.o.z, own));
                    const f3 pd = sq::mk(lane_pull(T.d.x, own), lane_pull(T.d.y, own), lane_pull(T.d.z, own));
                    if (!act) tri0 = 0;                                         // lanes past the last unit test triangle 0 and drop the answer
                    stamp(4);
                    f3 v0[NT], e1[NT], e2[NT];
                    G.template get_run<NT>(tri0, v0, e1, e2);                   // a part-filled unit tests the records after it and drops the answers
                    float tt[NT]; bool hit[NT]; unsigned long long hmk[NT], hm = 0;
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const bool valid = act && (k == 0 || tri0 + k < end_o);
                        hit[k] = moller_trumbore_flat(po, pd, v0[k], e1[k], e2[k], tt[k]) & valid;
                    }
                    stamp(5);
#pragma unroll
                    for (int k = 0; k < NT; ++k) { hmk[k] = sq_ballot(hit[k]); hm |= hmk[k]; if (PROFILE) pl_hit += (int)hit[k]; }
                    while (hm) {                                                // accepted hits in leaf order: lane by lane, a lane's triangles in turn
                        const int l = __ffsll((long long)hm) - 1;
                        hm &= hm - 1;
                        const int so = __builtin_amdgcn_readlane(own, l);
                        const int stri = __builtin_amdgcn_readlane(tri0, l);
#pragma unroll
                        for (int k = 0; k < NT; ++k) if ((hmk[k] >> l) & 1ull) {
                            const float st = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tt[k]), l));
                            if (lane_tag == so && (T.R.tri < 0 || dist_gt(T.o, T.d, T.R.t, st, T.safe))) { T.R.t = st; T.R.tri = stri + k; }
                        }
                    }
                    stamp(6);
                }
                const int done = min(U, nwin << 6);                             // units tested
                if (PROFILE) { pf_leaf += nwin; pf_outer += NT * done; }
                if constexpr (kFlat) {
                    const bool fin = u > 0 && incl <= done;                     // the Leaf equation is finished: R is its value
                    const int adv = (u > 0 && incl > done && start < done) ? NT * (done - start) : 0;   // whole units come first
                    T.mode = fin ? M_UNWIND : T.mode; lf_first += adv; lf_cnt -= adv;
                } else
                if (u > 0) {
                    if (incl <= done) T.mode = M_UNWIND;                        // the Leaf equation is finished: R is its value
                    else if (start < done) { lf_first += NT * (done - start); lf_cnt -= NT * (done - start); }   // whole units come first
                }
                carry = done < U;
                stamp(7);
                continue;
            }
            const int c = (T.mode == M_LEAFQ) ? lf_cnt : 0;
            if (sq_ballot(c > 0) == 0) continue;
            const int incl = wave_scan_add(c);                              // pairs of lanes 0..lane
            const int P = __builtin_amdgcn_readlane(incl, 63);              // pairs queued in the wave
            const int start = incl - c;                                     // this lane's first pair
            const int tb = lf_first - start;                                // triangle of pair p = p + tb
            int nwin = P >> 6;
            const int rem = P & 63;
            const bool others = sq_ballot(T.mode == M_UNWIND || T.mode == M_DESCEND) != 0;
            if (rem && (!others || carry || rem >= A.flush_min)) ++nwin;
            stamp(3);
            // KW windows per step: their LDS round trips (head table, owner pulls, index record, vertices) are issued
            // together and waited for once, and the independent arithmetic of KW triangle tests interleaves.
            for (int w0 = 0; w0 < nwin; w0 += KW) {
                int own[KW], tri[KW]; f3 po[KW], pd[KW]; float t[KW]; bool hit[KW];
                // who owns pair base + lane?  Every owner whose pairs reach into a window writes its lane number at the
                // window position of its first pair there; a max-scan spreads it over the owner's run.
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const int base = (w0 + k) << 6;
                    const int h = start - base;
                    if (c > 0 && h < 64 && incl > base) tab[k * 64 + (h > 0 ? h : 0)] = (uint8_t)lane_tag;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = 0; k < KW; ++k) { own[k] = tab[k * 64 + lane]; tab[k * 64 + lane] = 0; }
#pragma unroll
                for (int k = 0; k < KW; ++k) own[k] = wave_scan_max(own[k]);
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const int src = own[k];                                 // ds_bpermute takes lane * 4 and ignores the two low bits
                    const int p = ((w0 + k) << 6) + lane;
                    tri[k] = p + lane_pull(tb, src);
                    po[k] = sq::mk(lane_pull(T.o.x, src), lane_pull(T.o.y, src), lane_pull(T.o.z, src));
                    pd[k] = sq::mk(lane_pull(T.d.x, src), lane_pull(T.d.y, src), lane_pull(T.d.z, src));
#ifdef SQ_EXTRA_PULLS   // timing experiment only (results unchanged): what do N more ds_bpermute per window cost, with no VALU attached?
                    { int sink;
#pragma unroll
                      for (int e = 0; e < SQ_EXTRA_PULLS; ++e) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(sink) : "v"(src), "v"(tri[k]));
                      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
                    hit[k] = p < P;                                         // so far: the pair exists
                    if (!hit[k]) tri[k] = 0;                                // lanes past the last pair test triangle 0 and drop the answer
                }
                if (PROFILE) { t[0] = po[0].x + pd[0].x + __int_as_float(tri[0]); asm volatile("" :: "v"(t[0])); }   // the pulls have arrived
                stamp(4);
                f3 v0[KW], e1[KW], e2[KW];
#pragma unroll
                for (int k = 0; k < KW; ++k) if (k == 0 || w0 + k < nwin) G.get1(tri[k], v0[k], e1[k], e2[k]);
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    if (k == 0 || w0 + k < nwin) { float tk; const bool ok = moller_trumbore_flat(po[k], pd[k], v0[k], e1[k], e2[k], tk); t[k] = tk; hit[k] = hit[k] & ok; }
                    else { t[k] = 0.0f; hit[k] = false; }
                }
                stamp(5);
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    unsigned long long hm = sq_ballot(hit[k]);
                    if (PROFILE) pl_hit += hit[k];
                    while (hm) {                                            // accepted hits in pair order
                        const int l = __ffsll((long long)hm) - 1;
                        hm &= hm - 1;
                        const int so = __builtin_amdgcn_readlane(own[k], l);
                        const float st = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t[k]), l));
                        const int stri = __builtin_amdgcn_readlane(tri[k], l);
                        if (PROFILE && lane_tag == so && T.R.tri >= 0 && !(!(T.R.t > st) && st < __builtin_inff() && T.safe)) ++pl_hslow;
                        if (lane_tag == so && (T.R.tri < 0 || dist_gt(T.o, T.d, T.R.t, st, T.safe))) { T.R.t = st; T.R.tri = stri; }
                    }
                }
                stamp(6);
            }
            const int done = min(P, nwin << 6);
            if (PROFILE) { pf_leaf += nwin; pf_outer += done; }
            if (c > 0) {
                if (incl <= done) T.mode = M_UNWIND;                        // the Leaf equation is finished: R is its value
                else if (start < done) { lf_first += done - start; lf_cnt = incl - done; }
            }
            carry = done < P;
            stamp(7);
        }
        if (lane == 0) atomicAdd(&A.stats[0], n_traced);
        if (A.diag && threadIdx.x == 0) atomicAdd(&A.stats[kDiagGauge], ~0ull);   // -1: the workgroup's first wave is leaving (the others follow within a ray's length)
        if (PROFILE) {
            if (lane == 0) { atomicAdd(&A.stats[1], pf_adv); atomicAdd(&A.stats[4], pf_leaf); atomicAdd(&A.stats[5], pf_outer); atomicAdd(&A.stats[7], pf_refill); }
            atomicAdd(&A.stats[2], (unsigned long long)pl_unw); atomicAdd(&A.stats[3], (unsigned long long)pl_desc);
            atomicAdd(&A.stats[6], (unsigned long long)pl_hit); atomicAdd(&A.stats[8], (unsigned long long)pl_ref);
            atomicAdd(&A.stats[9], (unsigned long long)prof.combine); atomicAdd(&A.stats[10], (unsigned long long)prof.recompute_lanes);
            atomicAdd(&A.stats[11], (unsigned long long)prof.recompute_waves); atomicAdd(&A.stats[12], (unsigned long long)prof.slowcmp_lanes);
            atomicAdd(&A.stats[13], (unsigned long long)prof.slowcmp_waves); atomicAdd(&A.stats[14], (unsigned long long)pl_hslow);
            if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&A.stats[16 + i], tsec[i]);
        }
        return;
    }
    for (;;) {
        if (PROFILE) ++pf_outer;
        const bool idle = (T.mode == M_DONE);
        if (idle && my_ray >= 0) { *reinterpret_cast<int2*>(A.org + my_ray) = make_int2(__float_as_int(T.R.t), T.R.tri); my_ray = -1; }
        const unsigned long long m = sq_ballot(idle);
        if (m) {
            refill(m, idle);
            if (exhausted && m == ~0ull) break;
        }
        // advance until (almost) every lane has a leaf to test or is finished
        for (;;) {
            const bool adv = (T.mode == M_DESCEND) || (T.mode == M_UNWIND);
            const unsigned long long am = sq_ballot(adv);
            if (am == 0) break;
            if (__popcll(am) <= A.straggler_lanes && sq_ballot(T.mode == M_LEAF) != 0) break;
            if (PROFILE) { ++pf_adv; pl_unw += (T.mode == M_UNWIND); }
            if (T.mode == M_UNWIND) trav_unwind(T, N, G, stk, BLOCK);
            if (PROFILE) pl_desc += (T.mode == M_DESCEND);
            if (T.mode == M_DESCEND) trav_descend(T, N, stk, BLOCK);
        }
        if (PROFILE) {
            const int cnt = (T.mode == M_LEAF) ? G.leaf(T.cur).y : 0;
            pl_tri += cnt; pf_leaf += wave_max(cnt);
        }
        if (T.mode == M_LEAF) trav_leaf(T, G);
    }
    if (lane == 0) atomicAdd(&A.stats[0], n_traced);
    if (A.diag && threadIdx.x == 0) atomicAdd(&A.stats[kDiagGauge], ~0ull);
    if (PROFILE) {
        if (lane == 0) { atomicAdd(&A.stats[1], pf_adv); atomicAdd(&A.stats[4], pf_leaf); atomicAdd(&A.stats[6], pf_outer); atomicAdd(&A.stats[7], pf_refill); }
        atomicAdd(&A.stats[2], (unsigned long long)pl_unw); atomicAdd(&A.stats[3], (unsigned long long)pl_desc);
        atomicAdd(&A.stats[5], (unsigned long long)pl_tri); atomicAdd(&A.stats[8], (unsigned long long)pl_ref);
    }
}

template <typename StackT, bool RESIDENT, int BLOCK, bool PROFILE, bool POOL>
__global__ void __launch_bounds__(BLOCK) sq_trace_rays(const SceneView S, const TraceArgs A) {
    trace_rays_body<StackT, RESIDENT, BLOCK, PROFILE, POOL>(S, A);
}
// The streaming pooled kernel once more, compiled for SIX waves per SIMD (at most 80 VGPRs; the plain build takes 86, i.e. four):
// three 512-thread workgroups per CU instead of two when their stacks and a share of the tree's top fit in a third of the LDS.
// The streaming form waits on memory two thirds of its time, and the extra waves are worth more than the handful of spilled
// registers and the smaller LDS node prefix: 25.9 -> 24.5 ms (82k triangles), 30.7 -> 29.4 ms (1M triangles) at 64 spp, same
// call (profiles/r03c_stream_incremental.txt, rows libc16 / libw6c16 with incremental = 0).
template <typename StackT>
__global__ void __launch_bounds__(kTraceBlock, 6) sq_trace_rays_dense(const SceneView S, const TraceArgs A) {
    trace_rays_body<StackT, false, kTraceBlock, false, true>(S, A);
}

// ---- diagnostics: primitives of the numeric spec evaluated on the device ----
__global__ void sq_debug_kernel(int op, const void* a, const void* b, long long n, void* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* fa = (const float*)a; const float* fb = (const float*)b; float* fo = (float*)out;
    switch (op) {
        case SQ_OP_SQRT: fo[i] = sq::fsqrt(fa[i]); break;
        case SQ_OP_DIV: fo[i] = fa[i] / fb[i]; break;
        case SQ_OP_SIN: fo[i] = sq::fsin(fa[i]); break;
        case SQ_OP_COS: fo[i] = sq::fcos(fa[i]); break;
        case SQ_OP_ACOS: fo[i] = sq::facos(fa[i]); break;
        case SQ_OP_ATAN: fo[i] = sq::fatan(fa[i]); break;
        case SQ_OP_UNIT_FLOAT: fo[i] = sq::unit_float(((const uint32_t*)a)[i]); break;
        case SQ_OP_TFGEN3: {
            uint32_t n0, n1, n2; sq::tfgen3(((const long long*)a)[i], n0, n1, n2);
            uint32_t* o = (uint32_t*)out + 3 * i; o[0] = n0; o[1] = n1; o[2] = n2; break;
        }
        case SQ_OP_TONEMAP: tonemap(sq::mk(fa[3 * i], fa[3 * i + 1], fa[3 * i + 2]), (uint8_t*)out + 3 * i); break;
        case SQ_OP_RCP_SWEEP: {                                            // all 65536 floats whose upper 16 bits are a[i]
            const uint32_t hi = ((const uint32_t*)a)[i] << 16; uint32_t bad = 0;
            for (uint32_t lo = 0; lo < 65536u; ++lo) {
                const float x = __uint_as_float(hi | lo);
                bad += __float_as_uint(rcp_midrange(x)) != __float_as_uint(1.0f / x);
            }
            ((uint32_t*)out)[i] = bad; break;
        }
        case SQ_OP_CULL_SLAB: {                                            // the culling slab test as the trace kernels run it
            const uint32_t* w = (const uint32_t*)a + 9 * i; const float* r = fa + 9 * i + 3;
            const f3 o = sq::mk(r[0], r[1], r[2]), d = sq::mk(r[3], r[4], r[5]);
            const f3 df = sq::mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), nodf = sq::mk(-o.x * df.x, -o.y * df.y, -o.z * df.z);
            ((uint32_t*)out)[i] = cull_slab_half(w[0], w[1], w[2], df, nodf) ? 1u : 0u; break;
        }
        default: break;
    }
}

// ----------------------------------------------------------------------------------------------
// Host: scene validation + upload
// ----------------------------------------------------------------------------------------------
struct sq_device_scene {
    int device = 0;
    SceneView view{};
    void* d_arena = nullptr;      // every d_* array below lives in this one allocation
    void *d_branches = nullptr, *d_leaves = nullptr, *d_tris = nullptr, *d_mats = nullptr, *d_verts = nullptr, *d_trix = nullptr, *d_rbranch = nullptr, *d_emitters = nullptr, *d_tri_mat = nullptr, *d_surfs = nullptr, *d_cull_child = nullptr, *d_cull16 = nullptr, *d_rtail = nullptr, *d_branches_m5 = nullptr;
    int height = 0; bool small_index = false; int n_cu = 256; int64_t n_grown = 0;
    // workspace (grow-only)
    Work work{}; void* d_work = nullptr; size_t work_bytes = 0; int64_t work_pixels = 0, work_slots = 0;
    // timing of the dominant kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0; int64_t launches = 0;
    // options
    int64_t opt_timing = 0, opt_variant = 2, opt_slots = 512ll << 20, opt_straggler = 8, opt_trace_blocks_per_cu = 0, opt_resident = 1, opt_profile = 0, opt_lds_node_kb = 32;
    const char* last_kernel = "sq_trace_rays";
    // second stream of the overlapped schedule (launch_frame) and its event pool
    hipStream_t aux = nullptr; std::vector<hipEvent_t> events;
    int64_t opt_overlap = 0, opt_aux_blocks_per_cu = 0;
    int64_t opt_pool = 1, opt_refill_min = 12, opt_flush_min = 40, opt_guided = 1, opt_primary_resident = 1, opt_pixel_major = -1, opt_cull = 1, opt_descend_extra = 2, opt_descend_lanes = 16, opt_primary_pooled = 0, opt_coresidency = 0, opt_trace_prio = 0, opt_aux_low_priority = 1, opt_aux_polite = 0, opt_incremental = 1, opt_primary_tiles = 1;
};

namespace {

// Walks the pre-order array once; checks that it is a well-formed tree, that every index is in
// range, and computes the height.  A malformed tree would otherwise fault on the GPU.
int validate_tree(const sq_scene& sc, int& height, std::vector<int32_t>& depth_of) {
    const int32_t n = sc.n_nodes;
    if (n < 1) return sq_set_error("scene has no nodes");
    depth_of.assign((size_t)n, 0);
    struct Fr { int32_t node, stage; };
    std::vector<Fr> st;
    st.push_back({ 0, 0 });
    int32_t next = 0;      // next unvisited pre-order index
    depth_of[0] = 1;
    height = 0;
    while (!st.empty()) {
        const int32_t node = st.back().node; const int stage = st.back().stage;
        const sq_node& nd = sc.nodes[node];
        const int kind = nd.kind & 3;
        if (stage == 0) {
            if (node != next) return sq_set_error("node %d is not in pre-order position (expected %d)", node, next);
            ++next;
            const int dep = depth_of[(size_t)node];
            if (dep > height) height = dep;
            if (kind == 3) {
                const int64_t cnt = nd.kind >> 2, first = nd.link;
                if (cnt < 0 || first < 0 || first + cnt > sc.n_tris) return sq_set_error("leaf %d has triangle range [%lld,+%lld) outside 0..%d", node, (long long)first, (long long)cnt, sc.n_tris);
                st.pop_back();
                continue;
            }
            if ((nd.kind >> 2) != 0) return sq_set_error("branch %d has stray bits in kind", node);
            if (node + 1 >= n) return sq_set_error("branch %d has no left child", node);
            st.back().stage = 1;
            depth_of[(size_t)node + 1] = dep + 1;
            st.push_back({ node + 1, 0 });
        } else if (stage == 1) {
            if (nd.link != next) return sq_set_error("branch %d: right child link %d, expected %d", node, nd.link, next);
            if (nd.link >= n) return sq_set_error("branch %d: right child %d out of range", node, nd.link);
            st.back().stage = 2;
            depth_of[(size_t)nd.link] = depth_of[(size_t)node] + 1;
            st.push_back({ nd.link, 0 });
        } else st.pop_back();
    }
    if (next != n) return sq_set_error("tree covers %d of %d nodes", next, n);
    return 0;
}

}  // namespace

extern "C" int sq_scene_upload(const sq_scene* sc, int32_t device, sq_device_scene** out) {
    if (!sc || !out) return sq_set_error("null argument");
    if (!sc->nodes || sc->n_nodes < 1) return sq_set_error("scene has no nodes");
    if (sc->n_tris < 0 || sc->n_mats < 0 || (sc->n_tris && !sc->tris) || (sc->n_mats && !sc->mats)) return sq_set_error("bad triangle/material arrays");
    for (int32_t i = 0; i < sc->n_tris; ++i)
        if (sc->tris[i].mat < 0 || sc->tris[i].mat >= sc->n_mats) return sq_set_error("triangle %d: material %d outside 0..%d", i, sc->tris[i].mat, sc->n_mats - 1);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return sq_set_error("no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return sq_set_error("device %d outside 0..%d", device, ndev - 1);
    int height = 0; std::vector<int32_t> depth;
    if (validate_tree(*sc, height, depth)) return 1;
    if (sc->height && sc->height != height) return sq_set_error("scene.height = %d but the tree has height %d", sc->height, height);

    // Re-pack.  Leaves keep pre-order numbering; branches are numbered breadth-first (stable within a
    // level) so the top of the tree is a prefix of the branch table.  Each branch carries its traversal box.
    const int32_t n = sc->n_nodes;
    std::vector<uint32_t> ref((size_t)n);
    int32_t nb = 0, nl = 0;
    {
        std::vector<std::vector<int32_t>> by_depth((size_t)height + 1);
        for (int32_t i = 0; i < n; ++i) {
            if ((sc->nodes[i].kind & 3) == 3) ref[(size_t)i] = (uint32_t)nl++ | kLeafBit;
            else by_depth[(size_t)depth[(size_t)i]].push_back(i);
        }
        for (auto& level : by_depth) for (int32_t i : level) ref[(size_t)i] = (uint32_t)nb++;
    }
    std::vector<DevBranch> br((size_t)nb);
    std::vector<int> br_axis((size_t)nb);
    std::vector<uint32_t> br_grown((size_t)nb, 0u);     // kGrownLeft | kGrownRight (sq_scene.h), for the incremental slab test
    bool incremental_ok = true;                         // no child interval is inverted anywhere in the tree
    std::vector<DevLeaf> lf((size_t)nl);
    std::vector<sq_bounds> box((size_t)n);
    box[0] = sc->root;
    for (int32_t i = 0; i < n; ++i) {                   // pre-order: parents come before children
        const sq_node& nd = sc->nodes[i];
        const int kind = nd.kind & 3;
        if (kind == 3) { lf[ref[(size_t)i] & ~kLeafBit] = { nd.link, nd.kind >> 2 }; continue; }
        const sq_bounds& b = box[(size_t)i];
        DevBranch& d = br[ref[(size_t)i]];
        for (int c = 0; c < 3; ++c) { d.lo[c] = b.lo[c]; d.hi[c] = b.hi[c]; }
        d.lmax = d.lmax2 = nd.lmax; d.rmin = d.rmin2 = nd.rmin; br_axis[ref[(size_t)i]] = kind;
        d.left = ref[(size_t)i + 1]; d.right = ref[(size_t)nd.link];
        {   // how the children's planes sit in this branch's interval on the split axis (NaN fails every comparison)
            const float lo = b.lo[kind], hi = b.hi[kind];
            if (!(lo <= hi) || !(lo <= nd.lmax) || !(nd.rmin <= hi)) incremental_ok = false;
            br_grown[ref[(size_t)i]] = (nd.lmax > hi ? kGrownLeft : 0u) | (nd.rmin < lo ? kGrownRight : 0u);
        }
        sq_bounds l = b, r = b;                          // src/BIH.hs:130-141
        l.hi[kind] = nd.lmax; r.lo[kind] = nd.rmin;
        box[(size_t)i + 1] = l; box[(size_t)nd.link] = r;
    }
    std::vector<DevTri> tr((size_t)sc->n_tris); std::vector<int32_t> tri_mat((size_t)sc->n_tris);
    for (int32_t i = 0; i < sc->n_tris; ++i) {
        const sq_tri& t = sc->tris[i]; DevTri& d = tr[(size_t)i];
        for (int c = 0; c < 3; ++c) { d.v0[c] = t.v0[c]; d.e1[c] = t.v1[c] - t.v0[c]; d.e2[c] = t.v2[c] - t.v0[c]; }
        tri_mat[(size_t)i] = t.mat;
    }
    tr.resize((size_t)sc->n_tris + GlobalTris::kRunPad);                   // zero triangles: get_run may read past the last one
    std::vector<DevSurf> sf((size_t)sc->n_tris);
    std::vector<DevMat> mt((size_t)sc->n_mats);
    bool nonneg = true;
    for (int32_t i = 0; i < sc->n_mats; ++i) {
        const sq_material& m = sc->mats[i];
        mt[(size_t)i] = { m.reflective, m.surf[0], m.surf[1], m.surf[2], m.emissive, m.emit[0], m.emit[1], m.emit[2] };
        const float comp[8] = { m.reflective, m.surf[0], m.surf[1], m.surf[2], m.emissive, m.emit[0], m.emit[1], m.emit[2] };
        for (float c : comp) { uint32_t bits; std::memcpy(&bits, &c, 4); if ((bits >> 31) || !(c == c) || c > 3.0e38f) nonneg = false; }
    }
    // The s == 0 shortcuts (absorbs()) replace `0 * L` by +0, which is only the reference's value while the nested
    // radiance L is finite (0 * inf = NaN, src/Lib.hs:135).  Components <= 3e38 do not bound the PRODUCTS: the emission
    // `emissive *^ emitColor` (src/Lib.hs:136) and `surf * L + e` one level down can overflow.  L <= max_s * max_e + max_e
    // with max_e the largest emission (as the fp32 product the kernels use) and max_s the largest surface component;
    // the shortcuts stay on only if that bound, evaluated in double, is comfortably finite in fp32.
    if (nonneg) {
        double max_e = 0.0, max_s = 0.0;
        for (int32_t i = 0; i < sc->n_mats; ++i) {
            const sq_material& m = sc->mats[i];
            for (int k = 0; k < 3; ++k) {
                const float e = m.emissive * m.emit[k];
                if (!(e - e == 0.0f)) nonneg = false;                    // the product itself is inf
                max_e = std::max(max_e, (double)e); max_s = std::max(max_s, (double)m.surf[k]);
            }
        }
        if (!(max_s * max_e + max_e <= 3.0e38)) nonneg = false;
    }
    for (int32_t i = 0; i < sc->n_tris; ++i) {                       // per-triangle shading record (surface_of)
        const DevTri& d = tr[(size_t)i]; const sq_material& m = sc->mats[sc->tris[i].mat]; DevSurf& o = sf[(size_t)i];
        const f3 nrm = sq::cross(sq::mk(d.e1[0], d.e1[1], d.e1[2]), sq::mk(d.e2[0], d.e2[1], d.e2[2]));
        const f3 em = sq::scale(m.emissive, sq::mk(m.emit[0], m.emit[1], m.emit[2]));
        o.n[0] = nrm.x; o.n[1] = nrm.y; o.n[2] = nrm.z; o.reflective = m.reflective;
        o.surf[0] = m.surf[0]; o.surf[1] = m.surf[1]; o.surf[2] = m.surf[2]; o.pad0 = 0;
        o.emit[0] = em.x; o.emit[1] = em.y; o.emit[2] = em.z; o.pad1 = 0;
    }
    // Indexed form for LDS residency: unique vertices (bitwise) + 16-bit indices, when they fit.
    std::vector<float> uverts; std::vector<uint16_t> trix;
    {
        struct Key { uint32_t a, b, c; bool operator==(const Key& o) const { return a == o.a && b == o.b && c == o.c; } };
        struct KeyHash { size_t operator()(const Key& k) const { return ((size_t)k.a * 0x9E3779B1u) ^ ((size_t)k.b * 0x85EBCA77u) ^ ((size_t)k.c * 0xC2B2AE3Du); } };
        std::unordered_map<Key, uint32_t, KeyHash> ids;
        bool fits = sc->n_mats <= 65535;
        std::vector<uint32_t> idx((size_t)sc->n_tris * 3);
        for (int32_t i = 0; i < sc->n_tris && fits; ++i) {
            const float* vs[3] = { sc->tris[i].v0, sc->tris[i].v1, sc->tris[i].v2 };
            for (int k = 0; k < 3; ++k) {
                Key key; std::memcpy(&key.a, &vs[k][0], 4); std::memcpy(&key.b, &vs[k][1], 4); std::memcpy(&key.c, &vs[k][2], 4);
                auto it = ids.find(key);
                if (it == ids.end()) {
                    if (ids.size() >= 65535) { fits = false; break; }
                    it = ids.emplace(key, (uint32_t)ids.size()).first;
                    uverts.insert(uverts.end(), vs[k], vs[k] + 3); uverts.push_back(0.0f);
                }
                idx[(size_t)i * 3 + k] = it->second;
            }
        }
        if (fits) {
            trix.resize((size_t)sc->n_tris * 4);
            for (int32_t i = 0; i < sc->n_tris; ++i) {
                for (int k = 0; k < 3; ++k) trix[(size_t)i * 4 + k] = (uint16_t)idx[(size_t)i * 3 + k];
                trix[(size_t)i * 4 + 3] = (uint16_t)sc->tris[i].mat;
            }
        } else { uverts.clear(); }
    }
    // Culling boxes (include/squigly_host.h: sq_cull_boxes), per pre-order node.
    std::vector<float> cbox((size_t)n * 6); float cull_limits[3] = { -1.0f, 0.25f, 1.5624f };
    if (sq_cull_boxes(sc, cbox.data(), cull_limits)) return 1;
    // Resident encoding of the branches (see sq_scene.h): needs encodable leaves and a 24-bit index space.
    std::vector<uint32_t> rbranch; uint32_t rroot = 0;
    {
        bool ok = !trix.empty() && nb < (1 << 24) && sc->n_tris < (1 << 24);
        for (int32_t i = 0; i < nl && ok; ++i) ok = lf[(size_t)i].count <= 31;
        auto enc = [&](uint32_t r) -> uint32_t {
            if (!(r & kLeafBit)) return r;
            const DevLeaf& L = lf[r & ~kLeafBit];
            return kLeafBit | ((uint32_t)L.count << 24) | (uint32_t)L.first;
        };
        if (ok) {
            rbranch.resize((size_t)nb * 10);
            for (int32_t i = 0; i < nb; ++i) {
                const DevBranch& d = br[(size_t)i];
                uint32_t* r = &rbranch[(size_t)i * 10];
                std::memcpy(r, d.lo, 12); std::memcpy(r + 3, &d.lmax, 4); std::memcpy(r + 4, d.hi, 12); std::memcpy(r + 7, &d.rmin, 4);
                r[8] = enc(d.left) | ((uint32_t)br_axis[(size_t)i] << 29); r[9] = enc(d.right) | (br_grown[(size_t)i] << 29);   // kAxisMask bits: axis | grown children
            }
            rroot = enc(ref[0]);
        } else { trix.clear(); }
    }
    // Streaming forms: the culling boxes of a branch's two children, with the branch (GlobalNodes / HybridNodes)
    std::vector<float> cull_child;
    if (cull_limits[0] >= 0.0f && nb > 0) {
        cull_child.resize((size_t)nb * 16);
        for (int32_t i = 0; i < n; ++i) {
            if ((sc->nodes[i].kind & 3) == 3) continue;
            float* o = &cull_child[(size_t)ref[(size_t)i] * 16];
            const float* l = &cbox[(size_t)(i + 1) * 6]; const float* r = &cbox[(size_t)sc->nodes[i].link * 6];
            o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = 0; o[4] = l[3]; o[5] = l[4]; o[6] = l[5]; o[7] = 0;
            o[8] = r[0]; o[9] = r[1]; o[10] = r[2]; o[11] = 0; o[12] = r[3]; o[13] = r[4]; o[14] = r[5]; o[15] = 0;
        }
    }
    std::vector<uint32_t> rtail;                                      // resident form: a return's data per branch, one quad
    for (size_t b = 0; b * 10 < rbranch.size(); ++b) { const uint32_t* r = &rbranch[b * 10]; rtail.insert(rtail.end(), { r[3], r[7], r[8], r[9] }); }
    std::vector<uint32_t> cull16;                                     // the same boxes as binary16 pairs, for the resident form
    if (!cull_child.empty()) {
        cull16.resize((size_t)nb * 8, 0u);
        for (int32_t b = 0; b < nb; ++b) for (int side = 0; side < 2; ++side) {
            const float* bx = &cull_child[(size_t)b * 16 + (size_t)side * 8];
            for (int c = 0; c < 3; ++c) cull16[(size_t)b * 8 + (size_t)side * 4 + (size_t)c] = sq_half_outward(bx[c], 0) | (sq_half_outward(bx[4 + c], 1) << 16);
        }
    }
    // Streaming form: leaf references carry (first, count) themselves when they fit, which saves the dependent
    // leaf-table load of every leaf visit.
    bool packed_leaves = sc->n_tris < (1 << 24);
    for (int32_t i = 0; i < nl && packed_leaves; ++i) packed_leaves = lf[(size_t)i].count <= 31;
    if (nb >= (1 << 29) || nl >= (1 << 29)) { sq_set_error("scene has %d branches and %d leaves; the device layout holds 2^29 of each", nb, nl); return 1; }
    uint32_t root_ref = ref[0];
    {
        auto enc = [&](uint32_t r) -> uint32_t {
            if (!packed_leaves || !(r & kLeafBit)) return r;
            const DevLeaf& L = lf[r & ~kLeafBit];
            return kLeafBit | ((uint32_t)L.count << 24) | (uint32_t)L.first;
        };
        for (int32_t i = 0; i < nb; ++i) {
            DevBranch& d = br[(size_t)i];
            d.left = enc(d.left) | ((uint32_t)br_axis[(size_t)i] << 29);   // kAxisMask bits
            d.right = enc(d.right);
        }
        root_ref = enc(root_ref);
    }
    // Streaming form: branch record + its children's binary16 culling boxes as ONE packed 80-byte record (SceneView::branches_m,
    // HybridNodes::kMerged).  Measured at 64 spp, same process, as a launch option: 1M-triangle scene
    // trace launches 23.04 -> 22.62 ms, 82k-triangle scene +-0; the same records padded to one 128-byte line each were 16-21 %
    // SLOWER (26.7 / 23.1 ms): the form lives on what stays in L2, i.e. on the table's footprint, not on lines per visit
    // (profiles/r03t_merged_branches_ab.txt)
    std::vector<uint32_t> br_m5;
    br_m5.assign((size_t)nb * 20, 0u);                    // (a scene without culling boxes keeps zeros there: they are never read)
    for (int32_t b = 0; b < nb; ++b) {
        std::memcpy(&br_m5[(size_t)b * 20], &br[(size_t)b], 48);
        if (!cull16.empty()) std::memcpy(&br_m5[(size_t)b * 20 + 12], &cull16[(size_t)b * 8], 32);
    }
    // Emissive triangles (for the last-bounce shortcut of sq_shade1).  Disabled (-1) when a material value is not
    // finite (then s*0 + e is not exactly +0 for non-emitters) or when the list is long enough to cost more than it saves.
    std::vector<int32_t> emitters; int32_t n_emitters = -1;
    {
        bool finite = true;
        std::vector<char> emits((size_t)sc->n_mats, 0);
        for (int32_t i = 0; i < sc->n_mats; ++i) {
            const sq_material& m = sc->mats[i];
            const float comp[8] = { m.reflective, m.surf[0], m.surf[1], m.surf[2], m.emissive, m.emit[0], m.emit[1], m.emit[2] };
            for (float c : comp) if (!(c - c == 0.0f)) finite = false;
            for (int k = 0; k < 3; ++k) { const float e = m.emissive * m.emit[k]; uint32_t bits; std::memcpy(&bits, &e, 4); if (bits != 0u) emits[(size_t)i] = 1; }
        }
        for (int32_t i = 0; i < sc->n_tris; ++i) if (emits[(size_t)sc->tris[i].mat]) emitters.push_back(i);
        if (finite && emitters.size() <= 64) n_emitters = (int32_t)emitters.size();
    }
    SQ_HIP(hipSetDevice(device));
    sq_device_scene* s = new sq_device_scene;
    s->device = device; s->height = height;
    s->small_index = nb < 0x8000 && sc->n_tris < 0x8000;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cu = prop.multiProcessorCount;
    // One arena for every array of the scene: one hipMalloc, one host-packed hipMemcpy and (sq_scene_free) one hipFree
    // instead of thirteen of each -- the one-shot calls upload and free a scene per frame, and at the CLI's default frame
    // (540 x 540 at 10 samples) those calls were a third of the call's time.  Every array starts on a 256-byte boundary.
    struct Piece { void** dst; const void* src; size_t bytes, off; };
    std::vector<Piece> pieces;
    size_t arena_bytes = 0;
    auto up = [&](void** dst, const void* src, size_t bytes) {
        pieces.push_back(Piece{ dst, src, bytes, arena_bytes });
        arena_bytes += ((bytes ? bytes : 16) + 255) & ~(size_t)255;
    };
    up(&s->d_branches, br.data(), br.size() * sizeof(DevBranch)); up(&s->d_leaves, lf.data(), lf.size() * sizeof(DevLeaf));
    up(&s->d_tris, tr.data(), tr.size() * sizeof(DevTri)); up(&s->d_tri_mat, tri_mat.data(), tri_mat.size() * sizeof(int32_t));
    up(&s->d_surfs, sf.data(), sf.size() * sizeof(DevSurf)); up(&s->d_mats, mt.data(), mt.size() * sizeof(DevMat));
    up(&s->d_verts, uverts.data(), uverts.size() * sizeof(float)); up(&s->d_trix, trix.data(), trix.size() * sizeof(uint16_t));
    up(&s->d_rbranch, rbranch.data(), rbranch.size() * sizeof(uint32_t));
    up(&s->d_emitters, emitters.data(), emitters.size() * sizeof(int32_t));
    if (!cull_child.empty()) up(&s->d_cull_child, cull_child.data(), cull_child.size() * sizeof(float));
    if (!cull16.empty()) up(&s->d_cull16, cull16.data(), cull16.size() * sizeof(uint32_t));
    up(&s->d_rtail, rtail.data(), rtail.size() * sizeof(uint32_t));
    if (!br_m5.empty()) up(&s->d_branches_m5, br_m5.data(), br_m5.size() * sizeof(uint32_t));
    {
        std::vector<unsigned char> staging(arena_bytes, 0);
        for (const Piece& pc : pieces) if (pc.bytes) std::memcpy(staging.data() + pc.off, pc.src, pc.bytes);
        if (hipMalloc(&s->d_arena, arena_bytes) != hipSuccess) { sq_scene_free(s); return sq_set_error("hipMalloc(%zu) for the scene failed", arena_bytes); }
        if (hipMemcpy(s->d_arena, staging.data(), arena_bytes, hipMemcpyHostToDevice) != hipSuccess) { sq_scene_free(s); return sq_set_error("hipMemcpy H2D of the scene failed"); }
        for (const Piece& pc : pieces) *pc.dst = (char*)s->d_arena + pc.off;
    }
    SceneView& v = s->view;
    v.branches = (const float4*)s->d_branches; v.leaves = (const int2*)s->d_leaves;
    v.tris = (const float*)s->d_tris; v.tri_mat = (const int32_t*)s->d_tri_mat; v.mats = (const float4*)s->d_mats; v.surfs = (const float4*)s->d_surfs;
    for (int c = 0; c < 3; ++c) { v.root_lo[c] = sc->root.lo[c]; v.root_hi[c] = sc->root.hi[c]; }
    v.root_ref = root_ref; v.packed_leaves = packed_leaves ? 1 : 0;
    v.n_branches = nb; v.n_leaves = nl; v.n_tris = sc->n_tris; v.n_mats = sc->n_mats;
    v.height = height; v.nonneg_materials = nonneg ? 1 : 0;
    {
        bool fin = true;
        auto ok = [](float c) { return c - c == 0.0f; };
        for (const DevBranch& d : br) for (int c = 0; c < 3; ++c) fin = fin && ok(d.lo[c]) && ok(d.hi[c]);
        for (const DevBranch& d : br) fin = fin && ok(d.lmax) && ok(d.rmin);
        for (int c = 0; c < 3; ++c) fin = fin && ok(sc->root.lo[c]) && ok(sc->root.hi[c]);
        for (int32_t i = 0; i < sc->n_tris && fin; ++i) for (int c = 0; c < 3; ++c) fin = fin && ok(sc->tris[i].v0[c]) && ok(sc->tris[i].v1[c]) && ok(sc->tris[i].v2[c]);
        v.finite_geometry = fin ? 1 : 0;
    }
    v.verts4 = (const float4*)s->d_verts; v.trix = trix.empty() ? nullptr : (const ushort4*)s->d_trix; v.n_verts = (int32_t)(uverts.size() / 4);
    v.rbranch = (const uint32_t*)s->d_rbranch; v.rroot = rroot;
    v.emitters = (const int32_t*)s->d_emitters; v.n_emitters = n_emitters;
    v.cull_o2max = cull_limits[0]; v.cull_d2min = cull_limits[1]; v.cull_d2max = cull_limits[2];
    v.incremental_ok = (incremental_ok && v.finite_geometry) ? 1 : 0;
    v.branches_m = (const float4*)s->d_branches_m5;
    s->n_grown = 0; for (uint32_t g : br_grown) s->n_grown += (g & 1u) + (g >> 1);
    v.cull_child = (const float4*)s->d_cull_child; v.cull_child16 = (const uint4*)s->d_cull16; v.rtail = (const uint4*)s->d_rtail;
    *out = s;
    return 0;
}

// Frame workspaces outlive the scene that allocated them: one block per device is kept for the next scene
// (sq_release_cached_memory() returns it).  The one-shot calls create and free a scene per frame, and a
// hipMalloc that follows the hipFree of a 32 GB block can wait seconds for the driver to scrub it.
namespace {
struct CachedBlock { void* ptr = nullptr; size_t bytes = 0; };
std::mutex g_cache_mutex;
CachedBlock g_cache[64];
void* cache_take(int device, size_t bytes, size_t* got) {   // a block of at least `bytes` (its size in *got), or nullptr
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    CachedBlock& c = g_cache[device & 63];
    if (!c.ptr || c.bytes < bytes) return nullptr;
    void* p = c.ptr; *got = c.bytes; c = CachedBlock{};
    return p;
}
void cache_give(int device, void* ptr, size_t bytes) { // keeps the larger block, frees the other
    void* drop = ptr;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        CachedBlock& c = g_cache[device & 63];
        if (bytes > c.bytes) { drop = c.ptr; c.ptr = ptr; c.bytes = bytes; }
    }
    if (drop) (void)hipFree(drop);
}
}  // namespace
extern "C" void sq_release_cached_memory(void) {
    for (int d = 0; d < 64; ++d) {
        size_t got = 0;
        void* p = cache_take(d, 0, &got);
        if (p && hipSetDevice(d) == hipSuccess) (void)hipFree(p);
    }
}

extern "C" void sq_scene_free(sq_device_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (auto& p : s->pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    (void)hipFree(s->d_arena);
    if (s->d_work) cache_give(s->device, s->d_work, s->work_bytes);
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    if (s->aux) (void)hipStreamDestroy(s->aux);
    delete s;
}

extern "C" int32_t sq_shard_rows(int32_t w, sq_shard sh) {
    if (w <= 0 || sh.row_block <= 0 || sh.n_shards <= 0 || sh.shard < 0 || sh.shard >= sh.n_shards) return -1;
    const int64_t nblocks = ((int64_t)w + sh.row_block - 1) / sh.row_block;
    int64_t rows = 0;
    for (int64_t b = sh.shard; b < nblocks; b += sh.n_shards) {
        const int64_t y0 = b * sh.row_block, y1 = (y0 + sh.row_block < w) ? y0 + sh.row_block : w;
        rows += y1 - y0;
    }
    return (int32_t)rows;
}
extern "C" int32_t sq_shard_global_row(int32_t j, sq_shard sh) {
    const int32_t blk = j / sh.row_block;
    return (blk * sh.n_shards + sh.shard) * sh.row_block + (j - blk * sh.row_block);
}

namespace {

// Carves the frame workspace out of one allocation (grow-only; allocation happens outside timed steps after warm-up).
// If the device cannot give `slots` sample slots the request is halved (down to one sample per pixel): the frame
// then simply runs in more batches.
int ensure_workspace(sq_device_scene* s, int64_t pixels, int64_t slots) {
    if (pixels <= s->work_pixels && slots <= s->work_slots && s->d_work) return 0;
    pixels = std::max(pixels, s->work_pixels); slots = std::max(slots, s->work_slots);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    if (s->d_work) { cache_give(s->device, s->d_work, s->work_bytes); s->d_work = nullptr; s->work_pixels = s->work_slots = 0; }
    size_t block_bytes = 0;
    size_t off = 0, o_cnt = 0, o_stats = 0, o_pix = 0, o_t0 = 0, o_tri0 = 0, o_sum = 0, o_mt = 0, o_mtri = 0, o_state = 0, o_org = 0, o_dir = 0, o_rad = 0;
    for (;;) {
        off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
        o_cnt = take(128 * sizeof(int32_t)); o_stats = take(kStatSlots * sizeof(unsigned long long));
        o_pix = take(pixels * 4); o_t0 = take(pixels * 4); o_tri0 = take(pixels * 4); o_sum = take(pixels * 12);
        o_mt = take(pixels * 4); o_mtri = take(pixels * 4);
        // + pixels: the mirror rays' spare region behind the sample slots (state and the two ray quads)
        o_state = take(slots + pixels); o_org = take((slots + pixels) * 16); o_dir = take((slots + pixels) * 16); o_rad = take(slots * 12);
        // every array has its own, ordered place in the block: a slip here would be a GPU fault, not an error code
        if (!(o_cnt < o_stats && o_stats < o_pix && o_pix < o_t0 && o_t0 < o_tri0 && o_tri0 < o_sum && o_sum < o_mt && o_mt < o_mtri &&
              o_mtri < o_state && o_state < o_org && o_org < o_dir && o_dir < o_rad && o_rad < off))
            return sq_set_error("internal error: frame workspace layout");
        block_bytes = off;
        if ((s->d_work = cache_take(s->device, off, &block_bytes)) != nullptr) break;
        if (hipMalloc(&s->d_work, off) == hipSuccess) break;
        (void)hipGetLastError();
        s->d_work = nullptr;
        if (slots <= pixels) return sq_set_error("hipMalloc(%zu B) for the frame workspace failed", off);
        slots = std::max<int64_t>(pixels, slots / 2);
    }
    char* base = (char*)s->d_work;
    Work& W = s->work;
    int32_t* cnt = (int32_t*)(base + o_cnt);
    W.n_active = cnt; W.head[0] = cnt + 16; W.head[1] = cnt + 32;      // separate cache lines
    W.stats = (unsigned long long*)(base + o_stats);
    if (hipMemset(W.stats, 0, kStatSlots * sizeof(unsigned long long)) != hipSuccess) return sq_set_error("hipMemset failed");
    W.px_pixel = (int32_t*)(base + o_pix); W.px_t0 = (float*)(base + o_t0); W.px_tri0 = (int32_t*)(base + o_tri0); W.px_sum = (float*)(base + o_sum);
    W.px_mt = (float*)(base + o_mt); W.px_mtri = (int32_t*)(base + o_mtri);
    W.state = (uint8_t*)(base + o_state); W.org = (float4*)(base + o_org); W.dir = (float4*)(base + o_dir); W.rad = (float*)(base + o_rad);
    W.slot_capacity = slots;
    s->work_bytes = block_bytes; s->work_pixels = pixels; s->work_slots = slots;
    return 0;
}

template <typename StackT>
int launch_frame(sq_device_scene* s, const Frame& F, hipStream_t stream) {
    SceneView S = s->view;
    if (!s->opt_cull) S.cull_o2max = -1.0f;                            // no ray is inside the culling limits: every leaf is tested
    if (!s->opt_incremental) S.incremental_ok = 0;                     // resident form: every branch step tests both children from the branch's own box
    const long long pixels = (long long)F.local_rows * F.h;
    const int stack_cap = std::max(S.height, 1);
    const size_t px_lds = (size_t)kBlock * stack_cap * sizeof(StackT);
    const long long px_blocks = (pixels + kBlock - 1) / kBlock;
    if (px_blocks > 0x7fffffffLL) return sq_set_error("image too large for one launch");
    if (px_lds > 160 * 1024) return sq_set_error("BIH height %d needs %zu B of LDS stack per workgroup (max 163840)", S.height, px_lds);
    auto timed = [&](auto&& fn, const char* name, hipStream_t on) -> int {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (s->opt_timing) { SQ_HIP(hipEventCreate(&e0)); SQ_HIP(hipEventCreate(&e1)); SQ_HIP(hipEventRecord(e0, on)); }
        fn();
        SQ_HIP(hipGetLastError());
        if (s->opt_timing) {
            SQ_HIP(hipEventRecord(e1, on)); s->pending.emplace_back(e0, e1); s->last_kernel = name;
            if (s->pending.size() > 8192) SQ_HIP(sq_kernel_timing(s, nullptr, nullptr, nullptr) ? hipErrorUnknown : hipSuccess);   // fold, bounded memory
        }
        return 0;
    };
    if (s->opt_variant == 1 || F.cast) {
        if (px_lds > 64 * 1024) SQ_HIP(hipFuncSetAttribute((const void*)sq_render_pixels<StackT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)px_lds));
        return timed([&] { hipLaunchKernelGGL(sq_render_pixels<StackT>, dim3((unsigned)px_blocks), dim3(kBlock), px_lds, stream, S, F); }, "sq_render_pixels", stream);
    }
    // ---- wavefront pipeline ----
    // at least one sample of every pixel per batch, never more slots than the frame has samples
    const int64_t slots = std::max<int64_t>(pixels, std::min<int64_t>(s->opt_slots, (int64_t)pixels * F.samples));
    if (ensure_workspace(s, pixels, slots)) return 1;
    const Work& W = s->work;
    const int64_t have_slots = W.slot_capacity;
    // Overlapped schedule: the sample batches alternate between two halves of the workspace ("tracks"); every
    // trace launch stays on the caller's stream, in the order T1(a) T1(b) T2(a) T2(b), while the per-sample
    // kernels (RNG + bounce, shading, accumulation) run on a second stream beside them, ordered by events.
    // Opt-in (sq_set_option "overlap"): measured +3 % on the headline frame (105.5 -> 102.3 ms) -- the kernels do
    // run side by side, but the chip is VALU-bound as a whole, so each slows the other down by what it gains;
    // and trace-launch durations then include that interference, which blurs the per-kernel roofline figure.
    const bool overlap = s->opt_overlap && F.samples >= 2 && have_slots >= 2 * pixels;
    const int tracks = overlap ? 2 : 1;
    const int64_t track_slots = have_slots / tracks;
    Work Wt[2] = { W, W };
    if (overlap) {
        Work& V = Wt[1];
        V.state += track_slots; V.org += track_slots; V.dir += track_slots; V.rad += 3 * track_slots;
        V.head[0] = W.n_active + 64 + 16; V.head[1] = W.n_active + 64 + 32;
    }
    // samples per batch: as many as a track holds, split evenly (few large launches: a small trace launch
    // wastes its ramp-up and drain, and the second-bounce launches only carry a few percent of the slots)
    const int max_batch = (int)std::max<int64_t>(1, std::min<int64_t>(F.samples, track_slots / pixels));
    const int n_batches = std::max(tracks, (F.samples + max_batch - 1) / max_batch);
    const int batch = (F.samples + n_batches - 1) / n_batches;
    if (F.out_avg) SQ_HIP(hipMemsetAsync(F.out_avg, 0, (size_t)pixels * 3 * sizeof(float), stream));   // pixels whose primary ray misses: black
    if (F.out_rgb) SQ_HIP(hipMemsetAsync(F.out_rgb, 0, (size_t)pixels * 3, stream));
    SQ_HIP(hipMemsetAsync(W.n_active, 0, 128 * sizeof(int32_t), stream));
    // persistent trace kernel geometry.  Resident form: the whole scene (branches, leaves, unique vertices,
    // 16-bit indexed triangles) plus every lane's stack fits in the 160 KB of one CU -> one 1024-thread
    // workgroup per CU, no global traffic except ray fetch and hit store.  Streaming form otherwise.
    const size_t lds_budget = 160 * 1024;
    bool resident = false;
    const bool pool = s->opt_pool != 0;
    TraceLds L{};
    if (s->opt_resident && S.trix) {
        L = trace_lds_layout(S.n_branches, true, S.n_verts, S.n_tris, kResidentBlock, stack_cap, (int)sizeof(StackT), pool);
        resident = L.total <= lds_budget && S.n_verts <= 4096;         // vertex byte offsets are 16-bit (ResidentTris)
    }
    int n_lds = S.n_branches, trace_blocks = 0, trace_threads = 0;
    const void* trace_fn = nullptr;
    if (resident) {
        trace_fn = pool ? (s->opt_profile ? (const void*)sq_trace_rays<StackT, true, kResidentBlock, true, true> : (const void*)sq_trace_rays<StackT, true, kResidentBlock, false, true>)
                        : (s->opt_profile ? (const void*)sq_trace_rays<StackT, true, kResidentBlock, true, false> : (const void*)sq_trace_rays<StackT, true, kResidentBlock, false, false>);
        trace_blocks = s->n_cu; trace_threads = kResidentBlock;
    } else {
        const size_t max_node_bytes = (size_t)s->opt_lds_node_kb * 1024;     // top of the tree; the rest of LDS buys occupancy
        const int n_lds_want = (int)std::min<size_t>((size_t)S.n_branches, max_node_bytes / 48);
        const TraceLds L0 = trace_lds_layout(0, false, S.n_verts, S.n_tris, kTraceBlock, stack_cap, (int)sizeof(StackT), pool);   // stacks, live lists, window tables
        if (L0.total > lds_budget) return sq_set_error("BIH height %d needs %u B of LDS per workgroup (max %zu)", S.height, L0.total, lds_budget);
        // Three workgroups per CU (the six-wave build) when a third of the LDS holds a workgroup's stacks plus at least 4 KB of
        // the tree's top (or all of it); otherwise two, or one, with up to lds_node_kb of tree each.
        const size_t third = (lds_budget / 3) & ~(size_t)2047;            // 52 KB: the hardware allocates LDS in granules, and 3 x 53.3 KB rounded up does not fit
        const bool dense_fits = L0.total + std::min<size_t>((size_t)S.n_branches * 48, 4096) + 16 <= third;
        const bool dense = pool && !s->opt_profile && (s->opt_trace_blocks_per_cu == 0 ? dense_fits : s->opt_trace_blocks_per_cu == 3 && dense_fits);
        int per_cu;
        if (dense) {
            per_cu = 3;
            n_lds = (int)std::min<size_t>((size_t)n_lds_want, (third - L0.total - 16) / 48);
        } else {
            n_lds = n_lds_want;
            while (n_lds > 0 && trace_lds_layout(n_lds, false, S.n_verts, S.n_tris, kTraceBlock, stack_cap, (int)sizeof(StackT), pool).total > lds_budget) n_lds /= 2;
        }
        L = trace_lds_layout(n_lds, false, S.n_verts, S.n_tris, kTraceBlock, stack_cap, (int)sizeof(StackT), pool);
        if (!dense) {
            per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, lds_budget / L.total));    // the plain build's 86 VGPRs allow four waves per SIMD = two workgroups
            if (s->opt_trace_blocks_per_cu > 0) per_cu = (int)std::min<int64_t>(s->opt_trace_blocks_per_cu, (int64_t)std::max<size_t>(1, lds_budget / L.total));
        }
        trace_fn = dense ? (const void*)sq_trace_rays_dense<StackT>
                 : pool ? (s->opt_profile ? (const void*)sq_trace_rays<StackT, false, kTraceBlock, true, true> : (const void*)sq_trace_rays<StackT, false, kTraceBlock, false, true>)
                        : (s->opt_profile ? (const void*)sq_trace_rays<StackT, false, kTraceBlock, true, false> : (const void*)sq_trace_rays<StackT, false, kTraceBlock, false, false>);
        trace_blocks = s->n_cu * per_cu; trace_threads = kTraceBlock;
    }
    const size_t tr_lds = L.total;
    if (tr_lds > 64 * 1024) SQ_HIP(hipFuncSetAttribute(trace_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tr_lds));
    static bool static_lds_checked = false;                           // per StackT instantiation; once per process is enough
    if (resident && !static_lds_checked) {
        // ResidentTris takes a vertex's byte offset for its LDS address: the vertex table must sit at LDS address 0, i.e. the
        // kernels that stage a resident scene must own no static __shared__ (their dynamic LDS then starts at 0).  Checked here,
        // where a violation is an error code, rather than by the device-side trap, where it would be a GPU abort.
        const void* fns[5] = { (const void*)sq_trace_rays<StackT, true, kResidentBlock, false, true>, (const void*)sq_trace_rays<StackT, true, kResidentBlock, true, true>,
                               (const void*)sq_trace_rays<StackT, true, kResidentBlock, false, false>, (const void*)sq_trace_rays<StackT, true, kResidentBlock, true, false>,
                               (const void*)sq_primary_resident<StackT> };
        for (const void* fn : fns) {
            hipFuncAttributes attr{};
            SQ_HIP(hipFuncGetAttributes(&attr, fn));
            if (attr.sharedSizeBytes != 0)
                return sq_set_error("internal error: a resident-scene kernel has %zu B of static LDS; its vertex table would not start at LDS address 0", (size_t)attr.sharedSizeBytes);
        }
        static_lds_checked = true;
    }
    // primary rays: once per pixel.  With a resident scene they are traced out of LDS as well.
    const bool primary_pooled = s->opt_primary_pooled && pool;          // ... or through the pooled trace kernel, below
    if (primary_pooled) {
    } else if (resident && s->opt_primary_resident) {
        const TraceLds Lp = trace_lds_layout(S.n_branches, true, S.n_verts, S.n_tris, kResidentBlock, stack_cap, (int)sizeof(StackT), false);
        SQ_HIP(hipFuncSetAttribute((const void*)sq_primary_resident<StackT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lp.total));
        const long long need = (primary_padded(F) + kResidentBlock - 1) / kResidentBlock;
        hipLaunchKernelGGL(sq_primary_resident<StackT>, dim3((unsigned)std::min<long long>(s->n_cu, need)), dim3(kResidentBlock), Lp.total, stream, S, F, W, stack_cap);
    } else {
        if (px_lds > 64 * 1024) SQ_HIP(hipFuncSetAttribute((const void*)sq_primary<StackT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)px_lds));
        hipLaunchKernelGGL(sq_primary<StackT>, dim3((unsigned)((primary_padded(F) + kBlock - 1) / kBlock)), dim3(kBlock), px_lds, stream, S, F, W);
    }
    SQ_HIP(hipGetLastError());
    const int aux_blocks = s->n_cu * (int)(s->opt_aux_blocks_per_cu ? s->opt_aux_blocks_per_cu : 8);
    // sq_accumulate: the grouped loop when the shard has no more pixels than the launch has threads (every thread at most one pixel)
    const bool acc_grouped = pixels <= (long long)aux_blocks * kBlock;
#define SQ_LAUNCH_ACCUMULATE(...) do { if (acc_grouped) hipLaunchKernelGGL(sq_accumulate<true>, __VA_ARGS__); else hipLaunchKernelGGL(sq_accumulate<false>, __VA_ARGS__); } while (0)
    // per-sample kernels that run one thread per active pixel: x covers the pixels, y splits a pixel's samples when the
    // frame has too few pixels to fill the chip (one rank's share of a frame, small frames)
    auto pp_grid = [&](int kc) {
        // Overlapped schedules, option "aux_polite" = n > 0: the per-sample kernels get n workgroups per CU in all (grid-stride
        // loops do the rest), few enough that a trace workgroup -- 16 waves, all of the CU's LDS, 4 x 104 VGPRs per SIMD -- can
        // always be placed beside them, whichever kernel reaches a CU first.
        if (overlap && s->opt_aux_polite > 0) return dim3((unsigned)(s->n_cu * (int)s->opt_aux_polite), 1u);
        const long long bx = std::max<long long>(1, std::min<long long>((pixels + kBlock - 1) / kBlock, 1 << 20));
        const long long want_threads = (long long)s->n_cu * 2048 * 2;
        const long long ks = std::max<long long>(1, std::min<long long>(std::min(kc, 64), (want_threads + pixels - 1) / std::max<long long>(pixels, 1)));
        return dim3((unsigned)bx, (unsigned)ks);
    };
    auto launch_trace = [&](const Work& W, int kc, int level, hipStream_t on, bool with_mirror_rays = false) -> int {
        // a launch with few slots (the per-pixel mirror rays) takes small reservations, or only a few waves get any
        const int max_chunk = resident ? kChunkResident : kChunkStreaming;
        const int64_t per_wave = pixels * (int64_t)kc / std::max(1, trace_blocks * (trace_threads / 64));
        const int chunk = (int)std::min<int64_t>(max_chunk, std::max<int64_t>(64, (per_wave / 8) / 64 * 64));
        // queue order: a pixel's samples in a row pays once the triangles no longer fit the L2s (rays that start at one point
        // share their first leaves: 1M-triangle scene +2.5 %), and costs 1-7 % below that (strided queue reads)
        const bool pixel_major = s->opt_pixel_major < 0 ? (!resident && (size_t)S.n_tris * sizeof(DevTri) > ((size_t)4 << 20)) : s->opt_pixel_major != 0;
        int guide_shift = 2;                                            // log2(4 x waves of the launch), rounded up
        while ((1ll << guide_shift) < 4ll * trace_blocks * (trace_threads / 64)) ++guide_shift;
        if (!((s->opt_guided >> level) & 1)) guide_shift = 62;         // bit 0: first bounce level (and the mirror / primary launches), bit 1: second
        TraceArgs A{ W.org, W.dir, W.state, level == 0 ? (int32_t)kRay1 : (int32_t)kRay2, (long long)s->work.slot_capacity, with_mirror_rays ? 1 : 0,
                     W.n_active, kc, W.head[level], n_lds, stack_cap, (int32_t)s->opt_straggler, chunk, guide_shift,
                     (int32_t)s->opt_refill_min, (int32_t)s->opt_flush_min, (int32_t)s->opt_descend_extra, (int32_t)s->opt_descend_lanes,
                     (int32_t)(s->opt_coresidency ? 1 : 0), (int32_t)s->opt_trace_prio, (int32_t)pixel_major, W.stats };
        return timed([&] {
            void* kargs[] = { (void*)&S, (void*)&A };
            (void)hipLaunchKernel(trace_fn, dim3(trace_blocks), dim3(trace_threads), kargs, tr_lds, on);
        }, "sq_trace_rays", on);
    };
    if (primary_pooled) {
        Work Wp = W; Wp.n_active = W.n_active + 48;                     // the launch's queue is the shard's pixels, not the active ones
        SQ_HIP(hipMemsetAsync(W.head[0], 0, 32 * sizeof(int32_t), stream));
        hipLaunchKernelGGL(sq_primary_gen, dim3(aux_blocks), dim3(kBlock), 0, stream, F, W, pixels);
        SQ_HIP(hipGetLastError());
        if (launch_trace(Wp, 1, 0, stream)) return 1;
        hipLaunchKernelGGL(sq_primary_store, dim3(aux_blocks), dim3(kBlock), 0, stream, W, pixels);
        SQ_HIP(hipGetLastError());
    }
    // once per frame: the depth-0 mirror ray of every active pixel (reused by every sample that mirrors).  In the plain
    // schedule these rays ride at the head of the first batch's first bounce launch (slots behind the sample slots,
    // dequeued first); the overlapped schedules give them a launch of their own, before the tracks split.
    const bool mirror_rides = !overlap;
    if (!mirror_rides) {
        SQ_HIP(hipMemsetAsync(W.head[0], 0, 32 * sizeof(int32_t), stream));
        hipLaunchKernelGGL(sq_mirror1_gen, dim3(aux_blocks), dim3(kBlock), 0, stream, S, F, W, 0ll);
        SQ_HIP(hipGetLastError());
        if (launch_trace(W, 1, 0, stream)) return 1;
        hipLaunchKernelGGL(sq_mirror1_store, dim3(aux_blocks), dim3(kBlock), 0, stream, W, 0ll);
        SQ_HIP(hipGetLastError());
    }
    auto k0_of = [&](int i) { return i * batch; };
    auto kc_of = [&](int i) { return std::max(0, std::min(batch, F.samples - i * batch)); };
    int n_real = 0;
    while (n_real < n_batches && kc_of(n_real) > 0) ++n_real;
    if (!overlap) {
        for (int i = 0; i < n_real; ++i) {
            const int k0 = k0_of(i), kc = kc_of(i);
            SQ_HIP(hipMemsetAsync(W.head[0], 0, 32 * sizeof(int32_t), stream));     // both dequeue cursors
            hipLaunchKernelGGL(sq_gen_bounce1, pp_grid(kc), dim3(kBlock), 0, stream, S, F, W, k0, kc);
            SQ_HIP(hipGetLastError());
            const bool front = mirror_rides && i == 0;
            if (front) hipLaunchKernelGGL(sq_mirror1_gen, dim3(aux_blocks), dim3(kBlock), 0, stream, S, F, W, (long long)W.slot_capacity);
            for (int level = 0; level < 2; ++level) {
                if (launch_trace(W, kc, level, stream, front && level == 0)) return 1;
                if (front && level == 0) hipLaunchKernelGGL(sq_mirror1_store, dim3(aux_blocks), dim3(kBlock), 0, stream, W, (long long)W.slot_capacity);
                if (level == 0) hipLaunchKernelGGL(sq_shade1, pp_grid(kc), dim3(kBlock), 0, stream, S, F, W, kc);
                SQ_HIP(hipGetLastError());
            }
            SQ_LAUNCH_ACCUMULATE( dim3(aux_blocks), dim3(kBlock), 0, stream, S, F, W, kc, (k0 + kc >= F.samples) ? 1 : 0);
            SQ_HIP(hipGetLastError());
        }
        return 0;
    }
    if (!s->aux) {
        // The second stream carries the per-sample kernels (overlap 1) or the odd batches (overlap 2).  Lowest priority: when a
        // trace launch and a per-sample kernel become ready together, the trace workgroups (one per CU, all of its LDS) must be
        // placed first and the per-sample blocks fill the wave slots beside them; the other way round the trace workgroups wait
        // for whole CUs to drain (rocprofv3 timeline, profiles/r03c_timeline_clocks_coresidency.txt).
        int least = 0, greatest = 0;
        SQ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (s->opt_aux_low_priority) SQ_HIP(hipStreamCreateWithPriority(&s->aux, hipStreamNonBlocking, least));    // `least` = numerically largest = lowest priority
        else SQ_HIP(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking));
    }
    const hipStream_t X = s->aux;
    size_t next_event = 0;
    auto new_event = [&](hipEvent_t* e) -> int {
        if (next_event == s->events.size()) { hipEvent_t n; SQ_HIP(hipEventCreateWithFlags(&n, hipEventDisableTiming)); s->events.push_back(n); }
        *e = s->events[next_event++];
        return 0;
    };
    if (s->opt_overlap == 2) {
        // Two pipelines: even batches run start to end on the caller's stream, odd batches on the second stream.  A trace
        // launch needs a whole CU's LDS per workgroup, so the two tracks' launches do not share CUs: the later one's
        // workgroups move in as the earlier one's finish, which fills the ramp-down of every launch (a ray takes
        // 150-300 us from fetch to hit, and a launch ends when its slowest rays do) and overlaps the per-sample kernels
        // of one track with the trace launches of the other.  Only the accumulation is ordered across tracks (src/Lib.hs:88).
        std::vector<hipEvent_t> eAcc((size_t)n_real);
        for (int i = 0; i < n_real; ++i) if (new_event(&eAcc[(size_t)i])) return 1;
        hipEvent_t e_setup2, e_done2;
        if (new_event(&e_setup2) || new_event(&e_done2)) return 1;
        SQ_HIP(hipEventRecord(e_setup2, stream));
        SQ_HIP(hipStreamWaitEvent(X, e_setup2, 0));
        for (int i = 0; i < n_real; ++i) {
            const hipStream_t on = (i & 1) ? X : stream;
            const Work& V = Wt[i & 1];
            const int k0 = k0_of(i), kc = kc_of(i);
            SQ_HIP(hipMemsetAsync(V.head[0], 0, 32 * sizeof(int32_t), on));
            hipLaunchKernelGGL(sq_gen_bounce1, pp_grid(kc), dim3(kBlock), 0, on, S, F, V, k0, kc);
            SQ_HIP(hipGetLastError());
            if (launch_trace(V, kc, 0, on)) return 1;
            hipLaunchKernelGGL(sq_shade1, pp_grid(kc), dim3(kBlock), 0, on, S, F, V, kc);
            SQ_HIP(hipGetLastError());
            if (launch_trace(V, kc, 1, on)) return 1;
            if (i > 0) SQ_HIP(hipStreamWaitEvent(on, eAcc[(size_t)i - 1], 0));
            SQ_LAUNCH_ACCUMULATE( dim3(aux_blocks), dim3(kBlock), 0, on, S, F, V, kc, i == n_real - 1 ? 1 : 0);
            SQ_HIP(hipGetLastError());
            SQ_HIP(hipEventRecord(eAcc[(size_t)i], on));
        }
        SQ_HIP(hipEventRecord(e_done2, X));
        SQ_HIP(hipStreamWaitEvent(stream, e_done2, 0));
        return 0;
    }
    // the four events of a batch: G = its rays are generated, T1 / T2 = a trace level is done, S1 = ray 2 is in the slots
    std::vector<hipEvent_t> eG((size_t)n_real), eT1((size_t)n_real), eS1((size_t)n_real), eT2((size_t)n_real);
    for (int i = 0; i < n_real; ++i) if (new_event(&eG[(size_t)i]) || new_event(&eT1[(size_t)i]) || new_event(&eS1[(size_t)i]) || new_event(&eT2[(size_t)i])) return 1;
    hipEvent_t e_setup, e_done;
    if (new_event(&e_setup) || new_event(&e_done)) return 1;
    SQ_HIP(hipEventRecord(e_setup, stream));
    SQ_HIP(hipStreamWaitEvent(X, e_setup, 0));
    auto gen = [&](int i) -> int {                          // on X
        const Work& V = Wt[i & 1];
        SQ_HIP(hipMemsetAsync(V.head[0], 0, 32 * sizeof(int32_t), X));
        hipLaunchKernelGGL(sq_gen_bounce1, pp_grid(kc_of(i)), dim3(kBlock), 0, X, S, F, V, k0_of(i), kc_of(i));
        SQ_HIP(hipGetLastError());
        SQ_HIP(hipEventRecord(eG[(size_t)i], X));
        return 0;
    };
    auto trace = [&](int i, int level) -> int {             // on the caller's stream
        SQ_HIP(hipStreamWaitEvent(stream, level == 0 ? eG[(size_t)i] : eS1[(size_t)i], 0));
        if (launch_trace(Wt[i & 1], kc_of(i), level, stream)) return 1;
        SQ_HIP(hipEventRecord(level == 0 ? eT1[(size_t)i] : eT2[(size_t)i], stream));
        return 0;
    };
    auto shade1 = [&](int i) -> int {                       // on X
        SQ_HIP(hipStreamWaitEvent(X, eT1[(size_t)i], 0));
        hipLaunchKernelGGL(sq_shade1, pp_grid(kc_of(i)), dim3(kBlock), 0, X, S, F, Wt[i & 1], kc_of(i));
        SQ_HIP(hipGetLastError());
        SQ_HIP(hipEventRecord(eS1[(size_t)i], X));
        return 0;
    };
    auto finish = [&](int i) -> int {                       // on X, in batch order: the per-pixel sum is ordered (src/Lib.hs:88)
        SQ_HIP(hipStreamWaitEvent(X, eT2[(size_t)i], 0));
        SQ_LAUNCH_ACCUMULATE( dim3(aux_blocks), dim3(kBlock), 0, X, S, F, Wt[i & 1], kc_of(i), i == n_real - 1 ? 1 : 0);
        SQ_HIP(hipGetLastError());
        if (i + 2 < n_real) return gen(i + 2);              // the track is free again
        return 0;
    };
    if (gen(0)) return 1;
    if (n_real > 1 && gen(1)) return 1;
    for (int a = 0; a < n_real; a += 2) {
        const int b = a + 1 < n_real ? a + 1 : -1;
        if (trace(a, 0)) return 1;
        if (b >= 0 && trace(b, 0)) return 1;
        if (shade1(a)) return 1;
        if (b >= 0 && shade1(b)) return 1;
        if (trace(a, 1)) return 1;
        if (b >= 0 && trace(b, 1)) return 1;
        if (finish(a)) return 1;
        if (b >= 0 && finish(b)) return 1;
    }
    SQ_HIP(hipEventRecord(e_done, X));
    SQ_HIP(hipStreamWaitEvent(stream, e_done, 0));
    return 0;
}

}  // namespace

extern "C" int sq_render_rows_device(sq_device_scene* s, const sq_camera* cam, int32_t samples, int32_t w, int32_t h,
                                     int32_t cast, sq_shard sh, float* d_avg, uint8_t* d_rgb, void* hip_stream) {
    if (!s || !cam) return sq_set_error("null argument");
    if (samples < 1 || w < 1 || h < 1) return sq_set_error("samples, width and height must be positive (got %d, %d, %d)", samples, w, h);
    const int32_t rows = sq_shard_rows(w, sh);
    if (rows < 0) return sq_set_error("bad shard {row_block=%d, shard=%d, n_shards=%d}", sh.row_block, sh.shard, sh.n_shards);
    if (rows == 0) return 0;                    // an empty shard (more shards than row blocks) has nothing to render
    if (!d_avg && !d_rgb) return sq_set_error("no output buffer");
    SQ_HIP(hipSetDevice(s->device));
    Frame F{};
    std::memcpy(F.cam_pos, cam->pos, sizeof F.cam_pos);
    std::memcpy(F.cam_rot, cam->rot, sizeof F.cam_rot);
    F.samples = samples; F.w = w; F.h = h; F.cast = cast ? 1 : 0;
    F.row_block = sh.row_block; F.shard = sh.shard; F.n_shards = sh.n_shards; F.local_rows = rows;
    {   // primary-ray tiles: as tall as the adjacency of local rows allows (8 x 8 on a whole image, 2 x 32 with blocks of 2 rows)
        const int rb = sh.n_shards <= 1 ? 8 : sh.row_block;
        F.tile_rows = rb >= 8 && rb % 8 == 0 ? 8 : rb >= 4 && rb % 4 == 0 ? 4 : rb >= 2 && rb % 2 == 0 ? 2 : 1;
        if (s->opt_primary_tiles == 0) F.tile_rows = 1;
        const int tw = 64 / F.tile_rows;
        F.tiles_x = (h + tw - 1) / tw;
    }
    F.out_avg = d_avg; F.out_rgb = d_rgb;
    F.diag = s->opt_coresidency ? std::max(1, s->n_cu - 8) : 0;   // "beside" = while all but a handful of the CUs hold a live trace workgroup
    hipStream_t stream = (hipStream_t)hip_stream;
    return s->small_index ? launch_frame<uint16_t>(s, F, stream) : launch_frame<uint32_t>(s, F, stream);
}

extern "C" int sq_kernel_timing(sq_device_scene* s, double* avg_ms, int64_t* launches, const char** name) {
    if (!s) return sq_set_error("null argument");
    SQ_HIP(hipSetDevice(s->device));
    for (auto& p : s->pending) {
        SQ_HIP(hipEventSynchronize(p.second));
        float ms = 0;
        SQ_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        s->total_ms += ms; s->launches++;
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    s->pending.clear();
    if (avg_ms) *avg_ms = s->launches ? s->total_ms / (double)s->launches : 0.0;
    if (launches) *launches = s->launches;
    if (name) *name = s->last_kernel;
    return 0;
}
extern "C" void sq_kernel_timing_reset(sq_device_scene* s) {
    if (!s) return;
    for (auto& p : s->pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    s->pending.clear(); s->total_ms = 0; s->launches = 0;
}
extern "C" int sq_get_stats(sq_device_scene* s, uint64_t* out, int32_t n, int32_t reset) {
    if (!s || !out || n < 0 || n > kStatSlots) return sq_set_error("bad argument");
    for (int i = 0; i < n; ++i) out[i] = 0;
    if (!s->d_work) return 0;
    SQ_HIP(hipSetDevice(s->device));
    SQ_HIP(hipDeviceSynchronize());
    SQ_HIP(hipMemcpy(out, s->work.stats, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) SQ_HIP(hipMemset(s->work.stats, 0, kStatSlots * sizeof(uint64_t)));
    return 0;
}
extern "C" int sq_set_option(sq_device_scene* s, const char* key, int64_t value) {
    if (!s || !key) return sq_set_error("null argument");
    if (!std::strcmp(key, "timing")) { s->opt_timing = value; return 0; }
    if (!std::strcmp(key, "variant")) { if (value != 1 && value != 2) return sq_set_error("variant must be 1 (per-pixel kernel) or 2 (wavefront)"); s->opt_variant = value; return 0; }
    if (!std::strcmp(key, "slots")) { if (value < 1 || value > (512ll << 20)) return sq_set_error("slots must be in 1..2^29"); s->opt_slots = value; return 0; }
    if (!std::strcmp(key, "straggler_lanes")) { if (value < 0 || value > 63) return sq_set_error("straggler_lanes must be in 0..63"); s->opt_straggler = value; return 0; }
    if (!std::strcmp(key, "resident")) { s->opt_resident = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "profile")) { s->opt_profile = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "lds_node_kb")) { if (value < 0 || value > 128) return sq_set_error("lds_node_kb must be in 0..128"); s->opt_lds_node_kb = value; return 0; }
    if (!std::strcmp(key, "trace_blocks_per_cu")) { if (value < 0 || value > 8) return sq_set_error("trace_blocks_per_cu must be in 0..8"); s->opt_trace_blocks_per_cu = value; return 0; }
    if (!std::strcmp(key, "overlap")) { if (value < 0 || value > 2) return sq_set_error("overlap must be 0, 1 or 2"); s->opt_overlap = value; return 0; }
    if (!std::strcmp(key, "pool")) { s->opt_pool = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "guided")) { if (value < 0 || value > 3) return sq_set_error("guided must be in 0..3"); s->opt_guided = value; return 0; }
    if (!std::strcmp(key, "primary_resident")) { s->opt_primary_resident = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "cull")) { s->opt_cull = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "incremental")) { s->opt_incremental = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "primary_tiles")) { s->opt_primary_tiles = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "primary_pooled")) { s->opt_primary_pooled = value != 0; return 0; }
    if (!std::strcmp(key, "coresidency")) { s->opt_coresidency = value != 0; return 0; }
    if (!std::strcmp(key, "aux_polite")) { if (value < 0 || value > 8) return sq_set_error("aux_polite must be in 0..8"); s->opt_aux_polite = value; return 0; }
    if (!std::strcmp(key, "trace_prio")) { if (value < 0 || value > 3) return sq_set_error("trace_prio must be in 0..3"); s->opt_trace_prio = value; return 0; }
    if (!std::strcmp(key, "aux_low_priority")) {           // takes effect when the second stream is created (first overlapped frame)
        s->opt_aux_low_priority = value != 0;
        if (s->aux) { (void)hipStreamSynchronize(s->aux); (void)hipStreamDestroy(s->aux); s->aux = nullptr; }
        return 0;
    }
    if (!std::strcmp(key, "descend_extra")) { if (value < 0 || value > 16) return sq_set_error("descend_extra must be in 0..16"); s->opt_descend_extra = value; return 0; }
    if (!std::strcmp(key, "descend_lanes")) { if (value < 1 || value > 64) return sq_set_error("descend_lanes must be in 1..64"); s->opt_descend_lanes = value; return 0; }
    if (!std::strcmp(key, "pixel_major")) { s->opt_pixel_major = value < 0 ? -1 : value != 0; return 0; }
    if (!std::strcmp(key, "refill_min")) { if (value < 1 || value > 64) return sq_set_error("refill_min must be in 1..64"); s->opt_refill_min = value; return 0; }
    if (!std::strcmp(key, "flush_min")) { if (value < 0 || value > 64) return sq_set_error("flush_min must be in 0..64"); s->opt_flush_min = value; return 0; }
    if (!std::strcmp(key, "aux_blocks_per_cu")) { if (value < 0 || value > 16) return sq_set_error("aux_blocks_per_cu must be in 0..16"); s->opt_aux_blocks_per_cu = value; return 0; }
    return sq_set_error("unknown option '%s'", key);
}

// ---- one-shot entry points (the drop-in for src/Lib.hs:73-74) ----
namespace {
// One shard of a one-shot call on one device: upload, render into a compact device buffer, copy back.
struct OneshotPart {
    int device = 0; sq_shard shard{ 1, 0, 1 };
    std::vector<float> avg; std::vector<uint8_t> rgb;
    int rc = 0; std::string error;
};
// direct_avg / direct_rgb: the caller's own image (only when this part is the whole frame, rows in order): the finished frame is
// copied straight into it -- the only step left that can fail is that copy itself -- instead of through a staging vector.
int oneshot_part(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h, int32_t cast,
                 bool want_avg, bool want_rgb, OneshotPart& P, float* direct_avg = nullptr, uint8_t* direct_rgb = nullptr) {
    const int32_t rows = sq_shard_rows(w, P.shard);
    if (rows <= 0) return 0;
    // SQ_ONESHOT_TIMING=1: host wall time of each stage of the call on stderr (what a host that binds the one-shot call pays
    // around the frame itself)
    const bool timing = std::getenv("SQ_ONESHOT_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    sq_device_scene* s = nullptr;
    if (sq_scene_upload(scene, P.device, &s)) return 1;
    const auto t1 = now();
    auto t2 = t1, t3 = t1, t4 = t1;
    const size_t npx = (size_t)rows * (size_t)h * 3;
    float* d_avg = nullptr; uint8_t* d_rgb = nullptr; hipStream_t stream = nullptr;
    auto body = [&]() -> int {
        SQ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (want_avg) SQ_HIP(hipMalloc((void**)&d_avg, npx * sizeof(float)));
        if (want_rgb) SQ_HIP(hipMalloc((void**)&d_rgb, npx));
        t2 = now();
        if (sq_render_rows_device(s, cam, samples, w, h, cast, P.shard, d_avg, d_rgb, stream)) return 1;
        SQ_HIP(hipStreamSynchronize(stream));
        t3 = now();
        // staged through private buffers so nothing is written to the caller's memory on failure
        if (want_avg) { float* dst = direct_avg; if (!dst) { P.avg.resize(npx); dst = P.avg.data(); } SQ_HIP(hipMemcpy(dst, d_avg, npx * sizeof(float), hipMemcpyDeviceToHost)); }
        if (want_rgb) { uint8_t* dst = direct_rgb; if (!dst) { P.rgb.resize(npx); dst = P.rgb.data(); } SQ_HIP(hipMemcpy(dst, d_rgb, npx, hipMemcpyDeviceToHost)); }
        t4 = now();
        return 0;
    };
    const int rc = body();
    (void)hipFree(d_avg); (void)hipFree(d_rgb);
    if (stream) (void)hipStreamDestroy(stream);
    sq_scene_free(s);
    if (timing) std::fprintf(stderr, "sq one-shot (device %d): upload %.2f ms, stream+buffers %.2f, render+sync %.2f, copy back %.2f, free %.2f\n",
                             P.device, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, now()));
    return rc;
}

// Devices a one-shot call spreads its rows over.  Default: every visible device when the frame is worth it
// (>= 2^24 samples), else device 0.  SQ_DEVICES="0,1,3" names them explicitly; an index may repeat (several
// shards on one device, which is how the threading is tested on a one-GPU box).
int oneshot_devices(int64_t total_samples, int32_t w, std::vector<int>& out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return sq_set_error("no HIP device available (this library has no CPU fallback)");
    out.clear();
    if (const char* env = std::getenv("SQ_DEVICES")) {
        const char* p = env;
        while (*p) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p || v < 0 || v >= ndev) return sq_set_error("SQ_DEVICES='%s': expected a comma-separated list of device indices in 0..%d", env, ndev - 1);
            out.push_back((int)v);
            p = end;
            if (*p == ',') ++p;
            else if (*p) return sq_set_error("SQ_DEVICES='%s': expected a comma-separated list of device indices in 0..%d", env, ndev - 1);
        }
        if (out.empty()) return sq_set_error("SQ_DEVICES is empty");
        if (out.size() > 64) return sq_set_error("SQ_DEVICES names more than 64 shards");
        return 0;
    }
    const int blocks = (w + kOneshotRowBlock - 1) / kOneshotRowBlock;
    const int n = total_samples >= ((int64_t)1 << 24) ? std::min(ndev, std::max(blocks, 1)) : 1;
    for (int d = 0; d < n; ++d) out.push_back(d);
    return 0;
}

// The foreign call of src/Lib.hs:73-74.  The reference host is ONE process, so this is where a node's GPUs are
// put to work for it: rows are cut into interleaved blocks of 2 (the sq_shard scheme), one host thread per
// device renders its shard, and the shards are de-interleaved into the caller's image.  No exchange between
// devices: a pixel depends only on (x, y, samples, w).
int render_oneshot(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h, int32_t cast,
                   float* out_avg, uint8_t* out_rgb) {
    if (!scene || !cam || (!out_avg && !out_rgb)) return sq_set_error("null argument");
    if (samples < 1 || w < 1 || h < 1) return sq_set_error("samples, width and height must be positive (got %d, %d, %d)", samples, w, h);
    std::vector<int> devices;
    if (oneshot_devices((int64_t)w * h * samples, w, devices)) return 1;
    const int G = (int)devices.size();
    std::vector<OneshotPart> parts((size_t)G);
    for (int g = 0; g < G; ++g) { parts[(size_t)g].device = devices[(size_t)g]; parts[(size_t)g].shard = sq_shard{ G == 1 ? w : kOneshotRowBlock, g, G }; }
    auto run = [&](OneshotPart& P) {
        P.rc = oneshot_part(scene, cam, samples, w, h, cast, out_avg != nullptr, out_rgb != nullptr, P);
        if (P.rc) P.error = sq_last_error();                      // the message is thread-local: carry it out
    };
    if (G == 1) {                                                  // one device renders every row in order: no staging, no de-interleave
        if (oneshot_part(scene, cam, samples, w, h, cast, out_avg != nullptr, out_rgb != nullptr, parts[0], out_avg, out_rgb))
            return sq_set_error("device %d (shard 0 of 1): %s", parts[0].device, std::string(sq_last_error()).c_str());
        return 0;
    }
    {
        std::vector<std::thread> threads;
        for (int g = 1; g < G; ++g) threads.emplace_back(run, std::ref(parts[(size_t)g]));
        run(parts[0]);
        for (auto& t : threads) t.join();
    }
    for (const OneshotPart& P : parts)
        if (P.rc) return sq_set_error("device %d (shard %d of %d): %s", P.device, P.shard.shard, G, P.error.c_str());
    const size_t row_px = (size_t)h * 3;
    for (const OneshotPart& P : parts) {
        const int32_t rows = sq_shard_rows(w, P.shard);
        for (int32_t j = 0; j < rows; ++j) {
            const size_t dst = (size_t)sq_shard_global_row(j, P.shard) * row_px, src = (size_t)j * row_px;
            if (out_avg) std::memcpy(out_avg + dst, P.avg.data() + src, row_px * sizeof(float));
            if (out_rgb) std::memcpy(out_rgb + dst, P.rgb.data() + src, row_px);
        }
    }
    return 0;
}

}  // namespace

extern "C" int sq_render_rgb8(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h, int32_t cast, uint8_t* out) {
    return render_oneshot(scene, cam, samples, w, h, cast, nullptr, out);
}
extern "C" int sq_render_f32(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h, int32_t cast, float* out_avg) {
    return render_oneshot(scene, cam, samples, w, h, cast, out_avg, nullptr);
}

extern "C" int sq_debug_eval(int32_t device, int32_t op, const void* a, const void* b, int64_t n, void* out) {
    if (!a || !out || n < 0 || op < 0 || op > SQ_OP_CULL_SLAB) return sq_set_error("bad argument");
    if (op == SQ_OP_DIV && !b) return sq_set_error("SQ_OP_DIV needs b");
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return sq_set_error("no such HIP device %d", device);
    SQ_HIP(hipSetDevice(device));
    const size_t in_sz = (size_t)n * (op == SQ_OP_TFGEN3 ? 8 : op == SQ_OP_TONEMAP ? 12 : op == SQ_OP_CULL_SLAB ? 36 : 4);
    const size_t out_sz = (size_t)n * (op == SQ_OP_TFGEN3 ? 12 : op == SQ_OP_TONEMAP ? 3 : 4);
    void *da = nullptr, *db = nullptr, *dout = nullptr;
    auto body = [&]() -> int {
        SQ_HIP(hipMalloc(&da, in_sz)); SQ_HIP(hipMalloc(&dout, out_sz));
        SQ_HIP(hipMemcpy(da, a, in_sz, hipMemcpyHostToDevice));
        if (op == SQ_OP_DIV) { SQ_HIP(hipMalloc(&db, in_sz)); SQ_HIP(hipMemcpy(db, b, in_sz, hipMemcpyHostToDevice)); }
        hipLaunchKernelGGL(sq_debug_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, da, db, (long long)n, dout);
        SQ_HIP(hipGetLastError());
        SQ_HIP(hipDeviceSynchronize());
        SQ_HIP(hipMemcpy(out, dout, out_sz, hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = body();
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return rc;
}

extern "C" int32_t sq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int32_t sq_abi_version(void) { return SQ_ABI_VERSION; }
#ifndef SQ_BUILD_ID
#define SQ_BUILD_ID "unknown"
#endif
extern "C" const char* sq_build_id(void) { return SQ_BUILD_ID; }
extern "C" const char* sq_last_error(void) { return sq_error_buffer(); }
