// sq_bih_device.hip — BIH.makeBIH (src/BIH.hs:62-99) on the GPU: sq_bih_build_device().
//
// Produces the same tree, bit for bit, as the host build in sq_host.cpp (and therefore as the reference):
// tree shape and leaf order decide traversal tie-breaks, so nothing here may be "approximately" the same.
// The reference recursion is rebuilt level by level (all nodes of one depth at once), element-parallel:
//
//   per level   k_centroid  one lane per triangle     centroid on its node's axis          Geometry.hs:181-182
//               k_plane     one wave per node         split plane = ORDERED fp32 sum / n   BIH.hs:89-90
//               k_flag+scan one lane per triangle     left = centroid < plane, stable rank BIH.hs:85-86,91-92
//               k_split     one lane per node         sizes, terminal branches             BIH.hs:70-75
//               k_scatter   one lane per triangle     stable partition + child boxes       BIH.hs:77-78, Geometry.hs:155-163
//               k_finish    one lane per node         lmax / rmin, next level's nodes      BIH.hs:93-96
//
// What keeps it exact:
//   * the split plane is a left-to-right fp32 sum.  fp32 addition is not associative, so the sum is a
//     serial chain by definition: a wave stages 1024 centroids at a time in LDS (coalesced loads, the next
//     tile in flight) and adds them in order.  The root of a 1M-triangle mesh is one chain of 1M adds
//     (about 5 ns per add); every other node of a level runs beside it.
//   * boxes are `foldl1 min` / `foldl1 max` with Haskell's min/max (ties keep the earlier / later operand,
//     which only shows in the sign of a zero).  They are reduced as 64-bit keys
//     (order-preserving value bits with -0 == +0) << 32 | (position in the fold), atomicMin / atomicMax after a wave-level
//     reduction, and the winning position is read back for the value: ties resolve exactly as the fold does.
//   * the partition is stable (exclusive scan of the left flags), so leaf order is the reference's.
// Non-finite vertex coordinates (impossible through the .obj grammar short of overflow) are refused: the
// folds above treat NaN the way Haskell's min/max do only in the host build.
#include <hip/hip_runtime.h>

#include <cstring>
#include <utility>
#include <vector>

#include "../../include/squigly_hip.h"
#include "sq_error.h"
#include "sq_host_types.h"
#include "sq_math.h"

#define SQ_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return sq_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

typedef unsigned long long u64;

constexpr int kB = 256;            // element-parallel kernels
constexpr int kLeafLimit = 15;     // `length geom < 15` is a Leaf (src/BIH.hs:69,80)
constexpr int kTile = 1024;        // centroids per LDS tile of the ordered sum
constexpr int kTriFloats = 10;     // sq_tri = 9 floats + material id
constexpr int kScanItems = 8;      // per thread
constexpr int kScanTile = kB * kScanItems;

// Build nodes, structure of arrays.  Ids are handed out level by level and children in pairs
// (left = id, right = id + 1), so a level is a contiguous id range.
struct Nodes {
    int32_t* begin; int32_t* count;
    int32_t* axis;                 // 0..2 = branch on X/Y/Z, 3 = leaf
    int32_t* left;                 // id of the left child, -1 for a leaf
    float* lmax; float* rmin; float* plane;
    float* box;                    // 6 per node: the node's own tight box (lo xyz, hi xyz), BIH.hs:77-78
    uint8_t* state;                // 0 = settled, 1 = splits at this level, 2 = ... into two non-empty sides
};
// Box keys of the children being created at this level, 3 per child, indexed by (child id - level base).
struct Keys { u64* kmin; u64* kmax; };

__device__ inline uint32_t order_bits(float v) {
    const uint32_t u = __float_as_uint(v);
    if (v == 0.0f) return 0x80000000u;                       // -0 and +0 compare equal in the fold
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline u64 wave_min64(u64 v) {
    for (int o = 32; o; o >>= 1) { const u64 w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
__device__ inline u64 wave_max64(u64 v) {
    for (int o = 32; o; o >>= 1) { const u64 w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// Folds the three vertices of triangle `t`, which sits at position `seq_tri` of its box's fold, into keys.
__device__ inline void tri_keys(const float* __restrict__ tris, int32_t t, uint32_t seq_tri, u64 kmn[3], u64 kmx[3]) {
    const float* p = tris + (size_t)t * kTriFloats;
    for (int c = 0; c < 3; ++c) { kmn[c] = ~0ull; kmx[c] = 0ull; }
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
            const u64 key = ((u64)order_bits(p[3 * k + c]) << 32) | (u64)(seq_tri * 3u + (uint32_t)k);
            kmn[c] = key < kmn[c] ? key : kmn[c];            // foldl1 min keeps the earlier of equals (Geometry.hs:156-158)
            kmx[c] = key > kmx[c] ? key : kmx[c];            // foldl1 max keeps the later of equals   (Geometry.hs:159-161)
        }
}
// Lanes with slot >= 0 contribute their keys to box `slot`; one atomic per (wave, slot, component).
__device__ inline void reduce_keys(int32_t slot, const u64 kmn[3], const u64 kmx[3], Keys K) {
    const int lane = (int)(threadIdx.x & 63);
    u64 todo = __ballot(slot >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int32_t s = __shfl(slot, leader, 64);
        const bool mine = slot == s;
        const u64 group = __ballot(mine);
        for (int c = 0; c < 3; ++c) {
            const u64 a = wave_min64(mine ? kmn[c] : ~0ull);
            const u64 b = wave_max64(mine ? kmx[c] : 0ull);
            if (lane == leader) { atomicMin(&K.kmin[(size_t)s * 3 + c], a); atomicMax(&K.kmax[(size_t)s * 3 + c], b); }
        }
        todo &= ~group;
    }
}
// The value a key stands for: vertex `seq % 3` of the triangle at position `seq / 3` of the box's range.
__device__ inline float key_value(u64 key, int c, const float* __restrict__ tris, const int32_t* __restrict__ ids, int32_t begin) {
    const uint32_t seq = (uint32_t)key;
    const int32_t t = ids[(size_t)begin + seq / 3u];
    return tris[(size_t)t * kTriFloats + 3 * (seq % 3u) + c];
}
// longestAxis (src/Geometry.hs:190-193): maximumBy keeps the later of equal maxima
__device__ inline int longest_axis(const float* box) {
    int best = 0;
    for (int c = 1; c < 3; ++c) {
        const float cur = box[3 + best] - box[best], cand = box[3 + c] - box[c];
        if (!sq::cmp_gt(cur, cand)) best = c;
    }
    return best;
}

__global__ void __launch_bounds__(kB) k_init(int32_t T, const float* __restrict__ tris, int32_t* ids, int32_t* node_of, Keys K, int* bad) {
    const int32_t i = (int32_t)(blockIdx.x * kB + threadIdx.x);
    u64 kmn[3], kmx[3];
    if (i < T) {
        ids[i] = i; node_of[i] = 0;
        const float* p = tris + (size_t)i * kTriFloats;
        bool ok = true;
        for (int k = 0; k < 9; ++k) ok = ok && (fabsf(p[k]) <= 3.402823466e38f);
        if (!ok) *bad = 1;
        tri_keys(tris, i, (uint32_t)i, kmn, kmx);
    }
    reduce_keys(i < T ? 0 : -1, kmn, kmx, K);                // the root box (makeBIH, src/BIH.hs:62-65)
}
__global__ void k_root(int32_t T, const float* __restrict__ tris, const int32_t* __restrict__ ids, Nodes N, Keys K, int32_t* next_active, int32_t* next_count) {
    float* box = N.box;
    for (int c = 0; c < 3; ++c) {
        box[c] = key_value(K.kmin[c], c, tris, ids, 0);
        box[3 + c] = key_value(K.kmax[c], c, tris, ids, 0);
    }
    N.begin[0] = 0; N.count[0] = T; N.left[0] = -1; N.lmax[0] = 0; N.rmin[0] = 0;
    if (T >= kLeafLimit) { N.axis[0] = longest_axis(box); N.state[0] = 1; next_active[0] = 0; *next_count = 1; }
    else { N.axis[0] = 3; N.state[0] = 0; *next_count = 0; }
}

// averagePoints of the triangle's vertices, one component (src/Geometry.hs:181-182): foldl from zero, then / 3
__global__ void __launch_bounds__(kB) k_centroid(int32_t T, const int32_t* __restrict__ ids, const int32_t* __restrict__ node_of, Nodes N,
                                                 const float* __restrict__ tris, float* __restrict__ cen) {
    const int32_t i = (int32_t)(blockIdx.x * kB + threadIdx.x);
    if (i >= T) return;
    const int32_t node = node_of[i];
    if (!N.state[node]) return;
    const float* p = tris + (size_t)ids[i] * kTriFloats + N.axis[node];
    cen[i] = (((0.0f + p[0]) + p[3]) + p[6]) / 3.0f;
}

// Split plane (src/BIH.hs:89-90): centroids summed left to right in fp32, divided by the count (itself a
// float that was incremented once per element, so it stops growing at 2^24).
__global__ void __launch_bounds__(64) k_plane(const int32_t* __restrict__ active, Nodes N, const float* __restrict__ cen) {
    __shared__ __attribute__((aligned(16))) float tile[2][kTile];
    const int32_t node = active[blockIdx.x];
    const int32_t b = N.begin[node], n = N.count[node];
    const int lane = (int)threadIdx.x;
    constexpr int kPer = kTile / 64;
    float r[kPer];
    const float* src = cen + b;
    for (int j = 0; j < kPer; ++j) { const int32_t e = j * 64 + lane; r[j] = e < n ? src[e] : 0.0f; }
    float sum = 0.0f;
    int buf = 0;
    for (int32_t t0 = 0; t0 < n; t0 += kTile, buf ^= 1) {
        for (int j = 0; j < kPer; ++j) tile[buf][j * 64 + lane] = r[j];      // zero padding is neutral: the sum is never -0
        __syncthreads();
        const int32_t t1 = t0 + kTile;
        if (t1 < n)
            for (int j = 0; j < kPer; ++j) { const int32_t e = t1 + j * 64 + lane; r[j] = e < n ? src[e] : 0.0f; }
        // 32 adds per group, the next group's LDS reads in flight meanwhile (two register sets, no copies).
        const int groups = (min(kTile, n - t0) + 31) >> 5;
        const float4* q = reinterpret_cast<const float4*>(tile[buf]);
        float4 A[8], B[8];
        auto fetch = [&](float4* dst, int g) {
            const float4* s = q + min(g, kTile / 32 - 1) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = s[j];
            __builtin_amdgcn_sched_barrier(0);               // keep these reads ahead of the adds that follow
        };
        auto fold = [&](const float4* v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sum = sum + v[j].x; sum = sum + v[j].y; sum = sum + v[j].z; sum = sum + v[j].w; }
        };
        fetch(A, 0);
        for (int g = 0; g < groups; g += 2) {
            fetch(B, g + 1);
            fold(A);
            if (g + 1 >= groups) break;
            fetch(A, g + 2);
            fold(B);
        }
    }
    if (lane == 0) N.plane[node] = sum / (float)min(n, 16777216);
}

__global__ void __launch_bounds__(kB) k_flag(int32_t T, const int32_t* __restrict__ node_of, Nodes N, const float* __restrict__ cen, uint32_t* __restrict__ flag) {
    const int32_t i = (int32_t)(blockIdx.x * kB + threadIdx.x);
    if (i >= T) return;
    const int32_t node = node_of[i];
    flag[i] = (N.state[node] && cen[i] < N.plane[node]) ? 1u : 0u;           // strict `<` goes left (src/BIH.hs:91)
}

// ---- exclusive scan of uint32 (three passes, recursive on the tile sums) ----
__device__ inline uint32_t block_exclusive(uint32_t v, uint32_t* total) {    // kB threads
    __shared__ uint32_t wsum[kB / 64];
    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (int k = 0; k < kB / 64; ++k) { if (k < w) base += wsum[k]; all += wsum[k]; }
    __syncthreads();
    *total = all;
    return base + inc - v;
}
__global__ void __launch_bounds__(kB) k_scan_sums(const uint32_t* __restrict__ in, uint32_t* __restrict__ sums, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t s = 0;
    for (int k = 0; k < kScanItems; ++k) if (base + k < n) s += in[base + k];
    uint32_t total;
    block_exclusive(s, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kB) k_scan_apply(const uint32_t* __restrict__ in, const uint32_t* __restrict__ offs, uint32_t* __restrict__ out, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems], s = 0;
    for (int k = 0; k < kScanItems; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint32_t total;
    uint32_t run = block_exclusive(s, &total) + (offs ? offs[blockIdx.x] : 0u);
    for (int k = 0; k < kScanItems; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

__global__ void __launch_bounds__(kB) k_split(const int32_t* __restrict__ active, int32_t n_active, int32_t base, Nodes N,
                                              const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, Keys K) {
    const int32_t r = (int32_t)(blockIdx.x * kB + threadIdx.x);
    if (r >= n_active) return;
    const int32_t node = active[r];
    const int32_t b = N.begin[node], n = N.count[node], ax = N.axis[node];
    const int32_t nl = (int32_t)(pos[b + n - 1] + flag[b + n - 1] - pos[b]), nr = n - nl;
    const int32_t c0 = base + 2 * r, c1 = c0 + 1;
    N.left[node] = c0;
    N.left[c0] = N.left[c1] = -1;
    N.lmax[c0] = N.rmin[c0] = N.lmax[c1] = N.rmin[c1] = 0.0f;
    N.state[c0] = N.state[c1] = 0;
    N.axis[c0] = N.axis[c1] = 3;
    const float* box = N.box + (size_t)node * 6;
    if (nl == 0 || nr == 0) {
        // Terminal branch (src/BIH.hs:70-75): an empty leaf beside a leaf of everything.  The populated side's
        // extreme is the node's own box face (same fold over the same vertices); the empty side takes the
        // maximumDef / minimumDef default (src/BIH.hs:93-96).
        const float face = nl == 0 ? box[ax] : box[3 + ax];
        N.lmax[node] = 0.001f + face;
        N.rmin[node] = (-0.001f) + face;
        N.begin[c0] = b;            N.count[c0] = nl == 0 ? 0 : n;
        N.begin[c1] = b + N.count[c0]; N.count[c1] = nl == 0 ? n : 0;
        N.state[node] = 1;
    } else {
        N.begin[c0] = b;      N.count[c0] = nl;
        N.begin[c1] = b + nl; N.count[c1] = nr;
        for (int c = 0; c < 3; ++c) {
            K.kmin[(size_t)(2 * r) * 3 + c] = ~0ull;     K.kmax[(size_t)(2 * r) * 3 + c] = 0ull;
            K.kmin[(size_t)(2 * r + 1) * 3 + c] = ~0ull; K.kmax[(size_t)(2 * r + 1) * 3 + c] = 0ull;
        }
        N.state[node] = 2;
    }
}

// Stable partition of every properly split node at once (src/BIH.hs:85-86), and the children's boxes.
__global__ void __launch_bounds__(kB) k_scatter(int32_t T, int32_t base, const int32_t* __restrict__ ids_in, const int32_t* __restrict__ node_in,
                                                int32_t* __restrict__ ids_out, int32_t* __restrict__ node_out, Nodes N,
                                                const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos,
                                                const float* __restrict__ tris, Keys K) {
    const int32_t i = (int32_t)(blockIdx.x * kB + threadIdx.x);
    int32_t slot = -1;
    u64 kmn[3], kmx[3];
    if (i < T) {
        const int32_t node = node_in[i], t = ids_in[i];
        if (N.state[node] == 2) {
            const int32_t b = N.begin[node], c0 = N.left[node];
            const int32_t nl = N.count[c0];
            const int32_t rank_l = (int32_t)(pos[i] - pos[b]);
            const bool left = flag[i] != 0;
            const int32_t within = left ? rank_l : (i - b) - rank_l;
            const int32_t to = left ? b + within : b + nl + within;
            const int32_t child = left ? c0 : c0 + 1;
            ids_out[to] = t; node_out[to] = child;
            slot = child - base;
            tri_keys(tris, t, (uint32_t)within, kmn, kmx);
        } else {
            ids_out[i] = t; node_out[i] = node;
        }
    }
    reduce_keys(slot, kmn, kmx, K);
}

__global__ void __launch_bounds__(kB) k_finish(const int32_t* __restrict__ active, int32_t n_active, int32_t base, Nodes N,
                                               const int32_t* __restrict__ ids, const float* __restrict__ tris, Keys K,
                                               int32_t* __restrict__ next_active, int32_t* __restrict__ next_count) {
    const int32_t r = (int32_t)(blockIdx.x * kB + threadIdx.x);
    if (r >= n_active) return;
    const int32_t node = active[r];
    if (N.state[node] == 2) {
        const int32_t ax = N.axis[node];
        for (int side = 0; side < 2; ++side) {
            const int32_t c = base + 2 * r + side, cb = N.begin[c];
            float* box = N.box + (size_t)c * 6;
            for (int k = 0; k < 3; ++k) {
                box[k] = key_value(K.kmin[(size_t)(2 * r + side) * 3 + k], k, tris, ids, cb);
                box[3 + k] = key_value(K.kmax[(size_t)(2 * r + side) * 3 + k], k, tris, ids, cb);
            }
            if (N.count[c] >= kLeafLimit) {
                N.axis[c] = longest_axis(box);
                N.state[c] = 1;
                next_active[atomicAdd(next_count, 1)] = c;
            }
        }
        // lmax = 0.001 + max of the left vertices, rmin = -0.001 + min of the right ones (src/BIH.hs:93-96):
        // the same folds as the children's boxes.
        N.lmax[node] = 0.001f + N.box[(size_t)(base + 2 * r) * 6 + 3 + ax];
        N.rmin[node] = (-0.001f) + N.box[(size_t)(base + 2 * r + 1) * 6 + ax];
    }
    N.state[node] = 0;
}

__global__ void __launch_bounds__(kB) k_gather(int32_t T, const int32_t* __restrict__ ids, const float* __restrict__ tris, float* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * kB + threadIdx.x;
    if (g >= (int64_t)T * kTriFloats) return;
    const int32_t i = (int32_t)(g / kTriFloats), k = (int32_t)(g % kTriFloats);
    out[g] = tris[(size_t)ids[i] * kTriFloats + k];
}

// One allocation carved into 256-byte aligned arrays (two passes: measure, then place).
struct Arena {
    char* base = nullptr; size_t used = 0;
    ~Arena() { if (base) (void)hipFree(base); }
    template <typename T> void take(T** p, size_t n) {
        *p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    }
};

inline unsigned grid_for(int64_t n, int per_block) { return (unsigned)std::max<int64_t>(1, (n + per_block - 1) / per_block); }

// out = exclusive scan of in[0..n); `tmp` holds the tile-sum pyramid
int exclusive_scan(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* tmp) {
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    if (nb <= 1) {
        k_scan_apply<<<1, kB>>>(in, nullptr, out, n);
        return 0;
    }
    k_scan_sums<<<(unsigned)nb, kB>>>(in, tmp, n);
    if (exclusive_scan(tmp, tmp, nb, tmp + nb)) return 1;        // in place: a thread reads its items before it writes them
    k_scan_apply<<<(unsigned)nb, kB>>>(in, tmp, out, n);
    return 0;
}

}  // namespace

extern "C" int sq_bih_build_device(const sq_mesh* mesh, int32_t device, sq_bih** outp) {
    if (!mesh || !outp) return sq_set_error("null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return sq_set_error("no HIP device available (sq_bih_build is the host build)");
    if (device < 0 || device >= ndev) return sq_set_error("device %d outside 0..%d", device, ndev - 1);
    const size_t Tn = mesh->tris.size();
    if (Tn > 700000000u) return sq_set_error("mesh has %zu triangles; the device build handles up to 7e8", Tn);
    const int32_t T = (int32_t)Tn;

    sq_bih* b = new sq_bih;
    b->mats = mesh->mats;
    if (T == 0) {                                                 // a single empty leaf under a zero box, as in the host build
        std::memset(&b->root, 0, sizeof b->root);
        b->nodes.push_back(sq_node{ 3, 0.0f, 0.0f, 0 });
        b->height = 1; b->leaves = 1; b->longest = 0;
        *outp = b;
        return 0;
    }
    struct Guard { sq_bih* p; ~Guard() { delete p; } } guard{ b };
    SQ_HIP(hipSetDevice(device));

    const int32_t max_active = T / kLeafLimit + 1;               // disjoint ranges of at least 15 triangles
    const int64_t cap = 2 * (int64_t)T + (int64_t)T / 4 + 16;    // leaves <= T + T/15, branches = leaves - 1
    Arena D;
    float *tris, *cen, *out_tris; int32_t *ids[2], *node_of[2], *active[2], *counter; uint32_t *flag, *pos, *scan_tmp; int* bad;
    Nodes N; Keys K;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) { const size_t total = D.used; SQ_HIP(hipMalloc((void**)&D.base, total)); D.used = 0; }
        D.take(&tris, (size_t)T * kTriFloats); D.take(&out_tris, (size_t)T * kTriFloats); D.take(&cen, (size_t)T);
        for (int k = 0; k < 2; ++k) { D.take(&ids[k], (size_t)T); D.take(&node_of[k], (size_t)T); D.take(&active[k], (size_t)max_active); }
        D.take(&flag, (size_t)T); D.take(&pos, (size_t)T); D.take(&scan_tmp, (size_t)T / kScanTile + 4096);
        D.take(&counter, 1); D.take(&bad, 1);
        D.take(&N.begin, (size_t)cap); D.take(&N.count, (size_t)cap); D.take(&N.axis, (size_t)cap); D.take(&N.left, (size_t)cap);
        D.take(&N.lmax, (size_t)cap); D.take(&N.rmin, (size_t)cap); D.take(&N.plane, (size_t)cap); D.take(&N.box, (size_t)cap * 6);
        D.take(&N.state, (size_t)cap);
        D.take(&K.kmin, (size_t)max_active * 6); D.take(&K.kmax, (size_t)max_active * 6);
    }

    static_assert(sizeof(sq_tri) == kTriFloats * sizeof(float), "sq_tri layout");
    SQ_HIP(hipMemcpy(tris, mesh->tris.data(), (size_t)T * sizeof(sq_tri), hipMemcpyHostToDevice));
    SQ_HIP(hipMemset(bad, 0, sizeof(int)));
    SQ_HIP(hipMemset(K.kmin, 0xff, 3 * sizeof(u64)));
    SQ_HIP(hipMemset(K.kmax, 0, 3 * sizeof(u64)));

    const unsigned egrid = grid_for(T, kB);
    k_init<<<egrid, kB>>>(T, tris, ids[0], node_of[0], K, bad);
    k_root<<<1, 1>>>(T, tris, ids[0], N, K, active[0], counter);
    int32_t n_active = 0; int h_bad = 0;
    SQ_HIP(hipMemcpy(&n_active, counter, sizeof n_active, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(&h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost));
    if (h_bad) return sq_set_error("the device BIH build needs finite vertex coordinates (use sq_bih_build)");

    int64_t n_nodes = 1;
    int cur = 0;
    while (n_active > 0) {
        if (n_nodes + 2 * (int64_t)n_active > cap) return sq_set_error("internal: node table overflow");
        const int32_t base = (int32_t)n_nodes;
        const unsigned agrid = grid_for(n_active, kB);
        k_centroid<<<egrid, kB>>>(T, ids[cur], node_of[cur], N, tris, cen);
        k_plane<<<(unsigned)n_active, 64>>>(active[cur], N, cen);
        k_flag<<<egrid, kB>>>(T, node_of[cur], N, cen, flag);
        if (exclusive_scan(flag, pos, T, scan_tmp)) return 1;
        k_split<<<agrid, kB>>>(active[cur], n_active, base, N, flag, pos, K);
        k_scatter<<<egrid, kB>>>(T, base, ids[cur], node_of[cur], ids[cur ^ 1], node_of[cur ^ 1], N, flag, pos, tris, K);
        SQ_HIP(hipMemsetAsync(counter, 0, sizeof(int32_t), 0));
        k_finish<<<agrid, kB>>>(active[cur], n_active, base, N, ids[cur ^ 1], tris, K, active[cur ^ 1], counter);
        n_nodes += 2 * (int64_t)n_active;
        SQ_HIP(hipMemcpy(&n_active, counter, sizeof n_active, hipMemcpyDeviceToHost));
        cur ^= 1;
    }
    k_gather<<<grid_for((int64_t)T * kTriFloats, kB), kB>>>(T, ids[cur], tris, out_tris);
    SQ_HIP(hipGetLastError());

    // Download, then number the nodes in pre-order (BIH.flatten, src/BIH.hs:50-52).
    const size_t nn = (size_t)n_nodes;
    std::vector<int32_t> begin(nn), count(nn), axis(nn), left(nn);
    std::vector<float> lmax(nn), rmin(nn);
    SQ_HIP(hipMemcpy(begin.data(), N.begin, nn * 4, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(count.data(), N.count, nn * 4, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(axis.data(), N.axis, nn * 4, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(left.data(), N.left, nn * 4, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(lmax.data(), N.lmax, nn * 4, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(rmin.data(), N.rmin, nn * 4, hipMemcpyDeviceToHost));
    float rootbox[6];
    SQ_HIP(hipMemcpy(rootbox, N.box, sizeof rootbox, hipMemcpyDeviceToHost));
    for (int c = 0; c < 3; ++c) { b->root.lo[c] = rootbox[c]; b->root.hi[c] = rootbox[3 + c]; }
    b->tris.resize((size_t)T);
    SQ_HIP(hipMemcpy(b->tris.data(), out_tris, (size_t)T * sizeof(sq_tri), hipMemcpyDeviceToHost));

    b->nodes.reserve(nn);
    struct Item { int32_t node, parent_slot, depth; };
    std::vector<Item> stack;
    stack.push_back({ 0, -1, 1 });
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const int32_t at = (int32_t)b->nodes.size();
        if (it.parent_slot >= 0) b->nodes[(size_t)it.parent_slot].link = at;      // the right child's pre-order index
        const size_t v = (size_t)it.node;
        if (left[v] < 0) {
            b->nodes.push_back(sq_node{ 3 | (count[v] << 2), 0.0f, 0.0f, begin[v] });
            b->leaves++;
            if (count[v] > b->longest) b->longest = count[v];
            if (it.depth > b->height) b->height = it.depth;
        } else {
            b->nodes.push_back(sq_node{ axis[v], lmax[v], rmin[v], -1 });
            stack.push_back({ left[v] + 1, at, it.depth + 1 });
            stack.push_back({ left[v], -1, it.depth + 1 });
        }
    }
    guard.p = nullptr;
    *outp = b;
    return 0;
}
