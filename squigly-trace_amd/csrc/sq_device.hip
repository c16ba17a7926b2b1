// sq_device.hip — gfx950 kernels and the render half of the C-ABI (include/squigly_hip.h).
//
// What runs on the GPU (all hand-written for CDNA4, wave64):
//   camera-ray generation            src/Lib.hs:107-114
//   BIH traversal + Moller-Trumbore  src/BIH.hs:101-141, src/Geometry.hs:117-177
//   bounce / scatter / mirror, RNG   src/Lib.hs:127-137,155-198 (+ tf-random's Threefish block)
//   ordered per-pixel accumulation   src/Lib.hs:85-88
//   atan tonemap                     src/Lib.hs:93-104
//
// The traversal is the reference's recursion with its call stack made explicit.  A stack frame
// is either FAR(branch) — "the far child of this branch is still to be visited" — or
// COMBINE(hit) — "the near child produced this hit, combine it with the far child's result".
// The register R plays the role of the value returned by the most recently finished call, so the
// early exit `isClose` (src/BIH.hs:114,121-123) and the tie-breaks of minimumBy (src/BIH.hs:109,115)
// see exactly the values the Haskell sees.  No fast-math, no FMA contraction.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/squigly_hip.h"
#include "sq_error.h"
#include "sq_math.h"

using sq::f3;

#define SQ_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return sq_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ----------------------------------------------------------------------------------------------
// Device-resident scene layout (HBM, read-only during a render)
// ----------------------------------------------------------------------------------------------
constexpr uint32_t kLeafBit = 0x80000000u;     // child reference: leaf index | kLeafBit, or branch index
constexpr uint32_t kCombineBit = 0x80000000u;  // stack word: triangle index | kCombineBit (then a second word: t)
constexpr int kBlock = 256;

struct DevBranch {          // 48 B, three 16-byte quads
    float lo[3]; float lmax;    // traversal box of THIS branch (root bounds clipped along the path, src/BIH.hs:130-141)
    float hi[3]; float rmin;
    int32_t axis; uint32_t left, right; int32_t pad;
};
struct DevLeaf { int32_t first, count; };
struct DevTri {             // 48 B: v0 | e1 = v1 - v0 | e2 = v2 - v0 (same rounding as src/Geometry.hs:130-131)
    float v0[3]; int32_t mat;
    float e1[3]; float pad1;
    float e2[3]; float pad2;
};
struct DevMat { float reflective, sr, sg, sb, emissive, er, eg, eb; };   // 32 B

struct SceneView {
    const float4* branches;   // 3 per branch
    const int2* leaves;
    const float4* tris;       // 3 per triangle
    const float4* mats;       // 2 per material
    float root_lo[3], root_hi[3];
    uint32_t root_ref;
    int32_t n_branches, n_leaves, n_tris, n_mats;
    int32_t stack_words;      // per-lane LDS stack capacity in 32-bit words
};

struct RenderParams {
    SceneView sc;
    float cam_pos[3]; float cam_rot[9];
    int32_t samples, w, h, cast;
    int32_t row_block, shard, n_shards, local_rows;
    float* out_avg; uint8_t* out_rgb;
};

// ----------------------------------------------------------------------------------------------
// Device code
// ----------------------------------------------------------------------------------------------
struct Hit { float t, dist; int32_t tri; };   // tri < 0 : Nothing

// intersectsBB (src/Geometry.hs:166-177) on precomputed df = 1/dir
__device__ __forceinline__ bool slab(float lx, float ly, float lz, float hx, float hy, float hz, f3 o, f3 df) {
    float t1 = (lx - o.x) * df.x, t2 = (hx - o.x) * df.x;
    float t3 = (ly - o.y) * df.y, t4 = (hy - o.y) * df.y;
    float t5 = (lz - o.z) * df.z, t6 = (hz - o.z) * df.z;
    float tmin = sq::hmax(sq::hmax(sq::hmin(t1, t2), sq::hmin(t3, t4)), sq::hmin(t5, t6));
    float tmax = sq::hmin(sq::hmin(sq::hmax(t1, t2), sq::hmax(t3, t4)), sq::hmax(t5, t6));
    return tmax > 0 && tmin < tmax;
}

// mollerTrumbore (src/Geometry.hs:117-142) on (v0, e1, e2)
__device__ __forceinline__ bool moller_trumbore(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& t_out, float& dist_out) {
    const float eps = 0.0001f;
    f3 h = sq::cross(d, e2);
    float a = sq::dot(e1, h);
    if (a > -eps && a < eps) return false;
    float f = 1.0f / a;
    f3 s = o - v0;
    float u = f * sq::dot(s, h);
    if (u < 0 || u > 1) return false;
    f3 q = sq::cross(s, e1);
    float v = f * sq::dot(d, q);
    if (v < 0 || u + v > 1) return false;
    float t = f * sq::dot(e2, q);
    if (!(t > eps)) return false;
    f3 p = o + sq::scale(t, d);
    t_out = t;
    dist_out = sq::norm(p - o);
    return true;
}

// intersectBIH (src/BIH.hs:101-141).  `stk` points at this lane's word 0; consecutive words of a
// lane are `stride` words apart (lane-minor layout: conflict-free ds_read_b32/ds_write_b32).
__device__ __forceinline__ Hit trace(const SceneView& S, f3 o, f3 d, uint32_t* stk, int stride) {
    Hit R; R.t = 0; R.dist = 0; R.tri = -1;
    const f3 df = sq::mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t cur = S.root_ref;
    int sp = 0;
    enum { DESCEND = 0, LEAF = 1, UNWIND = 2, DONE = 3 };
    int mode = (cur & kLeafBit) ? LEAF : DESCEND;
    if (mode == DESCEND &&
        !slab(S.root_lo[0], S.root_lo[1], S.root_lo[2], S.root_hi[0], S.root_hi[1], S.root_hi[2], o, df))
        mode = DONE;                                                    // src/BIH.hs:112 at the root
    while (mode != DONE) {
        while (mode == DESCEND) {                                       // Branch equation, src/BIH.hs:111-141
            const float4 q0 = S.branches[3 * cur], q1 = S.branches[3 * cur + 1], q2 = S.branches[3 * cur + 2];
            const int ax = __float_as_int(q2.x);
            const uint32_t left = __float_as_uint(q2.y), right = __float_as_uint(q2.z);
            const float lmax = q0.w, rmin = q1.w;
            // left = bbox with hi[ax] := lmax ; right = bbox with lo[ax] := rmin   (src/BIH.hs:130-141)
            const bool iL = slab(q0.x, q0.y, q0.z, ax == 0 ? lmax : q1.x, ax == 1 ? lmax : q1.y, ax == 2 ? lmax : q1.z, o, df);
            const bool iR = slab(ax == 0 ? rmin : q0.x, ax == 1 ? rmin : q0.y, ax == 2 ? rmin : q0.z, q1.x, q1.y, q1.z, o, df);
            if (iL && iR) {
                const bool l2r = sq::axis_of(d, ax) > 0;                // src/BIH.hs:127
                stk[sp * stride] = cur; ++sp;                           // FAR(cur)
                cur = l2r ? left : right;
            } else if (iL) cur = left;
            else if (iR) cur = right;
            else { R.tri = -1; mode = UNWIND; break; }                  // src/BIH.hs:119
            if (cur & kLeafBit) mode = LEAF;
        }
        if (mode == LEAF) {                                             // Leaf equation, src/BIH.hs:105-109
            const int2 lf = S.leaves[cur & ~kLeafBit];
            R.tri = -1;
            for (int i = lf.x; i < lf.x + lf.y; ++i) {
                const float4 a = S.tris[3 * i], b = S.tris[3 * i + 1], c = S.tris[3 * i + 2];
                float t, dist;
                if (moller_trumbore(o, d, sq::mk(a.x, a.y, a.z), sq::mk(b.x, b.y, b.z), sq::mk(c.x, c.y, c.z), t, dist)) {
                    // minimumBy (comparing dist): replace only when compare best new == GT
                    if (R.tri < 0 || sq::cmp_gt(R.dist, dist)) { R.t = t; R.dist = dist; R.tri = i; }
                }
            }
            mode = UNWIND;
        }
        while (mode == UNWIND) {
            if (sp == 0) { mode = DONE; break; }
            --sp;
            const uint32_t e = stk[sp * stride];
            if (e & kCombineBit) {                                      // minimumByMay over [near, far], src/BIH.hs:115,120
                --sp;
                const float nt = __uint_as_float(stk[sp * stride]);
                const int32_t ntri = (int32_t)(e & ~kCombineBit);
                const f3 np = o + sq::scale(nt, d);
                const float ndist = sq::norm(np - o);
                if (R.tri < 0 || !sq::cmp_gt(ndist, R.dist)) { R.t = nt; R.dist = ndist; R.tri = ntri; }
            } else {                                                    // back in branch e, near child returned R
                const float4 q0 = S.branches[3 * e], q1 = S.branches[3 * e + 1], q2 = S.branches[3 * e + 2];
                const int ax = __float_as_int(q2.x);
                const bool l2r = sq::axis_of(d, ax) > 0;
                if (R.tri >= 0) {
                    const float p = sq::axis_of(o, ax) + R.t * sq::axis_of(d, ax);   // projectToAxis ax (intersectPoint v)
                    const bool close = l2r ? (p < q1.w) : (p > q0.w);  // src/BIH.hs:121-123
                    if (close) continue;                                // src/BIH.hs:114: return near
                    stk[sp * stride] = __float_as_uint(R.t); ++sp;      // COMBINE(R)
                    stk[sp * stride] = (uint32_t)R.tri | kCombineBit; ++sp;
                }
                cur = l2r ? __float_as_uint(q2.z) : __float_as_uint(q2.y);          // far child
                mode = (cur & kLeafBit) ? LEAF : DESCEND;
            }
        }
    }
    return R;
}

struct Surface {            // what shading needs from a hit triangle
    f3 n;                   // normal = e1 x e2, un-normalised (src/Geometry.hs:79-80)
    float reflective; f3 surf; f3 emit;   // emit = emissive *^ emitColor (src/Lib.hs:136)
};
__device__ __forceinline__ Surface surface_of(const SceneView& S, int tri) {
    const float4 a = S.tris[3 * tri], b = S.tris[3 * tri + 1], c = S.tris[3 * tri + 2];
    const int m = __float_as_int(a.w);
    const float4 m0 = S.mats[2 * m], m1 = S.mats[2 * m + 1];
    Surface s;
    s.n = sq::cross(sq::mk(b.x, b.y, b.z), sq::mk(c.x, c.y, c.z));
    s.reflective = m0.x; s.surf = sq::mk(m0.y, m0.z, m0.w);
    s.emit = sq::scale(m1.x, sq::mk(m1.y, m1.z, m1.w));
    return s;
}

// bounceRay (src/Lib.hs:155-181): x and u are the SAME draw nu; v is the next draw nv.
__device__ __forceinline__ f3 bounce_dir(f3 d, const Surface& s, uint32_t nu, uint32_t nv) {
    const float x = sq::unit_float(nu);
    if (s.reflective < x) {                                             // scatterRay, src/Lib.hs:166-172
        const float u = x, v = sq::unit_float(nv);
        const float th = 2 * sq::kPi * u;
        const float ph = sq::facos(2 * v - 1);
        float sth, cth, sph, cph;
        sq::fsincos(th, sth, cth); sq::fsincos(ph, sph, cph);
        const f3 nd = sq::mk(cth * sph, sth * sph, cph);                // randomVector, src/Lib.hs:192-198
        const float old_ = sq::hsignum(sq::dot(d, s.n)), new_ = sq::hsignum(sq::dot(nd, s.n));
        return (old_ == new_) ? -nd : nd;
    }
    const f3 dn = sq::normalize(s.n);                                   // reflectRay, src/Lib.hs:176-181
    return d - sq::scale(2 * sq::dot(dn, d), dn);
}

// rgbFloatToPixelRGB (src/Lib.hs:93-104).  floor :: Float -> Word8 wraps mod 256 and maps NaN/Inf to 0.
__device__ __forceinline__ uint8_t to_word8(float f) {
    if (!(f == f) || f == __builtin_inff() || f == -__builtin_inff()) return 0;
    const double fl = __builtin_floor((double)f);
    double md = fl - 256.0 * __builtin_floor(fl / 256.0);
    return (uint8_t)(int)md;
}
__device__ __forceinline__ void tonemap(f3 c, uint8_t* out) {
    const float mx = sq::hmax(sq::hmax(c.x, c.y), c.z), mn = sq::hmin(sq::hmin(c.x, c.y), c.z);
    const float lightness = 0.5f * (mx + mn);
    const float intensity = sq::fatan(lightness) / (sq::kPi / 2);
    const f3 s = sq::scale(intensity / mx, c);
    out[0] = to_word8(s.x * 255); out[1] = to_word8(s.y * 255); out[2] = to_word8(s.z * 255);
}

// makeRay (src/Lib.hs:107-114) + rotVert (src/Geometry.hs:104-107)
__device__ __forceinline__ f3 primary_dir(const RenderParams& P, int y, int x) {
    const float ww = (float)P.w, hh = (float)P.h;
    const float xoffs = ((float)x - (ww / 2)) / ww;
    const float yoffs = ((hh / 2) - (float)y) / hh;
    const float v[3] = { 1.0f, xoffs, yoffs };
    float o[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { float r = 0.0f; for (int k = 0; k < 3; ++k) r = v[k] * P.cam_rot[3 * k + j] + r; o[j] = r; }
    return sq::mk(o[0], o[1], o[2]);
}

// One lane per pixel.  The primary ray is traced once per pixel: every sample of a pixel shoots the
// same primary ray (src/Lib.hs:81-87), so its intersection is the same value each time.
__global__ void __launch_bounds__(kBlock) sq_render_pixels(const RenderParams P) {
    extern __shared__ uint32_t lds_stack[];
    const int tid = threadIdx.x;
    uint32_t* stk = lds_stack + tid;
    const long long pix = (long long)blockIdx.x * kBlock + tid;
    const long long total = (long long)P.local_rows * P.h;
    if (pix >= total) return;
    const int j = (int)(pix / P.h), x = (int)(pix % P.h);
    const int blk = j / P.row_block;
    const int y = (blk * P.n_shards + P.shard) * P.row_block + (j - blk * P.row_block);
    const SceneView& S = P.sc;
    const f3 o0 = sq::mk(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    const f3 d0 = primary_dir(P, y, x);
    const int n = P.samples;
    f3 sum = sq::mk(0, 0, 0);                                           // sum = foldl (+) 0
    const Hit h0 = trace(S, o0, d0, stk, kBlock);
    if (h0.tri >= 0) {
        const Surface s0 = surface_of(S, h0.tri);
        const f3 p0 = o0 + sq::scale(h0.t, d0);
        if (P.cast) {                                                   // raycast, src/Lib.hs:141-151
            const f3 light = sq::mk(0, 3, -1);
            const float dl = sq::norm(p0 - light);
            const Hit sh = trace(S, p0, light - p0, stk, kBlock);
            f3 c = sq::mk(0, 0, 0);
            if (!(sh.tri >= 0 && !(sh.dist > dl))) {
                const float4 a = S.tris[3 * h0.tri];
                const float4 m0 = S.mats[2 * __float_as_int(a.w)];
                c = sq::scale(2 / dl, sq::mk(m0.y, m0.z, m0.w));
            }
            for (int k = 0; k < n; ++k) sum = sum + c;
        } else {
            const long long rix = (long long)n * ((long long)x + (long long)y * (long long)P.w);   // src/Lib.hs:85
#pragma unroll 1
            for (int k = 0; k < n; ++k) {                               // raytrace gen scene ray 0, src/Lib.hs:127-137
                uint32_t n0, n1, n2;
                sq::tfgen3(rix + k, n0, n1, n2);
                f3 L1 = sq::mk(0, 0, 0);
                const f3 d1 = bounce_dir(d0, s0, n0, n1);
                const Hit h1 = trace(S, p0, d1, stk, kBlock);
                if (h1.tri >= 0) {
                    const Surface s1 = surface_of(S, h1.tri);
                    const f3 p1 = p0 + sq::scale(h1.t, d1);
                    const f3 d2 = bounce_dir(d1, s1, n1, n2);
                    const Hit h2 = trace(S, p1, d2, stk, kBlock);
                    f3 L2 = sq::mk(0, 0, 0);
                    if (h2.tri >= 0) {
                        const Surface s2 = surface_of(S, h2.tri);
                        L2 = s2.surf * sq::mk(0, 0, 0) + s2.emit;
                    }
                    L1 = s1.surf * L2 + s1.emit;
                }
                const f3 L0 = s0.surf * L1 + s0.emit;
                sum = sum + L0;
            }
        }
    } else {
        for (int k = 0; k < n; ++k) sum = sum + sq::mk(0, 0, 0);
    }
    const f3 avg = sq::scale(1 / (float)n, sum);                        // src/Lib.hs:88
    if (P.out_avg) { float* o = P.out_avg + pix * 3; o[0] = avg.x; o[1] = avg.y; o[2] = avg.z; }
    if (P.out_rgb) tonemap(avg, P.out_rgb + pix * 3);
}

// ----------------------------------------------------------------------------------------------
// Host: scene validation + upload
// ----------------------------------------------------------------------------------------------
struct sq_device_scene {
    int device = 0;
    SceneView view{};
    void *d_branches = nullptr, *d_leaves = nullptr, *d_tris = nullptr, *d_mats = nullptr;
    int height = 0;
    // timing of the dominant kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0; int64_t launches = 0;
    int64_t opt_timing = 1;
};

namespace {

// Walks the pre-order array once; checks that it is a well-formed tree, that every index is in
// range, and computes the height.  A malformed tree would otherwise fault on the GPU.
int validate_tree(const sq_scene& sc, int& height, std::vector<int32_t>& depth_of) {
    const int32_t n = sc.n_nodes;
    if (n < 1) return sq_set_error("scene has no nodes");
    depth_of.assign((size_t)n, 0);
    struct Frame { int32_t node, stage; };
    std::vector<Frame> st;
    st.push_back({ 0, 0 });
    int32_t next = 0;      // next unvisited pre-order index
    depth_of[0] = 1;
    height = 0;
    while (!st.empty()) {
        Frame& f = st.back();
        const sq_node& nd = sc.nodes[f.node];
        const int kind = nd.kind & 3;
        if (f.stage == 0) {
            if (f.node != next) return sq_set_error("node %d is not in pre-order position (expected %d)", f.node, next);
            ++next;
            const int dep = depth_of[(size_t)f.node];
            if (dep > height) height = dep;
            if (kind == 3) {
                const int64_t cnt = nd.kind >> 2, first = nd.link;
                if (cnt < 0 || first < 0 || first + cnt > sc.n_tris) return sq_set_error("leaf %d has triangle range [%lld,+%lld) outside 0..%d", f.node, (long long)first, (long long)cnt, sc.n_tris);
                st.pop_back();
                continue;
            }
            if ((nd.kind >> 2) != 0) return sq_set_error("branch %d has stray bits in kind", f.node);
            if (f.node + 1 >= n) return sq_set_error("branch %d has no left child", f.node);
            f.stage = 1;
            depth_of[(size_t)f.node + 1] = dep + 1;
            st.push_back({ f.node + 1, 0 });
        } else if (f.stage == 1) {
            if (nd.link != next) return sq_set_error("branch %d: right child link %d, expected %d", f.node, nd.link, next);
            if (nd.link >= n) return sq_set_error("branch %d: right child %d out of range", f.node, nd.link);
            f.stage = 2;
            depth_of[(size_t)nd.link] = depth_of[(size_t)f.node] + 1;
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

    // Re-pack: branches and leaves get their own dense tables; each branch carries its traversal box.
    const int32_t n = sc->n_nodes;
    std::vector<uint32_t> ref((size_t)n);
    int32_t nb = 0, nl = 0;
    for (int32_t i = 0; i < n; ++i) ref[(size_t)i] = ((sc->nodes[i].kind & 3) == 3) ? ((uint32_t)nl++ | kLeafBit) : (uint32_t)nb++;
    std::vector<DevBranch> br((size_t)nb);
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
        d.lmax = nd.lmax; d.rmin = nd.rmin; d.axis = kind; d.pad = 0;
        d.left = ref[(size_t)i + 1]; d.right = ref[(size_t)nd.link];
        sq_bounds l = b, r = b;                          // src/BIH.hs:130-141
        l.hi[kind] = nd.lmax; r.lo[kind] = nd.rmin;
        box[(size_t)i + 1] = l; box[(size_t)nd.link] = r;
    }
    std::vector<DevTri> tr((size_t)sc->n_tris);
    for (int32_t i = 0; i < sc->n_tris; ++i) {
        const sq_tri& t = sc->tris[i]; DevTri& d = tr[(size_t)i];
        for (int c = 0; c < 3; ++c) { d.v0[c] = t.v0[c]; d.e1[c] = t.v1[c] - t.v0[c]; d.e2[c] = t.v2[c] - t.v0[c]; }
        d.mat = t.mat; d.pad1 = d.pad2 = 0;
    }
    std::vector<DevMat> mt((size_t)sc->n_mats);
    for (int32_t i = 0; i < sc->n_mats; ++i) {
        const sq_material& m = sc->mats[i];
        mt[(size_t)i] = { m.reflective, m.surf[0], m.surf[1], m.surf[2], m.emissive, m.emit[0], m.emit[1], m.emit[2] };
    }
    SQ_HIP(hipSetDevice(device));
    sq_device_scene* s = new sq_device_scene;
    s->device = device; s->height = height;
    auto up = [&](void** dst, const void* src, size_t bytes) -> int {
        if (hipMalloc(dst, bytes ? bytes : 16) != hipSuccess) return sq_set_error("hipMalloc(%zu) failed", bytes);
        if (bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return sq_set_error("hipMemcpy H2D failed");
        return 0;
    };
    if (up(&s->d_branches, br.data(), br.size() * sizeof(DevBranch)) || up(&s->d_leaves, lf.data(), lf.size() * sizeof(DevLeaf)) ||
        up(&s->d_tris, tr.data(), tr.size() * sizeof(DevTri)) || up(&s->d_mats, mt.data(), mt.size() * sizeof(DevMat))) {
        sq_scene_free(s);
        return 1;
    }
    SceneView& v = s->view;
    v.branches = (const float4*)s->d_branches; v.leaves = (const int2*)s->d_leaves;
    v.tris = (const float4*)s->d_tris; v.mats = (const float4*)s->d_mats;
    for (int c = 0; c < 3; ++c) { v.root_lo[c] = sc->root.lo[c]; v.root_hi[c] = sc->root.hi[c]; }
    v.root_ref = ref[0];
    v.n_branches = nb; v.n_leaves = nl; v.n_tris = sc->n_tris; v.n_mats = sc->n_mats;
    v.stack_words = 2 * height + 2;        // one frame per level, a COMBINE frame is two words
    *out = s;
    return 0;
}

extern "C" void sq_scene_free(sq_device_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (auto& p : s->pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    (void)hipFree(s->d_branches); (void)hipFree(s->d_leaves); (void)hipFree(s->d_tris); (void)hipFree(s->d_mats);
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

extern "C" int sq_render_rows_device(sq_device_scene* s, const sq_camera* cam, int32_t samples, int32_t w, int32_t h,
                                     int32_t cast, sq_shard sh, float* d_avg, uint8_t* d_rgb, void* hip_stream) {
    if (!s || !cam) return sq_set_error("null argument");
    if (samples < 1 || w < 1 || h < 1) return sq_set_error("samples, width and height must be positive (got %d, %d, %d)", samples, w, h);
    const int32_t rows = sq_shard_rows(w, sh);
    if (rows < 0) return sq_set_error("bad shard {row_block=%d, shard=%d, n_shards=%d}", sh.row_block, sh.shard, sh.n_shards);
    if (!d_avg && !d_rgb) return sq_set_error("no output buffer");
    if (rows == 0) return 0;
    SQ_HIP(hipSetDevice(s->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    RenderParams P{};
    P.sc = s->view;
    std::memcpy(P.cam_pos, cam->pos, sizeof P.cam_pos);
    std::memcpy(P.cam_rot, cam->rot, sizeof P.cam_rot);
    P.samples = samples; P.w = w; P.h = h; P.cast = cast ? 1 : 0;
    P.row_block = sh.row_block; P.shard = sh.shard; P.n_shards = sh.n_shards; P.local_rows = rows;
    P.out_avg = d_avg; P.out_rgb = d_rgb;
    const long long total = (long long)rows * h;
    const long long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return sq_set_error("image too large for one launch");
    const size_t lds = (size_t)kBlock * (size_t)s->view.stack_words * sizeof(uint32_t);
    if (lds > 160 * 1024) return sq_set_error("BIH height %d needs %zu B of LDS stack per workgroup (max 163840)", s->height, lds);
    if (lds > 64 * 1024)
        SQ_HIP(hipFuncSetAttribute((const void*)sq_render_pixels, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->opt_timing) {
        SQ_HIP(hipEventCreate(&e0)); SQ_HIP(hipEventCreate(&e1));
        SQ_HIP(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(sq_render_pixels, dim3((unsigned)blocks), dim3(kBlock), lds, stream, P);
    SQ_HIP(hipGetLastError());
    if (s->opt_timing) {
        SQ_HIP(hipEventRecord(e1, stream));
        s->pending.emplace_back(e0, e1);
    }
    return 0;
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
    if (name) *name = "sq_render_pixels";
    return 0;
}
extern "C" void sq_kernel_timing_reset(sq_device_scene* s) {
    if (!s) return;
    for (auto& p : s->pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    s->pending.clear(); s->total_ms = 0; s->launches = 0;
}
extern "C" int sq_set_option(sq_device_scene* s, const char* key, int64_t value) {
    if (!s || !key) return sq_set_error("null argument");
    if (!std::strcmp(key, "timing")) { s->opt_timing = value; return 0; }
    return sq_set_error("unknown option '%s'", key);
}

// ---- one-shot entry points (the drop-in for src/Lib.hs:73-74) ----
namespace {
int render_oneshot(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h, int32_t cast,
                   float* out_avg, uint8_t* out_rgb) {
    if (!scene || !cam || (!out_avg && !out_rgb)) return sq_set_error("null argument");
    if (samples < 1 || w < 1 || h < 1) return sq_set_error("samples, width and height must be positive (got %d, %d, %d)", samples, w, h);
    sq_device_scene* s = nullptr;
    if (sq_scene_upload(scene, 0, &s)) return 1;
    const size_t npx = (size_t)w * (size_t)h * 3;
    float* d_avg = nullptr; uint8_t* d_rgb = nullptr;
    std::vector<float> h_avg; std::vector<uint8_t> h_rgb;
    int rc = 0;
    auto body = [&]() -> int {
        if (out_avg) SQ_HIP(hipMalloc((void**)&d_avg, npx * sizeof(float)));
        if (out_rgb) SQ_HIP(hipMalloc((void**)&d_rgb, npx));
        sq_shard whole = { w, 0, 1 };
        if (sq_render_rows_device(s, cam, samples, w, h, cast, whole, d_avg, d_rgb, nullptr)) return 1;
        SQ_HIP(hipDeviceSynchronize());
        // stage through private buffers so nothing is written to the caller's memory on failure
        if (out_avg) { h_avg.resize(npx); SQ_HIP(hipMemcpy(h_avg.data(), d_avg, npx * sizeof(float), hipMemcpyDeviceToHost)); }
        if (out_rgb) { h_rgb.resize(npx); SQ_HIP(hipMemcpy(h_rgb.data(), d_rgb, npx, hipMemcpyDeviceToHost)); }
        return 0;
    };
    rc = body();
    (void)hipFree(d_avg); (void)hipFree(d_rgb);
    sq_scene_free(s);
    if (rc) return rc;
    if (out_avg) std::memcpy(out_avg, h_avg.data(), npx * sizeof(float));
    if (out_rgb) std::memcpy(out_rgb, h_rgb.data(), npx);
    return 0;
}
}  // namespace

extern "C" int sq_render_rgb8(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h, int32_t cast, uint8_t* out) {
    return render_oneshot(scene, cam, samples, w, h, cast, nullptr, out);
}
extern "C" int sq_render_f32(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h, int32_t cast, float* out_avg) {
    return render_oneshot(scene, cam, samples, w, h, cast, out_avg, nullptr);
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
        default: break;
    }
}
extern "C" int sq_debug_eval(int32_t device, int32_t op, const void* a, const void* b, int64_t n, void* out) {
    if (!a || !out || n < 0 || op < 0 || op > SQ_OP_TONEMAP) return sq_set_error("bad argument");
    if (op == SQ_OP_DIV && !b) return sq_set_error("SQ_OP_DIV needs b");
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return sq_set_error("no such HIP device %d", device);
    SQ_HIP(hipSetDevice(device));
    const size_t in_sz = (size_t)n * (op == SQ_OP_TFGEN3 ? 8 : op == SQ_OP_TONEMAP ? 12 : 4);
    const size_t out_sz = (size_t)n * (op == SQ_OP_TFGEN3 ? 12 : op == SQ_OP_TONEMAP ? 3 : 4);
    void *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = 0;
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
    rc = body();
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return rc;
}

extern "C" int32_t sq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int32_t sq_abi_version(void) { return SQ_ABI_VERSION; }
extern "C" const char* sq_last_error(void) { return sq_error_buffer(); }
