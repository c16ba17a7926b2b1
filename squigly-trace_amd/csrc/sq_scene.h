// sq_scene.h — device-resident scene layout and the per-ray primitives shared by all kernels.
// Citations are relative to the reference repository root.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "sq_math.h"

namespace sqd {
using sq::f3;

constexpr uint32_t kLeafBit = 0x80000000u;   // child reference: leaf index | kLeafBit, or branch index

// Builtin vector types and explicit LDS (address space 3) pointers: a load through an LDS-qualified
// pointer is always a ds_read, never a flat load, and cannot be merged with a global pointer.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned short v4us __attribute__((ext_vector_type(4)));
#define SQ_LDS __attribute__((address_space(3)))
template <typename T> __device__ __forceinline__ SQ_LDS T* to_lds(void* generic) { return (SQ_LDS T*)generic; }

// HBM layout (read-only during a render).  Branches are numbered breadth-first so that the top
// of the tree is a prefix of the table (that prefix is what gets staged in LDS).
struct DevBranch {          // 48 B, three 16-byte quads
    float lo[3]; float lmax;    // traversal box of THIS branch: root bounds clipped along the path (src/BIH.hs:130-141)
    float hi[3]; float rmin;
    float lmax2, rmin2;         // the third quad alone serves a return into this branch (planes, axis, children):
    uint32_t left, right;       //   one 16-byte load instead of three.  Bits 30..29 of `left` hold the split axis.
};
struct DevLeaf { int32_t first, count; };
struct DevTri {             // 36 B: v0 | e1 = v1 - v0 | e2 = v2 - v0 (same rounding as src/Geometry.hs:130-131).
    float v0[3];            // Unpadded on purpose: the streaming kernel is bound by L2/MALL/HBM bytes on large
    float e1[3];            // scenes; the material index lives in its own array because only shading needs it.
    float e2[3];
};
struct DevMat { float reflective, sr, sg, sb, emissive, er, eg, eb; };   // 32 B
struct DevSurf {            // 48 B, three quads: everything shading needs from a hit triangle behind ONE index
    float n[3], reflective;     // normal = e1 x e2 (src/Geometry.hs:79-80) | Material.reflective
    float surf[3], pad0;        // surfColor
    float emit[3], pad1;        // emissive *^ emitColor (src/Lib.hs:136), the same fp32 products as on the device
};

struct SceneView {
    const float4* branches;   // 3 quads per branch
    const int2* leaves;
    const float* tris;        // 9 floats per triangle (DevTri)
    const int32_t* tri_mat;   // material index per triangle
    const float4* mats;       // 2 quads per material
    const float4* surfs;      // 3 quads per triangle (DevSurf): normal and material values, one dependent load instead of three
    float root_lo[3], root_hi[3];
    uint32_t root_ref;
    int32_t packed_leaves;    // 1: a leaf reference is kLeafBit | count << 24 | first (count <= 31, < 2^24 triangles), no table lookup
    int32_t n_branches, n_leaves, n_tris, n_mats;
    // Resident (LDS) form of the same scene, present when it can be encoded (16-bit vertex indices, leaves
    // of at most 31 triangles, < 2^24 triangles/branches): see "resident encoding" below.
    const float4* verts4;     // unique vertices, 16 B each
    const ushort4* trix;      // per triangle: vertex indices i0,i1,i2 and material; nullptr if not encodable
    const uint32_t* rbranch;  // 10 words per branch: lo.xyz,lmax | hi.xyz,rmin | left word, right word
    uint32_t rroot;           // root reference in resident encoding
    int32_t n_verts;
    int32_t height;           // BIH.height; a traversal never holds more than height-1 frames
    int32_t nonneg_materials; // 1 if every material component is >= +0 (enables the exact s == 0 shortcuts)
    int32_t finite_geometry;  // 1 if every vertex and box coordinate is finite (v_min/v_max slabs need NaN-free planes)
    const int32_t* emitters;  // triangles whose emission `emissive *^ emitColor` is not exactly (+0,+0,+0)
    int32_t n_emitters;       // their number, or -1 when the last-bounce shortcut is disabled (see sq_shade1)
    // Leaf culling (sq_cull_boxes, include/squigly_host.h): a ray inside these limits that misses a leaf's culling box
    // skips the leaf's triangles -- mollerTrumbore would reject them all.  cull_o2max < 0: nothing is culled.
    float cull_o2max, cull_d2min, cull_d2max;
    const float4* cull_child; // streaming form: 4 quads per branch, the culling boxes (lo, hi) of its left and of its right child; or nullptr
    const uint4* rtail;       // resident form: (lmax, rmin, left word, right word) per branch, what a return into a branch needs
    const uint4* cull_child16; // 2 quads per branch, the same boxes as binary16 pairs (x, y, z, unused), left child then right; or nullptr
    const float4* branches_m;  // streaming form: the 48-byte branch record and the 32 bytes of its children's binary16 culling boxes as ONE
                               //   packed record of 5 quads (HybridNodes::kMerged): a visit touches 1.5 lines on average instead of 2.3
    int32_t incremental_ok;    // 1: every child interval of the tree is regular or grown (kGrownLeft / kGrownRight) and the geometry is
                               // finite: the resident form may carry (tmin, tmax) down the tree (trav_descend)
};

struct Hit { float t; int32_t tri; };   // tri < 0 : Nothing.  dist is derived from t on demand (hit_dist)

// Wave ballot of a Bool.  HIP's __ballot takes an int, and the bool -> int -> (!= 0) round trip survives into the ISA as a
// v_cndmask 0/1 + v_cmp_ne pair per call (two 4-cycle VALU instructions) wherever the predicate already lives in an SGPR
// pair; the builtin takes the i1 as it is.  SQ_BALLOT_BUILTIN=0 restores __ballot (A/B).
#ifndef SQ_BALLOT_BUILTIN
#define SQ_BALLOT_BUILTIN 1
#endif
__device__ __forceinline__ unsigned long long sq_ballot(bool p) {
#if SQ_BALLOT_BUILTIN
    return __builtin_amdgcn_ballot_w64(p);
#else
    return __ballot(p);
#endif
}

// intersectsBB (src/Geometry.hs:166-177) with df = 1/dir precomputed (the reference recomputes the same value)
__device__ __forceinline__ bool slab(float lx, float ly, float lz, float hx, float hy, float hz, f3 o, f3 df) {
    const float t1 = (lx - o.x) * df.x, t2 = (hx - o.x) * df.x;
    const float t3 = (ly - o.y) * df.y, t4 = (hy - o.y) * df.y;
    const float t5 = (lz - o.z) * df.z, t6 = (hz - o.z) * df.z;
    const float tmin = sq::hmax(sq::hmax(sq::hmin(t1, t2), sq::hmin(t3, t4)), sq::hmin(t5, t6));
    const float tmax = sq::hmin(sq::hmin(sq::hmax(t1, t2), sq::hmax(t3, t4)), sq::hmax(t5, t6));
    return tmax > 0 && tmin < tmax;
}

// The same test with hardware min/max.  Haskell's min/max differ from v_min_f32/v_max_f32 only when an
// operand is NaN (and in the sign of a zero result, which no comparison below can see).  When the ray's
// origin, direction and 1/direction are all finite, (bound - o) * df is finite or +-inf, never NaN, so
// both forms return the same Bool.  `safe` rays use this form; any other ray uses slab().
// v_min_f32 / v_max_f32 / v_max3_f32 / v_min3_f32 written out: through the builtins the compiler first "canonicalises"
// operands it cannot prove quiet (six extra v_max x,x per branch step); the instructions themselves need no such help,
// and for the NaN-free operands of a safe ray they return the plain minimum / maximum.
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ bool slab_fast(float lx, float ly, float lz, float hx, float hy, float hz, f3 o, f3 df) {
    const float t1 = (lx - o.x) * df.x, t2 = (hx - o.x) * df.x;
    const float t3 = (ly - o.y) * df.y, t4 = (hy - o.y) * df.y;
    const float t5 = (lz - o.z) * df.z, t6 = (hz - o.z) * df.z;
    const float tmin = vmax3(vmin(t1, t2), vmin(t3, t4), vmin(t5, t6));
    const float tmax = vmin3(vmax(t1, t2), vmax(t3, t4), vmax(t5, t6));
    return tmax > 0 && tmin < tmax;
}
__device__ __forceinline__ bool finite3(f3 v) {
    return __builtin_isfinite(v.x) && __builtin_isfinite(v.y) && __builtin_isfinite(v.z);
}

#ifndef SQ_FAST_RCP
#define SQ_FAST_RCP 1
#endif
// mollerTrumbore (src/Geometry.hs:117-142) on (v0, e1, e2)
__device__ __forceinline__ bool moller_trumbore(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& t_out) {
    const float eps = 0.0001f;
    const f3 h = sq::cross(d, e2);
    const float a = sq::dot(e1, h);
    if (a > -eps && a < eps) return false;
    const float f = 1.0f / a;
    const f3 s = o - v0;
    const float u = f * sq::dot(s, h);
    if (u < 0 || u > 1) return false;
    const f3 q = sq::cross(s, e1);
    const float v = f * sq::dot(d, q);
    if (v < 0 || u + v > 1) return false;
    const float t = f * sq::dot(e2, q);
    if (!(t > eps)) return false;
    t_out = t;
    return true;
}
// 1/a for the determinant of the pooled triangle test, where the full IEEE division sequence (11 VALU instructions) is a
// seventh of the arithmetic: v_rcp_f32 (1 ulp) and two Newton steps in FMA form.  The second step is Markstein's final
// correction, which rounds correctly when its input is within an ulp; that this holds for EVERY float with
// 2^-14 <= |a| <= 2^100 (all intermediates normal there) is not argued but checked: SQ_OP_RCP_SWEEP compares it with
// 1.0f / a for all 1.9e9 of them (tests/test_gpu_parity.py).  Outside that range the caller must divide.
constexpr float kRcpMidLo = 0x1p-14f, kRcpMidHi = 0x1p100f;
__device__ __forceinline__ float rcp_midrange(float a) {
    float r = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-a, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// The same function without early exits: every value is produced by the same expression as above (a zero `a` makes
// f infinite and the later values NaN or infinite, which the conjunction then rejects exactly as the first guard does).
// For a full wave of unrelated (ray, triangle) pairs some lane nearly always reaches the last guard, so the exits
// only cost branches there.
__device__ __forceinline__ bool moller_trumbore_flat(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& t_out) {
    const float eps = 0.0001f;
    const f3 h = sq::cross(d, e2);
    const float a = sq::dot(e1, h);
#if SQ_FAST_RCP
    // |a| < eps fails g1 whatever f is (eps > kRcpMidLo), so only a huge, infinite or NaN determinant needs the division;
    // the test is wave-uniform so that the common case pays no divergence
    float f;
    if (sq_ballot(!(__builtin_fabsf(a) <= kRcpMidHi))) f = 1.0f / a; else f = rcp_midrange(a);
#else
    const float f = 1.0f / a;
#endif
    const f3 s = o - v0;
    const float u = f * sq::dot(s, h);
    const f3 q = sq::cross(s, e1);
    const float v = f * sq::dot(d, q);
    const float t = f * sq::dot(e2, q);
    const bool g1 = !(a > -eps && a < eps), g2 = !(u < 0 || u > 1), g3 = !(v < 0 || u + v > 1), g4 = t > eps;
    t_out = t;
    return g1 & g2 & g3 & g4;
}
// dist of an Intersection (src/Geometry.hs:134,141): norm ((o + t *^ d) - o), from the rounded hit point.
__device__ __forceinline__ float hit_dist(f3 o, f3 d, float t) {
    const f3 p = o + sq::scale(t, d);
    return sq::norm(p - o);
}
// `compare (dist a) (dist b) == GT` for two hits of the SAME ray, given their t.  Haskell's `compare` on Floats
// answers GT whenever neither `<` nor `==` holds, so a NaN distance compares GT both ways (sq::cmp_gt).
// hit_dist is monotone non-decreasing in t: t*d_k, o_k + (.), (.) - o_k, squaring of a value whose sign is
// fixed by d_k, the two additions and the square root are each monotone under round-to-nearest.  So for a ray
// with finite origin and direction and two finite t (then no distance is NaN: overflow only makes +inf),
// ta <= tb implies dist(a) <= dist(b), i.e. not GT, and the distances are only evaluated when ta > tb
// (where equal rounded distances still give "not GT", exactly as the reference's tie rule needs).
// A hit can carry t = +inf (f * dot overflows and still passes `t > eps`); its point has a NaN wherever the
// direction has a zero, its distance is NaN, and the shortcut must not be taken.
__device__ __forceinline__ bool dist_gt(f3 o, f3 d, float ta, float tb, bool finite_ray) {
    if (!(ta > tb) && tb < __builtin_inff() && finite_ray) return false;   // ta <= tb < inf: both finite
    return sq::cmp_gt(hit_dist(o, d, ta), hit_dist(o, d, tb));
}

// Culling slab test (my own test, not the reference's intersectsBB): plane values in FMA form, l * (1/d) + (-o/d).
// v_fma_mix_f32 takes the plane straight from a packed pair of binary16 values (exact conversion, one rounding).
__device__ __forceinline__ float fma_mix_lo(uint32_t h, float b, float c) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fma_mix_hi(uint32_t h, float b, float c) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ bool cull_slab_half(uint32_t wx, uint32_t wy, uint32_t wz, f3 df, f3 nodf) {   // w = lo | hi << 16 per axis
    const float t1 = fma_mix_lo(wx, df.x, nodf.x), t2 = fma_mix_hi(wx, df.x, nodf.x);
    const float t3 = fma_mix_lo(wy, df.y, nodf.y), t4 = fma_mix_hi(wy, df.y, nodf.y);
    const float t5 = fma_mix_lo(wz, df.z, nodf.z), t6 = fma_mix_hi(wz, df.z, nodf.z);
    const float tmin = vmax3(vmin(t1, t2), vmin(t3, t4), vmin(t5, t6));
    const float tmax = vmin3(vmax(t1, t2), vmax(t3, t4), vmax(t5, t6));
    return tmax > 0 && tmin < tmax;
}
__device__ __forceinline__ bool cull_slab(const float* b, f3 df, f3 nodf) {                                 // b = lo.xyz, hi.xyz
    const float t1 = __builtin_fmaf(b[0], df.x, nodf.x), t2 = __builtin_fmaf(b[3], df.x, nodf.x);
    const float t3 = __builtin_fmaf(b[1], df.y, nodf.y), t4 = __builtin_fmaf(b[4], df.y, nodf.y);
    const float t5 = __builtin_fmaf(b[2], df.z, nodf.z), t6 = __builtin_fmaf(b[5], df.z, nodf.z);
    const float tmin = vmax3(vmin(t1, t2), vmin(t3, t4), vmin(t5, t6));
    const float tmax = vmin3(vmax(t1, t2), vmax(t3, t4), vmax(t5, t6));
    return tmax > 0 && tmin < tmax;
}

struct Surface {            // what shading needs from a hit triangle
    f3 n;                   // normal = e1 x e2, un-normalised (src/Geometry.hs:79-80)
    float reflective; f3 surf; f3 emit;   // emit = emissive *^ emitColor (src/Lib.hs:136)
};
__device__ __forceinline__ Surface surface_of(const SceneView& S, int tri) {
    const float4* q = S.surfs + 3 * (size_t)tri;
    const float4 a = q[0], b = q[1], c = q[2];
    Surface s;
    s.n = sq::mk(a.x, a.y, a.z); s.reflective = a.w;
    s.surf = sq::mk(b.x, b.y, b.z);
    s.emit = sq::mk(c.x, c.y, c.z);
    return s;
}

// bounceRay (src/Lib.hs:155-160): `ref < x` scatters, otherwise mirrors; x is the first draw of `gen`.
__device__ __forceinline__ bool scatters(const Surface& s, uint32_t nu) { return s.reflective < sq::unit_float(nu); }
// scatterRay (src/Lib.hs:166-172): u is the SAME draw nu as x (same `gen`); v is the next draw nv.
__device__ __forceinline__ f3 scatter_dir(f3 d, const Surface& s, uint32_t nu, uint32_t nv) {
    const float u = sq::unit_float(nu), v = sq::unit_float(nv);
    const float th = 2 * sq::kPi * u;
    const float ph = sq::facos(2 * v - 1);
    float sth, cth, sph, cph;
    sq::fsincos(th, sth, cth); sq::fsincos(ph, sph, cph);
    const f3 nd = sq::mk(cth * sph, sth * sph, cph);                    // randomVector, src/Lib.hs:192-198
    const float old_ = sq::hsignum(sq::dot(d, s.n)), new_ = sq::hsignum(sq::dot(nd, s.n));
    return (old_ == new_) ? -nd : nd;
}
// reflectRay (src/Lib.hs:176-181): no random input, so it is the same ray for every sample that mirrors.
__device__ __forceinline__ f3 mirror_dir(f3 d, const Surface& s) {
    const f3 dn = sq::normalize(s.n);
    return d - sq::scale(2 * sq::dot(dn, d), dn);
}
__device__ __forceinline__ f3 bounce_dir(f3 d, const Surface& s, uint32_t nu, uint32_t nv) {
    return scatters(s, nu) ? scatter_dir(d, s, nu, nv) : mirror_dir(d, s);
}

// rgbFloatToPixelRGB (src/Lib.hs:93-104).  floor :: Float -> Word8 wraps mod 256 and maps NaN/Inf to 0.
__device__ __forceinline__ uint8_t to_word8(float f) {
    if (!(f == f) || f == __builtin_inff() || f == -__builtin_inff()) return 0;
    const double fl = __builtin_floor((double)f);
    const double md = fl - 256.0 * __builtin_floor(fl / 256.0);
    return (uint8_t)(int)md;
}
__device__ __forceinline__ void tonemap(f3 c, uint8_t* out) {
    const float mx = sq::hmax(sq::hmax(c.x, c.y), c.z), mn = sq::hmin(sq::hmin(c.x, c.y), c.z);
    const float lightness = 0.5f * (mx + mn);
    const float intensity = sq::fatan(lightness) / (sq::kPi / 2);
    const f3 s = sq::scale(intensity / mx, c);
    out[0] = to_word8(s.x * 255); out[1] = to_word8(s.y * 255); out[2] = to_word8(s.z * 255);
}

// makeRay (src/Lib.hs:107-114) + rotVert (src/Geometry.hs:104-107): dims = (w :. h), ix = (y :. x)
__device__ __forceinline__ f3 primary_dir(const float* rot, int w, int h, int y, int x) {
    const float ww = (float)w, hh = (float)h;
    const float xoffs = ((float)x - (ww / 2)) / ww;
    const float yoffs = ((hh / 2) - (float)y) / hh;
    const float v[3] = { 1.0f, xoffs, yoffs };
    float o[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { float r = 0.0f; for (int k = 0; k < 3; ++k) r = v[k] * rot[3 * k + j] + r; o[j] = r; }
    return sq::mk(o[0], o[1], o[2]);
}

// ----------------------------------------------------------------------------------------------
// intersectBIH (src/BIH.hs:101-141) as a state machine: the reference's recursion with its call
// stack made explicit.  A frame is one stack word:
//     FAR(b)      = b            "branch b descended into its near child; the far child is pending"
//     COMBINE(i)  = i | flag     "the near child of this branch returned the hit on triangle i; combine
//                                 it with what the far child returns" (src/BIH.hs:115,120)
// R is the value returned by the most recently finished call.  A COMBINE frame stores only the
// triangle: its t is recomputed with moller_trumbore (same bits); dist is always derived from t.
// NodeSrc supplies branch quads (LDS-staged or global); StackT is uint16_t when indices fit 15 bits.
// ----------------------------------------------------------------------------------------------
enum : int { M_DESCEND = 0, M_LEAF = 1, M_UNWIND = 2, M_DONE = 3,
              M_LEAFQ = 4 };   // pooled leaf phase only: the leaf is open and its untested triangles are queued in the wave's pair pool

template <typename StackT> struct StackTraits;
template <> struct StackTraits<uint16_t> { static constexpr uint32_t flag = 0x8000u; };
template <> struct StackTraits<uint32_t> { static constexpr uint32_t flag = 0x80000000u; };

// One branch as the traversal needs it.  Child references are opaque to the traversal except for
// kLeafBit; the triangle source of the same kernel knows how to turn a leaf reference into a range.
struct BranchData { v4f q0, q1; int axis; uint32_t left, right; };   // q0 = lo.xyz,lmax ; q1 = hi.xyz,rmin

struct BranchTail { float lmax, rmin; int axis; uint32_t left, right; uint32_t grown; };   // what a return into a branch needs
constexpr uint32_t kAxisMask = 0x60000000u;     // bits 30..29 of a branch's LEFT reference word: the split axis
// Resident form, bits 30..29 of the RIGHT reference word: "the left / the right child's box GROWS", i.e. the plane that
// replaces one of this branch's own (lmax for hi[axis], rmin for lo[axis], src/BIH.hs:130-141) lies outside this branch's box.
// That happens where lmax = max + 0.001 or rmin = min - 0.001 (src/BIH.hs:92-95) passes a plane of the root box that no
// ancestor has clipped yet (5 of scene.obj's 1278 children).  Used by the incremental slab test (trav_descend).
constexpr uint32_t kGrownLeft = 1u, kGrownRight = 2u;
__device__ __forceinline__ BranchTail unpack_tail(v4f q2) {
    const uint32_t l = __float_as_uint(q2.z);
    return BranchTail{ q2.x, q2.y, (int)((l >> 29) & 3u), l & ~kAxisMask, __float_as_uint(q2.w), 0u };
}
__device__ __forceinline__ BranchData unpack_branch(v4f q0, v4f q1, v4f q2) {
    const BranchTail t = unpack_tail(q2);
    return BranchData{ q0, q1, t.axis, t.left, t.right };
}
// Culling boxes of the streaming forms: the box of a CHILD sits with its parent (4 quads per branch: left lo, left hi, right
// lo, right hi), so that a child -- leaf or branch, whose box is the union of its subtree's -- is tested before anything of
// its own is read.  fp32: a 1M-triangle scene's leaves are too small for binary16 planes.
struct CullBoxes32 { float4 llo, lhi, rlo, rhi; };
__device__ __forceinline__ CullBoxes32 cull_load32(const float4* cull, uint32_t parent) {
    const float4* p = cull + 4 * (size_t)parent;
    return CullBoxes32{ p[0], p[1], p[2], p[3] };
}
__device__ __forceinline__ bool cull_test32(const CullBoxes32& c, bool left, f3 df, f3 nodf) {
    const float4 lo = left ? c.llo : c.rlo, hi = left ? c.lhi : c.rhi;
    const float b[6] = { lo.x, lo.y, lo.z, hi.x, hi.y, hi.z };
    return cull_slab(b, df, nodf);
}
struct GlobalNodes {            // every branch read from HBM/L2
    static constexpr bool kBoxInRegisters = false;
    static constexpr bool kIncremental = false;
    static constexpr bool kCull = true;
    const float4* g;
    const float4* cull; bool cull_on;
    using CullBoxes = CullBoxes32;
    __device__ __forceinline__ CullBoxes cull_load(uint32_t parent) const { return cull_load32(cull, parent); }
    __device__ __forceinline__ bool cull_test(const CullBoxes& c, bool left, f3 df, f3 nodf) const { return cull_test32(c, left, df, nodf); }
    __device__ __forceinline__ BranchData load(uint32_t b) const {
        const float4 a = g[3 * b], c = g[3 * b + 1], d = g[3 * b + 2];
        return unpack_branch(v4f{ a.x, a.y, a.z, a.w }, v4f{ c.x, c.y, c.z, c.w }, v4f{ d.x, d.y, d.z, d.w });
    }
    __device__ __forceinline__ BranchTail tail(uint32_t b) const {
        const float4 d = g[3 * b + 2];
        return unpack_tail(v4f{ d.x, d.y, d.z, d.w });
    }
};
struct HybridNodes {            // first n_lds branches (top of the tree) in LDS, the rest from HBM/L2
    // kBoxInRegisters = true carries the traversal box in six registers (a child's box is its parent's with one plane
    // replaced, src/BIH.hs:130-141 -- the same floats as the precomputed boxes), so that a visit reads ONE quad (planes,
    // axis, children) instead of three and only a return into a branch re-reads that branch's box: 45 % fewer branch
    // loads.  Measured (one MI355X, both builds in the same run, bits unchanged): 82k-triangle scene 54.8 -> 57.6 ms,
    // 1M-triangle scene 105.7 -> 107.5 ms, scene.obj streamed 86.5 -> 89.0 ms.  The counters say why: these scenes are bound by
    // VALU issue, not by branch loads (4.2 cycles per VALU instruction per SIMD on the 82k scene, L2 hit rate 99.9 %,
    // profiles/r02a_pmc_c3.txt), and tracking the box costs VALU.  Left off (-DSQ_BOX_IN_REGISTERS=1 builds it).
#ifndef SQ_BOX_IN_REGISTERS
#define SQ_BOX_IN_REGISTERS 0
#endif
    static constexpr bool kBoxInRegisters = SQ_BOX_IN_REGISTERS != 0;
    // (the incremental slab test below, ResidentNodes, was first built for this form in round 3: exact, and no faster,
    // because this form follows its memory system and not its instruction count -- profiles/r03c_stream_incremental.txt)
    static constexpr bool kIncremental = false;
    static constexpr bool kCull = true;
    // SQ_STREAM_CULL16 (default): the children's culling boxes as binary16 pairs (2 quads per branch, as in the resident form)
    // instead of fp32 (4 quads): a branch visit reads 5 quads instead of 7 and the table is half the size (4.2 MB instead of
    // 8.3 MB on the 1M-triangle scene).  Coarser boxes cull a little less (planes move outwards by up to 2^-10 of their
    // magnitude), but both stand-in scenes render faster: 27.5 -> 25.2 ms (82k triangles) and 32.7 -> 30.3 ms (1M triangles)
    // at 64 spp, same call (profiles/r03a_stream_variants.txt).  -DSQ_STREAM_CULL16=0 builds the fp32 form.
#ifndef SQ_STREAM_CULL16
#define SQ_STREAM_CULL16 1
#endif
    // SQ_STREAM_MERGED (needs SQ_STREAM_CULL16): the 48-byte branch record and the 32 bytes of its children's binary16 culling
    // boxes come from ONE packed 80-byte record (SceneView::branches_m): 1.5 cache lines per visit on average instead of 2.3, and
    // one table pointer in scalar registers instead of two (the kernel spills SGPRs).  A compile-time choice: as a launch option it
    // cost three more SGPRs and a select per load, and the six-wave build lost 12 % to the extra spills
    // (profiles/r03v_bisect_streaming.txt).
#ifndef SQ_STREAM_MERGED
#define SQ_STREAM_MERGED 1
#endif
    static constexpr bool kMerged = (SQ_STREAM_MERGED != 0) && (SQ_STREAM_CULL16 != 0);
    static constexpr uint32_t kStride = kMerged ? 5u : 3u;      // quads per record in `g`
    const SQ_LDS v4f* l; const float4* g; uint32_t n_lds;      // g = SceneView::branches_m (kMerged) or SceneView::branches
    bool cull_on;
#if SQ_STREAM_CULL16
    const uint4* cull16;                                       // !kMerged: the culling boxes' own table
    struct CullBoxes { uint4 l, r; };
    __device__ __forceinline__ CullBoxes cull_load(uint32_t parent) const {
        if constexpr (kMerged) {
            const float4* p = g + (size_t)parent * kStride;
            const float4 a = p[3], c = p[4];
            return CullBoxes{ uint4{ __float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w) }, uint4{ __float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), __float_as_uint(c.w) } };
        } else return CullBoxes{ cull16[2 * (size_t)parent], cull16[2 * (size_t)parent + 1] };
    }
    __device__ __forceinline__ bool cull_test(const CullBoxes& c, bool left, f3 df, f3 nodf) const {
        const uint4 w = left ? c.l : c.r;
        return cull_slab_half(w.x, w.y, w.z, df, nodf);
    }
#else
    const float4* cull;
    using CullBoxes = CullBoxes32;
    __device__ __forceinline__ CullBoxes cull_load(uint32_t parent) const { return cull_load32(cull, parent); }
    __device__ __forceinline__ bool cull_test(const CullBoxes& c, bool left, f3 df, f3 nodf) const { return cull_test32(c, left, df, nodf); }
#endif
    __device__ __forceinline__ BranchData load(uint32_t b) const {
        if (b < n_lds) return unpack_branch(l[3 * b], l[3 * b + 1], l[3 * b + 2]);
        const float4* p = g + (size_t)b * kStride;
        const float4 a = p[0], c = p[1], d = p[2];
        return unpack_branch(v4f{ a.x, a.y, a.z, a.w }, v4f{ c.x, c.y, c.z, c.w }, v4f{ d.x, d.y, d.z, d.w });
    }
    __device__ __forceinline__ BranchTail tail(uint32_t b) const {
        if (b < n_lds) return unpack_tail(l[3 * b + 2]);
        const float4 d = g[(size_t)b * kStride + 2];
        return unpack_tail(v4f{ d.x, d.y, d.z, d.w });
    }
};
// Resident encoding (whole scene in LDS).  A branch is 40 B: two 16-B quads and two reference words.
//   reference word: bit 31 = leaf; leaf: bits 28..24 = triangle count (<= 31), bits 23..0 = first triangle;
//                   branch: bits 23..0 = branch index.  Bits 30..29 of the LEFT word hold the split axis.
constexpr uint32_t kResAxisMask = kAxisMask;
#ifndef SQ_RES_BOX_IN_REGISTERS
#define SQ_RES_BOX_IN_REGISTERS 0
#endif
#ifndef SQ_RES_INCREMENTAL
#define SQ_RES_INCREMENTAL 0      // measured and rejected in the resident form too (see kIncremental below)
#endif
#ifndef SQ_RES_TAIL_W
#define SQ_RES_TAIL_W 0           // a return reads lmax and rmin as two 4-byte LDS reads instead of the two whole quads
#endif
struct ResidentNodes {
    // Round 1 read a branch's own box with every visit ("LDS reads are cheap and VALU is what binds").  After the pooled
    // windows and the culling boxes it is the number of memory instructions that the frame time follows, so the box can
    // ride in six registers instead (a child's box is its parent's with one plane replaced, src/BIH.hs:130-141): a branch
    // step then reads ONE 16-byte tail record (lmax, rmin, left word, right word) instead of two quads and a reference
    // pair; only a return re-reads the branch's own box (24 bytes, behind the tails).  Same 40 bytes per branch.
    static constexpr bool kBoxInRegisters = SQ_RES_BOX_IN_REGISTERS != 0;
    // Incremental slab test (round 3, trav_descend / trav_unwind): a safe ray carries (tmin, tmax) of intersectsBB for the box
    // of its current branch; a child's values follow from ONE new plane -- 2 fp32 operations and one min or max instead of
    // 24 and 16 -- and a return into a branch computes the far child's pair from that branch's box.  The trace kernel is
    // bound by VALU issue, and the two traversal slab tests were a third of a branch step's instructions.
    // `incr`: the tree allows it (no inverted interval anywhere, finite geometry: checked at upload) and option "incremental".
    static constexpr bool kIncremental = SQ_RES_INCREMENTAL != 0;
    // LDS layout with one 16-byte tail record per branch (lmax, rmin, left word, right word) and the 24-byte (lo.xyz, hi.xyz)
    // box records behind them; otherwise (lo, lmax) of all branches, then (hi, rmin) of all branches, then the reference pairs
    static constexpr bool kTailLayout = kBoxInRegisters || kIncremental;
    // (lo, lmax) of all branches, then (hi, rmin) of all branches: a wave's read of either spreads over all 16 bank slots
    // kTailLayout: `quads` holds the tail records and `boxes` the 24-byte (lo.xyz, hi.xyz) records; quads_hi / refs unused
    const SQ_LDS v4f* quads; const SQ_LDS v4f* quads_hi;
    const SQ_LDS v2i* refs;       // 1 per branch
    const SQ_LDS v2f* boxes;
    // Culling boxes (sq_cull_boxes) of a branch's two children -- leaves and whole subtrees -- as binary16 pairs, two quads
    // per branch in GLOBAL memory (20 KB for scene.obj: L1-resident).  The LDS has no room for them, and more to the point
    // the kernel is held by the CU's LDS pipe while its vector-memory path idles: the same boxes in the vertices' unused w
    // words cost 70.0 ms per headline frame against 62.5 ms from global memory (same run).
    static constexpr bool kCull = true;
    bool cull_on;
    const uint4* cull16;
    struct CullBoxes { uint4 l, r; };
    __device__ __forceinline__ CullBoxes cull_load(uint32_t parent) const { return CullBoxes{ cull16[2 * parent], cull16[2 * parent + 1] }; }
    __device__ __forceinline__ bool cull_test(const CullBoxes& c, bool left, f3 df, f3 nodf) const {
        const uint4 w = left ? c.l : c.r;
        return cull_slab_half(w.x, w.y, w.z, df, nodf);
    }
    __device__ __forceinline__ v4f q0(uint32_t b) const { return quads[b]; }
    __device__ __forceinline__ v4f q1(uint32_t b) const { return quads_hi[b]; }
    // the branch's own traversal box, lo.xyz and hi.xyz (kTailLayout: three 8-byte reads)
    __device__ __forceinline__ void box(uint32_t b, f3& lo, f3& hi) const {
        if constexpr (kTailLayout) {
            const v2f p0 = boxes[3 * b], p1 = boxes[3 * b + 1], p2 = boxes[3 * b + 2];
            lo = sq::mk(p0.x, p0.y, p1.x); hi = sq::mk(p1.y, p2.x, p2.y);
        } else {
            const v4f a = q0(b), c = q1(b);
            lo = sq::mk(a.x, a.y, a.z); hi = sq::mk(c.x, c.y, c.z);
        }
    }
    __device__ __forceinline__ BranchData load(uint32_t b) const {
        if constexpr (kTailLayout) {
            const v4f t = quads[b];
            const v2f p0 = boxes[3 * b], p1 = boxes[3 * b + 1], p2 = boxes[3 * b + 2];
            const uint32_t l = __float_as_uint(t.z);
            return BranchData{ v4f{ p0.x, p0.y, p1.x, t.x }, v4f{ p1.y, p2.x, p2.y, t.y }, (int)((l >> 29) & 3u), l & ~kResAxisMask, __float_as_uint(t.w) & ~kResAxisMask };
        }
        const v2i r = refs[b];
        return BranchData{ q0(b), q1(b), (int)(((uint32_t)r.x >> 29) & 3u), (uint32_t)r.x & ~kResAxisMask, (uint32_t)r.y & ~kResAxisMask };
    }
#ifndef SQ_RES_TAIL_GLOBAL
#define SQ_RES_TAIL_GLOBAL 0      // measured: 64.2 ms per headline frame with the global tail against 62.5 ms with the three LDS reads (same run)
#endif
    const uint4* rtail;           // global copy of (lmax, rmin, left word, right word) per branch: what a return needs, in one load
    bool incr;
    __device__ __forceinline__ BranchTail tail(uint32_t b) const {
        if (SQ_RES_TAIL_GLOBAL) {   // one global load (10 KB table, L1) instead of three LDS reads: the LDS pipe is what binds
            const uint4 t = rtail[b];
            return BranchTail{ __uint_as_float(t.x), __uint_as_float(t.y), (int)((t.z >> 29) & 3u), t.z & ~kResAxisMask, t.w & ~kResAxisMask, (t.w >> 29) & 3u };
        }
        if constexpr (kTailLayout) {
            const v4f t = quads[b];
            const uint32_t l = __float_as_uint(t.z), r = __float_as_uint(t.w);
            return BranchTail{ t.x, t.y, (int)((l >> 29) & 3u), l & ~kResAxisMask, r & ~kResAxisMask, (r >> 29) & 3u };
        }
        const v2i r = refs[b];
#if SQ_RES_TAIL_W
        const float lmax = reinterpret_cast<const SQ_LDS float*>(quads)[4 * b + 3], rmin = reinterpret_cast<const SQ_LDS float*>(quads_hi)[4 * b + 3];
        return BranchTail{ lmax, rmin, (int)(((uint32_t)r.x >> 29) & 3u), (uint32_t)r.x & ~kResAxisMask, (uint32_t)r.y & ~kResAxisMask, ((uint32_t)r.y >> 29) & 3u };
#endif
        return BranchTail{ q0(b).w, q1(b).w, (int)(((uint32_t)r.x >> 29) & 3u), (uint32_t)r.x & ~kResAxisMask, (uint32_t)r.y & ~kResAxisMask, ((uint32_t)r.y >> 29) & 3u };
    }
};

// Triangle sources: (v0, e1, e2) of triangle i.  e1 = v1 - v0 and e2 = v2 - v0 are the reference's
// edge1/edge2 (src/Geometry.hs:130-131) whether they were subtracted at upload or here.
// SQ_STREAM_NT (timing experiment, bits unchanged): bit 0 = triangle runs of the streaming form are loaded with the
// non-temporal hint (a triangle is read by the few rays that open its leaf: let the L2 keep the tree instead)
#ifndef SQ_STREAM_NT
#define SQ_STREAM_NT 0
#endif
struct GlobalTris {
    static constexpr bool kPairLoads = true;
    const float* t; const int2* leaves; bool packed, deep;   // deep: 8 triangles' loads in flight (scene beyond L2)
    __device__ __forceinline__ void get(int i, f3& v0, f3& e1, f3& e2) const {
        const float* p = t + 9 * (size_t)i;
        v0 = sq::mk(p[0], p[1], p[2]); e1 = sq::mk(p[3], p[4], p[5]); e2 = sq::mk(p[6], p[7], p[8]);
    }
    // N consecutive triangles (leaf order is memory order) as one run of 9N dwords: 16-byte loads at 4-byte
    // alignment, 9 per 4 triangles instead of 12.  Every lane reads its own addresses, so the L1 spends a tag
    // lookup per lane per instruction: fewer, wider loads are what counts.
    template <int N>
    __device__ __forceinline__ void get_n(int i, f3* v0, f3* e1, f3* e2) const {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        const float* p = t + 9 * (size_t)i;
        float w[9 * N + 3];
        constexpr int Q = (9 * N) / 4, R = (9 * N) % 4;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#if (SQ_STREAM_NT & 1)
            const f4u x = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(p + 4 * q));
#else
            const f4u x = *reinterpret_cast<const f4u*>(p + 4 * q);
#endif
            w[4 * q] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
        }
#if (SQ_STREAM_NT & 1)
        if constexpr (R >= 2) { const f2u x = __builtin_nontemporal_load(reinterpret_cast<const f2u*>(p + 4 * Q)); w[4 * Q] = x.x; w[4 * Q + 1] = x.y; }
        if constexpr (R == 1 || R == 3) w[9 * N - 1] = __builtin_nontemporal_load(p + 9 * N - 1);
#else
        if constexpr (R >= 2) { const f2u x = *reinterpret_cast<const f2u*>(p + 4 * Q); w[4 * Q] = x.x; w[4 * Q + 1] = x.y; }
        if constexpr (R == 1 || R == 3) w[9 * N - 1] = p[9 * N - 1];
#endif
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float* c = w + 9 * k;
            v0[k] = sq::mk(c[0], c[1], c[2]); e1[k] = sq::mk(c[3], c[4], c[5]); e2[k] = sq::mk(c[6], c[7], c[8]);
        }
    }
    __device__ __forceinline__ void get1(int i, f3& v0, f3& e1, f3& e2) const { get_n<1>(i, &v0, &e1, &e2); }   // two 16-byte loads + one dword
    static constexpr int kRunPad = 3;            // zero triangles after the last one (upload), so a run may start at any triangle
    template <int N>
    __device__ __forceinline__ void get_run(int i, f3* v0, f3* e1, f3* e2) const { get_n<N>(i, v0, e1, e2); }
    __device__ __forceinline__ int2 leaf(uint32_t ref) const {
        if (packed) return make_int2((int)(ref & 0xFFFFFFu), (int)((ref >> 24) & 31u));   // saves a dependent load per leaf visit
        return leaves[ref & ~kLeafBit];
    }
};
struct ResidentTris {           // whole scene resident in LDS: 16-bit indexed triangles + unique vertices (16 B each)
    static constexpr bool kPairLoads = false;
    // The vertex table sits at the very start of the workgroup's LDS and a triangle record holds its three vertices as
    // BYTE offsets into it (index * 16, which fits 16 bits for the <= 4096 vertices the resident form takes): a
    // vertex address is the record field itself, no shift and no base add (integer VALU ops cost 4.2 cycles here
    // against 2.5 for an fp32 multiply, tools/ubench/op_rate.hip).
    const SQ_LDS v4us* trix;
    __device__ __forceinline__ v4f vertex(unsigned off) const { return *reinterpret_cast<const SQ_LDS v4f*>((uintptr_t)off); }   // LDS address = offset (table at LDS address 0)
    __device__ __forceinline__ void get(int i, f3& v0, f3& e1, f3& e2) const { get_indexed(trix[i], v0, e1, e2); }
    __device__ __forceinline__ void get1(int i, f3& v0, f3& e1, f3& e2) const {    // three 16-bit reads: the offsets arrive zero-extended
        const volatile SQ_LDS unsigned short* r = reinterpret_cast<const volatile SQ_LDS unsigned short*>(trix + i);   // volatile: keep them apart
        const unsigned o0 = r[0], o1 = r[1], o2 = r[2];
        const v4f a = vertex(o0), b = vertex(o1), c = vertex(o2);
        v0 = sq::mk(a.x, a.y, a.z);
        e1 = sq::mk(b.x, b.y, b.z) - v0;
        e2 = sq::mk(c.x, c.y, c.z) - v0;
    }
    // N consecutive records, one 8-byte read each, offsets unpacked by VALU -- a third of the LDS instructions of 16-bit reads.  kRunPad zero records follow the last triangle, so a run
    // that starts at any triangle is readable.
    static constexpr int kRunPad = 3;
    template <int N>
    __device__ __forceinline__ void get_run(int i, f3* v0, f3* e1, f3* e2) const {
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        // volatile: separate ds_read_b64 (2 LDS cycles each) rather than the merged ds_read2_b64 (8)
        const volatile SQ_LDS v2u* r = reinterpret_cast<const volatile SQ_LDS v2u*>(trix + i);
        v2u rec[N];
#pragma unroll
        for (int k = 0; k < N; ++k) rec[k] = r[k];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const v4f a = vertex(rec[k].x & 0xffffu), b = vertex(rec[k].x >> 16), c = vertex(rec[k].y & 0xffffu);
            v0[k] = sq::mk(a.x, a.y, a.z); e1[k] = sq::mk(b.x, b.y, b.z) - v0[k]; e2[k] = sq::mk(c.x, c.y, c.z) - v0[k];
        }
    }
    __device__ __forceinline__ v4us index(int i) const { return trix[i]; }
    __device__ __forceinline__ void get_indexed(v4us r, f3& v0, f3& e1, f3& e2) const {
        const v4f a = vertex(r.x), b = vertex(r.y), c = vertex(r.z);
        v0 = sq::mk(a.x, a.y, a.z);
        e1 = sq::mk(b.x, b.y, b.z) - v0;
        e2 = sq::mk(c.x, c.y, c.z) - v0;
    }
    __device__ __forceinline__ int2 leaf(uint32_t ref) const { return make_int2((int)(ref & 0xFFFFFFu), (int)((ref >> 24) & 31u)); }
};

// ---- wave64 prefix scans on the DPP network (gfx9 DPP: row_shr inside rows of 16 lanes, then row_bcast:15 and
// row_bcast:31 carry a row's total into the rows above).  Six VALU instructions, no LDS.  EXEC must be all ones.
__device__ __forceinline__ int wave_scan_add(int v) {       // inclusive
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ int wave_scan_max(int v) {       // inclusive, values >= 0
    // Written out: the compiler splits the two row_bcast steps into v_mov + v_mov_dpp + v_max.  A DPP read of a VGPR
    // needs two wait states after the VALU write.  Lanes a step does not reach (bound_ctrl, row_mask) keep their value.
    asm volatile(
        "s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int lane_pull(int v, int byte_addr) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }
__device__ __forceinline__ float lane_pull(float v, int byte_addr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }

// PROFILE builds: how often the rare, expensive side paths of a return run.  `lanes` counts lanes, `waves` counts
// executions (the first active lane of each execution adds one), which is what the SIMD pays for.
struct TravProf { unsigned combine, recompute_lanes, recompute_waves, slowcmp_lanes, slowcmp_waves; };
__device__ __forceinline__ bool first_active_lane() {
    const unsigned long long m = __ballot(1);
    return (m & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0;
}

struct Trav {
    f3 o, d, df;
    uint32_t cur;
    int sp, mode;
    Hit R;
    bool safe;          // o, d, 1/d and every box coordinate finite: slab_fast and the dist_gt shortcut are exact for this ray
    int csp;            // stack index of the COMBINE frame whose t is cached below, or -1
    float ct;
    f3 blo, bhi;        // NodeSrc::kBoxInRegisters: traversal box of the node `cur` (unused otherwise)
    bool cull;          // the ray is inside the limits of sq_cull_boxes: a node whose culling box it misses returns Nothing
    f3 nodf;            // -o * (1/d), for the culling slab test in FMA form
    // NodeSrc::kIncremental, safe rays: tmin / tmax of intersectsBB (src/Geometry.hs:166-177) for the box of branch `cur`, valid
    // while tvalid; otherwise the next branch step recomputes them from that branch's own box
    float tn, tf; bool tvalid;
};

// tmin / tmax of the slab test of a box for a SAFE ray (finite o, d, 1/d, finite planes): the values the reference computes
// (src/Geometry.hs:166-177), up to the sign of a zero (v_min / v_max against Haskell's min / max, see slab_fast), which no
// comparison can see.
__device__ __forceinline__ void slab_interval(f3 lo, f3 hi, f3 o, f3 df, float& tn, float& tf) {
    const float t1 = (lo.x - o.x) * df.x, t2 = (hi.x - o.x) * df.x;
    const float t3 = (lo.y - o.y) * df.y, t4 = (hi.y - o.y) * df.y;
    const float t5 = (lo.z - o.z) * df.z, t6 = (hi.z - o.z) * df.z;
    tn = vmax3(vmin(t1, t2), vmin(t3, t4), vmin(t5, t6));
    tf = vmin3(vmax(t1, t2), vmax(t3, t4), vmax(t5, t6));
}

__device__ __forceinline__ void trav_begin(Trav& T, const SceneView& S, uint32_t root_ref, f3 o, f3 d) {
    T.o = o; T.d = d; T.df = sq::mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    T.cur = root_ref; T.sp = 0; T.R.t = 0; T.R.tri = -1;
    T.safe = S.finite_geometry && finite3(o) && finite3(d) && finite3(T.df);
    T.csp = -1; T.ct = 0;
    T.nodf = sq::mk(-o.x * T.df.x, -o.y * T.df.y, -o.z * T.df.z);
    {   // limits of the culling lemma (sq_cull_boxes): |o|^2 <= o2max, d2min <= |d|^2 <= d2max, everything finite
        const float oo = sq::dot(o, o), dd = sq::dot(d, d);
        T.cull = T.safe && finite3(T.nodf) && oo <= S.cull_o2max && dd >= S.cull_d2min && dd <= S.cull_d2max;
    }
    T.blo = sq::mk(S.root_lo[0], S.root_lo[1], S.root_lo[2]); T.bhi = sq::mk(S.root_hi[0], S.root_hi[1], S.root_hi[2]);
    T.mode = (T.cur & kLeafBit) ? M_LEAF : M_DESCEND;
    // incremental slab test: the root's pair (the same six products as slab() below; only meaningful for a safe ray)
    slab_interval(T.blo, T.bhi, o, T.df, T.tn, T.tf); T.tvalid = true;
    if (T.mode == M_DESCEND &&
        !slab(S.root_lo[0], S.root_lo[1], S.root_lo[2], S.root_hi[0], S.root_hi[1], S.root_hi[2], o, T.df))
        T.mode = M_DONE;                                                // src/BIH.hs:112 at the root
}

// Near-child prefetch (resident form, pooled kernel; -DSQ_DESCEND_PREFETCH=bits, 0 = off): the kernel is bound by the
// latency of its dependent steps, and a branch step's chain is "branch data -> slab tests -> choose the child -> the child's
// data".  Which child is NEAR only depends on the ray's direction sign on the split axis, known as soon as the reference
// words arrive: its data (bit 1: the two quads + reference words from LDS; bit 2: its children's culling boxes from global
// memory) is requested before the slab tests and used by the next branch step if the ray does descend into it.  The data
// is a pure function of the branch index, so a prefetched record is valid for any ray that reaches that branch.
#ifndef SQ_DESCEND_PREFETCH
#define SQ_DESCEND_PREFETCH 0
#endif
#ifndef SQ_UNIFORM_SAFE
#define SQ_UNIFORM_SAFE 0
#endif
#ifndef SQ_CULL_ONE_REGION
#define SQ_CULL_ONE_REGION 1      // 54.3 -> 53.85 ms per headline frame, same call, two rounds (profiles/r03r_ab_control_flow.txt)
#endif
struct BranchPf { uint32_t idx; BranchData B; ResidentNodes::CullBoxes cb; bool cb_ok; };

// One Branch equation (src/BIH.hs:111-141).  Pre: mode == M_DESCEND.
template <typename NodeSrc, typename StackT>
__device__ __forceinline__ void trav_descend(Trav& T, const NodeSrc& N, SQ_LDS StackT* stk, int stride, BranchPf* pf = nullptr) {
    if constexpr (NodeSrc::kIncremental) {
        // Incremental slab test.  The box of a child is its parent's with ONE plane replaced (src/BIH.hs:130-141), and
        // t(plane) = (plane - o_a) * (1/d_a), the very product intersectsBB forms, is monotone in the plane under
        // round-to-nearest (non-decreasing for d_a > 0, non-increasing for d_a < 0).  For a safe ray (no NaN anywhere) and a
        // REGULAR child -- the new plane lies within the parent's interval [lo_a, hi_a] on the split axis -- the child's
        // per-axis (near, far) pair on that axis is therefore contained in the parent's, the other two axes are the
        // parent's, and with t = t(new plane):
        //     left child  [lo_a, lmax]:  d_a > 0: tmax' = min(tmax, t)   d_a < 0: tmin' = max(tmin, t)   (the other one unchanged)
        //     right child [rmin, hi_a]:  d_a > 0: tmin' = max(tmin, t)   d_a < 0: tmax' = min(tmax, t)
        // are the reference's values for the child bit for bit: a max / min over the three axes in which one term is replaced
        // by one that is at least as tight.  A GROWN child (kGrownLeft / kGrownRight) has a box that contains its parent's; the
        // parent passed `tmax > 0 && tmin < tmax` (or the ray would not be here), the child's tmin is no larger and its tmax
        // no smaller, so the child passes too -- but its own pair is not derivable from the parent's: the ray is marked and
        // recomputes it from that child's own box if it descends into it.  A return into a FAR frame computes the far child's
        // pair from the branch's own box with the one plane replaced (trav_unwind), which is right for regular and grown
        // children alike.  Trees with an inverted interval anywhere (never built by makeBIH; possible through the C-ABI) and
        // unsafe rays take the plain path below.
        if (N.incr && T.safe) {
            typename NodeSrc::CullBoxes cbx;
            const bool use_cull_i = T.cull && N.cull_on;
            if (use_cull_i) cbx = N.cull_load(T.cur);                       // both children's culling boxes, requested first
            const BranchTail B = N.tail(T.cur);
            if (!T.tvalid) {                                                // rare: the ray descended into a grown child
                f3 lo, hi; N.box(T.cur, lo, hi);
                slab_interval(lo, hi, T.o, T.df, T.tn, T.tf);
                T.tvalid = true;
            }
            const float oa = sq::axis_of(T.o, B.axis), dfa = sq::axis_of(T.df, B.axis);
            const bool pos = sq::axis_of(T.d, B.axis) > 0;                  // leftToRight, src/BIH.hs:127
            const float tl = (B.lmax - oa) * dfa, tr = (B.rmin - oa) * dfa;
            const float tnL = pos ? T.tn : vmax(T.tn, tl), tfL = pos ? vmin(T.tf, tl) : T.tf;
            const float tnR = pos ? vmax(T.tn, tr) : T.tn, tfR = pos ? T.tf : vmin(T.tf, tr);
            const bool gL = (B.grown & kGrownLeft) != 0, gR = (B.grown & kGrownRight) != 0;
            bool iL = gL || (tfL > 0 && tnL < tfL), iR = gR || (tfR > 0 && tnR < tfR);
            if (use_cull_i) { iL = iL && N.cull_test(cbx, true, T.df, T.nodf); iR = iR && N.cull_test(cbx, false, T.df, T.nodf); }
            bool went_left;
            if (iL && iR) {
                stk[T.sp * stride] = (StackT)T.cur; ++T.sp;                 // FAR(cur)
                went_left = pos;
            } else if (iL) went_left = true;
            else if (iR) went_left = false;
            else { T.R.tri = -1; T.mode = M_UNWIND; return; }               // src/BIH.hs:119
            T.cur = went_left ? B.left : B.right;
            T.tn = went_left ? tnL : tnR; T.tf = went_left ? tfL : tfR;
            T.tvalid = !(went_left ? gL : gR);
            if (T.cur & kLeafBit) T.mode = M_LEAF;
            return;
        }
    }
    v4f q0, q1; int ax; uint32_t left, right;
    constexpr bool kPf = (SQ_DESCEND_PREFETCH != 0) && std::is_same<NodeSrc, ResidentNodes>::value;
    bool have = false;
    if constexpr (kPf) have = pf != nullptr && pf->idx == T.cur;
    // Culling (sq_cull_boxes): a child whose culling box the ray misses returns Nothing without being visited -- for a leaf,
    // mollerTrumbore would reject every triangle (src/BIH.hs:105-109); for a branch, every leaf below it.  The boxes of
    // both children are requested first, so that their latency overlaps the branch's own reads and slab tests.
    typename NodeSrc::CullBoxes cb;
    bool use_cull = false;
    if constexpr (NodeSrc::kCull) {
        use_cull = T.cull && N.cull_on;
        if constexpr (kPf && (SQ_DESCEND_PREFETCH & 2)) {
            if (use_cull) { if (have && pf->cb_ok) cb = pf->cb; else cb = N.cull_load(T.cur); }
        } else if (use_cull) cb = N.cull_load(T.cur);
    }
    if constexpr (kPf && (SQ_DESCEND_PREFETCH & 1)) {
        BranchData B;
        if (have) B = pf->B; else B = N.load(T.cur);
        q0 = B.q0; q1 = B.q1; ax = B.axis; left = B.left; right = B.right;
    } else
    if constexpr (NodeSrc::kBoxInRegisters) {
        const BranchTail B = N.tail(T.cur);
        q0 = v4f{ T.blo.x, T.blo.y, T.blo.z, B.lmax }; q1 = v4f{ T.bhi.x, T.bhi.y, T.bhi.z, B.rmin };
        ax = B.axis; left = B.left; right = B.right;
    } else {
        const BranchData B = N.load(T.cur);
        q0 = B.q0; q1 = B.q1; ax = B.axis; left = B.left; right = B.right;
    }
    if constexpr (kPf) if (pf != nullptr) {                             // request the near child's data before the slab tests
        const uint32_t near = (sq::axis_of(T.d, ax) > 0) ? left : right;   // src/BIH.hs:127
        pf->idx = near;                                                 // a leaf reference never equals a branch's T.cur
        if (!(near & kLeafBit)) {
            if constexpr ((SQ_DESCEND_PREFETCH & 1) != 0) pf->B = N.load(near);
            if constexpr ((SQ_DESCEND_PREFETCH & 2) != 0) { pf->cb_ok = use_cull; if (use_cull) pf->cb = N.cull_load(near); }
        }
    }
    const float lmax = q0.w, rmin = q1.w;
    // left = bbox with hi[ax] := lmax ; right = bbox with lo[ax] := rmin   (src/BIH.hs:130-141)
    const float lhx = ax == 0 ? lmax : q1.x, lhy = ax == 1 ? lmax : q1.y, lhz = ax == 2 ? lmax : q1.z;
    const float rlx = ax == 0 ? rmin : q0.x, rly = ax == 1 ? rmin : q0.y, rlz = ax == 2 ? rmin : q0.z;
    bool iL, iR;
#if SQ_UNIFORM_SAFE
    // slab() is right for every ray and slab_fast() for safe ones: the choice is made per WAVE (one scalar branch instead of
    // a divergent if / else with its exec-mask bookkeeping -- the trace kernel's time follows its scalar and branch
    // instructions as much as its VALU instructions, profiles/r03q_pmc_incremental*.txt)
    if (sq_ballot(!T.safe) == 0) {
#else
    if (T.safe) {
#endif
        iL = slab_fast(q0.x, q0.y, q0.z, lhx, lhy, lhz, T.o, T.df);
        iR = slab_fast(rlx, rly, rlz, q1.x, q1.y, q1.z, T.o, T.df);
    } else {
        iL = slab(q0.x, q0.y, q0.z, lhx, lhy, lhz, T.o, T.df);
        iR = slab(rlx, rly, rlz, q1.x, q1.y, q1.z, T.o, T.df);
    }
    if constexpr (NodeSrc::kCull) {
        // With a child known to return Nothing the Branch equation reduces to the other child's value, whatever isClose says
        // (src/BIH.hs:113-119: near = Nothing -> far; far = Nothing -> near in all three alternatives), i.e. to the
        // single-child equations, and no frame is pushed: "intersects" below means "intersects and may hold a hit".
#if SQ_CULL_ONE_REGION
        // both tests for every lane that loaded the boxes, in one exec region (no short-circuit: nearly every wave has a lane
        // that needs each of them anyway)
        if (use_cull) { const bool cL = N.cull_test(cb, true, T.df, T.nodf), cR = N.cull_test(cb, false, T.df, T.nodf); iL = iL & cL; iR = iR & cR; }
#else
        if (use_cull) { iL = iL && N.cull_test(cb, true, T.df, T.nodf); iR = iR && N.cull_test(cb, false, T.df, T.nodf); }
#endif
    }
    bool went_left;
    if (iL && iR) {
        const bool l2r = sq::axis_of(T.d, ax) > 0;                      // src/BIH.hs:127
        stk[T.sp * stride] = (StackT)T.cur; ++T.sp;                     // FAR(cur)
        went_left = l2r;
    } else if (iL) went_left = true;
    else if (iR) went_left = false;
    else { T.R.tri = -1; T.mode = M_UNWIND; return; }                   // src/BIH.hs:119
    T.cur = went_left ? left : right;
    if constexpr (NodeSrc::kBoxInRegisters) {                           // the chosen child's box (src/BIH.hs:130-141)
        if (went_left) T.bhi = sq::mk(lhx, lhy, lhz); else T.blo = sq::mk(rlx, rly, rlz);
    }
    if (T.cur & kLeafBit) T.mode = M_LEAF;
}

// One triangle of a Leaf equation folded into R with minimumBy's rule (src/BIH.hs:105-109).
__device__ __forceinline__ void leaf_fold(Trav& T, f3 v0, f3 e1, f3 e2, int i) {
    float t;
    if (moller_trumbore(T.o, T.d, v0, e1, e2, t)) {
        if (T.R.tri < 0 || dist_gt(T.o, T.d, T.R.t, t, T.safe)) { T.R.t = t; T.R.tri = i; }   // replace only on GT
    }
}
// The whole Leaf equation, triangles in leaf order.  Pre: mode == M_LEAF.
// (A software-pipelined form that kept the next triangle's loads in flight measured 2 % slower with
// the scene in LDS and 28 % slower from L2: the waves already hide that latency.)
template <typename TriSrc>
__device__ __forceinline__ void trav_leaf(Trav& T, const TriSrc& G) {
    const int2 lf = G.leaf(T.cur);
    T.R.tri = -1;
    int i = lf.x;
    const int end = lf.x + lf.y;
    if constexpr (TriSrc::kPairLoads) {          // from L2/HBM: several triangles' loads in flight per iteration
        if (G.deep) for (; i + 7 < end; i += 8) {
            f3 v0[8], e1[8], e2[8];
            G.template get_n<8>(i, v0, e1, e2);
#pragma unroll
            for (int k = 0; k < 8; ++k) leaf_fold(T, v0[k], e1[k], e2[k], i + k);
        }
        for (; i + 3 < end; i += 4) {
            f3 v0[4], e1[4], e2[4];
            G.template get_n<4>(i, v0, e1, e2);
#pragma unroll
            for (int k = 0; k < 4; ++k) leaf_fold(T, v0[k], e1[k], e2[k], i + k);
        }
        for (; i + 1 < end; i += 2) {
            f3 v0[2], e1[2], e2[2];
            G.template get_n<2>(i, v0, e1, e2);
            leaf_fold(T, v0[0], e1[0], e2[0], i);
            leaf_fold(T, v0[1], e1[1], e2[1], i + 1);
        }
    }
    if constexpr (!TriSrc::kPairLoads) {
        // LDS-resident: a triangle costs two dependent LDS round trips (index record, then vertices); the next
        // triangle's index record is read one iteration ahead (8 bytes; the read past the last record stays in LDS).
        if (i < end) {
            auto r = G.index(i);
            for (; i < end; ++i) {
                const auto cur = r;
                r = G.index(i + 1);
                __builtin_amdgcn_sched_barrier(0);
                f3 v0, e1, e2;
                G.get_indexed(cur, v0, e1, e2);
                leaf_fold(T, v0, e1, e2, i);
            }
        }
    } else
    for (; i < end; ++i) {
        f3 v0, e1, e2;
        G.get(i, v0, e1, e2);
        leaf_fold(T, v0, e1, e2, i);
    }
    T.mode = M_UNWIND;
}

// The COMBINE half of a return: the popped frame `e` holds the hit the near child had returned; R is the far child's.
// minimumByMay over [near, far] (src/BIH.hs:115,120).  T.sp already points at the popped frame.
template <typename TriSrc, typename StackT>
__device__ __forceinline__ void trav_unwind_combine(Trav& T, const TriSrc& G, uint32_t e, TravProf* prof = nullptr) {
    constexpr uint32_t flag = StackTraits<StackT>::flag;
    const int32_t ntri = (int32_t)(e & ~flag);
    float nt = T.ct;
    if (prof) ++prof->combine;
    if (T.csp != T.sp) {                                                // not the cached (newest) frame: same bits from MT
        if (prof) { ++prof->recompute_lanes; prof->recompute_waves += first_active_lane(); }
        f3 v0, e1, e2;
        G.get(ntri, v0, e1, e2);
        (void)moller_trumbore(T.o, T.d, v0, e1, e2, nt);
    }
    T.csp = -1;
    if (prof && T.R.tri >= 0 && !(!(nt > T.R.t) && T.R.t < __builtin_inff() && T.safe)) { ++prof->slowcmp_lanes; prof->slowcmp_waves += first_active_lane(); }
    if (T.R.tri < 0 || !dist_gt(T.o, T.d, nt, T.R.t, T.safe)) { T.R.t = nt; T.R.tri = ntri; }   // ties keep near
}
// Return to the caller of the call that just produced R: pop one frame.  Pre: mode == M_UNWIND.
template <typename NodeSrc, typename TriSrc, typename StackT>
__device__ __forceinline__ void trav_unwind(Trav& T, const NodeSrc& N, const TriSrc& G, SQ_LDS StackT* stk, int stride, TravProf* prof = nullptr) {
    constexpr uint32_t flag = StackTraits<StackT>::flag;
    if (T.sp == 0) { T.mode = M_DONE; return; }
    --T.sp;
    const uint32_t e = stk[T.sp * stride];
    if (e & flag) { trav_unwind_combine<TriSrc, StackT>(T, G, e, prof); return; }
    BranchTail B;                                                       // back in branch e: its near child returned R
    if constexpr (NodeSrc::kBoxInRegisters) {
        const BranchData D = N.load(e);                                 // this branch's own box again; the far child's follows below
        T.blo = sq::mk(D.q0.x, D.q0.y, D.q0.z); T.bhi = sq::mk(D.q1.x, D.q1.y, D.q1.z);
        B = BranchTail{ D.q0.w, D.q1.w, D.axis, D.left, D.right, 0u };
    } else B = N.tail(e);
    const int ax = B.axis;
    const bool l2r = sq::axis_of(T.d, ax) > 0;
    if (T.R.tri >= 0) {
        const float p = sq::axis_of(T.o, ax) + T.R.t * sq::axis_of(T.d, ax);   // projectToAxis ax (intersectPoint near)
#if SQ_BALLOT_BUILTIN
        const bool close = (l2r & (p < B.rmin)) | (!l2r & (p > B.lmax));   // isClose, src/BIH.hs:121-123 (mask arithmetic, no select of Bools)
#else
        const bool close = l2r ? (p < B.rmin) : (p > B.lmax);           // isClose, src/BIH.hs:121-123
#endif
        if (close) return;                                              // src/BIH.hs:114: the branch returns near
        stk[T.sp * stride] = (StackT)((uint32_t)T.R.tri | flag);        // COMBINE(R)
        T.csp = T.sp; T.ct = T.R.t; ++T.sp;
    }
    T.cur = l2r ? B.right : B.left;                                     // the far child
    if constexpr (NodeSrc::kBoxInRegisters) {
        if (l2r) { if (ax == 0) T.blo.x = B.rmin; else if (ax == 1) T.blo.y = B.rmin; else T.blo.z = B.rmin; }
        else     { if (ax == 0) T.bhi.x = B.lmax; else if (ax == 1) T.bhi.y = B.lmax; else T.bhi.z = B.lmax; }
    }
    if constexpr (NodeSrc::kIncremental) {
        // incremental slab test: the far child's (tmin, tmax) from its own box = this branch's box with lo[ax] := rmin (right
        // child) or hi[ax] := lmax (left child), src/BIH.hs:130-141 -- regular or grown, the same expressions as the reference
        if (N.incr && T.safe && !(T.cur & kLeafBit)) {
            f3 lo, hi; N.box(e, lo, hi);
            const float rx = (l2r && ax == 0) ? B.rmin : lo.x, ry = (l2r && ax == 1) ? B.rmin : lo.y, rz = (l2r && ax == 2) ? B.rmin : lo.z;
            const float hx = (!l2r && ax == 0) ? B.lmax : hi.x, hy = (!l2r && ax == 1) ? B.lmax : hi.y, hz = (!l2r && ax == 2) ? B.lmax : hi.z;
            slab_interval(sq::mk(rx, ry, rz), sq::mk(hx, hy, hz), T.o, T.df, T.tn, T.tf);
            T.tvalid = true;
        } else T.tvalid = false;
    }
    // (no culling test here: a FAR frame is only pushed for a far child whose culling box the ray hits, trav_descend)
    T.mode = (T.cur & kLeafBit) ? M_LEAF : M_DESCEND;
}


// ---- "flat" branch and return steps (resident form, pooled kernel; SQ_FLAT_STEPS) ----------------------------------------------
// The same equations as trav_descend / the FAR half of trav_unwind, written as ONE exec region each: every decision is a select on
// values computed for all lanes of the step, and the two stack stores are unconditional (a store that the reference's control flow
// would not make lands above the top of the stack or on the frame that was just popped, where nothing reads it).  Round 3 measured
// that this kernel's time follows its scalar / exec-mask instructions as much as its VALU instructions: the nested
// ifs of the plain forms cost about three scalar or branch instructions per nesting level and execution.
// Preconditions, checked per WAVE by the caller: every lane taking the step has a safe ray (slab_fast is exact for it), and the
// scene has culling boxes (N.cull_on).  Arithmetic: expression for expression that of the plain forms.
#ifndef SQ_FLAT_STEPS
#define SQ_FLAT_STEPS 1
#endif
#ifndef SQ_FLAT_STREAM
#define SQ_FLAT_STREAM 0          // the same steps in the streaming form's pooled kernel (HybridNodes): A/B switch
#endif
template <typename NodeSrc, typename StackT>
__device__ __forceinline__ void trav_descend_flat(Trav& T, const NodeSrc& N, SQ_LDS StackT* stk, int stride) {
    const typename NodeSrc::CullBoxes cb = N.cull_load(T.cur);          // both children's culling boxes, requested first
    const BranchData B = N.load(T.cur);
    const v4f q0 = B.q0, q1 = B.q1; const int ax = B.axis;
    const float lmax = q0.w, rmin = q1.w;
    // left = bbox with hi[ax] := lmax ; right = bbox with lo[ax] := rmin   (src/BIH.hs:130-141)
    const float lhx = ax == 0 ? lmax : q1.x, lhy = ax == 1 ? lmax : q1.y, lhz = ax == 2 ? lmax : q1.z;
    const float rlx = ax == 0 ? rmin : q0.x, rly = ax == 1 ? rmin : q0.y, rlz = ax == 2 ? rmin : q0.z;
    const bool sL = slab_fast(q0.x, q0.y, q0.z, lhx, lhy, lhz, T.o, T.df);
    const bool sR = slab_fast(rlx, rly, rlz, q1.x, q1.y, q1.z, T.o, T.df);
    const bool cL = N.cull_test(cb, true, T.df, T.nodf), cR = N.cull_test(cb, false, T.df, T.nodf);
    // (logical operators on values that are already computed: they stay i1 mask arithmetic in scalar registers, where `|` / `&`
    // on bools are integer operations on 0 / 1 words in vector registers)
    const bool nocull = !T.cull;                                        // a ray outside the lemma's limits visits what the reference visits
    const bool iL = sL && (cL || nocull), iR = sR && (cR || nocull);
    const bool both = iL && iR, any = iL || iR;
    const bool l2r = sq::axis_of(T.d, ax) > 0;                          // src/BIH.hs:127
    stk[T.sp * stride] = (StackT)T.cur;                                 // FAR(cur) if both children are visited; else a dead word above the top
    T.sp += both ? 1 : 0;
    const bool went_left = (both && l2r) || (!both && iL);
    const uint32_t next = went_left ? B.left : B.right;
    T.cur = any ? next : T.cur;
    T.R.tri = any ? T.R.tri : -1;                                       // src/BIH.hs:119: neither child -> Nothing
    T.mode = any ? ((next & kLeafBit) ? M_LEAF : M_DESCEND) : M_UNWIND;
}
// The FAR half of a return (the popped frame is a branch whose near child returned R).  Pre: mode == M_UNWIND, the popped word `e`
// has no COMBINE flag; T.sp already points at the popped frame.
template <typename NodeSrc, typename StackT>
__device__ __forceinline__ void trav_unwind_far_flat(Trav& T, const NodeSrc& N, SQ_LDS StackT* stk, int stride, uint32_t e) {
    constexpr uint32_t flag = StackTraits<StackT>::flag;
    const BranchTail B = N.tail(e);
    const int ax = B.axis;
    const float da = sq::axis_of(T.d, ax);
    const bool l2r = da > 0;
    const bool hit = T.R.tri >= 0;
    const float p = sq::axis_of(T.o, ax) + T.R.t * da;                  // projectToAxis ax (intersectPoint near); unused without a hit
    const bool close = hit && ((l2r && p < B.rmin) || (!l2r && p > B.lmax));   // isClose, src/BIH.hs:121-123: the branch returns near
    const bool push = hit && !close;                                    // COMBINE(R) (src/BIH.hs:115,120)
    stk[T.sp * stride] = (StackT)((uint32_t)T.R.tri | flag);            // lands on the frame just popped: dead unless pushed
    T.csp = push ? T.sp : T.csp; T.ct = push ? T.R.t : T.ct;
    T.sp += push ? 1 : 0;
    const uint32_t far = l2r ? B.right : B.left;
    T.cur = close ? T.cur : far;
    T.mode = close ? M_UNWIND : ((far & kLeafBit) ? M_LEAF : M_DESCEND);
}

// Whole query, one ray per lane (used where rays of a wave are coherent: primary and shadow rays).
template <typename NodeSrc, typename TriSrc, typename StackT>
__device__ __forceinline__ Hit trace_one(const SceneView& S, const NodeSrc& N, const TriSrc& G, uint32_t root_ref, f3 o, f3 d, SQ_LDS StackT* stk, int stride) {
    Trav T;
    trav_begin(T, S, root_ref, o, d);
    while (T.mode != M_DONE) {
        while (T.mode == M_DESCEND) trav_descend(T, N, stk, stride);
        if (T.mode == M_LEAF) trav_leaf(T, G);
        while (T.mode == M_UNWIND) trav_unwind(T, N, G, stk, stride);
    }
    return T.R;
}
template <typename NodeSrc, typename StackT>
__device__ __forceinline__ Hit trace_one(const SceneView& S, const NodeSrc& N, f3 o, f3 d, SQ_LDS StackT* stk, int stride) {
    const GlobalTris G{ S.tris, S.leaves, S.packed_leaves != 0, false };
    return trace_one(S, N, G, S.root_ref, o, d, stk, stride);
}

}  // namespace sqd
