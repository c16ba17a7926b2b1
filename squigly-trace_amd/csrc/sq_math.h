// sq_math.h — scalar arithmetic of the sampling path, usable from host C++ and gfx950 device code.
//
// Everything here is written so that the x86 host compiler and the gfx950 device compiler
// produce bit-identical binary32 results: no FMA contraction (-ffp-contract=off), IEEE
// divide/sqrt (-fhip-fp32-correctly-rounded-divide-sqrt), Haskell's Ord-class min/max, and
// transcendental functions evaluated in binary64 with a fixed operation order then rounded
// once ("crd" spec; constants from tools/gen_math_consts.py).
// Citations are relative to the reference repository root.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SQ_HD __host__ __device__ __forceinline__
#else
#define SQ_HD inline
#endif

namespace sq {

struct f3 { float x, y, z; };

SQ_HD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
SQ_HD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }            // src/V3.hs:8
SQ_HD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }            // src/V3.hs:9
SQ_HD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }                                 // src/V3.hs:12
SQ_HD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }            // Num default: a + negate b
SQ_HD f3 scale(float r, f3 v) { return mk(r * v.x, r * v.y, r * v.z); }                   // (*^) src/V3.hs:18-19
SQ_HD f3 cross(f3 p, f3 q) {                                                              // src/V3.hs:21-22
    return mk(p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x);
}
SQ_HD float dot(f3 p, f3 q) { return (p.x * q.x + p.y * q.y) + p.z * q.z; }               // src/V3.hs:25-26
// IEEE square root on both sides.  (HIP's __fsqrt_rn is the *native* 1-ulp v_sqrt_f32; the builtin
// lowers to llvm.sqrt, which -fhip-fp32-correctly-rounded-divide-sqrt expands to a correctly rounded one.)
SQ_HD float fsqrt(float x) { return __builtin_sqrtf(x); }
SQ_HD float norm(f3 v) { return fsqrt(dot(v, v)); }                                       // src/V3.hs:28-32
SQ_HD f3 normalize(f3 v) { float n = norm(v); return mk(v.x / n, v.y / n, v.z / n); }     // src/V3.hs:34-37
SQ_HD float axis_of(f3 v, int ax) { return ax == 0 ? v.x : (ax == 1 ? v.y : v.z); }       // projectToAxis, src/Geometry.hs:200-205

// Ord Float class defaults: max x y = if x <= y then y else x ; min x y = if x <= y then x else y.
// NaN in either argument selects per the comparison, unlike fminf/fmaxf.
SQ_HD float hmax(float x, float y) { return (x <= y) ? y : x; }
SQ_HD float hmin(float x, float y) { return (x <= y) ? x : y; }
// compare a b == GT for Float (LT if a<b, EQ if a==b, otherwise GT — so NaN gives GT)
SQ_HD bool cmp_gt(float a, float b) { return !(a < b) && !(a == b); }
// signum :: Float: 1 if x>0, -1 if x<0, x itself otherwise (-0, NaN)
SQ_HD float hsignum(float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : x); }

// ---------------- "crd" transcendental spec ----------------
namespace crd {
SQ_HD double poly_sin(double r) {
    double z = r * r, p;
    p = 0x1.952c77030ad4ap-49;
    p = p * z + -0x1.ae7f3e733b81fp-41;
    p = p * z + 0x1.6124613a86d09p-33;
    p = p * z + -0x1.ae64567f544e4p-26;
    p = p * z + 0x1.71de3a556c734p-19;
    p = p * z + -0x1.a01a01a01a01ap-13;
    p = p * z + 0x1.1111111111111p-7;
    p = p * z + -0x1.5555555555555p-3;
    return r + (r * z) * p;
}
SQ_HD double poly_cos(double r) {
    double z = r * r, p;
    p = 0x1.ae7f3e733b81fp-45;
    p = p * z + -0x1.93974a8c07c9dp-37;
    p = p * z + 0x1.1eed8eff8d898p-29;
    p = p * z + -0x1.27e4fb7789f5cp-22;
    p = p * z + 0x1.a01a01a01a01ap-16;
    p = p * z + -0x1.6c16c16c16c17p-10;
    p = p * z + 0x1.5555555555555p-5;
    return (1.0 - 0.5 * z) + (z * z) * p;
}
SQ_HD int reduce(double x, double& r) {
    if (__builtin_fabs(x) <= 0x1.921fb54442d18p-1) { r = x; return 0; }
    double fn = __builtin_floor(x * 0x1.45f306dc9c883p-1 + 0.5);
    r = (x - fn * 0x1.921fb54400000p+0) - fn * 0x1.0b4611a626331p-34;
    // fn is integral; from 2^54 on it is a multiple of 4, and beyond the range of long long (or NaN) the conversion is
    // undefined in C++ (found by a UBSan build of the host side on a camera file with an absurd angle): quadrant 0 there,
    // which is both fn mod 4 and what the x86 conversion's 0x8000000000000000 gave
    if (!(__builtin_fabs(fn) < 0x1p62)) return 0;
    return (int)((long long)fn & 3);
}
// sin and cos of the same argument share the reduction and both polynomials
SQ_HD void sincos(double x, double& s, double& c) {
    double r; int n = reduce(x, r);
    double ps = poly_sin(r), pc = poly_cos(r);
    s = (n == 0) ? ps : (n == 1) ? pc : (n == 2) ? -ps : -pc;
    c = (n == 0) ? pc : (n == 1) ? -ps : (n == 2) ? -pc : ps;
}
SQ_HD double asin_tail(double z) {   // sum_{k=1..24} C(2k,k)/(4^k(2k+1)) z^k
    double p = 0x1.3275586c5f2f0p-9;
    p = p * z + 0x1.464c0950f7d47p-9;  p = p * z + 0x1.5c5f56efaaaabp-9;  p = p * z + 0x1.750de64d7d05fp-9;
    p = p * z + 0x1.90cb77f60c7cep-9;  p = p * z + 0x1.b026f57b13b14p-9;  p = p * z + 0x1.d3d2a8e0dd67dp-9;
    p = p * z + 0x1.fcaf8fb6db6dbp-9;  p = p * z + 0x1.15ee9d45d1746p-8;  p = p * z + 0x1.31683bdef7bdfp-8;
    p = p * z + 0x1.51ba308d3dcb1p-8;  p = p * z + 0x1.782dda12f684cp-8;  p = p * z + 0x1.a6863d70a3d71p-8;
    p = p * z + 0x1.df3bd37a6f4dfp-8;  p = p * z + 0x1.12ef3cf3cf3cfp-7;  p = p * z + 0x1.3fde50d79435ep-7;
    p = p * z + 0x1.7a87878787878p-7;  p = p * z + 0x1.c99999999999ap-7;  p = p * z + 0x1.1c4ec4ec4ec4fp-6;
    p = p * z + 0x1.6e8ba2e8ba2e9p-6;  p = p * z + 0x1.f1c71c71c71c7p-6;  p = p * z + 0x1.6db6db6db6db7p-5;
    p = p * z + 0x1.3333333333333p-4;  p = p * z + 0x1.5555555555555p-3;
    return p * z;
}
SQ_HD double dsqrt(double x) { return __builtin_sqrt(x); }
SQ_HD double acos_(double x) {
    double ax = __builtin_fabs(x);
    if (ax <= 0.5) { double z = x * x; return 0x1.921fb54442d18p+0 - (x + x * asin_tail(z)); }
    double z = (1.0 - ax) * 0.5;
    double s = dsqrt(z);
    double t = 2.0 * (s + s * asin_tail(z));
    return (x > 0) ? t : (0x1.921fb54442d18p+1 - t);
}
SQ_HD double atan_tab(int k) {
    switch (k) {
        case 0: return 0.0;
        case 1: return 0x1.fd5ba9aac2f6ep-4;  case 2: return 0x1.f5b75f92c80ddp-3;
        case 3: return 0x1.6f61941e4def1p-2;  case 4: return 0x1.dac670561bb4fp-2;
        case 5: return 0x1.1e00babdefeb4p-1;  case 6: return 0x1.4978fa3269ee1p-1;
        case 7: return 0x1.700a7c5784634p-1;  default: return 0x1.921fb54442d18p-1;
    }
}
SQ_HD double atan_(double x) {
    double ax = __builtin_fabs(x);
    bool inv = ax > 1.0;
    double y = inv ? 1.0 / ax : ax;
    double kf = __builtin_floor(y * 8.0 + 0.5);
    double c = kf * 0.125;
    double t = (y - c) / (1.0 + y * c);
    double z = t * t, p;
    p = -0x1.1111111111111p-4;
    p = p * z + 0x1.3b13b13b13b14p-4;
    p = p * z + -0x1.745d1745d1746p-4;
    p = p * z + 0x1.c71c71c71c71cp-4;
    p = p * z + -0x1.2492492492492p-3;
    p = p * z + 0x1.999999999999ap-3;
    p = p * z + -0x1.5555555555555p-2;
    double r = atan_tab((int)kf) + (t + (t * z) * p);
    if (inv) r = 0x1.921fb54442d18p+0 - r;
    return (x < 0) ? -r : r;
}
}  // namespace crd

SQ_HD void fsincos(float x, float& s, float& c) { double ds, dc; crd::sincos((double)x, ds, dc); s = (float)ds; c = (float)dc; }
SQ_HD float fsin(float x) { float s, c; fsincos(x, s, c); return s; }
SQ_HD float fcos(float x) { float s, c; fsincos(x, s, c); return c; }
SQ_HD float facos(float x) { return (float)crd::acos_((double)x); }
SQ_HD float fatan(float x) { return (float)crd::atan_((double)x); }
constexpr float kPi = 3.14159265358979323846f;   // pi :: Float

// ---------------- TFGen (tf-random 0.5): first three outputs of mkTFGen seed ----------------
// Threefish-256 (Skein 1.3), key = (seed,0,0,0), zero tweak, zero counter block; outputs are the
// low then high halves of the ciphertext words (src/Lib.hs:86,134,185).
// 64-bit rotate by a compile-time amount.  On the device it is two v_alignbit_b32 on the register halves
// (the generic shift/or form compiles to 64-bit shifts, which issue at a quarter of the rate).
template <int R> SQ_HD uint64_t rotl64c(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (R == 32) return ((uint64_t)lo << 32) | hi;
    if (R < 32) {
        const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32 - R), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
        return ((uint64_t)nh << 32) | nl;
    }
    const uint32_t nh = __builtin_amdgcn_alignbit(lo, hi, 64 - R), nl = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
    return ((uint64_t)nh << 32) | nl;
#else
    return (v << R) | (v >> (64 - R));
#endif
}
#define SQ_TF_MIX(a, b, r) a += b; b = rotl64c<r>(b) ^ a
#define SQ_TF_ROUNDS8(s)                                                              \
    x0 += k[(s) % 5]; x1 += k[((s) + 1) % 5]; x2 += k[((s) + 2) % 5]; x3 += k[((s) + 3) % 5] + (uint64_t)(s); \
    SQ_TF_MIX(x0, x1, 14); SQ_TF_MIX(x2, x3, 16);                                     \
    SQ_TF_MIX(x0, x3, 52); SQ_TF_MIX(x2, x1, 57);                                     \
    SQ_TF_MIX(x0, x1, 23); SQ_TF_MIX(x2, x3, 40);                                     \
    SQ_TF_MIX(x0, x3, 5);  SQ_TF_MIX(x2, x1, 37);                                     \
    x0 += k[((s) + 1) % 5]; x1 += k[((s) + 2) % 5]; x2 += k[((s) + 3) % 5]; x3 += k[((s) + 4) % 5] + (uint64_t)((s) + 1); \
    SQ_TF_MIX(x0, x1, 25); SQ_TF_MIX(x2, x3, 33);                                     \
    SQ_TF_MIX(x0, x3, 46); SQ_TF_MIX(x2, x1, 12);                                     \
    SQ_TF_MIX(x0, x1, 58); SQ_TF_MIX(x2, x3, 22);                                     \
    SQ_TF_MIX(x0, x3, 32); SQ_TF_MIX(x2, x1, 32)
SQ_HD void threefish256_key_only(uint64_t seed, uint64_t out[4]) {
    // The word permutation (0,3,2,1) is folded into the operand order of the unrolled rounds.
    uint64_t k[5] = { seed, 0, 0, 0, 0x1BD11BDAA9FC1A22ULL ^ seed };
    uint64_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
    SQ_TF_ROUNDS8(0);  SQ_TF_ROUNDS8(2);  SQ_TF_ROUNDS8(4);  SQ_TF_ROUNDS8(6);  SQ_TF_ROUNDS8(8);
    SQ_TF_ROUNDS8(10); SQ_TF_ROUNDS8(12); SQ_TF_ROUNDS8(14); SQ_TF_ROUNDS8(16);
    x0 += k[18 % 5]; x1 += k[19 % 5]; x2 += k[20 % 5]; x3 += k[21 % 5] + 18ULL;
    out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}
SQ_HD void tfgen3(int64_t seed, uint32_t& n0, uint32_t& n1, uint32_t& n2) {
    uint64_t c[4];
    threefish256_key_only((uint64_t)seed, c);
    n0 = (uint32_t)c[0]; n1 = (uint32_t)(c[0] >> 32); n2 = (uint32_t)c[1];
}
// randomR (0,1) (src/Lib.hs:183-188): fromIntegral n / fromIntegral (maxBound :: Word32), then 0 + 1*p
SQ_HD float unit_float(uint32_t n) {
    float p = (float)n / 4294967296.0f;   // float(0xFFFFFFFF) rounds to 2^32
    return 0.0f + (1.0f - 0.0f) * p;
}

}  // namespace sq
