/* sq_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See sq_oracle.h.
 *
 * A plain-C, deliberately literal restatement of the reference's Haskell: recursive tree,
 * recursive traversal, per-sample recursion, same expression trees, same tie-breaks.
 * It shares NO source with squigly-trace_amd/csrc (which is an iterative, flattened,
 * GPU-shaped design); the two only share the written numeric spec in DESIGN.md.
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, no -ffast-math, no -march).
 *
 * All file:line citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "sq_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char* sqo_last_error(void) { return g_err; }
static int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return 1;
}
void sqo_free(void* p) { free(p); }

/* ======================= V3 (src/V3.hs) ======================= */
typedef sqo_v3 V3;
static inline V3 v3(float x, float y, float z) { V3 v = { x, y, z }; return v; }
static inline V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }       /* V3.hs:8 */
static inline V3 vmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }       /* V3.hs:9 */
static inline V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }                             /* V3.hs:12 */
/* V3 defines no (-): the Num default is x - y = x + negate y (bit-identical to IEEE sub). */
static inline V3 vsub(V3 a, V3 b) { return vadd(a, vneg(b)); }
static inline V3 vscale(float r, V3 v) { return v3(r * v.x, r * v.y, r * v.z); }         /* V3.hs:18-19 */
static inline V3 cross(V3 p, V3 q) {                                                     /* V3.hs:21-22 */
    float a = p.x, b = p.y, c = p.z, d = q.x, e = q.y, f = q.z;
    return v3(b * f - c * e, c * d - a * f, a * e - b * d);
}
static inline float dot(V3 p, V3 q) { return (p.x * q.x + p.y * q.y) + p.z * q.z; }     /* V3.hs:25-26 */
static inline float norm(V3 v) { return sqrtf(dot(v, v)); }                              /* V3.hs:28-32 */
static inline V3 normalize(V3 v) { float n = norm(v); return v3(v.x / n, v.y / n, v.z / n); } /* V3.hs:34-37 */
/* Ord Float class defaults (GHC.Classes): max x y = if x <= y then y else x ; min x y = if x <= y then x else y */
static inline float hs_max(float x, float y) { return (x <= y) ? y : x; }
static inline float hs_min(float x, float y) { return (x <= y) ? x : y; }
static inline float proj(int ax, V3 v) { return ax == 0 ? v.x : (ax == 1 ? v.y : v.z); } /* Geometry.hs:200-205 */
/* Float signum (GHC.Float): x>0 -> 1 ; x<0 -> -1 ; otherwise x (keeps -0 and NaN) */
static inline float hs_signum(float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : x); }

/* ======================= transcendental spec "crd" =======================
 * Float sin/cos/acos/atan in the reference are GHC primops that call the host libm
 * (sinf, cosf, acosf, atanf).  libm results differ between glibc versions and are not
 * available on the GPU, so the repo fixes ONE spec: convert the binary32 argument to
 * binary64, evaluate with the exact operation sequence below (no FMA, no reassociation),
 * round once to binary32.  Constants come from tools/gen_math_consts.py (integer
 * arithmetic).  tests/test_oracle_math.py measures agreement with the host libm. */
static const double PIO4    = 0x1.921fb54442d18p-1;
static const double PIO2    = 0x1.921fb54442d18p+0;
static const double PI_D    = 0x1.921fb54442d18p+1;
static const double INVPIO2 = 0x1.45f306dc9c883p-1;
static const double PIO2_1  = 0x1.921fb54400000p+0;   /* first 33 bits of pi/2 */
static const double PIO2_1T = 0x1.0b4611a626331p-34;  /* pi/2 - PIO2_1 */

static double ksin(double r) {      /* Taylor to r^17, Horner in r^2, |r| <= pi/4 */
    double z = r * r, p;
    p = 0x1.952c77030ad4ap-49;              /* +1/17! */
    p = p * z + -0x1.ae7f3e733b81fp-41;     /* -1/15! */
    p = p * z + 0x1.6124613a86d09p-33;      /* +1/13! */
    p = p * z + -0x1.ae64567f544e4p-26;     /* -1/11! */
    p = p * z + 0x1.71de3a556c734p-19;      /* +1/9!  */
    p = p * z + -0x1.a01a01a01a01ap-13;     /* -1/7!  */
    p = p * z + 0x1.1111111111111p-7;       /* +1/5!  */
    p = p * z + -0x1.5555555555555p-3;      /* -1/3!  */
    return r + (r * z) * p;
}
static double kcos(double r) {      /* Taylor to r^16 */
    double z = r * r, p;
    p = 0x1.ae7f3e733b81fp-45;              /* +1/16! */
    p = p * z + -0x1.93974a8c07c9dp-37;     /* -1/14! */
    p = p * z + 0x1.1eed8eff8d898p-29;      /* +1/12! */
    p = p * z + -0x1.27e4fb7789f5cp-22;     /* -1/10! */
    p = p * z + 0x1.a01a01a01a01ap-16;      /* +1/8!  */
    p = p * z + -0x1.6c16c16c16c17p-10;     /* -1/6!  */
    p = p * z + 0x1.5555555555555p-5;       /* +1/4!  */
    return (1.0 - 0.5 * z) + (z * z) * p;
}
static int reduce_pio2(double x, double* r) {
    if (fabs(x) <= PIO4) { *r = x; return 0; }
    double fn = floor(x * INVPIO2 + 0.5);
    *r = (x - fn * PIO2_1) - fn * PIO2_1T;
    if (!(fabs(fn) < 0x1p62)) return 0;     /* fn is a multiple of 4 from 2^54 on; the conversion below is undefined beyond long long (UBSan, host camera angles) */
    return (int)((long long)fn & 3);
}
double sqo_sin_d(double x) {
    double r; int n = reduce_pio2(x, &r);
    switch (n) { case 0: return ksin(r); case 1: return kcos(r); case 2: return -ksin(r); default: return -kcos(r); }
}
double sqo_cos_d(double x) {
    double r; int n = reduce_pio2(x, &r);
    switch (n) { case 0: return kcos(r); case 1: return -ksin(r); case 2: return -kcos(r); default: return ksin(r); }
}
static double kasin_tail(double z) { /* sum_{k>=1} a_k z^k, a_k = C(2k,k)/(4^k (2k+1)), z <= 1/4 */
    static const double A[24] = {
        0x1.5555555555555p-3, 0x1.3333333333333p-4, 0x1.6db6db6db6db7p-5, 0x1.f1c71c71c71c7p-6,
        0x1.6e8ba2e8ba2e9p-6, 0x1.1c4ec4ec4ec4fp-6, 0x1.c99999999999ap-7, 0x1.7a87878787878p-7,
        0x1.3fde50d79435ep-7, 0x1.12ef3cf3cf3cfp-7, 0x1.df3bd37a6f4dfp-8, 0x1.a6863d70a3d71p-8,
        0x1.782dda12f684cp-8, 0x1.51ba308d3dcb1p-8, 0x1.31683bdef7bdfp-8, 0x1.15ee9d45d1746p-8,
        0x1.fcaf8fb6db6dbp-9, 0x1.d3d2a8e0dd67dp-9, 0x1.b026f57b13b14p-9, 0x1.90cb77f60c7cep-9,
        0x1.750de64d7d05fp-9, 0x1.5c5f56efaaaabp-9, 0x1.464c0950f7d47p-9, 0x1.3275586c5f2f0p-9 };
    double p = A[23];
    for (int k = 22; k >= 0; k--) p = p * z + A[k];
    return p * z;
}
double sqo_acos_d(double x) {       /* x in [-1,1] */
    double ax = fabs(x);
    if (ax <= 0.5) {                /* acos x = pi/2 - asin x,  asin x = x + x*T(x^2) */
        double z = x * x;
        return PIO2 - (x + x * kasin_tail(z));
    }
    double z = (1.0 - ax) * 0.5;    /* acos|x| = 2 asin(sqrt z) */
    double s = sqrt(z);
    double t = 2.0 * (s + s * kasin_tail(z));
    return (x > 0) ? t : (PI_D - t);
}
double sqo_atan_d(double x) {
    static const double TAB[9] = {  /* atan(k/8) */
        0x0.0p+0, 0x1.fd5ba9aac2f6ep-4, 0x1.f5b75f92c80ddp-3, 0x1.6f61941e4def1p-2, 0x1.dac670561bb4fp-2,
        0x1.1e00babdefeb4p-1, 0x1.4978fa3269ee1p-1, 0x1.700a7c5784634p-1, 0x1.921fb54442d18p-1 };
    double ax = fabs(x);
    int inv = ax > 1.0;
    double y = inv ? 1.0 / ax : ax;           /* y in [0,1] (1/inf = 0) */
    double kf = floor(y * 8.0 + 0.5);
    double c = kf * 0.125;
    double t = (y - c) / (1.0 + y * c);       /* |t| <= 1/16 */
    double z = t * t, p;
    p = -0x1.1111111111111p-4;                /* -1/15 */
    p = p * z + 0x1.3b13b13b13b14p-4;         /* +1/13 */
    p = p * z + -0x1.745d1745d1746p-4;        /* -1/11 */
    p = p * z + 0x1.c71c71c71c71cp-4;         /* +1/9  */
    p = p * z + -0x1.2492492492492p-3;        /* -1/7  */
    p = p * z + 0x1.999999999999ap-3;         /* +1/5  */
    p = p * z + -0x1.5555555555555p-2;        /* -1/3  */
    int k = (kf >= 0.0 && kf <= 8.0) ? (int)kf : 0;   /* NaN input: any entry, the sum is NaN anyway */
    double r = TAB[k] + (t + (t * z) * p);
    if (inv) r = PIO2 - r;
    return (x < 0) ? -r : r;
}
float sqo_sinf(float x, int m)  { return m == SQO_TRIG_LIBM ? sinf(x)  : (float)sqo_sin_d((double)x); }
float sqo_cosf(float x, int m)  { return m == SQO_TRIG_LIBM ? cosf(x)  : (float)sqo_cos_d((double)x); }
float sqo_acosf(float x, int m) { return m == SQO_TRIG_LIBM ? acosf(x) : (float)sqo_acos_d((double)x); }
float sqo_atanf(float x, int m) { return m == SQO_TRIG_LIBM ? atanf(x) : (float)sqo_atan_d((double)x); }
static const float PI_F = 3.14159265358979323846f;   /* pi :: Float */

/* ======================= TFGen (tf-random 0.5) =======================
 * Third-party arithmetic, source NOT under /root/reference: restated from the published
 * algorithm (Threefish-256 of Skein 1.3; TFGen block = E_key(b,i,m,0), 8 x Word32 out). */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
void sqo_threefish256(const uint64_t key[4], const uint64_t tweak[2], const uint64_t pt[4],
                      int variant, uint64_t out[4]) {
    static const int R[8][2] = { {14,16},{52,57},{23,40},{5,37},{25,33},{46,12},{58,22},{32,32} };
    uint64_t k[5], t[3], x0 = pt[0], x1 = pt[1], x2 = pt[2], x3 = pt[3];
    k[4] = (variant & 2) ? 0x5555555555555555ULL : 0x1BD11BDAA9FC1A22ULL;
    for (int i = 0; i < 4; i++) { k[i] = key[i]; k[4] ^= key[i]; }
    t[0] = tweak[0]; t[1] = tweak[1]; t[2] = t[0] ^ t[1];
    for (int d = 0; d < 72; d++) {
        if ((d & 3) == 0) {
            int s = d >> 2;
            x0 += k[s % 5]; x1 += k[(s + 1) % 5] + t[s % 3];
            x2 += k[(s + 2) % 5] + t[(s + 1) % 3]; x3 += k[(s + 3) % 5] + (uint64_t)s;
        }
        x0 += x1; x1 = rotl64(x1, R[d & 7][0]) ^ x0;
        x2 += x3; x3 = rotl64(x3, R[d & 7][1]) ^ x2;
        uint64_t tmp = x1; x1 = x3; x3 = tmp;            /* word permutation (0,3,2,1) */
    }
    { int s = 18;
      x0 += k[s % 5]; x1 += k[(s + 1) % 5] + t[s % 3];
      x2 += k[(s + 2) % 5] + t[(s + 1) % 3]; x3 += k[(s + 3) % 5] + (uint64_t)s; }
    out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}
/* mkTFGen n = seedTFGen (fromIntegral n,0,0,0): key=(n,0,0,0), counter block (b=0,i=0,m=0,0) */
void sqo_tfgen_words(int64_t seed, int variant, uint32_t out8[8]) {
    uint64_t key[4] = { (uint64_t)seed, 0, 0, 0 }, tw[2] = { 0, 0 }, pt[4] = { 0, 0, 0, 0 }, c[4];
    sqo_threefish256(key, tw, pt, variant, c);
    for (int i = 0; i < 4; i++) {
        uint32_t lo = (uint32_t)c[i], hi = (uint32_t)(c[i] >> 32);
        if (variant & 1) { out8[2 * i] = hi; out8[2 * i + 1] = lo; }
        else             { out8[2 * i] = lo; out8[2 * i + 1] = hi; }
    }
}

/* ======================= Geometry (src/Geometry.hs) ======================= */
/* Data.Matrix (matrix-0.3.5) product: each entry is a dot product folded left-to-right
 * from an accumulator of 0:  r = 0; r = a_k*b_k + r.  (Geometry.hs:91,105) */
static void mat3_mul(const float a[9], const float b[9], float o[9]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        float r = 0.0f;
        for (int k = 0; k < 3; k++) r = a[3 * i + k] * b[3 * k + j] + r;
        o[3 * i + j] = r;
    }
}
void sqo_rot_matrix_rads(float alp, float bet, float gam, int m, float out9[9]) {    /* Geometry.hs:90-102 */
    float ca = sqo_cosf(alp, m), sa = sqo_sinf(alp, m), cb = sqo_cosf(bet, m), sb = sqo_sinf(bet, m);
    float cg = sqo_cosf(gam, m), sg = sqo_sinf(gam, m);
    float A[9] = { ca, -sa, 0, sa, ca, 0, 0, 0, 1 };
    float B[9] = { cb, 0, sb, 0, 1, 0, -sb, 0, cb };
    float C[9] = { 1, 0, 0, 0, cg, -sg, 0, sg, cg };
    float BC[9];
    mat3_mul(B, C, BC);          /* foldr1 (*) [A,B,C] = A * (B * C) */
    mat3_mul(A, BC, out9);
}
sqo_v3 sqo_rot_vert(sqo_v3 v, const float m[9]) {                                     /* Geometry.hs:104-107 */
    float a[3] = { v.x, v.y, v.z }, o[3];
    for (int j = 0; j < 3; j++) { float r = 0.0f; for (int k = 0; k < 3; k++) r = a[k] * m[3 * k + j] + r; o[j] = r; }
    return v3(o[0], o[1], o[2]);
}
static inline V3 tri_normal(const sqo_triangle* t) { return cross(vsub(t->b, t->a), vsub(t->c, t->a)); } /* Geometry.hs:79-80 */

int sqo_moller_trumbore(sqo_v3 rayVert, sqo_v3 rayDir, const sqo_triangle* tri, sqo_v3* point, float* dist) {
    /* Geometry.hs:117-142.  Backtick operators and *^ carry no fixity declaration => infixl 9,
     * tighter than * and +:  v = f * (rayDir.q), t = f * (edge2.q), outInter = rayVert + (t *^ rayDir). */
    const float eps = 0.0001f;
    V3 vertex0 = tri->a, vertex1 = tri->b, vertex2 = tri->c;
    V3 edge1 = vsub(vertex1, vertex0), edge2 = vsub(vertex2, vertex0);
    V3 h = cross(rayDir, edge2);
    float a = dot(edge1, h);
    if (a > -eps && a < eps) return 0;
    float f = 1.0f / a;
    V3 s = vsub(rayVert, vertex0);
    float u = f * dot(s, h);
    if (u < 0 || u > 1) return 0;
    V3 q = cross(s, edge1);
    float v = f * dot(rayDir, q);
    if (v < 0 || u + v > 1) return 0;
    float t = f * dot(edge2, q);
    if (t > eps) {
        V3 outInter = vadd(rayVert, vscale(t, rayDir));
        *point = outInter;
        *dist = norm(vsub(outInter, rayVert));
        return 1;
    }
    return 0;
}
int sqo_intersects_bb(const sqo_bounds* b, sqo_v3 o, sqo_v3 d) {                      /* Geometry.hs:166-177 */
    float dfx = 1.0f / d.x, dfy = 1.0f / d.y, dfz = 1.0f / d.z;
    float t1 = (b->lo.x - o.x) * dfx, t2 = (b->hi.x - o.x) * dfx;
    float t3 = (b->lo.y - o.y) * dfy, t4 = (b->hi.y - o.y) * dfy;
    float t5 = (b->lo.z - o.z) * dfz, t6 = (b->hi.z - o.z) * dfz;
    float tmin = hs_max(hs_max(hs_min(t1, t2), hs_min(t3, t4)), hs_min(t5, t6));
    float tmax = hs_min(hs_min(hs_max(t1, t2), hs_max(t3, t4)), hs_max(t5, t6));
    return tmax > 0 && tmin < tmax;
}
static sqo_bounds get_bounds_tris(const sqo_triangle* t, int n) {                     /* Geometry.hs:155-163,195-197 */
    /* minimum/maximum = foldl1 min/max over the vertex list [a0,b0,c0,a1,...] */
    sqo_bounds b; int first = 1;
    for (int i = 0; i < n; i++) {
        const V3* vs[3] = { &t[i].a, &t[i].b, &t[i].c };
        for (int k = 0; k < 3; k++) {
            V3 v = *vs[k];
            if (first) { b.lo = v; b.hi = v; first = 0; }
            else {
                b.lo = v3(hs_min(b.lo.x, v.x), hs_min(b.lo.y, v.y), hs_min(b.lo.z, v.z));
                b.hi = v3(hs_max(b.hi.x, v.x), hs_max(b.hi.y, v.y), hs_max(b.hi.z, v.z));
            }
        }
    }
    if (first) { b.lo = v3(0, 0, 0); b.hi = v3(0, 0, 0); }  /* Haskell: minimum [] errors; unreachable for n>0 */
    return b;
}
static int longest_axis(sqo_bounds b) {                                               /* Geometry.hs:185-193 */
    /* maximumBy (comparing snd): foldl1, keeps the LATER element unless the earlier is strictly GT */
    float dims[3] = { b.hi.x - b.lo.x, b.hi.y - b.lo.y, b.hi.z - b.lo.z };
    int best = 0;
    for (int i = 1; i < 3; i++) {
        float x = dims[best], y = dims[i];
        int gt = !(x < y) && !(x == y);       /* compare x y == GT */
        if (!gt) best = i;
    }
    return best;
}
static float centroid_ax(const sqo_triangle* t, int ax) {                             /* Geometry.hs:181-182 on `vertices tri` */
    float s = 0.0f;                            /* sum = foldl (+) 0 */
    s = s + proj(ax, t->a); s = s + proj(ax, t->b); s = s + proj(ax, t->c);
    return s / 3.0f;                           /* genericLength [a,b,c] :: Float */
}

/* ======================= BIH (src/BIH.hs) ======================= */
typedef struct node {
    int kind;                /* 0,1,2 = Branch on X,Y,Z ; 3 = Leaf */
    float lmax, rmin;        /* BIHN payload (BIH.hs:37) */
    struct node *l, *r;
    sqo_triangle* tris; int n; int first;   /* Leaf (Vector Triangle); first = offset in flatten order */
} node;
struct sqo_bih { sqo_bounds bounds; node* tree; int n_tris; int n_nodes; sqo_triangle* flat; /* BIH.flatten order */ };

static node* mk_leaf(const sqo_triangle* g, int n) {
    node* nd = (node*)calloc(1, sizeof *nd);
    nd->kind = 3; nd->n = n;
    if (n) { nd->tris = (sqo_triangle*)malloc((size_t)n * sizeof *g); memcpy(nd->tris, g, (size_t)n * sizeof *g); }
    return nd;
}
static node* bih_build(sqo_bounds bbox, const sqo_triangle* geom, int n) {            /* BIH.hs:67-80 */
    const int leafLimit = 15;
    if (n < leafLimit) return mk_leaf(geom, n);
    /* split (BIH.hs:82-99) */
    int ax = longest_axis(bbox);
    float* cen = (float*)malloc((size_t)n * sizeof(float));
    float sum = 0.0f, cnt = 0.0f;
    for (int i = 0; i < n; i++) { cen[i] = centroid_ax(&geom[i], ax); sum = sum + cen[i]; cnt = cnt + 1.0f; }
    float splitPlane = sum / cnt;              /* averagePoints of the centroids, projected (BIH.hs:89-90) */
    sqo_triangle* left = (sqo_triangle*)malloc((size_t)n * sizeof *left);
    sqo_triangle* right = (sqo_triangle*)malloc((size_t)n * sizeof *right);
    int nl = 0, nr = 0;
    for (int i = 0; i < n; i++) { if (cen[i] < splitPlane) left[nl++] = geom[i]; else right[nr++] = geom[i]; }
    free(cen);
    float leftSide = proj(ax, bbox.lo), rightSide = proj(ax, bbox.hi);
    float lm = leftSide, rm = rightSide;       /* maximumDef / minimumDef */
    for (int i = 0; i < nl; i++) {
        float c[3] = { proj(ax, left[i].a), proj(ax, left[i].b), proj(ax, left[i].c) };
        for (int k = 0; k < 3; k++) lm = (i == 0 && k == 0) ? c[k] : hs_max(lm, c[k]);
    }
    for (int i = 0; i < nr; i++) {
        float c[3] = { proj(ax, right[i].a), proj(ax, right[i].b), proj(ax, right[i].c) };
        for (int k = 0; k < 3; k++) rm = (i == 0 && k == 0) ? c[k] : hs_min(rm, c[k]);
    }
    float lmax = 0.001f + lm;                  /* BIH.hs:93 */
    float rmin = (-0.001f) + rm;               /* BIH.hs:95 */
    node* nd = (node*)calloc(1, sizeof *nd);
    nd->kind = ax; nd->lmax = lmax; nd->rmin = rmin;
    if (nl == 0)      { nd->l = mk_leaf(NULL, 0);  nd->r = mk_leaf(right, nr); }       /* BIH.hs:70-72 */
    else if (nr == 0) { nd->l = mk_leaf(left, nl); nd->r = mk_leaf(NULL, 0); }         /* BIH.hs:73-75 */
    else {                                                                             /* BIH.hs:76-78 */
        nd->l = bih_build(get_bounds_tris(left, nl), left, nl);
        nd->r = bih_build(get_bounds_tris(right, nr), right, nr);
    }
    free(left); free(right);
    return nd;
}
static void flatten(const node* nd, sqo_triangle* out);
static void number_leaves(node* nd, int* off, int* count) {
    (*count)++;
    if (nd->kind == 3) { nd->first = *off; *off += nd->n; return; }
    number_leaves(nd->l, off, count); number_leaves(nd->r, off, count);
}
sqo_bih* sqo_make_bih(const sqo_triangle* tris, int n) {                              /* BIH.hs:62-65 */
    sqo_bih* b = (sqo_bih*)calloc(1, sizeof *b);
    b->bounds = get_bounds_tris(tris, n);
    b->tree = bih_build(b->bounds, tris, n);
    int off = 0, cnt = 0; number_leaves(b->tree, &off, &cnt);
    b->n_tris = off; b->n_nodes = cnt;
    b->flat = (sqo_triangle*)malloc((size_t)(off ? off : 1) * sizeof(sqo_triangle));
    flatten(b->tree, b->flat);
    return b;
}
static void free_node(node* nd) { if (!nd) return; free_node(nd->l); free_node(nd->r); free(nd->tris); free(nd); }
void sqo_free_bih(sqo_bih* b) { if (b) { free_node(b->tree); free(b->flat); free(b); } }
static int height(const node* nd) { if (nd->kind == 3) return 1; int a = height(nd->l), c = height(nd->r); return 1 + (a > c ? a : c); }
static int num_leaves(const node* nd) { return nd->kind == 3 ? 1 : num_leaves(nd->l) + num_leaves(nd->r); }
static int longest_leaf(const node* nd) { if (nd->kind == 3) return nd->n; int a = longest_leaf(nd->l), c = longest_leaf(nd->r); return a > c ? a : c; }
int sqo_bih_height(const sqo_bih* b) { return height(b->tree); }
int sqo_bih_num_leaves(const sqo_bih* b) { return num_leaves(b->tree); }
int sqo_bih_longest_leaf(const sqo_bih* b) { return longest_leaf(b->tree); }
int sqo_bih_num_nodes(const sqo_bih* b) { return b->n_nodes; }
int sqo_bih_num_tris(const sqo_bih* b) { return b->n_tris; }
void sqo_bih_bounds(const sqo_bih* b, sqo_bounds* out) { *out = b->bounds; }
static void flatten(const node* nd, sqo_triangle* out) {                              /* BIH.hs:50-52 */
    if (nd->kind == 3) { if (nd->n) memcpy(out + nd->first, nd->tris, (size_t)nd->n * sizeof *out); return; }
    flatten(nd->l, out); flatten(nd->r, out);
}
int sqo_bih_flatten(const sqo_bih* b, sqo_triangle* out) { flatten(b->tree, out); return b->n_tris; }
static void preorder(const node* nd, int* i, int32_t* kind, float* a, float* bb, int32_t* cnt) {
    int me = (*i)++;
    kind[me] = nd->kind; a[me] = nd->lmax; bb[me] = nd->rmin; cnt[me] = nd->kind == 3 ? nd->n : 0;
    if (nd->kind != 3) { preorder(nd->l, i, kind, a, bb, cnt); preorder(nd->r, i, kind, a, bb, cnt); }
}
int sqo_bih_preorder(const sqo_bih* b, int32_t* kind, float* a, float* bb, int32_t* cnt) {
    int i = 0; preorder(b->tree, &i, kind, a, bb, cnt); return i;
}

/* compare (dist x) (dist y) == GT  (GHC.Classes Ord Float: LT if x<y, EQ if x==y, else GT) */
static inline int dist_gt(float x, float y) { return !(x < y) && !(x == y); }
/* minimumBy f = foldl1 (\x y -> case f x y of GT -> y ; _ -> x) : keeps the EARLIER on ties */
static inline sqo_hit min_by_dist(sqo_hit x, sqo_hit y) { return dist_gt(x.dist, y.dist) ? y : x; }

static sqo_counters g_null_counters;
static sqo_hit isect_rec(sqo_bounds bbox, const node* nd, V3 o, V3 d, sqo_counters* c) {   /* BIH.hs:104-141 */
    sqo_hit none; memset(&none, 0, sizeof none); none.tri = -1;
    if (nd->kind == 3) {                                                               /* BIH.hs:105-109 */
        c->leaf_visits++;
        sqo_hit best = none;
        for (int i = 0; i < nd->n; i++) {
            sqo_hit h = none;
            c->tri_tests++;
            if (sqo_moller_trumbore(o, d, &nd->tris[i], &h.point, &h.dist)) {
                h.hit = 1; h.tri = nd->first + i;
                best = best.hit ? min_by_dist(best, h) : h;
            }
        }
        return best;
    }
    c->branch_visits++;
    int ax = nd->kind; float lmax = nd->lmax, rmin = nd->rmin;
    c->slab_tests++;
    if (!sqo_intersects_bb(&bbox, o, d)) return none;                                  /* BIH.hs:112 */
    sqo_bounds left = bbox, right = bbox;                                              /* BIH.hs:130-141 */
    if (ax == 0) { left.hi.x = lmax; right.lo.x = rmin; }
    else if (ax == 1) { left.hi.y = lmax; right.lo.y = rmin; }
    else { left.hi.z = lmax; right.lo.z = rmin; }
    /* Guards are evaluated lazily, in order: `intersectsLeft && intersectsRight` evaluates
     * intersectsRight only if intersectsLeft holds; the third guard then evaluates it otherwise.
     * Either way both get evaluated at most once; count them as the reference would force them. */
    c->slab_tests++;
    int iL = sqo_intersects_bb(&left, o, d);
    c->slab_tests++;
    int iR = sqo_intersects_bb(&right, o, d);
    if (iL && iR) {                                                                    /* BIH.hs:113-116 */
        int leftToRight = proj(ax, d) > 0;                                             /* BIH.hs:127 */
        sqo_hit near = leftToRight ? isect_rec(left, nd->l, o, d, c) : isect_rec(right, nd->r, o, d, c);
        if (near.hit) {
            float p = proj(ax, near.point);
            int isClose = leftToRight ? (p < rmin) : (p > lmax);                       /* BIH.hs:121-123 */
            if (isClose) return near;
            sqo_hit far = leftToRight ? isect_rec(right, nd->r, o, d, c) : isect_rec(left, nd->l, o, d, c);
            return far.hit ? min_by_dist(near, far) : near;                            /* BIH.hs:115,120 */
        }
        return leftToRight ? isect_rec(right, nd->r, o, d, c) : isect_rec(left, nd->l, o, d, c); /* BIH.hs:116 */
    }
    if (iL) return isect_rec(left, nd->l, o, d, c);                                    /* BIH.hs:117 */
    if (iR) return isect_rec(right, nd->r, o, d, c);                                   /* BIH.hs:118 */
    return none;                                                                       /* BIH.hs:119 */
}
void sqo_intersect_bih(const sqo_bih* b, sqo_v3 o, sqo_v3 d, sqo_hit* out, sqo_counters* c) { /* BIH.hs:101-102 */
    if (!c) c = &g_null_counters;
    c->rays++;
    *out = isect_rec(b->bounds, b->tree, o, d, c);
    if (out->hit) c->hits++;
#ifdef SQO_RAY_HOOK              /* experiments under tests/ that include this file observe every ray and its result */
    SQO_RAY_HOOK(b, o, d, out);
#endif
}
static void naive_rec(const node* nd, V3 o, V3 d, sqo_hit* best) {
    if (nd->kind == 3) {
        for (int i = 0; i < nd->n; i++) {
            sqo_hit h; memset(&h, 0, sizeof h);
            if (sqo_moller_trumbore(o, d, &nd->tris[i], &h.point, &h.dist)) {
                h.hit = 1; h.tri = nd->first + i;
                *best = best->hit ? min_by_dist(*best, h) : h;
            }
        }
        return;
    }
    naive_rec(nd->l, o, d, best); naive_rec(nd->r, o, d, best);
}
void sqo_intersect_naive(const sqo_bih* b, sqo_v3 o, sqo_v3 d, sqo_hit* out) {       /* Geometry.hs:110-115 */
    memset(out, 0, sizeof *out); out->tri = -1;
    naive_rec(b->tree, o, d, out);
}

/* ======================= Lib (src/Lib.hs) ======================= */
typedef struct { const sqo_bih* b; int trig, rngv; sqo_counters* c; } ctx_t;

void sqo_make_ray(int w, int h, int y, int x, const sqo_camera* cam, sqo_v3* o, sqo_v3* d) {  /* Lib.hs:107-114 */
    float ww = (float)w, hh = (float)h;
    float xoffs = ((float)x - (ww / 2)) / ww;
    float yoffs = ((hh / 2) - (float)y) / hh;
    *d = sqo_rot_vert(v3(1, xoffs, yoffs), cam->rot);
    *o = cam->pos;
}
static inline float random01(uint32_t n) {                                            /* Lib.hs:183-188 with (lo,hi)=(0,1) */
    float p = (float)n / (float)0xFFFFFFFFu;   /* fromIntegral n / fromIntegral (maxBound :: Word32) */
    float lo = 0.0f, hi = 1.0f, r = hi - lo;
    return lo + r * p;
}
sqo_v3 sqo_random_vector(uint32_t n_u, uint32_t n_v, int m) {                         /* Lib.hs:192-198 */
    float u = random01(n_u), v = random01(n_v);
    float th = 2 * PI_F * u;
    float ph = sqo_acosf(2 * v - 1, m);
    float sph = sqo_sinf(ph, m);
    return v3(sqo_cosf(th, m) * sph, sqo_sinf(th, m) * sph, sqo_cosf(ph, m));
}
static V3 raytrace(const ctx_t* cx, const uint32_t* words, V3 o, V3 d, int bounces) { /* Lib.hs:127-137 */
    V3 black = v3(0, 0, 0);
    if (bounces > 2) return black;
    sqo_hit inter;
    sqo_counters before = *cx->c;
    sqo_intersect_bih(cx->b, o, d, &inter, cx->c);
    if (bounces >= 1) {
        cx->c->b_rays++; cx->c->b_hits += cx->c->hits - before.hits;
        cx->c->b_branch_visits += cx->c->branch_visits - before.branch_visits;
        cx->c->b_tri_tests += cx->c->tri_tests - before.tri_tests;
    }
    if (!inter.hit) return black;
    const sqo_triangle* tri = &cx->b->flat[inter.tri];
    sqo_material mat = tri->mat;
    V3 rec = black;
    if (bounces + 1 <= 2) {            /* newRay is a lazy thunk: forced only if the callee passes its guard */
        /* bounceRay (Lib.hs:155-160): x = first output of THIS gen; scatter reuses the same gen */
        float x = random01(words[0]);
        V3 no, nd;
        no = inter.point;
        if (mat.reflective < x) {      /* scatterRay (Lib.hs:166-172) */
            V3 newDir = sqo_random_vector(words[0], words[1], cx->trig);
            V3 nrm = tri_normal(tri);
            float old = hs_signum(dot(d, nrm));
            float new_ = hs_signum(dot(newDir, nrm));
            nd = (old == new_) ? vneg(newDir) : newDir;
        } else {                       /* reflectRay (Lib.hs:176-181) */
            V3 dn = normalize(tri_normal(tri));
            nd = vsub(d, vscale(2 * dot(dn, d), dn));
        }
        rec = raytrace(cx, words + 1, no, nd, bounces + 1);   /* newGen = snd (next gen) */
    }
    V3 nextBounce = vmul(mat.surf, rec);
    V3 emitContribution = vscale(mat.emissive, mat.emit);
    return vadd(nextBounce, emitContribution);
}
static V3 raycast(const ctx_t* cx, V3 o, V3 d) {                                      /* Lib.hs:141-151 */
    V3 black = v3(0, 0, 0);
    sqo_hit inter;
    sqo_intersect_bih(cx->b, o, d, &inter, cx->c);
    if (!inter.hit) return black;
    sqo_material mat = cx->b->flat[inter.tri].mat;
    V3 light = v3(0, 3, -1);
    V3 sdir = vsub(light, inter.point);                   /* a `to` b = Ray a (b - a), Geometry.hs:146-147 */
    float distanceToLight = norm(vsub(inter.point, light));
    sqo_hit sh;
    sqo_intersect_bih(cx->b, inter.point, sdir, &sh, cx->c);
    if (sh.hit && !(sh.dist > distanceToLight)) return black;    /* guard (Lib.hs:148-150) */
    return vscale(2 / distanceToLight, mat.surf);
}
void sqo_tonemap(sqo_v3 c, int m, uint8_t out3[3]) {                                  /* Lib.hs:93-104 */
    float r = c.x, g = c.y, b = c.z;
    float maxComponent = hs_max(hs_max(r, g), b);
    float minComponent = hs_min(hs_min(r, g), b);
    float lightness = 0.5f * (maxComponent + minComponent);
    float intensity = sqo_atanf(lightness, m) / (PI_F / 2);
    V3 s = vscale(intensity / maxComponent, c);
    float comp[3] = { s.x * 255, s.y * 255, s.z * 255 };
    for (int i = 0; i < 3; i++) {
        /* floor :: Float -> Word8 goes through properFraction/Integer and wraps mod 256;
         * NaN and +-Inf decode to multiples of 2^105, i.e. 0 (this is how black pixels,
         * 0/0 = NaN, come out black). min 255 is then a no-op on Word8. */
        float f = comp[i];
        uint8_t o;
        if (isnan(f) || isinf(f)) o = 0;
        else {
            double fl = floor((double)f);
            double md = fmod(fl, 256.0); if (md < 0) md += 256.0;
            o = (uint8_t)md;
        }
        out3[i] = o < 255 ? o : 255;
    }
}
static V3 render_pixel_avg(const ctx_t* cx, const sqo_camera* cam, int n, int cast, int w, int h, int y, int x) {
    /* Lib.hs:79-89.  dims = (w :. h): the array has w rows, h columns; ix = (y :. x). */
    V3 o, d;
    sqo_make_ray(w, h, y, x, cam, &o, &d);
    int64_t rix = (int64_t)n * ((int64_t)x + (int64_t)y * (int64_t)w);
    V3 sum = v3(0, 0, 0);                      /* sum = foldl (+) (fromInteger 0) */
    for (int k = 0; k < n; k++) {
        V3 oc;
        if (cast) oc = raycast(cx, o, d);
        else {
            uint32_t words[8];
            sqo_tfgen_words(rix + k, cx->rngv, words);
            oc = raytrace(cx, words, o, d, 0);
        }
        if (cx->c) cx->c->samples++;
        sum = vadd(sum, oc);
    }
    return vscale(1 / (float)n, sum);
}
void sqo_sample_radiance(const sqo_bih* b, const sqo_camera* cam, int n, int w, int h, int y, int x, int k,
                         int trig, int rngv, float out3[3]) {
    sqo_counters c; memset(&c, 0, sizeof c);
    ctx_t cx = { b, trig, rngv, &c };
    V3 o, d; sqo_make_ray(w, h, y, x, cam, &o, &d);
    int64_t rix = (int64_t)n * ((int64_t)x + (int64_t)y * (int64_t)w);
    uint32_t words[8]; sqo_tfgen_words(rix + k, rngv, words);
    V3 r = raytrace(&cx, words, o, d, 0);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

typedef struct {
    const sqo_bih* b; const sqo_camera* cam; int n, w, h, cast, y0, y1, ystep, trig, rngv;
    float* avg; uint8_t* rgb; int* next_row; pthread_mutex_t* mu; sqo_counters c;
} job_t;
static void* worker(void* arg) {                                                      /* role of massiv Par, Lib.hs:73-74 */
    job_t* j = (job_t*)arg;
    ctx_t cx = { j->b, j->trig, j->rngv, &j->c };
    /* work item = 32 columns of one row (a row of a 1M-triangle scene at 256 spp is seconds of work: whole rows would
       leave most threads idle on the few-row samples the tests and bench.py time) */
    const int cb = 32, per_row = (j->h + cb - 1) / cb;
    for (;;) {
        pthread_mutex_lock(j->mu); int item = (*j->next_row)++; pthread_mutex_unlock(j->mu);
        int r = item / per_row, x0 = (item % per_row) * cb, x1 = x0 + cb < j->h ? x0 + cb : j->h;
        int y = j->y0 + r * j->ystep;
        if (y >= j->y1) break;
        for (int x = x0; x < x1; x++) {
            V3 a = render_pixel_avg(&cx, j->cam, j->n, j->cast, j->w, j->h, y, x);
            size_t off = ((size_t)r * (size_t)j->h + (size_t)x) * 3;
            if (j->avg) { j->avg[off] = a.x; j->avg[off + 1] = a.y; j->avg[off + 2] = a.z; }
            if (j->rgb) sqo_tonemap(a, j->trig, j->rgb + off);
        }
    }
    return NULL;
}
int sqo_render_rows_strided(const sqo_bih* b, const sqo_camera* cam, int n, int w, int h, int cast, int y0, int y1,
                    int ystep, int threads, int trig, int rngv, float* avg, uint8_t* rgb, sqo_counters* counters) {
    if (!b || !cam) return fail("null scene/camera");
    if (n <= 0 || w <= 0 || h <= 0 || y0 < 0 || y1 > w || y0 > y1 || ystep < 1) return fail("bad dimensions/samples");
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER; int next = 0;
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof *jobs);
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof *th);
    for (int t = 0; t < threads; t++) {
        job_t j = { b, cam, n, w, h, cast, y0, y1, ystep, trig, rngv, avg, rgb, &next, &mu, { 0 } };
        jobs[t] = j;
        if (threads == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    sqo_counters tot; memset(&tot, 0, sizeof tot);
    for (int t = 0; t < threads; t++) {
        if (threads > 1) pthread_join(th[t], NULL);
        tot.samples += jobs[t].c.samples; tot.rays += jobs[t].c.rays; tot.branch_visits += jobs[t].c.branch_visits;
        tot.slab_tests += jobs[t].c.slab_tests; tot.leaf_visits += jobs[t].c.leaf_visits;
        tot.tri_tests += jobs[t].c.tri_tests; tot.hits += jobs[t].c.hits;
        tot.b_rays += jobs[t].c.b_rays; tot.b_branch_visits += jobs[t].c.b_branch_visits;
        tot.b_tri_tests += jobs[t].c.b_tri_tests; tot.b_hits += jobs[t].c.b_hits;
    }
    if (counters) *counters = tot;
    free(jobs); free(th);
    return 0;
}
int sqo_render_rows(const sqo_bih* b, const sqo_camera* cam, int n, int w, int h, int cast, int y0, int y1,
                    int threads, int trig, int rngv, float* avg, uint8_t* rgb, sqo_counters* counters) {
    return sqo_render_rows_strided(b, cam, n, w, h, cast, y0, y1, 1, threads, trig, rngv, avg, rgb, counters);
}
int sqo_render(const sqo_bih* b, const sqo_camera* cam, int n, int w, int h, int cast, int threads, int trig,
               int rngv, float* avg, uint8_t* rgb, sqo_counters* counters) {
    return sqo_render_rows(b, cam, n, w, h, cast, 0, w, threads, trig, rngv, avg, rgb, counters);
}

/* ======================= Obj loader (src/Obj.hs) =======================
 * A character-level restatement of the Parsec grammar.  Where the Haskell would build a
 * lazy `read` thunk on malformed digits (and crash only if forced), this returns an error. */
typedef struct { const char* s; size_t n, i; } P;
static int is_space(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }
static int is_digit(int c) { return c >= '0' && c <= '9'; }
static int is_alnum(int c) { return is_digit(c) || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
static int pk(const P* p) { return p->i < p->n ? (unsigned char)p->s[p->i] : -1; }
static void spaces(P* p) { while (is_space(pk(p))) p->i++; }
/* Parsec `string`: 0 = matched; 1 = failed without consuming; 2 = failed after consuming */
static int p_string(P* p, const char* lit) {
    size_t k = 0;
    while (lit[k]) { if (pk(p) != (unsigned char)lit[k]) return k == 0 ? 1 : 2; p->i++; k++; }
    return 0;
}
static int p_fractional(P* p, float* out) {                                           /* Obj.hs:115-121 */
    char buf[400]; size_t k = 0; int nd1 = 0, nd2 = 0, dot_ = 0;
    if (pk(p) == '-') { buf[k++] = '-'; p->i++; }
    while (is_digit(pk(p)) && k < sizeof buf - 2) { buf[k++] = (char)pk(p); p->i++; nd1++; }
    if (pk(p) == '.') { buf[k++] = '.'; p->i++; dot_ = 1; }
    while (is_digit(pk(p)) && k < sizeof buf - 2) { buf[k++] = (char)pk(p); p->i++; nd2++; }
    buf[k] = 0;
    if (nd1 == 0 || (dot_ && nd2 == 0)) return fail("read: no parse on number \"%s\" at offset %zu", buf, p->i);
    *out = strtof(buf, NULL);                  /* correctly rounded, as read/fromRational */
    return 0;
}
static int p_vec3(P* p, V3* v) {                                                      /* Obj.hs:166-171 */
    if (p_fractional(p, &v->x)) return 1;
    spaces(p);
    if (p_fractional(p, &v->y)) return 1;
    spaces(p);
    if (p_fractional(p, &v->z)) return 1;
    spaces(p);
    return 0;
}
static int p_word(P* p, char* out, size_t cap) {                                      /* Obj.hs:129-130 */
    size_t k = 0;
    while (pk(p) >= 0 && !is_space(pk(p))) { if (k + 1 < cap) out[k++] = (char)pk(p); p->i++; }
    out[k] = 0;
    if (k == 0) return fail("expected a word at offset %zu", p->i);
    spaces(p);
    return 0;
}
typedef struct { int i1, i2, i3; } face_t;
typedef struct { size_t v0, nv; char mtl[256]; size_t f0, nf; } object_t;
typedef struct { V3* verts; size_t nv, cv; face_t* faces; size_t nf, cf; object_t* objs; size_t no, co; char mtllib[256]; } objfile_t;
static void objfile_free(objfile_t* f) { free(f->verts); free(f->faces); free(f->objs); }
#define PUSH(arr, n, c, val) do { if ((n) == (c)) { (c) = (c) ? (c) * 2 : 64; (arr) = realloc((arr), (c) * sizeof *(arr)); } (arr)[(n)++] = (val); } while (0)

static int parse_obj(const char* text, size_t len, objfile_t* f) {                    /* Obj.hs:96-144 */
    memset(f, 0, sizeof *f);
    P p = { text, len, 0 };
    int r = p_string(&p, "mtllib");                                                   /* Obj.hs:126-127 */
    if (r) return fail("obj: expected \"mtllib\" at offset %zu", p.i);
    spaces(&p);
    if (p_word(&p, f->mtllib, sizeof f->mtllib)) return 1;
    for (;;) {                                                                        /* many parseObj */
        if (pk(&p) != 'o') break;              /* objectName: char 'o' fails without consuming => many stops */
        p.i++; spaces(&p);
        size_t k = 0;
        while (is_alnum(pk(&p)) || pk(&p) == '.' || pk(&p) == '_') { p.i++; k++; }
        if (k == 0) return fail("obj: empty object name at offset %zu", p.i);
        spaces(&p);
        object_t ob; memset(&ob, 0, sizeof ob);
        ob.v0 = f->nv; ob.f0 = f->nf;
        while (pk(&p) == 'v') {                                                       /* many vertex (Obj.hs:109-110) */
            p.i++; spaces(&p);
            V3 v; if (p_vec3(&p, &v)) return 1;
            V3 sw = v3(v.x, v.z, v.y);                                                /* swapYZ (Obj.hs:112-113) */
            PUSH(f->verts, f->nv, f->cv, sw); ob.nv++;
        }
        if (p_string(&p, "usemtl")) return fail("obj: expected \"usemtl\" at offset %zu", p.i);  /* Obj.hs:123-124 */
        spaces(&p);
        if (p_word(&p, ob.mtl, sizeof ob.mtl)) return 1;
        /* optional parseS (Obj.hs:132-133): try (string "s on") <|> string "s off" */
        if (pk(&p) == 's') {
            size_t save = p.i;
            if (p_string(&p, "s on") != 0) {
                p.i = save;
                if (p_string(&p, "s off") != 0) return fail("obj: bad smoothing line at offset %zu", p.i);
            }
            spaces(&p);
        }
        while (pk(&p) == 'f') {                                                       /* many face (Obj.hs:135-144) */
            p.i++; spaces(&p);
            int idx[3];
            for (int q = 0; q < 3; q++) {
                if (!is_digit(pk(&p))) return fail("obj: expected face index at offset %zu", p.i);
                long long val = 0;
                while (is_digit(pk(&p))) { val = val * 10 + (pk(&p) - '0'); if (val > 2000000000LL) val = 2000000000LL; p.i++; }
                spaces(&p);
                idx[q] = (int)val;
            }
            face_t fc = { idx[0], idx[1], idx[2] };
            PUSH(f->faces, f->nf, f->cf, fc); ob.nf++;
        }
        PUSH(f->objs, f->no, f->co, ob);
    }
    return 0;
}
typedef struct { char name[256]; sqo_material mat; } namedmat_t;
static int parse_sq(const char* text, size_t len, namedmat_t** out, size_t* n_out) {  /* Obj.hs:146-161 */
    P p = { text, len, 0 };
    namedmat_t* arr = NULL; size_t n = 0, c = 0;
    for (;;) {
        int r = p_string(&p, "newmtl ");
        if (r == 1) break;                     /* failed without consuming: many stops */
        if (r == 2) { free(arr); return fail("sq: expected \"newmtl \" at offset %zu", p.i); }
        namedmat_t m; memset(&m, 0, sizeof m);
        if (p_word(&p, m.name, sizeof m.name)) { free(arr); return 1; }
        spaces(&p);
        if (p_string(&p, "reflective ")) { free(arr); return fail("sq: expected \"reflective \" at offset %zu", p.i); }
        if (p_fractional(&p, &m.mat.reflective)) { free(arr); return 1; }
        spaces(&p);
        if (p_vec3(&p, &m.mat.surf)) { free(arr); return 1; }
        spaces(&p);
        if (p_string(&p, "emissive ")) { free(arr); return fail("sq: expected \"emissive \" at offset %zu", p.i); }
        if (p_fractional(&p, &m.mat.emissive)) { free(arr); return 1; }
        spaces(&p);
        if (p_vec3(&p, &m.mat.emit)) { free(arr); return 1; }
        spaces(&p);
        PUSH(arr, n, c, m);
    }
    *out = arr; *n_out = n;
    return 0;
}
int sqo_mtllib_of_text(const char* obj_text, size_t obj_len, char* name, size_t cap) {
    objfile_t f; if (parse_obj(obj_text, obj_len, &f)) { objfile_free(&f); return 1; }
    snprintf(name, cap, "%s", f.mtllib); objfile_free(&f); return 0;
}
int sqo_tris_from_text(const char* obj_text, size_t obj_len, const char* sq_text, size_t sq_len,
                       sqo_triangle** out, int* n_out) {                              /* Obj.hs:49-58,73-86 */
    objfile_t f; namedmat_t* mats = NULL; size_t nm = 0;
    *out = NULL; *n_out = 0;
    if (parse_obj(obj_text, obj_len, &f)) { objfile_free(&f); return 1; }
    if (parse_sq(sq_text, sq_len, &mats, &nm)) { objfile_free(&f); return 1; }
    sqo_triangle* tris = NULL; size_t nt = 0, ct = 0;
    /* matches = [(obj, mat) | obj <- objs, mat <- mats, mtl obj == fst mat]  (Obj.hs:75) */
    for (size_t oi = 0; oi < f.no; oi++) for (size_t mi = 0; mi < nm; mi++) {
        if (strcmp(f.objs[oi].mtl, mats[mi].name) != 0) continue;
        for (size_t k = 0; k < f.objs[oi].nf; k++) {                                  /* makeTris (Obj.hs:80-86) */
            face_t fc = f.faces[f.objs[oi].f0 + k];
            int ids[3] = { fc.i1, fc.i2, fc.i3 };
            for (int q = 0; q < 3; q++)
                if (ids[q] < 1 || (size_t)ids[q] > f.nv) {                            /* (!!) index error */
                    free(tris); free(mats); objfile_free(&f);
                    return fail("obj: face index %d out of range 1..%zu", ids[q], f.nv);
                }
            sqo_triangle t; t.a = f.verts[ids[0] - 1]; t.b = f.verts[ids[1] - 1]; t.c = f.verts[ids[2] - 1];
            t.mat = mats[mi].mat;
            PUSH(tris, nt, ct, t);
        }
    }
    free(mats); objfile_free(&f);
    *out = tris; *n_out = (int)nt;
    return 0;
}
static char* slurp(const char* path, size_t* len) {
    FILE* fp = fopen(path, "rb"); if (!fp) { fail("cannot open %s", path); return NULL; }
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    char* buf = (char*)malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, fp) != (size_t)n) { fclose(fp); free(buf); fail("short read on %s", path); return NULL; }
    fclose(fp); buf[n] = 0; *len = (size_t)n; return buf;
}
int sqo_tris_from_obj(const char* obj_path, const char* mtl_dir, sqo_triangle** out, int* n_out) {
    size_t ol, sl; char* ot = slurp(obj_path, &ol); if (!ot) return 1;
    char name[256];
    if (sqo_mtllib_of_text(ot, ol, name, sizeof name)) { free(ot); return 1; }
    char path[1024]; snprintf(path, sizeof path, "%s/%s", mtl_dir, name);            /* "./data/" ++ mtllib' (Obj.hs:52) */
    char* st = slurp(path, &sl); if (!st) { free(ot); return 1; }
    int r = sqo_tris_from_text(ot, ol, st, sl, out, n_out);
    free(ot); free(st); return r;
}
int sqo_camera_from_text(const char* text, size_t len, int trig_mode, sqo_camera* cam) {             /* Obj.hs:67-70 */
    P p = { text, len, 0 };
    V3 pos, e;
    if (p_vec3(&p, &pos)) return 1;
    if (p_vec3(&p, &e)) return 1;
    cam->pos = pos;
    sqo_rot_matrix_rads(e.x, e.y, e.z, trig_mode, cam->rot);
    return 0;
}
int sqo_load_camera(const char* path, int trig_mode, sqo_camera* cam) {
    size_t l; char* t = slurp(path, &l); if (!t) return 1;
    int r = sqo_camera_from_text(t, l, trig_mode, cam); free(t); return r;
}
