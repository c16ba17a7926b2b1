/* sq_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of rrruko/squigly-trace's per-pixel sampling path, written directly
 * from the Haskell source (citations are relative to /root/reference).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; nothing
 * under squigly-trace_amd/ links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" against GHC output.  No GHC/stack/cabal toolchain exists
 * in the build or GPU images, the reference ships no tests/golden vectors (test/Spec.hs:1-2
 * is a stub) and its RNG (tf-random) is a third-party package whose source is not on disk.
 * What IS pinned: Threefish-256 by the public Skein-1.3 known-answer vectors, BIH shape by
 * the reference's own --debug statistics as restated in SURVEY.md App. C, and the image
 * statistically by render/example.png (tests/test_oracle_*.py).
 */
#ifndef SQ_ORACLE_H
#define SQ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } sqo_v3;                                   /* src/V3.hs:5 */
typedef struct { float reflective; sqo_v3 surf; float emissive; sqo_v3 emit; } sqo_material; /* src/Color.hs:78-83 */
typedef struct { sqo_v3 a, b, c; sqo_material mat; } sqo_triangle;          /* src/Geometry.hs:49-54 */
typedef struct { sqo_v3 pos; float rot[9]; } sqo_camera;                    /* src/Geometry.hs:41 (row-major 3x3) */
typedef struct { sqo_v3 lo, hi; } sqo_bounds;                               /* src/Geometry.hs:153 */
typedef struct { sqo_v3 point; float dist; int tri; int hit; } sqo_hit;     /* src/Geometry.hs:71-75; tri = index in flatten order */
typedef struct sqo_bih sqo_bih;                                             /* src/BIH.hs:41-44 */

typedef struct {
    uint64_t samples;        /* (pixel,sample) pairs evaluated */
    uint64_t rays;           /* calls of Scene.intersect (src/Lib.hs:131,143,150) */
    uint64_t branch_visits;  /* intersectBIH' Branch equations entered (src/BIH.hs:111) */
    uint64_t slab_tests;     /* intersectsBB calls (src/Geometry.hs:166) */
    uint64_t leaf_visits;    /* intersectBIH' Leaf equations entered (src/BIH.hs:105) */
    uint64_t tri_tests;      /* mollerTrumbore calls (src/Geometry.hs:117) */
    uint64_t hits;           /* rays that returned Just (one material fetch each, src/Lib.hs:132) */
    /* the same four for bounce rays only (raytrace called with bounces >= 1, src/Lib.hs:135) */
    uint64_t b_rays, b_branch_visits, b_tri_tests, b_hits;
} sqo_counters;

/* trig_mode: how Float sin/cos/acos/atan are evaluated.
 *   SQO_TRIG_CRD  : the repo's "crd" spec (evaluate in binary64 with a fixed operation
 *                   sequence, round once to binary32) — identical on host and device.
 *   SQO_TRIG_LIBM : the host libm's sinf/cosf/acosf/atanf, which is what GHC's Float
 *                   instances call (informational: quantifies crd-vs-libm differences). */
enum { SQO_TRIG_CRD = 0, SQO_TRIG_LIBM = 1 };
/* rng_variant: TFGen layout candidates of SURVEY.md Appendix B.
 *   bit0 = 1: extract the HIGH 32 bits of each 64-bit word first (default 0: low first)
 *   bit1 = 1: key-schedule parity 0x5555555555555555 (Skein <= 1.2) instead of C240 (1.3) */
enum { SQO_RNG_DEFAULT = 0 };

const char* sqo_last_error(void);

/* ---- loaders (src/Obj.hs) ---- */
int  sqo_tris_from_text(const char* obj_text, size_t obj_len, const char* sq_text, size_t sq_len,
                        sqo_triangle** out, int* n_out);                    /* Obj.hs:49-58,73-86 given both texts */
int  sqo_mtllib_of_text(const char* obj_text, size_t obj_len, char* name, size_t cap); /* Obj.hs:126-127 */
int  sqo_tris_from_obj(const char* obj_path, const char* mtl_dir, sqo_triangle** out, int* n_out);
int  sqo_camera_from_text(const char* text, size_t len, int trig_mode, sqo_camera* cam);   /* Obj.hs:60-70 */
int  sqo_load_camera(const char* path, int trig_mode, sqo_camera* cam);
void sqo_free(void* p);

/* ---- geometry / math units ---- */
void sqo_rot_matrix_rads(float a, float b, float g, int trig_mode, float out9[9]); /* Geometry.hs:90-102 */
sqo_v3 sqo_rot_vert(sqo_v3 v, const float m9[9]);                           /* Geometry.hs:104-107 */
int  sqo_intersects_bb(const sqo_bounds* b, sqo_v3 o, sqo_v3 d);            /* Geometry.hs:166-177 */
int  sqo_moller_trumbore(sqo_v3 o, sqo_v3 d, const sqo_triangle* t, sqo_v3* point, float* dist); /* Geometry.hs:117-142 */
float sqo_sinf(float x, int trig_mode);
float sqo_cosf(float x, int trig_mode);
float sqo_acosf(float x, int trig_mode);
float sqo_atanf(float x, int trig_mode);
double sqo_sin_d(double x); double sqo_cos_d(double x); double sqo_acos_d(double x); double sqo_atan_d(double x);
void sqo_threefish256(const uint64_t key[4], const uint64_t tweak[2], const uint64_t pt[4],
                      int rng_variant, uint64_t out[4]);                    /* tf-random cbits, Skein 1.3 */
void sqo_tfgen_words(int64_t seed, int rng_variant, uint32_t out8[8]);      /* mkTFGen seed; 8 x next (Lib.hs:86,134,185) */
void sqo_tonemap(sqo_v3 c, int trig_mode, uint8_t out3[3]);                 /* Lib.hs:93-104 */
void sqo_make_ray(int w, int h, int y, int x, const sqo_camera* cam, sqo_v3* o, sqo_v3* d); /* Lib.hs:107-114 */
sqo_v3 sqo_random_vector(uint32_t n_u, uint32_t n_v, int trig_mode);        /* Lib.hs:183-198 */

/* ---- BIH (src/BIH.hs) ---- */
sqo_bih* sqo_make_bih(const sqo_triangle* tris, int n);                     /* BIH.hs:62-99 */
void sqo_free_bih(sqo_bih* b);
int  sqo_bih_height(const sqo_bih* b);                                      /* BIH.hs:46-48 */
int  sqo_bih_num_leaves(const sqo_bih* b);                                  /* BIH.hs:54-56 */
int  sqo_bih_longest_leaf(const sqo_bih* b);                                /* BIH.hs:58-60 */
int  sqo_bih_num_nodes(const sqo_bih* b);
int  sqo_bih_num_tris(const sqo_bih* b);
void sqo_bih_bounds(const sqo_bih* b, sqo_bounds* out);                     /* BIH.hs:42 */
int  sqo_bih_flatten(const sqo_bih* b, sqo_triangle* out);                  /* BIH.hs:50-52 */
/* pre-order dump: kind[i] = 0/1/2 branch on X/Y/Z, 3 = leaf; a[i],bb[i] = lmax,rmin (branch) ;
 * cnt[i] = leaf triangle count (leaf) */
int  sqo_bih_preorder(const sqo_bih* b, int32_t* kind, float* a, float* bb, int32_t* cnt);
void sqo_intersect_bih(const sqo_bih* b, sqo_v3 o, sqo_v3 d, sqo_hit* out, sqo_counters* c);   /* BIH.hs:101-141 */
void sqo_intersect_naive(const sqo_bih* b, sqo_v3 o, sqo_v3 d, sqo_hit* out);                  /* Geometry.hs:110-115 over flatten order */

/* ---- the hot path (src/Lib.hs:68-198) ---- */
/* out_avg: [w][h][3] float = `avg` of Lib.hs:88 (may be NULL); out_rgb: [w][h][3] u8 = Lib.hs:89 (may be NULL).
 * Image has w ROWS and h COLUMNS (massiv Ix2 quirk, Lib.hs:70-71,80). */
int  sqo_render(const sqo_bih* b, const sqo_camera* cam, int samples, int w, int h, int cast,
                int threads, int trig_mode, int rng_variant,
                float* out_avg, uint8_t* out_rgb, sqo_counters* counters);
/* rows [y0,y1) only, same output layout offset to row y0 (out has (y1-y0)*h*3 entries) */
int  sqo_render_rows(const sqo_bih* b, const sqo_camera* cam, int samples, int w, int h, int cast,
                     int y0, int y1, int threads, int trig_mode, int rng_variant,
                     float* out_avg, uint8_t* out_rgb, sqo_counters* counters);
/* rows y0, y0+ystep, ... (< y1); output rows are compact in that order */
int  sqo_render_rows_strided(const sqo_bih* b, const sqo_camera* cam, int samples, int w, int h, int cast,
                     int y0, int y1, int ystep, int threads, int trig_mode, int rng_variant,
                     float* out_avg, uint8_t* out_rgb, sqo_counters* counters);
/* one sample's radiance: raytrace (mkTFGen (n*(x+y*w)+k)) scene ray 0  (Lib.hs:84-87) */
void sqo_sample_radiance(const sqo_bih* b, const sqo_camera* cam, int samples, int w, int h,
                         int y, int x, int k, int trig_mode, int rng_variant, float out3[3]);

#ifdef __cplusplus
}
#endif
#endif
