/* squigly_hip.h — C-ABI of libsquigly_hip.so: the MI355X drop-in for squigly-trace's
 * per-pixel sampling loop.
 *
 * The reference (rrruko/squigly-trace) has no FFI.  Its hot path sits behind
 *     render :: Scene a -> Camera -> Settings -> IO ()            (src/Lib.hs:68-75)
 * and the narrowest seam is src/Lib.hs:73-74,
 *     img = computeAs S (makeArray Par (w :. h) (renderPixel scene cam samples cast dims))
 * i.e. "fill a w-rows x h-columns buffer of Pixel RGB Word8".  sq_render_rgb8() below is
 * what a `foreign import ccall safe` at that line binds (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain C, no exceptions, no ownership transfer.  Every function returns 0 on success,
 *    non-zero on failure; sq_last_error() then holds a message (thread-local).
 *  - all input arrays are caller-owned, read-only, valid for the duration of the call.
 *  - nothing is written to an output buffer on failure.
 *  - there is NO CPU fallback: without a usable HIP device every render entry point fails.
 *  - citations are relative to the reference repository root.
 */
#ifndef SQUIGLY_HIP_H
#define SQUIGLY_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SQ_ABI_VERSION 1

/* ---- data model (kept from the reference) ---- */
typedef struct { float lo[3], hi[3]; } sq_bounds;            /* Bounds, src/Geometry.hs:153 */

/* One BIH tree node, nodes stored in PRE-ORDER (a branch's left child is the next node).
 * Tree a b / BIHNode, src/BIH.hs:26,37-40.
 *   kind & 3 : 0,1,2 = Branch splitting on X,Y,Z ; 3 = Leaf
 *   branch   : lmax, rmin = BIHN payload ; link = index of the RIGHT child
 *   leaf     : link = index of its first triangle in `tris` (BIH.flatten order, src/BIH.hs:50-52);
 *              kind >> 2 = number of triangles (0 allowed: src/BIH.hs:70-75) */
typedef struct { int32_t kind; float lmax, rmin; int32_t link; } sq_node;

typedef struct { float v0[3], v1[3], v2[3]; int32_t mat; } sq_tri;          /* Triangle, src/Geometry.hs:49-54 (material by index) */
typedef struct { float reflective, surf[3], emissive, emit[3]; } sq_material; /* Material, src/Color.hs:78-83 */
typedef struct { float pos[3]; float rot[9]; } sq_camera;    /* Camera, src/Geometry.hs:41; rot row-major = rotMatrixRads a b g (:90-102) */

typedef struct {
    sq_bounds          root;      /* bounds of BIH, src/BIH.hs:42 */
    const sq_node*     nodes;     int32_t n_nodes;
    const sq_tri*      tris;      int32_t n_tris;    /* leaf order */
    const sq_material* mats;      int32_t n_mats;
    int32_t            height;    /* BIH.height, src/BIH.hs:46-48 (sizes the traversal stack); 0 = compute */
} sq_scene;

/* ---- one-shot drop-in for src/Lib.hs:73-74 ---- */
/* The reference host is one process, so the one-shot calls put a node's GPUs to work themselves: rows are cut
 * into interleaved blocks of 2 rows (the sq_shard scheme below), one host thread per device renders its shard, and
 * the shards are de-interleaved into `out`; no exchange between devices is needed (a pixel depends only on
 * x, y, samples, w).  Devices: all visible ones when the frame has >= 2^24 samples, else device 0; the
 * environment variable SQ_DEVICES="0,1,3" names them explicitly (an index may repeat).
 * out: w*h*3 bytes, row-major, w ROWS x h COLUMNS (massiv `w :. h`, src/Lib.hs:70-71,80), RGB8 =
 * rgbFloatToPixelRGB of each pixel (src/Lib.hs:93-104).  cast != 0 selects raycast (src/Lib.hs:141-151).
 * With one device the finished frame is copied straight into `out`; with several, each shard is staged and de-interleaved.  Either way
 * nothing is written to `out` unless every device's render succeeded.  SQ_ONESHOT_TIMING=1 prints the call's stages (scene
 * upload, buffers, render, copy back, free) on stderr. */
int sq_render_rgb8(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h,
                   int32_t cast, uint8_t* out);
/* out_avg: w*h*3 floats = the pre-tonemap `avg` of src/Lib.hs:88 (for tolerance checks). */
int sq_render_f32(const sq_scene* scene, const sq_camera* cam, int32_t samples, int32_t w, int32_t h,
                  int32_t cast, float* out_avg);

/* Frame workspaces (15 GB for a 1080p frame at 256 spp, at most 24 GB) are kept, one block per device, when a scene is
 * freed, so that repeated one-shot calls do not re-allocate them (a hipMalloc right after the hipFree of a block
 * that large can wait seconds for the driver to scrub it).  This hands them back to the driver. */
void sq_release_cached_memory(void);

/* ---- resident API: scene stays in HBM, output stays on the device (bench, multi-GPU) ---- */
typedef struct sq_device_scene sq_device_scene;
int  sq_scene_upload(const sq_scene* scene, int32_t device, sq_device_scene** out);
void sq_scene_free(sq_device_scene* s);

/* Row sharding (role of massiv's Par scheduler, src/Lib.hs:73): the w image rows are cut into
 * blocks of `row_block` rows and block b belongs to shard (b % n_shards).  A shard renders its rows
 * into a COMPACT buffer of sq_shard_rows() rows, local row j <-> global row sq_shard_global_row(). */
typedef struct { int32_t row_block, shard, n_shards; } sq_shard;
int32_t sq_shard_rows(int32_t w, sq_shard sh);
int32_t sq_shard_global_row(int32_t local_row, sq_shard sh);

/* d_avg (float, rows*h*3) and d_rgb (uint8, rows*h*3) are DEVICE pointers on the scene's device;
 * either may be NULL.  hip_stream is a hipStream_t (NULL = the null stream); the call only enqueues
 * work on it and returns (no synchronisation). */
int sq_render_rows_device(sq_device_scene* s, const sq_camera* cam, int32_t samples, int32_t w, int32_t h,
                          int32_t cast, sq_shard sh, float* d_avg, uint8_t* d_rgb, void* hip_stream);

/* Timing of the dominant kernel measured with hipEvents on the stream it was launched on:
 * average duration in ms over the launches since the last reset, and the launch count. */
int  sq_kernel_timing(sq_device_scene* s, double* avg_ms, int64_t* launches, const char** kernel_name);
void sq_kernel_timing_reset(sq_device_scene* s);
/* Cumulative statistics of the trace kernel since the last reset (synchronises the device), n <= 32:
 * out[0] = rays traced; out[1..23] = lane-occupancy counters, rare-path counts and per-section wave cycles of the
 * profile build, filled only with option "profile" = 1 (tools/gpu_pool.py prints them). */
int  sq_get_stats(sq_device_scene* s, uint64_t* out, int32_t n, int32_t reset);
/* Tunables; every setting produces identical bits.  Keys:
 *   "variant"            1 = one-lane-per-pixel kernel, 2 = wavefront pipeline (default)
 *   "slots"              sample slots of the frame workspace (default 512 Mi at 45 B each = 24 GB of the
 *                        288 GB; a frame with fewer samples allocates only what it needs)
 *   "resident"           1 = keep the whole scene in LDS when it fits (default), 0 = always stream
 *   "lds_node_kb"        streaming form: KB of LDS for the top of the tree (default 32; the six-wave build takes what its third of the LDS leaves)
 *   "pool"               1 = pooled trace kernel (default): a wave tests the triangles of all its open leaves as a pool of
 *                        (ray, triangle) pairs spread over its 64 lanes; 0 = every lane walks its own leaf
 *   "refill_min"         pooled kernel: idle lanes a wave collects before it fetches new rays (default 12)
 *   "flush_min"          pooled kernel: a trailing part-filled window of pairs runs at once from this many pairs on,
 *                        otherwise it waits one iteration for more (default 40)
 *   "pixel_major"        order in which a trace launch takes its queue: 0 = slot order (neighbouring pixels, one sample each),
 *                        1 = all samples of a pixel in a row (a wave's rays start at one surface point), -1 = choose (default:
 *                        1 when the triangles exceed the 4 MB L2s, else 0)
 *   "primary_resident"   1 = with a resident scene the primary rays are traced out of LDS too (default), 0 = from L2
 *   "primary_pooled"     1 = the primary rays go through the pooled trace kernel (one slot per pixel, one launch) instead of the
 *                        one-ray-per-lane pass: one rank's share of the headline frame at 8 ranks 8.20 -> 8.12 ms, the whole
 *                        frame 54.2 -> 54.4 ms; default 0
 *   "guided"             bit 0 (default 1): queue reservations of the first-bounce launches shrink towards the end of the queue, so that a
 *                        launch does not end with a few waves still working through a full reservation; bit 1: the same for the
 *                        second-bounce launches -- off by default since round 3: their queue is 8 % live, a shrunken reservation brings
 *                        a handful of rays for the same atomic and scan round trip, and every such launch took 0.34 ms longer with it
 *                        (one rank's share at 8 ranks: 1166 -> 825 us; whole frame: 4886 -> 4542 us); 0 = fixed size everywhere
 *   "straggler_lanes"    pool = 0: lanes still traversing when a wave turns to its leaves (default 8)
 *   "trace_blocks_per_cu" streaming form: 512-thread workgroups per CU.  0 (default) = three, with the kernel compiled for six waves per
 *                        SIMD (80 VGPRs), when a workgroup's stacks plus at least 4 KB of the tree's top fit in a third of the LDS, else
 *                        two with the plain build (86 VGPRs, up to lds_node_kb of tree); 1 / 2 = the plain build; 3 = the six-wave build if it fits
 *   "timing"             1 = bracket the dominant kernel with hipEvents for sq_kernel_timing (default 0)
 *   "profile"            1 = lane-occupancy counters in sq_get_stats (slower)
 *   "overlap"            0 = one stream (default; per-kernel durations stay clean for the roofline)
 *                        1 = two sample batches in flight: trace launches on the caller's stream, the per-sample
 *                            kernels beside them on an internal stream
 *                        2 = two pipelines: even and odd sample batches run start to end on two streams, so that one
 *                            track's launches fill the other's ramp-downs (1 and 2: -3 % on the whole headline frame, nothing on
 *                            half a frame or less, so both stay opt-in)
 *   "aux_blocks_per_cu"  workgroups per CU of the per-sample kernels (0 = default 8)
 *   "coresidency"        diagnostic, default 0: the trace kernel keeps a gauge of its live workgroups and every wave of the per-sample
 *                        kernels (RNG + bounce, shading) records whether it started / ended while (CUs - 8) or more of them were live,
 *                        i.e. beside a resident trace workgroup on its own CU; sq_get_stats slots 24..27 = gauge, per-sample waves,
 *                        started beside, ended beside (tools/gpu_overlap.py prints them per schedule)
 *   "descend_extra"      pooled trace kernel: further branch steps (default 2) a lane that keeps descending takes within one iteration,
 *   "descend_lanes"      each taken only while at least this many lanes (default 16) of the wave want one
 *   "cull"               1 (default): a ray inside the limits of sq_cull_boxes (squigly_host.h) that misses a leaf's culling box
 *                        skips the leaf's triangle tests -- the reference's mollerTrumbore would reject them all, so no bit
 *                        changes; 0: every leaf the reference visits is tested
 *   "primary_tiles"      1 (default): the primary rays of a shard are enumerated in tiles (8 x 8 pixels on a whole image, 2 x 32 with
 *                        row blocks of 2) so that the 64 rays of a wave stay together in both image directions; 0: 64 pixels of a row
 *   "incremental"        only in builds with -DSQ_RES_INCREMENTAL=1 (measured and rejected): the resident form carries
 *                        (tmin, tmax) of the reference's slab test down the tree instead of testing both children from scratch */
int  sq_set_option(sq_device_scene* s, const char* key, int64_t value);

/* Diagnostics for the numeric spec (tests only): evaluate one primitive on the device for n inputs.
 *   SQ_OP_SQRT/SIN/COS/ACOS/ATAN : a = n floats -> out = n floats        (b unused)
 *   SQ_OP_DIV                    : a, b = n floats -> out = a/b
 *   SQ_OP_UNIT_FLOAT             : a = n uint32 -> out = n floats (randomR (0,1), src/Lib.hs:183-188)
 *   SQ_OP_TFGEN3                 : a = n int64 seeds -> out = 3n uint32 (first three outputs of mkTFGen seed)
 *   SQ_OP_TONEMAP                : a = 3n floats -> out = 3n bytes (src/Lib.hs:93-104)
 *   SQ_OP_RCP_SWEEP              : a = n uint32 (upper 16 bits of a float) -> out = n uint32: among the 65536 floats x with
 *                                  those upper bits, how many have the triangle test's short reciprocal != 1.0f / x
 *   SQ_OP_CULL_SLAB              : a = 9n words (a culling box as three packed binary16 pairs lo | hi << 16 for x, y, z; ray
 *                                  origin; ray direction) -> out = n uint32: 1 if the ray passes the kernels' culling slab test
 * a, b, out are HOST pointers. */
enum { SQ_OP_SQRT = 0, SQ_OP_DIV = 1, SQ_OP_SIN = 2, SQ_OP_COS = 3, SQ_OP_ACOS = 4, SQ_OP_ATAN = 5,
       SQ_OP_UNIT_FLOAT = 6, SQ_OP_TFGEN3 = 7, SQ_OP_TONEMAP = 8, SQ_OP_RCP_SWEEP = 9, SQ_OP_CULL_SLAB = 10 };
int sq_debug_eval(int32_t device, int32_t op, const void* a, const void* b, int64_t n, void* out);

int32_t     sq_device_count(void);
int32_t     sq_abi_version(void);
/* Identity of this build: the first 16 hex digits of a SHA-256 over the library's sources, the C headers and the compiler
 * flags (squigly-trace_amd/build.py: source_id).  Profile files under profiles/ are stamped with it, and bench.py refuses to
 * price a roofline with counters that were collected on a different build ("unknown" = built without build.py). */
const char* sq_build_id(void);
const char* sq_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
