/* squigly_host.h — host side ABOVE the render boundary, as a C-ABI.
 *
 * In the reference these are Haskell functions (src/Obj.hs, src/BIH.hs, src/Geometry.hs) that a
 * Haskell host would keep calling unchanged.  No GHC exists in the build or GPU images, so the
 * same functions are provided in C++ (squigly-trace_amd/csrc/sq_host.cpp) with identical
 * arithmetic, and produce exactly the arrays that squigly_hip.h consumes.
 * All citations are relative to the reference repository root.
 */
#ifndef SQUIGLY_HOST_H
#define SQUIGLY_HOST_H
#include "squigly_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A loaded triangle soup in loader order (Obj.trisFromObj, src/Obj.hs:49-58). */
typedef struct sq_mesh sq_mesh;
int  sq_mesh_from_obj(const char* obj_path, const char* mtl_dir, sq_mesh** out);   /* mtl_dir plays "./data/" of src/Obj.hs:52 */
int  sq_mesh_from_text(const char* obj_text, size_t obj_len, const char* sq_text, size_t sq_len, sq_mesh** out);
int  sq_mesh_from_arrays(const sq_tri* tris, int32_t n_tris, const sq_material* mats, int32_t n_mats, sq_mesh** out);
int32_t sq_mesh_num_tris(const sq_mesh* m);
int32_t sq_mesh_num_materials(const sq_mesh* m);
const sq_tri*      sq_mesh_tris(const sq_mesh* m);
const sq_material* sq_mesh_materials(const sq_mesh* m);
void sq_mesh_free(sq_mesh* m);
/* What `--debug` prints while loading (src/Obj.hs:55-57): `print (head objs)` and `print mats`, in the text of Haskell's
 * derived Show instances (records, lists, `show :: Float -> String`).  The strings belong to the mesh; both are empty
 * for a mesh that did not come from .obj/.sq text, and *first_object is empty when the file has no object (the reference's
 * `head objs` throws there: the CLIs print "Prelude.head: empty list" and exit 1).  Names are escaped per GHC's showLitChar on
 * the UTF-8 decoded text (control characters by name, \DEL, \SO\&H, code points > 127 in decimal with \& before a digit). */
void sq_mesh_debug_show(const sq_mesh* m, const char** first_object, const char** materials);

/* Obj.loadCamera (src/Obj.hs:60-70): "px py pz\nalpha beta gamma" -> position + rotMatrixRads. */
int  sq_camera_from_file(const char* path, sq_camera* cam);
int  sq_camera_from_text(const char* text, size_t len, sq_camera* cam);
void sq_rot_matrix_rads(float alpha, float beta, float gamma, float out9[9]);     /* src/Geometry.hs:90-102 */

/* BIH.makeBIH (src/BIH.hs:62-99) + flatten (:50-52) into the pre-order arrays of squigly_hip.h.
 * The returned object owns the arrays; sq_bih_scene() fills an sq_scene that points into it. */
typedef struct sq_bih sq_bih;
int  sq_bih_build(const sq_mesh* mesh, sq_bih** out);
/* The same build on HIP device `device` (squigly-trace_amd/csrc/sq_bih_device.hip): level-synchronous,
 * element-parallel, and bit-identical to sq_bih_build -- same split planes (ordered fp32 sums), same stable
 * partitions, same boxes -- because tree shape and leaf order decide traversal tie-breaks.  Fails (non-zero,
 * sq_last_error) without a GPU or when a vertex coordinate is not finite; it never falls back to the host. */
int  sq_bih_build_device(const sq_mesh* mesh, int32_t device, sq_bih** out);
void sq_bih_scene(const sq_bih* b, sq_scene* out);
int32_t sq_bih_height(const sq_bih* b);        /* BIH.height       src/BIH.hs:46-48 */
int32_t sq_bih_num_leaves(const sq_bih* b);    /* BIH.numLeaves    src/BIH.hs:54-56 */
int32_t sq_bih_longest_leaf(const sq_bih* b);  /* BIH.longestLeaf  src/BIH.hs:58-60 */
void sq_bih_free(sq_bih* b);

/* Culling boxes (no counterpart in the reference; an exact reduction of its triangle tests).  boxes: 6 floats per node
 * of scene->nodes (lo.xyz, hi.xyz; a branch gets the union of its children).  A ray with ray_limits[1] <= |d|^2 <=
 * ray_limits[2], |o|^2 <= ray_limits[0] and finite o, d, 1/d, o/d that FAILS the fp32 slab test of a node's box is
 * rejected by the reference's mollerTrumbore (src/Geometry.hs:117-142) for every triangle of that node, so the node
 * returns Nothing (src/BIH.hs:105-109) without testing them.  The error analysis is in csrc/sq_host.cpp; leaves it does
 * not cover get an infinite box, and ray_limits[0] < 0 says no ray may be culled.  sq_scene_upload() calls this. */
int  sq_cull_boxes(const sq_scene* scene, float* boxes, float ray_limits[3]);
/* The binary16 encoding the resident kernels keep such a box in: the nearest value >= x (up != 0) or <= x, never subnormal. */
uint32_t sq_half_outward(float x, int32_t up);

#ifdef __cplusplus
}
#endif
#endif
