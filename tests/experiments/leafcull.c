/* Experiment (test infrastructure, uses the oracle): how many leaf visits / triangle tests / branch visits of the
 * reference's traversal (BIH.hs:101-141) would a conservative tight-box test remove, results unchanged?
 *   gcc -O2 -ffp-contract=off -o /tmp/leafcull tests/experiments/leafcull.c -lm -lpthread && /tmp/leafcull [spp] [rowstep] [margin]
 */
#include <stdint.h>
struct sqo_bih; struct sqo_hit_s;
static void ray_hook(const void* b, float ox, float oy, float oz, float dx, float dy, float dz, const void* out);
#define SQO_RAY_HOOK(b, o, d, out) ray_hook((b), (o).x, (o).y, (o).z, (d).x, (d).y, (d).z, (out))
#include "../../oracle/sq_oracle.c"
#include <stdio.h>

static float g_margin = 1e-3f;
typedef struct { sqo_bounds box; int ready; } tbox;
static tbox* g_leafbox;      /* by node->first for leaves */
typedef struct nbox { const node* nd; sqo_bounds box; int depth, pre, bfs; } nbox;
static nbox* g_nb; static int g_nnb;

static sqo_bounds tri_bounds(const node* nd) {
    sqo_bounds b; int init = 0;
    if (nd->kind == 3) {
        for (int i = 0; i < nd->n; i++) { const sqo_triangle* t = &nd->tris[i]; const sqo_v3* vs[3] = { &t->a, &t->b, &t->c };
            for (int k = 0; k < 3; k++) { sqo_v3 v = *vs[k];
                if (!init) { b.lo = v; b.hi = v; init = 1; }
                if (v.x < b.lo.x) b.lo.x = v.x; if (v.y < b.lo.y) b.lo.y = v.y; if (v.z < b.lo.z) b.lo.z = v.z;
                if (v.x > b.hi.x) b.hi.x = v.x; if (v.y > b.hi.y) b.hi.y = v.y; if (v.z > b.hi.z) b.hi.z = v.z; } }
        if (!init) { b.lo = v3(1, 1, 1); b.hi = v3(-1, -1, -1); }
        return b;
    }
    sqo_bounds l = tri_bounds(nd->l), r = tri_bounds(nd->r);
    b = l;
    if (r.lo.x < b.lo.x) b.lo.x = r.lo.x; if (r.lo.y < b.lo.y) b.lo.y = r.lo.y; if (r.lo.z < b.lo.z) b.lo.z = r.lo.z;
    if (r.hi.x > b.hi.x) b.hi.x = r.hi.x; if (r.hi.y > b.hi.y) b.hi.y = r.hi.y; if (r.hi.z > b.hi.z) b.hi.z = r.hi.z;
    return b;
}
static double g_omax2;
static float subtree_margin(const node* nd, double S) {
    if (nd->kind != 3) { float a = subtree_margin(nd->l, S), b = subtree_margin(nd->r, S); return a > b ? a : b; }
    double m = 0;
    for (int i = 0; i < nd->n; i++) { const sqo_triangle* t = &nd->tris[i];
        double e1[3] = { (double)t->b.x - t->a.x, (double)t->b.y - t->a.y, (double)t->b.z - t->a.z }, e2[3] = { (double)t->c.x - t->a.x, (double)t->c.y - t->a.y, (double)t->c.z - t->a.z };
        double E1 = sqrt(e1[0]*e1[0]+e1[1]*e1[1]+e1[2]*e1[2]), E2 = sqrt(e2[0]*e2[0]+e2[1]*e2[1]+e2[2]*e2[2]);
        double V0 = sqrt((double)t->a.x*t->a.x + (double)t->a.y*t->a.y + (double)t->a.z*t->a.z);
        double u = 5.9604644775390625e-8;
        double mt = S * (u / 1e-4) * E1 * E2 * 1.25 * (g_omax2 + V0 + E1 + E2) + 4 * u * (E1 + E2) + 8 * u * 16; if (mt > m) m = mt; }
    return (float)m;
}
static float half_to_float(unsigned short h) { int s = h >> 15, e = (h >> 10) & 31, m = h & 1023; float v;
    if (e == 0) v = ldexpf((float)m, -24); else if (e == 31) v = m ? NAN : INFINITY; else v = ldexpf((float)(m + 1024), e - 25); return s ? -v : v; }
static float half_down(float x) {   /* largest fp16 value <= x (binary search over the ordered encodings) */
    if (x != x) return x;
    /* order key: map half bits to a monotone integer */
    int lo = -0x7c00, hi = 0x7c00;   /* keys: negative halves = -(bits&0x7fff), positives = bits ; +-inf included */
    while (lo < hi) { int mid = (lo + hi + 1) >> 1 ; if (mid > hi) mid = hi; unsigned short b = mid >= 0 ? (unsigned short)mid : (unsigned short)(0x8000 | (-mid)); float v = half_to_float(b); if (v <= x) lo = mid; else hi = mid - 1; }
    unsigned short b = lo >= 0 ? (unsigned short)lo : (unsigned short)(0x8000 | (-lo)); return half_to_float(b);
}
static float half_up(float x) { return -half_down(-x); }
static int g_half = 0;
static int g_depth_now = 0;
static void collect(const node* nd) {
    g_nb[g_nnb].nd = nd; g_nb[g_nnb].depth = g_depth_now; g_nb[g_nnb].pre = g_nnb; g_nb[g_nnb].box = tri_bounds(nd);
    sqo_bounds* b = &g_nb[g_nnb].box; float m = g_margin;
    if (g_margin < 0) {          /* worst-case margin: 64 (u/eps) P (S + E1 + E2), S = -g_margin, over the subtree's triangles */
        m = subtree_margin(nd, -g_margin);
    }
    b->lo.x -= m; b->lo.y -= m; b->lo.z -= m; b->hi.x += m; b->hi.y += m; b->hi.z += m;
    if (g_half) { b->lo.x = half_down(b->lo.x); b->lo.y = half_down(b->lo.y); b->lo.z = half_down(b->lo.z); b->hi.x = half_up(b->hi.x); b->hi.y = half_up(b->hi.y); b->hi.z = half_up(b->hi.z); }
    g_nnb++;
    if (nd->kind != 3) { g_depth_now++; collect(nd->l); collect(nd->r); g_depth_now--; }
}
static const sqo_bounds* box_of(const node* nd) {   /* small tree: linear probe is fine for scene.obj, hash by pointer otherwise */
    static const node* last; static const sqo_bounds* lastb;
    if (nd == last) return lastb;
    for (int i = 0; i < g_nnb; i++) if (g_nb[i].nd == nd) { last = nd; lastb = &g_nb[i].box; return lastb; }
    return 0;
}
/* pointer-indexed lookup table instead of the linear probe */
#include <stdlib.h>
typedef struct { const node* k; int v; } slot;
static slot* g_tab; static size_t g_tabn;
static void tab_build(void) { g_tabn = 1; while (g_tabn < (size_t)g_nnb * 4) g_tabn <<= 1; g_tab = calloc(g_tabn, sizeof *g_tab);
    for (int i = 0; i < g_nnb; i++) { size_t h = ((uintptr_t)g_nb[i].nd >> 4) * 2654435761u & (g_tabn - 1); while (g_tab[h].k) h = (h + 1) & (g_tabn - 1); g_tab[h].k = g_nb[i].nd; g_tab[h].v = i; } }
static int g_b0 = 0;
static int bfs_of(const node* nd) { size_t h = ((uintptr_t)nd >> 4) * 2654435761u & (g_tabn - 1); while (g_tab[h].k != nd) h = (h + 1) & (g_tabn - 1); return g_nb[g_tab[h].v].bfs; }
static const sqo_bounds* box_fast(const node* nd) { size_t h = ((uintptr_t)nd >> 4) * 2654435761u & (g_tabn - 1); while (g_tab[h].k != nd) h = (h + 1) & (g_tabn - 1); return &g_nb[g_tab[h].v].box; }

typedef struct { uint64_t rays, leaf, tri, branch, leaf_culled, sub_culled, mism; } stats;
static __thread stats S[5];
static stats G[5]; static pthread_mutex_t gmu = PTHREAD_MUTEX_INITIALIZER;

/* variant 1: cull leaves by their tight box; variant 2: cull every child (leaf or branch) by its tight box */
static sqo_hit rec(int variant, sqo_bounds bbox, const node* nd, V3 o, V3 d, stats* s) {
    sqo_hit none; memset(&none, 0, sizeof none); none.tri = -1;
    if (variant >= 1 && (nd->kind == 3 || variant == 2 || variant == 4 || (variant == 3 && bfs_of(nd) >= g_b0))) {
        if (!sqo_intersects_bb(box_fast(nd), o, d)) { if (nd->kind == 3) s->leaf_culled++; else s->sub_culled++; return none; }
    }
    if (nd->kind == 3) {
        s->leaf++;
        sqo_hit best = none;
        for (int i = 0; i < nd->n; i++) { sqo_hit h = none; s->tri++;
            if (sqo_moller_trumbore(o, d, &nd->tris[i], &h.point, &h.dist)) { h.hit = 1; h.tri = nd->first + i; best = best.hit ? min_by_dist(best, h) : h; } }
        return best;
    }
    s->branch++;
    int ax = nd->kind; float lmax = nd->lmax, rmin = nd->rmin;
    if (!sqo_intersects_bb(&bbox, o, d)) return none;
    sqo_bounds left = bbox, right = bbox;
    if (ax == 0) { left.hi.x = lmax; right.lo.x = rmin; } else if (ax == 1) { left.hi.y = lmax; right.lo.y = rmin; } else { left.hi.z = lmax; right.lo.z = rmin; }
    int iL = sqo_intersects_bb(&left, o, d), iR = sqo_intersects_bb(&right, o, d);
    if (iL && iR) {
        int l2r = proj(ax, d) > 0;
        sqo_hit near = l2r ? rec(variant, left, nd->l, o, d, s) : rec(variant, right, nd->r, o, d, s);
        if (near.hit) {
            float p = proj(ax, near.point);
            int isClose = l2r ? (p < rmin) : (p > lmax);
            if (isClose) return near;
            if (variant == 4) {   /* entry parameter of the far child's culling box against the near hit's t (generous slack) */
                const node* fc = l2r ? nd->r : nd->l; const sqo_bounds* fb = box_fast(fc);
                float tn = 0; { float dx = near.point.x - o.x, dy = near.point.y - o.y, dz = near.point.z - o.z; float dd = d.x*d.x+d.y*d.y+d.z*d.z; tn = (dx*d.x+dy*d.y+dz*d.z)/dd; }
                float t1 = (fb->lo.x - o.x)/d.x, t2 = (fb->hi.x - o.x)/d.x, t3 = (fb->lo.y - o.y)/d.y, t4 = (fb->hi.y - o.y)/d.y, t5 = (fb->lo.z - o.z)/d.z, t6 = (fb->hi.z - o.z)/d.z;
                float tmin = fmaxf(fmaxf(fminf(t1,t2), fminf(t3,t4)), fminf(t5,t6));
                if (tmin - 1e-3f > tn * 1.001f) { s->sub_culled++; return near; }
            }
            sqo_hit far = l2r ? rec(variant, right, nd->r, o, d, s) : rec(variant, left, nd->l, o, d, s);
            return far.hit ? min_by_dist(near, far) : near;
        }
        return l2r ? rec(variant, right, nd->r, o, d, s) : rec(variant, left, nd->l, o, d, s);
    }
    if (iL) return rec(variant, left, nd->l, o, d, s);
    if (iR) return rec(variant, right, nd->r, o, d, s);
    return none;
}
static void ray_hook(const void* bv, float ox, float oy, float oz, float dx, float dy, float dz, const void* outv) {
    const sqo_bih* b = bv; const sqo_hit* ref = outv; V3 o = v3(ox, oy, oz), d = v3(dx, dy, dz);
    for (int v = 0; v < 5; v++) {
        S[v].rays++;
        sqo_hit h = rec(v, b->bounds, b->tree, o, d, &S[v]);
        if (h.hit != ref->hit || (h.hit && (h.tri != ref->tri || memcmp(&h.dist, &ref->dist, 4)))) S[v].mism++;
    }
    if (S[0].rays % 4096 == 0) { pthread_mutex_lock(&gmu); for (int v = 0; v < 5; v++) { uint64_t* a = (uint64_t*)&G[v]; uint64_t* t = (uint64_t*)&S[v]; for (int k = 0; k < 7; k++) { a[k] += t[k]; t[k] = 0; } } pthread_mutex_unlock(&gmu); }
}
int main(int argc, char** argv) {
    int spp = argc > 1 ? atoi(argv[1]) : 16, step = argc > 2 ? atoi(argv[2]) : 40; if (argc > 3) g_margin = (float)atof(argv[3]);
    const char* obj = argc > 4 ? argv[4] : "data/scene.obj"; const char* dir = argc > 5 ? argv[5] : "data";
    const char* camf = argc > 6 ? argv[6] : "data/camera";
    int w = argc > 7 ? atoi(argv[7]) : 1920, h = argc > 8 ? atoi(argv[8]) : 1080;
    sqo_triangle* tris; int n;
    if (sqo_tris_from_obj(obj, dir, &tris, &n)) { fprintf(stderr, "%s\n", sqo_last_error()); return 1; }
    sqo_bih* b = sqo_make_bih(tris, n);
    { sqo_bounds rb = tri_bounds(b->tree); double vx = fmax(fabs(rb.lo.x), fabs(rb.hi.x)), vy = fmax(fabs(rb.lo.y), fabs(rb.hi.y)), vz = fmax(fabs(rb.lo.z), fabs(rb.hi.z)); g_omax2 = 2 * sqrt(vx*vx+vy*vy+vz*vz); g_half = getenv("HALF") != 0; printf("omax2 %g half %d\n", g_omax2, g_half); }
    g_nb = malloc(sizeof(nbox) * (size_t)(2 * b->n_nodes + 4)); collect(b->tree);
    { int nbr = 0, maxd = 0; for (int i = 0; i < g_nnb; i++) if (g_nb[i].depth > maxd) maxd = g_nb[i].depth;
      for (int dd = 0; dd <= maxd; dd++) for (int i = 0; i < g_nnb; i++) if (g_nb[i].nd->kind != 3 && g_nb[i].depth == dd) g_nb[i].bfs = nbr++;
      for (int i = 0; i < g_nnb; i++) if (g_nb[i].nd->kind == 3) g_nb[i].bfs = 1 << 30;
      g_b0 = getenv("B0") ? atoi(getenv("B0")) : 209; printf("branches %d, b0 %d\n", nbr, g_b0); }
    tab_build();
    sqo_camera cam; sqo_load_camera(camf, 0, &cam);
    sqo_counters c;
    sqo_render_rows_strided(b, &cam, spp, w, h, 0, step / 2, w, step, 8, 0, 0, 0, 0, &c);
    const char* names[5] = { "reference", "leaf boxes", "leaf+subtree boxes", "leaf+deep subtrees", "subtree + far skip" };
    for (int v = 0; v < 5; v++) { stats* s = &G[v];
        printf("%-20s rays %llu  per ray: branch %.2f leaf %.2f tri %.2f  culled leaves %.2f subtrees %.2f  mismatches %llu\n", names[v],
               (unsigned long long)s->rays, (double)s->branch / s->rays, (double)s->leaf / s->rays, (double)s->tri / s->rays,
               (double)s->leaf_culled / s->rays, (double)s->sub_culled / s->rays, (unsigned long long)s->mism); }
    return 0;
}
