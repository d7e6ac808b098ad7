// order_experiment.c -- CPU experiment behind DESIGN.md "order-aware traversal" (VERDICT r2 item 5): how many box tests per ray
// does a NEAR-FIRST visiting order need on a spheres-only scene, against the fixed (reference) order the kernels walk today?
//
// For spheres the reference's answer is the minimum of (t, leaf ordinal) over the leaves whose own box passes
// (sphere.cuh:66 accepts t < limit, so the first of equal hits wins; bvh.cuh:95-106), so any visiting order that tests a
// superset of the contributing leaves and applies that rule returns the reference's record.  This program only COUNTS:
//   mode "fixed"   the depth-first array as given (skip links), limit = closest hit so far            -> tests per ray, per-node passes
//   mode "near"    classic stack traversal of the same tree, nearer child (by box entry distance) first
//   mode "octant"  eight depth-first arrays of the same tree, children ordered once per direction octant (stackless again)
// and checks that every mode finds the same (t, ordinal) for every ray.  Inputs are binary files written by
// tools/order_experiment.py: nodes (rt_node, 32 B: bmin[3], skip, bmax[3], prim), spheres (32 B), rays (8 floats: o, d, time, t).
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float bmin[3]; int32_t skip; float bmax[3]; int32_t prim; } node_t;
typedef struct { float c0[3]; float radius; float vel[3]; int32_t mat; } sphere_t;
typedef struct { float o[3], d[3], tm, t; } ray_t;

static node_t* nodes; static int n_nodes;
static sphere_t* spheres; static int n_spheres;
static ray_t* rays; static int n_rays;
static int* leaf_ord;   // ordinal of a leaf node in the ORIGINAL depth-first order, by sphere index (stable under re-ordering)

static void* slurp(const char* path, size_t rec, int* n) {
    FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    void* p = malloc(sz); if (fread(p, 1, sz, f) != (size_t)sz) exit(1); fclose(f);
    *n = (int)(sz / rec); return p;
}

// aabb::hit (aabb.cuh:45-61) with the interval returned
static int slab(const node_t* b, const ray_t* r, const float inv[3], float tmin, float tmax, float* t_in) {
    for (int a = 0; a < 3; ++a) {
        float t0 = (b->bmin[a] - r->o[a]) * inv[a], t1 = (b->bmax[a] - r->o[a]) * inv[a];
        if (inv[a] < 0.0f) { float t = t0; t0 = t1; t1 = t; }
        tmin = t0 > tmin ? t0 : tmin; tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return 0;
    }
    *t_in = tmin; return 1;
}
static float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[0], b[0], a[1] * b[1])); }
// sphere::hit (sphere.cuh:51-89), exclusive bounds
static int sphere_hit(const sphere_t* s, const ray_t* r, float tmin, float tmax, float* t_out) {
    float cc[3], oc[3];
    for (int a = 0; a < 3; ++a) { cc[a] = fmaf(r->tm, s->vel[a], s->c0[a]); oc[a] = r->o[a] - cc[a]; }
    const float A = dot3(r->d, r->d), B = dot3(oc, r->d), C = fmaf(-s->radius, s->radius, dot3(oc, oc));
    const float disc = fmaf(B, B, -(A * C));
    if (disc <= 0.0f) return 0;
    const float sq = sqrtf(disc);
    float t = (-B - sq) / A; if (t > tmin && t < tmax) { *t_out = t; return 1; }
    t = (-B + sq) / A; if (t > tmin && t < tmax) { *t_out = t; return 1; }
    return 0;
}

typedef struct { float t; int ord; } hit_t;
static void consider(const node_t* leaf, const ray_t* r, float tmin, hit_t* best) {
    float t;
    const int si = leaf->prim & 0x0FFFFFFF;
    // (t, ordinal) minimum: a later visit in another order may find an equal t with a lower ordinal
    if (sphere_hit(&spheres[si], r, tmin, FLT_MAX, &t)) {
        const int ord = leaf_ord[si];
        if (t < best->t || (t == best->t && ord < best->ord)) { best->t = t; best->ord = ord; }
    }
}

// fixed order over a depth-first array; pass[] (optional) += 1 per passing box
// inclusive: a box is entered if it can hold something closer OR equally close (another order may meet the lower ordinal later)
static hit_t walk_fixed(const node_t* nd, int n, const ray_t* r, unsigned long long* tests, double* pass, int inclusive) {
    const float inv[3] = {1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2]};
    hit_t best = {FLT_MAX, -1};
    int i = 0;
    while (i < n) {
        float t_in; ++*tests;
        int next = nd[i].skip;
        const float lim = (inclusive && best.ord >= 0) ? nextafterf(best.t, FLT_MAX) : best.t;
        if (slab(&nd[i], r, inv, 0.001f, lim, &t_in)) {
            if (pass) pass[i] += 1.0;
            if (nd[i].prim >= 0) consider(&nd[i], r, 0.001f, &best); else next = i + 1;
        }
        i = next;
    }
    return best;
}

// classic near-first stack traversal of a BINARY tree given as a depth-first array
static hit_t walk_near(const node_t* nd, int n, const ray_t* r, unsigned long long* tests) {
    const float inv[3] = {1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2]};
    hit_t best = {FLT_MAX, -1};
    int stack[128]; float stack_t[128]; int sp = 0;
    float t_in; ++*tests;
    if (!slab(&nd[0], r, inv, 0.001f, FLT_MAX, &t_in)) return best;
    int cur = 0;
    for (;;) {
        if (nd[cur].prim >= 0) consider(&nd[cur], r, 0.001f, &best);
        else {
            const int a = cur + 1, b = nd[a].skip;   // the two children
            float ta = 0, tb = 0;
            const float lim = best.ord >= 0 ? nextafterf(best.t, FLT_MAX) : FLT_MAX;
            *tests += 2;
            const int ha = slab(&nd[a], r, inv, 0.001f, lim, &ta), hb = (b < nd[cur].skip) ? slab(&nd[b], r, inv, 0.001f, lim, &tb) : 0;
            if (ha && hb) { const int nearc = ta <= tb ? a : b, farc = ta <= tb ? b : a; stack[sp] = farc; stack_t[sp++] = ta <= tb ? tb : ta; cur = nearc; continue; }
            if (ha) { cur = a; continue; }
            if (hb) { cur = b; continue; }
        }
        // pop: skip entries that can no longer hold anything closer (no box test: the entry distance is remembered)
        for (;;) {
            if (sp == 0) return best;
            --sp;
            if (stack_t[sp] <= best.t) { cur = stack[sp]; break; }
        }
    }
}

// the same tree with every interior node's children re-ordered for direction octant `oct` (bit a set: d[a] < 0): the
// child whose box starts earlier along the axis on which the children's centres differ most comes first
static int emit_ordered(const node_t* nd, int i, int oct, node_t* out, int at) {
    const int me = at++;
    out[me] = nd[i];
    if (nd[i].prim < 0) {
        int nk = 0;
        for (int c = i + 1; c < nd[i].skip; c = nd[c].skip) ++nk;
        int* kids = (int*)malloc(sizeof(int) * nk);
        float* key = (float*)malloc(sizeof(float) * nk);   // insertion sort by the octant's key
        nk = 0;
        for (int c = i + 1; c < nd[i].skip; c = nd[c].skip) kids[nk++] = c;
        int axis = 0; float spread = -1.f;
        for (int a = 0; a < 3; ++a) {
            float lo = FLT_MAX, hi = -FLT_MAX;
            for (int k = 0; k < nk; ++k) { const float c = 0.5f * (nd[kids[k]].bmin[a] + nd[kids[k]].bmax[a]); lo = fminf(lo, c); hi = fmaxf(hi, c); }
            if (hi - lo > spread) { spread = hi - lo; axis = a; }
        }
        for (int k = 0; k < nk; ++k) { const float c = 0.5f * (nd[kids[k]].bmin[axis] + nd[kids[k]].bmax[axis]); key[k] = ((oct >> axis) & 1) ? -c : c; }
        for (int k = 1; k < nk; ++k) { const int kk = kids[k]; const float kv = key[k]; int j = k - 1; while (j >= 0 && key[j] > kv) { kids[j + 1] = kids[j]; key[j + 1] = key[j]; --j; } kids[j + 1] = kk; key[j + 1] = kv; }
        for (int k = 0; k < nk; ++k) at = emit_ordered(nd, kids[k], oct, out, at);
        free(kids); free(key);
    }
    out[me].skip = at;
    return at;
}

// a walk array may be a forest (the collapse drops a root that always passes): its top-level nodes are ordered like children
static void emit_forest(const node_t* nd, int n, int oct, node_t* out) {
    node_t* tmp = (node_t*)malloc(sizeof(node_t) * (n + 1));
    memcpy(tmp + 1, nd, sizeof(node_t) * n);
    for (int i = 1; i <= n; ++i) tmp[i].skip += 1;
    memset(&tmp[0], 0, sizeof(node_t)); tmp[0].prim = -1; tmp[0].skip = n + 1;   // a virtual root in front
    node_t* o2 = (node_t*)malloc(sizeof(node_t) * (n + 1));
    emit_ordered(tmp, 0, oct, o2, 0);
    for (int i = 0; i < n; ++i) { out[i] = o2[i + 1]; out[i].skip -= 1; }
    free(tmp); free(o2);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: order_experiment <mode> nodes.bin spheres.bin rays.bin [pass_out.bin] [octant-only]\n"); return 2; }
    const char* mode = argv[1];
    nodes = (node_t*)slurp(argv[2], sizeof(node_t), &n_nodes);
    spheres = (sphere_t*)slurp(argv[3], sizeof(sphere_t), &n_spheres);
    rays = (ray_t*)slurp(argv[4], sizeof(ray_t), &n_rays);
    const char* reference = getenv("ORDER_REFERENCE_NODES");   // leaf ordinals come from the reference's own array
    int n_ref = 0;
    node_t* ref = reference ? (node_t*)slurp(reference, sizeof(node_t), &n_ref) : nodes;
    if (!reference) n_ref = n_nodes;
    leaf_ord = (int*)malloc(sizeof(int) * (n_spheres > 0 ? n_spheres : 1));
    { int q = 0; for (int i = 0; i < n_ref; ++i) if (ref[i].prim >= 0) leaf_ord[ref[i].prim & 0x0FFFFFFF] = q++; }
    const int only_oct = argc > 6 ? atoi(argv[6]) : -1;
    unsigned long long tests = 0, used = 0, mismatches = 0;
    double* pass = (double*)calloc((size_t)n_nodes, sizeof(double));
    node_t* arr[8] = {0};
    if (!strcmp(mode, "octant")) for (int o = 0; o < 8; ++o) { arr[o] = (node_t*)malloc(sizeof(node_t) * n_nodes); emit_forest(nodes, n_nodes, o, arr[o]); }
    for (int k = 0; k < n_rays; ++k) {
        const ray_t* r = &rays[k];
        const int oct = (r->d[0] < 0 ? 1 : 0) | (r->d[1] < 0 ? 2 : 0) | (r->d[2] < 0 ? 4 : 0);
        if (only_oct >= 0 && oct != only_oct) continue;
        if (!(fabsf(1.0f / r->d[0]) < INFINITY && fabsf(1.0f / r->d[1]) < INFINITY && fabsf(1.0f / r->d[2]) < INFINITY)) continue;
        ++used;
        hit_t h;
        if (!strcmp(mode, "fixed")) h = walk_fixed(nodes, n_nodes, r, &tests, pass, 0);
        else if (!strcmp(mode, "near")) h = walk_near(nodes, n_nodes, r, &tests);
        else h = walk_fixed(arr[oct], n_nodes, r, &tests, pass, 1);
        if (h.t != r->t) ++mismatches;   // the oracle's closest hit for this ray
    }
    printf("%s: %d nodes, %llu rays, %.3f box tests per ray, %llu rays whose closest t differs from the oracle's\n", mode, n_nodes, used, (double)tests / (double)(used ? used : 1), mismatches);
    if (argc > 5 && strcmp(argv[5], "-")) { FILE* f = fopen(argv[5], "wb"); fwrite(pass, sizeof(double), n_nodes, f); fclose(f); }
    if (!strcmp(mode, "octant") && argc > 5 && strcmp(argv[5], "-") && only_oct >= 0) {
        // the octant's own array, for the collapse planner (python side)
        char path[512]; snprintf(path, sizeof(path), "%s.nodes", argv[5]);
        FILE* f = fopen(path, "wb"); fwrite(arr[only_oct], sizeof(node_t), n_nodes, f); fclose(f);
    }
    return 0;
}
