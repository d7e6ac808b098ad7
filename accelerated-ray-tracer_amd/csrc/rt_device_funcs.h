// rt_device_funcs.h -- every device function of the render path on gfx950 (MI355X): the reference's
// hit / scatter / texture / camera routines (src/*.cuh) restated for the flattened scene of include/rt_abi.h.
// Shared by all kernel translation units (rt_kernel_*.hip, rt_staged_*.hip); internal to librt_mi355x.so.
//
// Design (see DESIGN.md for the measurements behind each choice):
//   * The scene arrives as flat arrays (include/rt_abi.h).  The BVH is the
//     reference's tree in depth-first order with skip links, so the fixed
//     left-then-right visiting order of bvh_node::hit (bvh.cuh:95-106) becomes
//     a stackless loop: box hit -> next node (or test the leaf's object), box
//     miss -> skip link.  Each node's box is tested at visit time against the
//     closest hit so far, which is exactly what the recursion does, so results
//     are bit-identical with no per-lane stack at all.
//   * Nodes (32 B) and spheres (32 B) are staged once per workgroup in LDS;
//     a persistent workgroup keeps them for the whole frame.
//   * One lane = one pixel; the cuRAND-compatible XORWOW state is six VGPRs
//     that never touch memory (the reference round-trips 96 B/pixel of
//     curandState through HBM, main.cu:116,126).
//   * The sample loop and the <=50-bounce loop are flattened into one state
//     machine per lane (path regeneration): every lane always has a live ray,
//     a lane that finishes its pixel pulls the next pixel from a global
//     counter, and the shading block runs only when a wave-wide ballot says
//     enough lanes are waiting for it.  This is pure re-scheduling: each
//     pixel still consumes its own random stream in the reference's order.
//   * No MFMA: there is no dense contraction anywhere on this path.
//
// Numerical contract: the arithmetic of the reference's real
// build.  nvcc's default -fmad=true contracts a*b+c into one FMA; those
// contractions are written out below with fmaf() and the file is built with
// -ffp-contract=off so the compiler neither adds nor removes one.  Every rule
// was checked against the reference's own output images, seven of which this
// code reproduces pixel for pixel.  Transcendentals (powf in gamma and
// Schlick, logf in the medium, __sinf in the noise texture, acos / atan2 in
// sphere uv) are evaluated in double and rounded once.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/rt_abi.h"
#include "rt_device.h"
#include "rt_xorwow.h"

#define DEV __device__ __forceinline__

namespace {

// ------------------------------------------------------------------ vec3 (vec3.cuh:8-158)
struct f3 { float x, y, z; };
DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV f3 operator*(float t, f3 v) { return mk3(t * v.x, t * v.y, t * v.z); }
DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
DEV f3 sdiv(f3 v, float t) { return mk3(v.x / t, v.y / t, v.z / t); }
// Contracted exactly where nvcc's default -fmad=true contracts the reference (DESIGN.md "numerical contract"): a
// product whose only use is an add/sub becomes one FMA with it; of two products under one add the first is fused;
// m0 + m1 + m2 = fma(m2, fma(m0, m1)).  The build itself runs with -ffp-contract=off: every FMA is written out.
DEV float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.x, b.x, a.y * b.y)); }
DEV f3 cross(f3 a, f3 b) { return mk3(fmaf(a.y, b.z, -(a.z * b.y)), -fmaf(a.x, b.z, -(a.z * b.x)), fmaf(a.x, b.y, -(a.y * b.x))); }
DEV float length(f3 v) { return sqrtf(fmaf(v.z, v.z, fmaf(v.x, v.x, v.y * v.y))); }
DEV f3 fma3(float t, f3 v, f3 a) { return mk3(fmaf(t, v.x, a.x), fmaf(t, v.y, a.y), fmaf(t, v.z, a.z)); }      // a + t*v
DEV f3 fma3(f3 t, f3 v, f3 a) { return mk3(fmaf(t.x, v.x, a.x), fmaf(t.y, v.y, a.y), fmaf(t.z, v.z, a.z)); }    // a + t*v, per component
DEV f3 unit_vector(f3 v) { return sdiv(v, length(v)); }

// correctly rounded fp32 transcendentals via double
__device__ __attribute__((noinline)) float cr_pow(float x, float y) { return (float)pow((double)x, (double)y); }
DEV float cr_pow5(float x) { double d = (double)x; return (float)(d * d * d * d * d); }
__device__ __attribute__((noinline)) float cr_log(float x) { return (float)log((double)x); }
__device__ __attribute__((noinline)) float cr_sin(float x) { return (float)sin((double)x); }
__device__ __attribute__((noinline)) float cr_acos(float x) { return (float)acos((double)x); }
__device__ __attribute__((noinline)) float cr_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

// ------------------------------------------------------------------ ray (ray.cuh:5-21)
// The ray's time is a double in the reference but is only ever consumed
// narrowed to float (sphere.cuh:54 through ray.cuh:16) or copied to the
// scattered ray, so the narrowed value is carried instead.
struct Ray { f3 o, d; float tm; };
DEV f3 ray_at(const Ray& r, float t) { return fma3(t, r.d, r.o); }   // A + t*B: one FMA per component

// what the traversal keeps about the closest hit; everything else is
// recomputed once per ray by resolve_hit() with the same expressions
struct HitInfo {
    float t;
    int32_t prim;   // resolved leaf: sphere / quad / medium ref
    int32_t inst;   // instance index the hit went through, or -1
};
struct HitRec { f3 p, n; int32_t mat; float u, v; };

struct SceneView {
    const rt_node* nodes;       // global or LDS
    const rt_sphere* spheres;   // global or LDS
    const rt_quad* quads;
    const rt_box* boxes;
    const rt_instance* instances;
    const rt_medium* media;
    const rt_material* materials;
    const rt_texture* textures;
    const uint8_t* images;
    int32_t n_nodes;
};

// ------------------------------------------------------------------ primitives
// sphere::hit (sphere.cuh:51-89).  Returns the accepted root or a negative
// value; exclusive bounds t > tmin && t < tmax.
// `a` = dot(r.d, r.d): the same value for every sphere a ray meets, so callers that test many compute it once
DEV bool sphere_test_a(const rt_sphere& s, const Ray& r, float a, float tmin, float tmax, float& t_out) {
    const f3 cc = fma3(r.tm, ld3(s.vel), ld3(s.c0));
    const f3 oc = r.o - cc;
    const float b = dot(oc, r.d);
    const float c = fmaf(-s.radius, s.radius, dot(oc, oc));
    const float disc = fmaf(b, b, -(a * c));
    if (disc <= 0.0f) return false;
    const float sq = sqrtf(disc);
    float t = (-b - sq) / a;
    if (t > tmin && t < tmax) { t_out = t; return true; }
    t = (-b + sq) / a;
    if (t > tmin && t < tmax) { t_out = t; return true; }
    return false;
}
DEV bool sphere_test(const rt_sphere& s, const Ray& r, float tmin, float tmax, float& t_out) {
    return sphere_test_a(s, r, dot(r.d, r.d), tmin, tmax, t_out);
}

// A record whose address is the same in every lane (scan mode, tier loops), read through the constant address space:
// the scene arrays are written by the host before the launch and never by a kernel, so the load may go through the
// scalar cache into SGPRs (s_load_dwordx*) instead of 64 identical vector loads.
template <typename T>
DEV T uniform_load(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "records are whole dwords");
    typedef const __attribute__((address_space(4))) uint32_t* const_words;
    const const_words w = (const_words)(unsigned long long)p;
    union { T value; uint32_t words[sizeof(T) / 4]; } u;
#pragma unroll
    for (unsigned int k = 0; k < sizeof(T) / 4; ++k) u.words[k] = w[k];
    return u.value;
}

// quad::hit (quad.cuh:60-90); inclusive bounds
DEV bool quad_test(const rt_quad& q, const Ray& r, float tmin, float tmax, float& t_out) {
    const f3 n = ld3(q.n);
    const float denom = dot(n, r.d);
    if (fabsf(denom) < 1e-8f) return false;
    const float t = (q.D - dot(n, r.o)) / denom;
    if (t < tmin || t > tmax) return false;
    const f3 P = ray_at(r, t);
    const f3 pl = P - ld3(q.Q);
    const f3 w = ld3(q.w);
    const float alpha = dot(w, cross(pl, ld3(q.v)));
    const float beta = dot(w, cross(ld3(q.u), pl));
    if (alpha < 0.f || alpha > 1.f || beta < 0.f || beta > 1.f) return false;
    t_out = t;
    return true;
}

// quad::hit for a quad whose unit normal is exactly +-e_C, whose u and v have one non-zero component each (on the other
// two axes) and whose w = n / dot(n, n) therefore has one too: an axis-aligned rectangle -- every wall of the Cornell box,
// every face of make_box (quad.cuh:145-162).  rt_scene_create verifies those exact zeros per quad and marks the quad
// (rt_quad.pad0) / the box (bit 30 of first_quad).  With them, the general expressions above collapse without changing a
// bit of what is compared: dot(n, x) = fma(n.z, x.z, fma(n.x, x.x, n.y * x.y)) is n_C * x_C plus exact zeros; dot(w, X) is
// w_C * X_C; X_C = cross(.,.)_C reads only the two in-plane components of pl.  (The only thing that can differ is the sign
// of a zero in denom, alpha or beta, which no comparison below sees: |denom| < 1e-8 misses either way, +-0 is neither < 0
// nor > 1.)  ~35 instructions instead of ~60.
template <int C> DEV float comp(const f3& v) { return C == 0 ? v.x : (C == 1 ? v.y : v.z); }
template <int C>
DEV bool quad_test_axis(const rt_quad& q, const Ray& r, float tmin, float tmax, float& t_out) {
    constexpr int A = (C + 1) % 3, B = (C + 2) % 3;
    const float s = q.n[C];
    const float denom = s * comp<C>(r.d);
    if (fabsf(denom) < 1e-8f) return false;
    const float t = (q.D - s * comp<C>(r.o)) / denom;
    if (t < tmin || t > tmax) return false;
    const float plA = fmaf(t, comp<A>(r.d), comp<A>(r.o)) - q.Q[A];     // ray.point_at_parameter(t) - Q, in-plane components
    const float plB = fmaf(t, comp<B>(r.d), comp<B>(r.o)) - q.Q[B];
    // component C of cross(pl, v) and of cross(u, pl), by the general formula of cross() for that component
    float xa, xb;
    if (C == 1) {   // cross(a, b).y = -fma(a.x, b.z, -(a.z * b.x)); here A = z, B = x
        xa = -fmaf(plB, q.v[A], -(plA * q.v[B]));
        xb = -fmaf(q.u[B], plA, -(q.u[A] * plB));
    } else {        // cross(a, b).x = fma(a.y, b.z, -(a.z * b.y)) (A = y, B = z); .z = fma(a.x, b.y, -(a.y * b.x)) (A = x, B = y)
        xa = fmaf(plA, q.v[B], -(plB * q.v[A]));
        xb = fmaf(q.u[A], plB, -(q.u[B] * plA));
    }
    const float alpha = q.w[C] * xa, beta = q.w[C] * xb;
    if (alpha < 0.f || alpha > 1.f || beta < 0.f || beta > 1.f) return false;
    t_out = t;
    return true;
}
// the quad's marking: 0 = general quad, 1 + C = axis-aligned with normal axis C
DEV int quad_axis_code(const rt_quad& q) { return __float_as_int(q.pad0); }

// sphere | quad | compound6 (quad.cuh:124-139: closest-hit scan over six faces,
// no box early-out).  On a hit, `leaf` is the resolved sphere/quad ref.
// UNIFORM: `ref` is the same in every lane (scan mode, tier loops), so a quad's axis code is a scalar branch.
template <bool UNIFORM = false>
DEV bool simple_test(const SceneView& sc, int32_t ref, const Ray& r, float tmin, float tmax, float& t_out, int32_t& leaf) {
    const int kind = RT_PRIM_KIND(ref), idx = RT_PRIM_INDEX(ref);
    if (kind == RT_PRIM_SPHERE) {
        leaf = ref;
        return sphere_test(sc.spheres[idx], r, tmin, tmax, t_out);
    }
    if (kind == RT_PRIM_QUAD) {
        leaf = ref;
        if (UNIFORM) {
            const rt_quad q = uniform_load(sc.quads + idx);
            const int code = quad_axis_code(q);
            if (code == 1) return quad_test_axis<0>(q, r, tmin, tmax, t_out);
            if (code == 2) return quad_test_axis<1>(q, r, tmin, tmax, t_out);
            if (code == 3) return quad_test_axis<2>(q, r, tmin, tmax, t_out);
            return quad_test(q, r, tmin, tmax, t_out);
        }
        return quad_test(sc.quads[idx], r, tmin, tmax, t_out);
    }
    // box
    const int first_raw = UNIFORM ? uniform_load(&sc.boxes[idx].first_quad) : sc.boxes[idx].first_quad, first = first_raw & 0x3FFFFFFF;
    bool any = false;
    float closest = tmax;
    // a face record: through the scalar cache when the box is the same in every lane
    auto face = [&](int f) -> rt_quad { return UNIFORM ? uniform_load(sc.quads + first + f) : sc.quads[first + f]; };
    if (first_raw & 0x40000000) {
        // make_box's faces in its order (front, right, back, left, top, bottom): normals along z, x, z, x, y, y -- verified
        // per box by rt_scene_create
        float t;
        if (quad_test_axis<2>(face(0), r, tmin, closest, t)) { any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + 0); }
        if (quad_test_axis<0>(face(1), r, tmin, closest, t)) { any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + 1); }
        if (quad_test_axis<2>(face(2), r, tmin, closest, t)) { any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + 2); }
        if (quad_test_axis<0>(face(3), r, tmin, closest, t)) { any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + 3); }
        if (quad_test_axis<1>(face(4), r, tmin, closest, t)) { any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + 4); }
        if (quad_test_axis<1>(face(5), r, tmin, closest, t)) { any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + 5); }
    } else {
        for (int f = 0; f < 6; ++f) {
            float t;
            if (quad_test(face(f), r, tmin, closest, t)) {
                any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + f);
            }
        }
    }
    t_out = closest;
    return any;
}

// ray into the object space of translate(rotate_y(.)) (hittable.cuh:58, 120-127)
DEV Ray to_object_space(const rt_instance& in, const Ray& r) {
    Ray q = r;
    if (in.flags & RT_INST_TRANSLATE) q.o = r.o - ld3(in.offset);
    if (in.flags & RT_INST_ROTATE_Y) {
        const float c = in.cos_t, s = in.sin_t;
        const f3 o = q.o, d = q.d;
        q.o = mk3(fmaf(c, o.x, -(s * o.z)), o.y, fmaf(s, o.x, c * o.z));
        q.d = mk3(fmaf(c, d.x, -(s * d.z)), d.y, fmaf(s, d.x, c * d.z));
    }
    return q;
}

// any leaf object except a medium: simple or instance-of-simple
template <bool UNIFORM = false>
DEV bool solid_test(const SceneView& sc, int32_t ref, const Ray& r, float tmin, float tmax, float& t_out, int32_t& leaf, int32_t& inst) {
    if (RT_PRIM_KIND(ref) == RT_PRIM_INSTANCE) {
        inst = RT_PRIM_INDEX(ref);
        const rt_instance in = UNIFORM ? uniform_load(sc.instances + inst) : sc.instances[inst];
        const Ray q = to_object_space(in, r);
        return simple_test<UNIFORM>(sc, in.child, q, tmin, tmax, t_out, leaf);
    }
    inst = -1;
    return simple_test<UNIFORM>(sc, ref, r, tmin, tmax, t_out, leaf);
}

// constant_medium::hit (constant_medium.cuh:36-64) behind the 4-argument
// fallback (:67-76), which is the only form a BVH ever calls (bvh.cuh:109-112):
// a private XORWOW seeded from a hash of the ray supplies the one uniform.
template <bool UNIFORM = false>
DEV bool medium_test(const SceneView& sc, const rt_medium& m, const Ray& r, float tmin, float tmax, float& t_out) {
    rt_xorwow fake;
    const uint32_t seed = 1337u ^ __float_as_uint(r.o.x) ^ __float_as_uint(r.o.y * 3.1f) ^ __float_as_uint(r.d.z * 5.7f);
    rt_xorwow_seed(fake, (uint64_t)seed);
    float t1, t2;
    int32_t leaf, inst;
    if (!solid_test<UNIFORM>(sc, m.boundary, r, -FLT_MAX, FLT_MAX, t1, leaf, inst)) return false;
    if (!solid_test<UNIFORM>(sc, m.boundary, r, t1 + 1e-4f, FLT_MAX, t2, leaf, inst)) return false;
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    const float ray_len = length(r.d);
    if (ray_len <= 0.0f || !isfinite(ray_len)) return false;
    const float distance_inside = (t2 - t1) * ray_len;
    const float U = fmaxf(1e-6f, rt_xorwow_uniform(fake));
    const float hit_distance = m.neg_inv_density * cr_log(U);
    if (hit_distance > distance_inside) return false;
    t_out = t1 + hit_distance / ray_len;
    return true;
}

// One leaf object, tested once.  The reference tests it twice (left == right
// in a single-object node, bvh.cuh:38-43,100-101); the second test runs with
// tmax = the first hit's t and either misses or reproduces the same record
// for every object kind, so one test gives the same result.
template <bool SPHERES_ONLY, bool UNIFORM = false>
DEV void leaf_test(const SceneView& sc, int32_t ref, const Ray& r, float tmin, HitInfo& best) {
    float t;
    if (SPHERES_ONLY) {
        if (sphere_test(sc.spheres[RT_PRIM_INDEX(ref)], r, tmin, best.t, t)) { best.t = t; best.prim = ref; best.inst = -1; }
        return;
    }
    int32_t leaf = ref, inst = -1;
    if (RT_PRIM_KIND(ref) == RT_PRIM_MEDIUM) {
        const rt_medium m = UNIFORM ? uniform_load(sc.media + RT_PRIM_INDEX(ref)) : sc.media[RT_PRIM_INDEX(ref)];
        if (medium_test<UNIFORM>(sc, m, r, tmin, best.t, t)) { best.t = t; best.prim = ref; best.inst = -1; }
        return;
    }
    if (solid_test<UNIFORM>(sc, ref, r, tmin, best.t, t, leaf, inst)) { best.t = t; best.prim = leaf; best.inst = inst; }
}

// aabb::hit (aabb.cuh:45-61) with 1/direction hoisted out of the node loop
// (same value as the reference's per-test 1.0f/dir) and the three per-axis
// early exits folded into the final comparison (tmin only grows and tmax only
// shrinks, so "some axis fails" == "the last axis fails").
DEV bool slab_test(const float4 lo_skip, const float4 hi_prim, const f3 o, const f3 inv, float tmin, float tmax) {
    float t0 = (lo_skip.x - o.x) * inv.x, t1 = (hi_prim.x - o.x) * inv.x;
    if (inv.x < 0.0f) { const float tmp = t0; t0 = t1; t1 = tmp; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    t0 = (lo_skip.y - o.y) * inv.y; t1 = (hi_prim.y - o.y) * inv.y;
    if (inv.y < 0.0f) { const float tmp = t0; t0 = t1; t1 = tmp; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    t0 = (lo_skip.z - o.z) * inv.z; t1 = (hi_prim.z - o.z) * inv.z;
    if (inv.z < 0.0f) { const float tmp = t0; t0 = t1; t1 = tmp; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    return !(tmax <= tmin);
}

// ------------------------------------------------------------------ hit record
DEV void sphere_uv(f3 p, float& u, float& v) {   // sphere.cuh:42-49
    const float PI_F = 3.141592654f;
    const float theta = cr_acos(-p.y);
    const float phi = cr_atan2(-p.z, p.x) + PI_F;
    u = phi / (2 * PI_F);
    v = theta / PI_F;
}

template <bool SPHERES_ONLY, bool NEED_UV>
DEV HitRec resolve_hit(const SceneView& sc, const Ray& r, const HitInfo& h) {
    HitRec rec;
    rec.u = 0.f; rec.v = 0.f;
    const int kind = SPHERES_ONLY ? RT_PRIM_SPHERE : RT_PRIM_KIND(h.prim);
    const int idx = RT_PRIM_INDEX(h.prim);
    if (!SPHERES_ONLY && kind == RT_PRIM_MEDIUM) {          // constant_medium.cuh:58-62
        rec.p = ray_at(r, h.t);
        rec.n = mk3(1, 0, 0);
        rec.mat = sc.media[idx].mat;
        return rec;
    }
    Ray q = r;
    rt_instance in;
    const bool through_instance = !SPHERES_ONLY && h.inst >= 0;
    if (through_instance) { in = sc.instances[h.inst]; q = to_object_space(in, r); }
    if (kind == RT_PRIM_SPHERE) {                           // sphere.cuh:68-74
        const rt_sphere s = sc.spheres[idx];
        const f3 cc = fma3(q.tm, ld3(s.vel), ld3(s.c0));
        rec.p = ray_at(q, h.t);
        rec.n = sdiv(rec.p - cc, s.radius);
        // sphere uv (acos / atan2, sphere.cuh:42-49) only where something reads it: rt_scene_create marks the materials
        // whose texture looks at (u, v) -- an image or a uv-offset texture, directly or under a checker -- in `pad`
        if (NEED_UV && sc.materials[s.mat].pad != 0.0f) sphere_uv(rec.n, rec.u, rec.v);
        rec.mat = s.mat;
    } else {                                                // quad.cuh:71-88
        const rt_quad qd = sc.quads[idx];
        const f3 P = ray_at(q, h.t);
        const f3 pl = P - ld3(qd.Q);
        const f3 w = ld3(qd.w);
        rec.u = dot(w, cross(pl, ld3(qd.v)));
        rec.v = dot(w, cross(ld3(qd.u), pl));
        f3 n = ld3(qd.n);
        if (dot(n, q.d) > 0.f) n = -n;
        rec.p = P; rec.n = n; rec.mat = qd.mat;
    }
    if (through_instance) {
        if (in.flags & RT_INST_ROTATE_Y) {                  // hittable.cuh:129-142
            const float c = in.cos_t, s = in.sin_t;
            const float px = fmaf(c, rec.p.x, s * rec.p.z);
            const float pz = fmaf(c, rec.p.z, -(s * rec.p.x));   // -s*x + c*z = c*z - s*x: the c*z product is the fused one
            const float nx = fmaf(c, rec.n.x, s * rec.n.z);
            const float nz = fmaf(c, rec.n.z, -(s * rec.n.x));
            rec.p = mk3(px, rec.p.y, pz);
            rec.n = unit_vector(mk3(nx, rec.n.y, nz));
            // "faces against the original ray": rotate_y::hit sees the translated ray
            f3 dir_seen = r.d;
            if (dot(rec.n, dir_seen) > 0.f) rec.n = -rec.n;
        }
        if (in.flags & RT_INST_TRANSLATE) rec.p = rec.p + ld3(in.offset);   // hittable.cuh:62
    }
    return rec;
}

// ------------------------------------------------------------------ perlin (perlin.cuh:6-83)
DEV uint32_t wanghash(uint32_t x) {
    x = (x ^ 61u) ^ (x >> 16); x *= 9u; x = x ^ (x >> 4); x *= 0x27d4eb2du; x = x ^ (x >> 15); return x;
}
DEV float u2m11(uint32_t h) { return fmaf((float)((h >> 8) & 0x00FFFFFFu), (1.0f / 8388607.5f), -1.0f); }
DEV f3 perlin_grad(int xi, int yi, int zi) {
    const uint32_t h = wanghash((uint32_t)xi * 73856093u ^ (uint32_t)yi * 19349663u ^ (uint32_t)zi * 83492791u);
    return unit_vector(mk3(u2m11(h), u2m11(wanghash(h)), u2m11(wanghash(h ^ 0x9e3779b9u))));
}
DEV float perlin_noise(f3 p) {
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f3 g = perlin_grad(i + a, j + b, k + c);
                const f3 weight = mk3(u - (float)a, v - (float)b, w - (float)c);
                const float s = (a ? uu : (1.0f - uu)) * (b ? vv : (1.0f - vv)) * (c ? ww : (1.0f - ww));
                accum = fmaf(s, dot(g, weight), accum);
            }
    return accum;
}
__device__ __attribute__((noinline)) float perlin_turb(f3 p, int depth) {
    float accum = 0.0f, weight = 1.0f;
    f3 temp = p;
    for (int i = 0; i < depth; ++i) {
        accum = fmaf(weight, perlin_noise(temp), accum);
        weight *= 0.5f;
        temp = mk3(temp.x * 2.0f, temp.y * 2.0f, temp.z * 2.0f);
    }
    return fabsf(accum);
}

// ------------------------------------------------------------------ textures (texture.cuh:7-76)
DEV float clamp01(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
// TEX: 1 = solid + checker only, 2 = every texture kind
template <int TEX>
DEV f3 texture_value(const SceneView& sc, int tex, float u, float v, f3 p) {
    rt_texture t = sc.textures[tex];
    while (t.kind == RT_TEX_CHECKER || (TEX >= 2 && t.kind == RT_TEX_UV_OFFSET)) {
        if (TEX >= 2 && t.kind == RT_TEX_UV_OFFSET) {       // texture.cuh:156-160
            float uu = u + t.scale; uu -= floorf(uu);
            float vv = v + t.p[0]; vv = fminf(fmaxf(vv, 0.f), 1.f);
            u = uu; v = vv;
            t = sc.textures[t.a];
        } else {                                            // texture.cuh:35-42
            const int xi = (int)floorf(t.scale * p.x);
            const int yi = (int)floorf(t.scale * p.y);
            const int zi = (int)floorf(t.scale * p.z);
            const bool is_even = ((xi + yi + zi) & 1) == 0;
            t = sc.textures[is_even ? t.a : t.b];
        }
    }
    if (TEX < 2 || t.kind == RT_TEX_SOLID) return ld3(t.color);
    if (t.kind == RT_TEX_IMAGE) {                           // texture.cuh:51-59
        if (t.a < 0 || t.b <= 0 || t.c <= 0) return mk3(0, 1, 1);
        u = clamp01(u); v = clamp01(v);
        int i = (int)(u * (float)t.b); if (i > t.b - 1) i = t.b - 1;
        int j = (int)((1.f - v) * (float)t.c); if (j > t.c - 1) j = t.c - 1;
        const uint8_t* px = sc.images + (size_t)t.a + (size_t)(j * t.b + i) * 3;
        const float inv255 = 1.f / 255.f;
        return mk3(inv255 * (float)px[0], inv255 * (float)px[1], inv255 * (float)px[2]);
    }
    if (t.kind == RT_TEX_NOODLE) {                          // texture.cuh:94-100
        const f3 dir = mk3(t.p[3], t.p[4], t.p[5]);
        const float uu = dot(p, dir);
        const float wig = perlin_turb(t.p[7] * p, t.a);
        const float stripes = fabsf(cr_sin(fmaf(t.scale, uu, t.p[6] * wig)));
        const float q = clamp01((stripes - 0.75f) / (0.98f - 0.75f));          // smoothstep, texture.cuh:78-82
        const float w = q * q * (3.0f - 2.0f * q);
        return fma3(1.f - w, mk3(t.p[0], t.p[1], t.p[2]), w * ld3(t.color));
    }
    if (t.kind == RT_TEX_FELT) {                            // texture.cuh:124-147
        const float m = perlin_noise(t.scale * p);
        const float phase = fmaf(p.x, t.p[1], 2.0f * perlin_turb(0.5f * p, 2));
        const float fibers = 0.5f * (1.0f + cr_sin(phase));
        float gain = fmaf(t.p[2], fibers - 0.5f, fmaf(t.p[0], m - 0.5f, 1.0f));
        gain = fminf(fmaxf(gain, 0.7f), 1.2f);
        return gain * ld3(t.color);
    }
    // noise (texture.cuh:67-72)
    const float s = cr_sin(fmaf(t.scale, p.z, 10.0f * perlin_turb(p, 7)));
    const float g = 0.5f * (1.0f + s);
    return mk3(g, g, g);
}

// ------------------------------------------------------------------ materials (material.cuh:10-201)
DEV f3 random_in_unit_sphere(rt_xorwow& g) {
    for (;;) {
        const float a = 2.0f * rt_xorwow_uniform(g) - 1.0f;
        const float b = 2.0f * rt_xorwow_uniform(g) - 1.0f;
        const float c = 2.0f * rt_xorwow_uniform(g) - 1.0f;
        const f3 p = mk3(a, b, c);
        if (dot(p, p) < 1.0f) return p;
    }
}
DEV f3 reflect(f3 v, f3 n) { return fma3(-(2.0f * dot(v, n)), n, v); }
DEV bool refract(f3 v, f3 n, float ni_over_nt, f3& refracted) {
    const f3 uv = unit_vector(v);
    const float dt = dot(uv, n);
    const float disc = fmaf(-(ni_over_nt * ni_over_nt), fmaf(-dt, dt, 1.0f), 1.0f);
    if (disc > 0.0f) {
        // ni*(uv - n*dt) - n*sqrt(disc): the second product is the one fused with the subtraction (tuned on the
        // reference's images)
        refracted = fma3(-sqrtf(disc), n, ni_over_nt * fma3(-dt, n, uv));
        return true;
    }
    return false;
}
DEV float schlick(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return fmaf(1.0f - r0, cr_pow5(1.0f - cosine), r0);
}

// dielectric::scatter (material.cuh:119-159): direction of the scattered ray; attenuation is (1,1,1)
DEV f3 dielectric_direction(f3 d_in, f3 n, float ior, rt_xorwow& g) {
    f3 outward_normal;
    const f3 reflected = reflect(d_in, n);
    float ni_over_nt, cosine, reflect_prob;
    f3 refracted = mk3(0.f, 0.f, 0.f);
    const float d_n = dot(d_in, n);
    if (d_n > 0.0f) {
        outward_normal = -n;
        ni_over_nt = ior;
        cosine = d_n / length(d_in);
        cosine = sqrtf(fmaxf(0.0f, fmaf(-(ior * ior), fmaf(-cosine, cosine, 1.0f), 1.0f)));
    } else {
        outward_normal = n;
        ni_over_nt = 1.0f / ior;
        cosine = -d_n / length(d_in);
    }
    if (refract(d_in, outward_normal, ni_over_nt, refracted)) reflect_prob = schlick(cosine, ior);
    else reflect_prob = 1.0f;
    return (rt_xorwow_uniform(g) < reflect_prob) ? reflected : refracted;
}

// emitted + scatter (main.cu:71-83).  Returns false when the path ends here.
template <int TEX>
DEV bool shade(const SceneView& sc, const Ray& in, const HitRec& rec, rt_xorwow& g, f3& emitted, f3& attenuation, Ray& out) {
    const rt_material m = sc.materials[rec.mat];
    emitted = mk3(0.f, 0.f, 0.f);
    out.o = rec.p;
    out.tm = in.tm;
    switch (m.kind) {
    case RT_MAT_LAMBERTIAN: {
        const f3 target = (rec.p + rec.n) + random_in_unit_sphere(g);
        out.d = target - rec.p;
        attenuation = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
        return true;
    }
    case RT_MAT_METAL: {
        const f3 reflected = reflect(unit_vector(in.d), rec.n);
        const f3 rs = random_in_unit_sphere(g);
        out.d = fma3(m.fuzz, rs, reflected);
        attenuation = ld3(m.albedo);
        return dot(out.d, rec.n) > 0.0f;
    }
    case RT_MAT_DIELECTRIC: {
        attenuation = mk3(1.0f, 1.0f, 1.0f);
        out.d = dielectric_direction(in.d, rec.n, m.ior, g);
        return true;
    }
    case RT_MAT_DIFFUSE_LIGHT: {
        emitted = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
        return false;
    }
    default: {   // isotropic
        out.d = random_in_unit_sphere(g);
        attenuation = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
        return true;
    }
    }
}

// ------------------------------------------------------------------ camera (camera.cuh:8-47)
DEV Ray camera_get_ray(const rt_camera& c, float s, float t, rt_xorwow& g) {
    f3 p;
    do {
        const float a = rt_xorwow_uniform(g);
        const float b = rt_xorwow_uniform(g);
        p = 2.0f * mk3(a, b, 0.0f) - mk3(1.0f, 1.0f, 0.0f);
    } while (dot(p, p) >= 1.0f);
    const f3 rd = c.lens_radius * p;
    const f3 cu = ld3(c.u), cv = ld3(c.v);
    const f3 offset = fma3(rd.x, cu, rd.y * cv);
    const double tm = fma((double)rt_xorwow_uniform(g), c.time1 - c.time0, c.time0);
    const f3 origin = ld3(c.origin);
    Ray r;
    r.o = origin + offset;
    r.d = (fma3(t, ld3(c.vertical), fma3(s, ld3(c.horizontal), ld3(c.lower_left_corner))) - origin) - offset;
    r.tm = (float)tm;
    return r;
}

DEV float apply_gamma(float c, float gamma) {   // main.cu:37-42
    if (gamma == 1.0f) return c;
    const float inv = 1.0f / gamma;
    return cr_pow(fmaxf(c, 0.0f), inv);
}

DEV f3 miss_color(const rt_frame_params& fp, const Ray& r) {   // main.cu:59-65
    f3 bg = mk3(fp.background[0], fp.background[1], fp.background[2]);
    if (fp.use_gradient_bg) {
        const f3 ud = unit_vector(r.d);
        const float t = 0.5f * (ud.y + 1.0f);
        bg = mk3(fmaf(t, 0.5f, 1.0f - t), fmaf(t, 0.7f, 1.0f - t), (1.0f - t) + t);   // (1-t)*1 folds to (1-t), t*1 to t
    }
    return bg;
}

// pixel of work item w (8x8 tiles, row-major tiles): returns false if outside
DEV bool work_to_pixel(const rt_frame_params& fp, uint32_t w, int& i, int& lrow) {
    // tile_order (optional): tiles sorted by descending cost from the prepass, so that the most expensive pixels -- whose
    // samples form the longest sequential chains -- start first.  Scheduling only.
    const uint32_t tile = fp.tile_order ? fp.tile_order[w >> 6] : (w >> 6), within = w & 63u;
    const uint32_t tx = tile % (uint32_t)fp.tiles_x, ty = tile / (uint32_t)fp.tiles_x;
    i = (int)(tx * 8u + (within & 7u));
    lrow = (int)(ty * 8u + (within >> 3));
    return i < fp.nx && lrow < fp.local_rows;
}
DEV int local_to_global_row(const rt_frame_params& fp, int lrow) {
    const int t = lrow / fp.tile_rows;
    return (fp.tile_first + t * fp.tile_stride) * fp.tile_rows + (lrow - t * fp.tile_rows);
}

DEV void store_pixel(const rt_frame_params& fp, int i, int lrow, f3 col) {   // main.cu:128-132
    const float k = (float)(1.0 / (double)(float)fp.ns);   // vec3::operator/=(float), vec3.cuh:145-153
    col = mk3(col.x * k, col.y * k, col.z * k);
    float* px = fp.fb + ((size_t)lrow * fp.nx + i) * 3;
    px[0] = apply_gamma(col.x, fp.gamma);
    px[1] = apply_gamma(col.y, fp.gamma);
    px[2] = apply_gamma(col.z, fp.gamma);
}

// stage nodes (+ spheres) into LDS; returns the view the traversal should use
template <int LDS_MODE>
DEV SceneView stage_scene(const rt_scene_dev& sd, unsigned char* lds) {
    SceneView v;
    v.nodes = sd.nodes; v.spheres = sd.spheres; v.quads = sd.quads; v.boxes = sd.boxes; v.instances = sd.instances;
    v.media = sd.media; v.materials = sd.materials; v.textures = sd.textures; v.images = sd.images; v.n_nodes = sd.n_nodes;
    if (LDS_MODE >= 1 && LDS_MODE <= 3) {
        float4* dst = reinterpret_cast<float4*>(lds);
        const float4* src = reinterpret_cast<const float4*>(sd.nodes);
        const int n16 = sd.n_nodes * 2;
        for (int k = threadIdx.x; k < n16; k += blockDim.x) dst[k] = src[k];
        v.nodes = reinterpret_cast<const rt_node*>(lds);
        if (LDS_MODE >= 2) {
            float4* dst2 = dst + n16;
            const float4* src2 = reinterpret_cast<const float4*>(sd.spheres);
            const int m16 = sd.n_spheres * 2;
            for (int k = threadIdx.x; k < m16; k += blockDim.x) dst2[k] = src2[k];
            v.spheres = reinterpret_cast<const rt_sphere*>(dst2);
            if (LDS_MODE >= 3) {   // materials + textures too: one to three dependent L2 round trips less per shaded hit
                float4* dst3 = dst2 + m16;
                const float4* src3 = reinterpret_cast<const float4*>(sd.materials);
                const int q16 = sd.n_materials * 2;
                for (int k = threadIdx.x; k < q16; k += blockDim.x) dst3[k] = src3[k];
                v.materials = reinterpret_cast<const rt_material*>(dst3);
                float4* dst4 = dst3 + q16;
                const float4* src4 = reinterpret_cast<const float4*>(sd.textures);
                const int r16 = sd.n_textures * 4;
                for (int k = threadIdx.x; k < r16; k += blockDim.x) dst4[k] = src4[k];
                v.textures = reinterpret_cast<const rt_texture*>(dst4);
            }
        }
        __syncthreads();
    }
    return v;
}

// closest hit over the whole world for one ray: bvh_node::hit from the root
// (bvh.cuh:95-106) as a stackless walk over the depth-first node array.
template <bool SPHERES_ONLY>
DEV bool trace(const SceneView& sc, const Ray& r, HitInfo& best, unsigned int* pass_count = nullptr) {
    const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const float tmin = 0.001f;   // main.cu:57
    best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    int i = 0;
    const int n = sc.n_nodes;
    while (i < n) {
        const float4 a = nodes4[2 * i], b = nodes4[2 * i + 1];
        int next = RT_NODE_SKIP(__float_as_int(a.w));   // skip link
        if (slab_test(a, b, r.o, inv, tmin, best.t)) {
            if (pass_count) atomicAdd(&pass_count[i], 1u);   // calibration pass only (rt_abi.hip, "collapse")
            const int32_t prim = __float_as_int(b.w);
            if (prim >= 0) leaf_test<SPHERES_ONLY>(sc, prim, r, tmin, best);
            else next = i + 1;
        }
        i = next;
    }
    return best.prim >= 0;
}

DEV bool inv_is_finite(const f3 inv) {
    return fabsf(inv.x) < INFINITY && fabsf(inv.y) < INFINITY && fabsf(inv.z) < INFINITY;
}

// the interval of the ray inside a box, no upper limit (rt_kernel_tier.h: trace_wave)
DEV void slab_interval(const float4 lo_skip, const float4 hi_prim, const f3 o, const f3 inv, float tmin, float& t_enter, float& t_exit) {
    float tmax = FLT_MAX;
    float t0 = (lo_skip.x - o.x) * inv.x, t1 = (hi_prim.x - o.x) * inv.x;
    if (inv.x < 0.0f) { const float tmp = t0; t0 = t1; t1 = tmp; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    t0 = (lo_skip.y - o.y) * inv.y; t1 = (hi_prim.y - o.y) * inv.y;
    if (inv.y < 0.0f) { const float tmp = t0; t0 = t1; t1 = tmp; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    t0 = (lo_skip.z - o.z) * inv.z; t1 = (hi_prim.z - o.z) * inv.z;
    if (inv.z < 0.0f) { const float tmp = t0; t0 = t1; t1 = tmp; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    t_enter = tmin; t_exit = tmax;
}

// aabb::hit for rays whose 1/d components are all finite: identical result to slab_test()
DEV bool slab_test_finite(const float4 lo_skip, const float4 hi_prim, const f3 o, const f3 inv, float tmin, float tmax) {
    const float x0 = (lo_skip.x - o.x) * inv.x, x1 = (hi_prim.x - o.x) * inv.x;
    const float y0 = (lo_skip.y - o.y) * inv.y, y1 = (hi_prim.y - o.y) * inv.y;
    const float z0 = (lo_skip.z - o.z) * inv.z, z1 = (hi_prim.z - o.z) * inv.z;
    const float nearx = fminf(x0, x1), farx = fmaxf(x0, x1);
    const float neary = fminf(y0, y1), fary = fmaxf(y0, y1);
    const float nearz = fminf(z0, z1), farz = fmaxf(z0, z1);
    const float t_in = fmaxf(fmaxf(fmaxf(nearx, neary), nearz), tmin);
    const float t_out = fminf(fminf(fminf(farx, fary), farz), tmax);
    return !(t_out <= t_in);
}

// ------------------------------------------------------------------ the walk loop's box test (round 3)
// The walk may visit a SUPERSET of the boxes the reference enters: interior boxes never change a result and
// every leaf the walk notes has its own box tested again, exactly, before its object test (stage B of the main kernel).  So
// inside the walk loop a cheaper test is exact as long as it passes whenever aabb::hit would: the slab bounds as ONE fma each,
//     (b - o) * x   ->   fma(b, x, -(o * x) -+ e)         (x = 1 / d of that axis, b = the box's lower or upper bound)
// 6 fma instead of 6 subtractions + 6 multiplications per box.  e is the widening that makes the result an outer bound of the
// reference's: with u = 2^-24, the reference's value is within 2.01 u |V| of V = (b - o) x (two roundings), the fused one within
// u (3 |o x| + |b x| + 2 e) of V -+ e (c0 = fl(o x), c = fl(c0 -+ e), one fma), and |V| <= (|b| + |o|) |x|; so
// e >= 5.1 u |x| (|o| + B) suffices for every box of a scene whose coordinates lie within +-B.  e = 2^-20 |x| (|o| + B) is used
// (three times that).  The lower bound of an axis gets - e, the upper + e -- which of lo / hi is which depends on the sign of x.
// Rays for which |x| (|o| + B) could overflow (a direction component below ~1e-27 of the others) take the reference's own
// form instead, like rays with a zero component (loose_ok() is part of `finite_inv`).  A false pass costs one box step; a NaN
// cannot occur (no product overflows), so fmin / fmax see numbers only.
struct LooseRay { f3 clo, chi; };   // per axis: the addend for the box's bmin and for its bmax
DEV bool loose_ok(const f3 inv, const f3 o, const float* bound) {
    const float ex = fabsf(inv.x) * (fabsf(o.x) + bound[0]), ey = fabsf(inv.y) * (fabsf(o.y) + bound[1]), ez = fabsf(inv.z) * (fabsf(o.z) + bound[2]);
    return fmaxf(fmaxf(ex, ey), ez) < 1e30f;
}
DEV LooseRay loose_setup(const f3 inv, const f3 o, const float* bound) {
    const float k = 9.5367431640625e-07f;   // 2^-20
    const float ex = k * (fabsf(inv.x) * (fabsf(o.x) + bound[0])), ey = k * (fabsf(inv.y) * (fabsf(o.y) + bound[1])), ez = k * (fabsf(inv.z) * (fabsf(o.z) + bound[2]));
    const float cx = -(o.x * inv.x), cy = -(o.y * inv.y), cz = -(o.z * inv.z);
    LooseRay r;
    // x > 0: bmin gives the near bound (- e), bmax the far one (+ e); x < 0: the other way round
    r.clo = mk3(inv.x < 0.0f ? cx + ex : cx - ex, inv.y < 0.0f ? cy + ey : cy - ey, inv.z < 0.0f ? cz + ez : cz - ez);
    r.chi = mk3(inv.x < 0.0f ? cx - ex : cx + ex, inv.y < 0.0f ? cy - ey : cy + ey, inv.z < 0.0f ? cz - ez : cz + ez);
    return r;
}
DEV bool slab_test_loose(const float4 lo_skip, const float4 hi_prim, const f3 inv, const LooseRay& lr, float tmin, float tmax) {
    const float x0 = fmaf(lo_skip.x, inv.x, lr.clo.x), x1 = fmaf(hi_prim.x, inv.x, lr.chi.x);
    const float y0 = fmaf(lo_skip.y, inv.y, lr.clo.y), y1 = fmaf(hi_prim.y, inv.y, lr.chi.y);
    const float z0 = fmaf(lo_skip.z, inv.z, lr.clo.z), z1 = fmaf(hi_prim.z, inv.z, lr.chi.z);
    const float nearx = fminf(x0, x1), farx = fmaxf(x0, x1);
    const float neary = fminf(y0, y1), fary = fmaxf(y0, y1);
    const float nearz = fminf(z0, z1), farz = fmaxf(z0, z1);
    const float t_in = fmaxf(fmaxf(fmaxf(nearx, neary), nearz), tmin);
    const float t_out = fminf(fminf(fminf(farx, fary), farz), tmax);
    return !(t_out <= t_in);
}

}  // namespace
