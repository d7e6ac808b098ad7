// rt_device.hip -- the render path on gfx950 (MI355X): render_init + render +
// color + every hit/scatter routine of the reference (src/main.cu:37-133 and
// the .cuh files it reaches), re-designed for CDNA4.
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
DEV float cr_pow(float x, float y) { return (float)pow((double)x, (double)y); }
DEV float cr_pow5(float x) { double d = (double)x; return (float)(d * d * d * d * d); }
DEV float cr_log(float x) { return (float)log((double)x); }
DEV float cr_sin(float x) { return (float)sin((double)x); }
DEV float cr_acos(float x) { return (float)acos((double)x); }
DEV float cr_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

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
DEV bool sphere_test(const rt_sphere& s, const Ray& r, float tmin, float tmax, float& t_out) {
    const f3 cc = fma3(r.tm, ld3(s.vel), ld3(s.c0));
    const f3 oc = r.o - cc;
    const float a = dot(r.d, r.d);
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

// sphere | quad | compound6 (quad.cuh:124-139: closest-hit scan over six faces,
// no box early-out).  On a hit, `leaf` is the resolved sphere/quad ref.
DEV bool simple_test(const SceneView& sc, int32_t ref, const Ray& r, float tmin, float tmax, float& t_out, int32_t& leaf) {
    const int kind = RT_PRIM_KIND(ref), idx = RT_PRIM_INDEX(ref);
    if (kind == RT_PRIM_SPHERE) {
        leaf = ref;
        return sphere_test(sc.spheres[idx], r, tmin, tmax, t_out);
    }
    if (kind == RT_PRIM_QUAD) {
        leaf = ref;
        return quad_test(sc.quads[idx], r, tmin, tmax, t_out);
    }
    // box
    const int first = sc.boxes[idx].first_quad;
    bool any = false;
    float closest = tmax;
    for (int f = 0; f < 6; ++f) {
        float t;
        if (quad_test(sc.quads[first + f], r, tmin, closest, t)) {
            any = true; closest = t; leaf = RT_PRIM_REF(RT_PRIM_QUAD, first + f);
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
DEV bool solid_test(const SceneView& sc, int32_t ref, const Ray& r, float tmin, float tmax, float& t_out, int32_t& leaf, int32_t& inst) {
    if (RT_PRIM_KIND(ref) == RT_PRIM_INSTANCE) {
        inst = RT_PRIM_INDEX(ref);
        const rt_instance in = sc.instances[inst];
        const Ray q = to_object_space(in, r);
        return simple_test(sc, in.child, q, tmin, tmax, t_out, leaf);
    }
    inst = -1;
    return simple_test(sc, ref, r, tmin, tmax, t_out, leaf);
}

// constant_medium::hit (constant_medium.cuh:36-64) behind the 4-argument
// fallback (:67-76), which is the only form a BVH ever calls (bvh.cuh:109-112):
// a private XORWOW seeded from a hash of the ray supplies the one uniform.
DEV bool medium_test(const SceneView& sc, const rt_medium& m, const Ray& r, float tmin, float tmax, float& t_out) {
    rt_xorwow fake;
    const uint32_t seed = 1337u ^ __float_as_uint(r.o.x) ^ __float_as_uint(r.o.y * 3.1f) ^ __float_as_uint(r.d.z * 5.7f);
    rt_xorwow_seed(fake, (uint64_t)seed);
    float t1, t2;
    int32_t leaf, inst;
    if (!solid_test(sc, m.boundary, r, -FLT_MAX, FLT_MAX, t1, leaf, inst)) return false;
    if (!solid_test(sc, m.boundary, r, t1 + 1e-4f, FLT_MAX, t2, leaf, inst)) return false;
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
template <bool SPHERES_ONLY>
DEV void leaf_test(const SceneView& sc, int32_t ref, const Ray& r, float tmin, HitInfo& best) {
    float t;
    if (SPHERES_ONLY) {
        if (sphere_test(sc.spheres[RT_PRIM_INDEX(ref)], r, tmin, best.t, t)) { best.t = t; best.prim = ref; best.inst = -1; }
        return;
    }
    int32_t leaf = ref, inst = -1;
    if (RT_PRIM_KIND(ref) == RT_PRIM_MEDIUM) {
        if (medium_test(sc, sc.media[RT_PRIM_INDEX(ref)], r, tmin, best.t, t)) { best.t = t; best.prim = ref; best.inst = -1; }
        return;
    }
    if (solid_test(sc, ref, r, tmin, best.t, t, leaf, inst)) { best.t = t; best.prim = leaf; best.inst = inst; }
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
        if (NEED_UV) sphere_uv(rec.n, rec.u, rec.v);
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
DEV float perlin_turb(f3 p, int depth) {
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
    if (LDS_MODE >= 1) {
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
DEV bool trace(const SceneView& sc, const Ray& r, HitInfo& best) {
    const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const float tmin = 0.001f;   // main.cu:57
    best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    int i = 0;
    const int n = sc.n_nodes;
    while (i < n) {
        const float4 a = nodes4[2 * i], b = nodes4[2 * i + 1];
        int next = __float_as_int(a.w);   // skip link
        if (slab_test(a, b, r.o, inv, tmin, best.t)) {
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

// One ray, the whole wave: the 64 lanes test 64 consecutive nodes of the depth-first array at once, then the walk the
// reference would take through them (box hit -> next index or leaf test, box miss -> skip link) is replayed with scalar
// bit tests on the ballot, v_readlane for the skip links.  Used by tier-1 waves, which hold a single pixel whose
// sequential chain bounds the frame time: a lone lane's traversal is LDS-latency bound (one dependent node read per
// step); here one pair of wide reads serves several steps.  Every box is still compared with the closest hit that the
// reference would have at that visit: the batch is abandoned as soon as a leaf test changes it.  All arguments are
// wave-uniform (every lane carries the same ray); the result is identical in all lanes.
template <bool SPHERES_ONLY>
DEV bool trace_wide(const SceneView& sc, const Ray& r, HitInfo& best) {
    const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const float tmin = 0.001f;
    best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const int n = sc.n_nodes;
    const int lane = (int)(threadIdx.x & 63u);
    int i = 0;
    while (i < n) {
        const int idx = i + lane;
        const bool valid = idx < n;
        const float4 a = nodes4[2 * (valid ? idx : 0)], b = nodes4[2 * (valid ? idx : 0) + 1];
        const bool pass = valid && slab_test(a, b, r.o, inv, tmin, best.t);
        const int my_skip = __float_as_int(a.w), my_prim = __float_as_int(b.w);
        const unsigned long long pass_mask = __ballot(pass);
        const unsigned long long leaf_mask = __ballot(valid && my_prim >= 0);
        const int end = (i + 64 < n) ? i + 64 : n;
        int j = i;
        while (j < end) {
            const int bit = j - i;
            if (!((pass_mask >> bit) & 1ull)) { j = __builtin_amdgcn_readlane(my_skip, bit); continue; }
            if ((leaf_mask >> bit) & 1ull) {
                const float before = best.t;
                const int32_t before_prim = best.prim;
                leaf_test<SPHERES_ONLY>(sc, __builtin_amdgcn_readlane(my_prim, bit), r, tmin, best);
                j = __builtin_amdgcn_readlane(my_skip, bit);
                if (best.t != before || best.prim != before_prim) break;   // later boxes must see the new limit
            } else {
                j = j + 1;
            }
        }
        i = j;
    }
    return best.prim >= 0;
}


// One ray, the whole workgroup ("tier 0", spheres-only scenes): every leaf of the tree is tested at once, one per
// thread -- its own box with no limit, then its sphere -- and the closest hit is the minimum of (t, node index) over the
// workgroup.  Why this is the reference's answer for finite 1/d:
//   * a leaf the reference tests has a box that passes with the limit of that moment, hence with none: the set tested
//     here is a superset of the reference's;
//   * a box contains its descendants' boxes and the slab test is monotone in the box, so a leaf whose own box passes has
//     ancestors that pass; the reference can only skip it because an earlier hit b satisfies b <= T_A <= T_leaf (entry
//     distances).  The winner here has t <= every other candidate's t, so the b in force when the reference reaches its
//     ancestors is >= t; if t > T_leaf (checked below) then b > T_A for every ancestor and the reference tests it too;
//   * the reference keeps a hit only if t < closest, i.e. the minimum with ties to the first visited = lowest index.
// A candidate hit at or before its own box's entry distance (rounding on a grazing ray) or a zero direction component
// falls back to the reference's walk, replayed by every thread.  One barrier per ray (slots are double-buffered).
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
DEV bool trace_group(const SceneView& sc, const Ray& r, HitInfo& best, const unsigned int* leaves, int n_leaves,
                     unsigned long long* slots, int& parity) {
    const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const float tmin = 0.001f;
    if (!inv_is_finite(inv)) return trace<true>(sc, r, best);   // workgroup-uniform
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    unsigned long long key = ~0ull;
    bool anomaly = false;
    for (int q = (int)threadIdx.x; q < n_leaves; q += (int)blockDim.x) {
        const int node = (int)leaves[q];
        const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
        float t_enter, t_exit;
        slab_interval(a, b, r.o, inv, tmin, t_enter, t_exit);
        if (!(t_exit <= t_enter)) {
            float t;
            if (sphere_test(sc.spheres[RT_PRIM_INDEX(__float_as_int(b.w))], r, tmin, FLT_MAX, t)) {
                if (!(t > t_enter)) anomaly = true;
                const unsigned long long k = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)node;
                if (k < key) key = k;
            }
        }
    }
    if (anomaly) key = 0ull;
    unsigned long long have = __ballot(key != ~0ull);
    unsigned long long wkey = ~0ull;
    const int lo = (int)(unsigned int)key, hi = (int)(unsigned int)(key >> 32);
    while (have != 0ull) {
        const int k = __ffsll((long long)have) - 1;
        have &= have - 1ull;
        const unsigned long long kk = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(hi, k) << 32) |
                                      (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(lo, k);
        if (kk < wkey) wkey = kk;
    }
    unsigned long long* mine = slots + parity * 16;
    if ((threadIdx.x & 63u) == 0u) mine[threadIdx.x >> 6] = wkey;
    __syncthreads();
    unsigned long long m = ~0ull;
    const int n_waves = (int)(blockDim.x >> 6);
    for (int w = 0; w < n_waves; ++w) { const unsigned long long v = mine[w]; if (v < m) m = v; }
    parity ^= 1;
    if (m == 0ull) return trace<true>(sc, r, best);   // a grazing hit at or before its box's entry: the reference's walk decides
    best.inst = -1;
    if (m == ~0ull) { best.t = FLT_MAX; best.prim = -1; return false; }
    best.t = __uint_as_float((unsigned int)(m >> 32));
    best.prim = sc.nodes[(int)(unsigned int)m].prim;
    return true;
}

}  // namespace

// =============================================================================
// Kernel A ("pixel"): the reference's own loop nest, one lane per pixel, one
// 8x8 tile per wave.  Kept as the simple form the persistent kernel is checked
// against on the GPU, and as the A/B baseline for the scheduling work.
// =============================================================================
template <bool SPHERES_ONLY, int TEX, bool NEED_UV, int LDS_MODE>
__global__ void __launch_bounds__(256) rt_render_pixel_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const SceneView sc = stage_scene<LDS_MODE>(sd, lds);

    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    int i, lrow;
    unsigned long long rays = 0;
    if (w < fp.work_items && work_to_pixel(fp, w, i, lrow)) {
        const int j = local_to_global_row(fp, lrow);
        rt_xorwow g;
        rt_xorwow_seed(g, fp.seed_base + (uint64_t)(j * fp.nx + i));   // render_init, main.cu:101-104
        f3 col = mk3(0, 0, 0);
        for (int s = 0; s < fp.ns; ++s) {
            const float u = ((float)i + rt_xorwow_uniform(g)) / (float)fp.nx;
            const float v = ((float)j + rt_xorwow_uniform(g)) / (float)fp.ny;
            Ray cur = camera_get_ray(sd.camera, u, v, g);
            f3 throughput = mk3(1, 1, 1), radiance = mk3(0, 0, 0);
            for (int bounce = 0; bounce < 50; ++bounce) {
                HitInfo h;
                ++rays;
                if (!trace<SPHERES_ONLY>(sc, cur, h)) {
                    radiance = fma3(throughput, miss_color(fp, cur), radiance);
                    break;
                }
                const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, cur, h);
                f3 emitted, attenuation;
                Ray scattered;
                const bool go_on = shade<TEX>(sc, cur, rec, g, emitted, attenuation, scattered);
                radiance = fma3(throughput, emitted, radiance);
                if (!go_on) break;
                throughput = throughput * attenuation;
                cur = scattered;
            }
            col = col + radiance;
        }
        store_pixel(fp, i, lrow, col);
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(fp.ray_counter, rays);
}

// =============================================================================
// Kernel B ("persistent"): flattened state machine with path regeneration.
//
// Per lane: TRAVERSE (walking the node array for the current ray) or WAIT
// (traversal finished; needs shading, and after shading either a bounce ray,
// a new sample, a new pixel, or nothing).  Each trip of the outer loop runs
// up to `steps_per_trip` node visits for the traversing lanes, then a ballot
// decides whether the shading block is worth running: when at least
// `shade_threshold` lanes wait, or nobody is traversing any more.
// =============================================================================
template <bool SPHERES_ONLY, int TEX, bool NEED_UV, int LDS_MODE>
__global__ void __launch_bounds__(RT_PERSISTENT_THREADS) rt_render_persistent_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const SceneView sc = stage_scene<LDS_MODE>(sd, lds);
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const int n_nodes = sc.n_nodes;
    const float tmin = 0.001f;

    // lane state
    rt_xorwow g = {0, 0, 0, 0, 0, 0};
    int px_i = 0, px_lrow = 0, px_j = 0, sample = 0, bounce = 0;
    f3 col = mk3(0, 0, 0), throughput = mk3(1, 1, 1), radiance = mk3(0, 0, 0);
    Ray cur; cur.o = mk3(0, 0, 0); cur.d = mk3(0, 0, 1); cur.tm = 0.f;
    f3 inv = mk3(0, 0, 0);
    HitInfo best; best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    int node = n_nodes;          // == n_nodes: traversal finished
    bool alive = true;           // lane still has (or may fetch) work
    bool have_pixel = false;     // lane owns a pixel
    bool pending_hit = false;    // a finished traversal is waiting for the shading block
    unsigned long long rays = 0;

    for (;;) {
        // ---------------- traversal steps
        for (int step = 0; step < fp.steps_per_trip; ++step) {
            if (node < n_nodes) {
                const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                int next = __float_as_int(a.w);
                if (slab_test(a, b, cur.o, inv, tmin, best.t)) {
                    const int32_t prim = __float_as_int(b.w);
                    if (prim >= 0) leaf_test<SPHERES_ONLY>(sc, prim, cur, tmin, best);
                    else next = node + 1;
                }
                node = next;
            }
        }
        const bool traversing = node < n_nodes;
        const bool waiting = alive && !traversing;
        const unsigned long long wait_mask = __ballot(waiting);
        const unsigned long long trav_mask = __ballot(traversing);
        if (wait_mask == 0ull && trav_mask == 0ull) break;   // every lane has drained
        if (trav_mask != 0ull && __popcll(wait_mask) < fp.shade_threshold) continue;

        // ---------------- shading / regeneration block
        if (waiting) {
            bool need_sample = !pending_hit;   // first trip, or path ended
            if (pending_hit) {
                pending_hit = false;
                if (best.prim < 0) {                                   // miss (main.cu:57-68)
                    radiance = fma3(throughput, miss_color(fp, cur), radiance);
                    need_sample = true;
                } else {
                    const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, cur, best);
                    f3 emitted, attenuation;
                    Ray scattered;
                    const bool go_on = shade<TEX>(sc, cur, rec, g, emitted, attenuation, scattered);
                    radiance = fma3(throughput, emitted, radiance);        // main.cu:71
                    ++bounce;
                    if (!go_on || bounce >= 50) need_sample = true;    // main.cu:54,76-80
                    else { throughput = throughput * attenuation; cur = scattered; }
                }
                if (need_sample) { col = col + radiance; ++sample; }   // main.cu:124
            }
            if (need_sample) {
                if (have_pixel && sample >= fp.ns) { store_pixel(fp, px_i, px_lrow, col); have_pixel = false; }
                while (!have_pixel && alive) {
                    const uint32_t w = atomicAdd(fp.work_counter, 1u);
                    if (w >= fp.work_items) { alive = false; break; }
                    if (work_to_pixel(fp, w, px_i, px_lrow)) {
                        px_j = local_to_global_row(fp, px_lrow);
                        rt_xorwow_seed(g, fp.seed_base + (uint64_t)(px_j * fp.nx + px_i));
                        col = mk3(0, 0, 0); sample = 0; have_pixel = true;
                    }
                }
                if (alive) {                                           // main.cu:121-123
                    const float u = ((float)px_i + rt_xorwow_uniform(g)) / (float)fp.nx;
                    const float v = ((float)px_j + rt_xorwow_uniform(g)) / (float)fp.ny;
                    cur = camera_get_ray(sd.camera, u, v, g);
                    throughput = mk3(1, 1, 1); radiance = mk3(0, 0, 0); bounce = 0;
                }
            }
            if (alive) {   // launch the traversal of `cur`
                inv = mk3(1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z);
                best.t = FLT_MAX; best.prim = -1; best.inst = -1;
                node = 0;
                pending_hit = true;
                ++rays;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(fp.ray_counter, rays);
}

// =============================================================================
// Kernel C ("parked"): kernel B with the leaf tests taken out of the node loop.
//
// Profiling kernel B (profiles/r01a_persistent_summary.txt) showed ~19 of 64
// lanes active per VALU instruction: about one node visit in twenty is a leaf,
// so in nearly every step some lane reaches one and the whole wave walks
// through the object test for two or three lanes.  Here a lane whose box test
// passes at a leaf PARKS (keeps its state, stops stepping); after the trip's
// node steps one leaf pass serves every parked lane together.  A parked lane
// is always served before its next box test, so each ray still sees exactly
// the reference's sequence of tests and limits: results stay bit-identical.
//
// The slab test uses min/max (v_min3/v_max3) instead of the reference's
// sign-select + ternaries.  For a finite 1/d the two forms select the same
// values (products are never NaN, near = min(t0,t1), far = max(t0,t1), and the
// running tmin/tmax are never NaN); a ray with a zero (or denormal-overflow)
// direction component keeps the reference's form, chosen per trip by ballot.
// =============================================================================
namespace {
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
}  // namespace

template <bool SPHERES_ONLY, int TEX, bool NEED_UV, int LDS_MODE>
__global__ void __launch_bounds__(RT_PERSISTENT_THREADS, (SPHERES_ONLY && TEX < 2) ? RT_PARKED_MIN_WAVES : 2) rt_render_parked_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const SceneView sc = stage_scene<LDS_MODE>(sd, lds);
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const int n_nodes = sc.n_nodes;
    const float tmin = 0.001f;

    rt_xorwow g = {0, 0, 0, 0, 0, 0};
    int px_i = 0, px_lrow = 0, px_j = 0, sample = 0, bounce = 0;
    f3 col = mk3(0, 0, 0), throughput = mk3(1, 1, 1), radiance = mk3(0, 0, 0);
    Ray cur; cur.o = mk3(0, 0, 0); cur.d = mk3(0, 0, 1); cur.tm = 0.f;
    f3 inv = mk3(1, 1, 1);
    HitInfo best; best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    int node = n_nodes;
    int32_t parked = -1;         // leaf primitive waiting for the leaf pass
    bool alive = true, have_pixel = false, pending_hit = false;
    bool finite_inv = true;
    unsigned int rays = 0;

    for (;;) {
        // ---------------- node steps (parked and finished lanes sit out)
        if (__ballot(!finite_inv && node < n_nodes) == 0ull) {
            for (int step = 0; step < fp.steps_per_trip; ++step) {
                if (node < n_nodes && parked < 0) {
                    const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                    int next = __float_as_int(a.w);
                    if (slab_test_finite(a, b, cur.o, inv, tmin, best.t)) {
                        const int32_t prim = __float_as_int(b.w);
                        if (prim >= 0) parked = prim;
                        else next = node + 1;
                    }
                    node = next;
                }
            }
        } else {   // some lane's ray has a zero direction component: the reference's own form for everyone
            for (int step = 0; step < fp.steps_per_trip; ++step) {
                if (node < n_nodes && parked < 0) {
                    const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                    int next = __float_as_int(a.w);
                    if (slab_test(a, b, cur.o, inv, tmin, best.t)) {
                        const int32_t prim = __float_as_int(b.w);
                        if (prim >= 0) parked = prim;
                        else next = node + 1;
                    }
                    node = next;
                }
            }
        }
        // ---------------- leaf pass
        const unsigned long long park_mask = __ballot(parked >= 0);
        const unsigned long long step_mask = __ballot(node < n_nodes && parked < 0);
        if (park_mask != 0ull && (__popcll(park_mask) >= fp.leaf_threshold || step_mask == 0ull)) {
            if (parked >= 0) {
                leaf_test<SPHERES_ONLY>(sc, parked, cur, tmin, best);
                parked = -1;
            }
        }
        const bool traversing = node < n_nodes || parked >= 0;
        const bool waiting = alive && !traversing;
        const unsigned long long wait_mask = __ballot(waiting);
        const unsigned long long trav_mask = __ballot(traversing);
        if (wait_mask == 0ull && trav_mask == 0ull) break;
        if (trav_mask != 0ull && __popcll(wait_mask) < fp.shade_threshold) continue;

        // ---------------- shading / regeneration block
        if (waiting) {
            bool need_sample = !pending_hit;
            if (pending_hit) {
                pending_hit = false;
                if (best.prim < 0) {
                    radiance = fma3(throughput, miss_color(fp, cur), radiance);
                    need_sample = true;
                } else {
                    const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, cur, best);
                    f3 emitted, attenuation;
                    Ray scattered;
                    const bool go_on = shade<TEX>(sc, cur, rec, g, emitted, attenuation, scattered);
                    radiance = fma3(throughput, emitted, radiance);
                    ++bounce;
                    if (!go_on || bounce >= 50) need_sample = true;
                    else { throughput = throughput * attenuation; cur = scattered; }
                }
                if (need_sample) { col = col + radiance; ++sample; }
            }
            if (need_sample) {
                if (have_pixel && sample >= fp.ns) { store_pixel(fp, px_i, px_lrow, col); have_pixel = false; }
                while (!have_pixel && alive) {
                    const uint32_t w = atomicAdd(fp.work_counter, 1u);
                    if (w >= fp.work_items) { alive = false; break; }
                    if (work_to_pixel(fp, w, px_i, px_lrow)) {
                        px_j = local_to_global_row(fp, px_lrow);
                        rt_xorwow_seed(g, fp.seed_base + (uint64_t)(px_j * fp.nx + px_i));
                        col = mk3(0, 0, 0); sample = 0; have_pixel = true;
                    }
                }
                if (alive) {
                    const float u = ((float)px_i + rt_xorwow_uniform(g)) / (float)fp.nx;
                    const float v = ((float)px_j + rt_xorwow_uniform(g)) / (float)fp.ny;
                    cur = camera_get_ray(sd.camera, u, v, g);
                    throughput = mk3(1, 1, 1); radiance = mk3(0, 0, 0); bounce = 0;
                }
            }
            if (alive) {
                inv = mk3(1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z);
                finite_inv = inv_is_finite(inv);
                best.t = FLT_MAX; best.prim = -1; best.inst = -1;
                node = 0;
                pending_hit = true;
                ++rays;
            }
        }
    }
    unsigned long long r64 = rays;
    for (int off = 32; off > 0; off >>= 1) r64 += __shfl_down(r64, off, 64);
    if ((threadIdx.x & 63) == 0 && r64) atomicAdd(fp.ray_counter, r64);
}

// =============================================================================
// Kernel D ("staged"): kernel C with the shading block cut into stages.
//
// Counters for kernel C (profiles/r01_bench_n1_parked_summary.txt, pmcsweep) put
// ~60 % of all issued VALU instructions in the shading block: ~1500
// instructions covering every material, the dielectric path, pixel
// finalisation and camera-ray generation, executed for ~24 waiting lanes of
// which each needs a fraction.  A microbenchmark (tools/ubench/valu_rate.hip)
// shows the SIMDs are close to issue-bound at ~3-4 cycles per VALU
// wave-instruction, so instructions issued for idle lanes are the cost.
//
// Here a lane's state is encoded in `node`:
//   0 <= node < n      walking the BVH (next box test)
//   node < 0           parked at a leaf; ~node is where the walk resumes
//   node == n + 0      traversal finished, hit/miss not yet classified
//   node == n + 1      path ended: needs accumulate + next sample / pixel + camera ray
//   node == n + 2      dielectric hit waiting for the (rare, long) dielectric stage
//   node == n + 3      new ray ready, needs per-ray setup (1/d etc.)
//   node == n + 4      no more work
// and each stage runs when a ballot finds enough lanes for it, or when no lane
// can walk any more.  Lambertian, metal and isotropic share one
// random_in_unit_sphere loop.  Stages only re-order work between lanes: every
// pixel still draws its own XORWOW stream in the reference's order.
// =============================================================================
// Diagnostic build only (-DRT_DIAG): per-stage execution counts, written to fp.diag (never to an output).
#ifdef RT_DIAG
#define DIAG_ADD(slot, value) do { const unsigned long long v_ = (unsigned long long)(value); if ((threadIdx.x & 63) == 0) diag_local[slot] += v_; } while (0)
#else
#define DIAG_ADD(slot, value) do { } while (0)
#endif

template <bool SPHERES_ONLY, int TEX, bool NEED_UV, int LDS_MODE>
__global__ void __launch_bounds__(RT_PERSISTENT_THREADS, (SPHERES_ONLY && TEX < 2) ? RT_PARKED_MIN_WAVES : 2) rt_render_staged_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef RT_DIAG
    unsigned long long diag_local[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const SceneView sc = stage_scene<LDS_MODE>(sd, lds);
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const int n_nodes = sc.n_nodes;
    const int ST_DONE = n_nodes, ST_NEWPATH = n_nodes + 1, ST_DIEL = n_nodes + 2, ST_SETUP = n_nodes + 3, ST_DEAD = n_nodes + 4;
    const float tmin = 0.001f;

    rt_xorwow g = {0, 0, 0, 0, 0, 0};
    int px_i = 0, px_lrow = 0, px_j = 0, sample = 0, bounce = 0;
    f3 col = mk3(0, 0, 0), throughput = mk3(1, 1, 1), radiance = mk3(0, 0, 0);
    Ray cur; cur.o = mk3(0, 0, 0); cur.d = mk3(0, 0, 1); cur.tm = 0.f;
    f3 inv = mk3(1, 1, 1);
    HitInfo best; best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    int node = ST_NEWPATH;       // every lane starts by asking for a pixel
    int32_t parked = -1;
    bool have_pixel = false, first = true, finite_inv = true;
    unsigned int rays = 0, rays_at_pixel_start = 0;
    // Sparse mode (see rt_abi.hip, "heavy tiles"): the first fp.sparse_wgs workgroups start by serving the queue of the
    // few dearest tiles with only every fp.sparse_stride-th lane, because a lane's rays advance ~2.5x faster in a wave
    // with few live lanes and those pixels' sequential chains bound the frame time.  When that queue is drained and the
    // wave's own heavy pixels are finished it becomes an ordinary wave.  Wave-uniform.
    bool sparse = (int)blockIdx.x < fp.sparse_wgs;
    // Tier-1 waves hold ONE pixel each -- the dearest pixels of the frame, whose sequential chains bound the frame time.
    // With a single live lane the state machine below is pure overhead, so they run the reference's plain loop nest
    // (as kernel A does) on pixels parked by part 1, one after another, and only then join the ordinary waves.
    // Tier-0 workgroups (spheres-only scenes) go one step further for the very dearest pixels: the whole workgroup holds
    // ONE pixel and every ray is traced by all its threads at once (trace_group()), which cuts the time per ray -- and
    // with it the sequential chain that bounds the frame and every multi-GPU partition of it -- several times over.
    const bool tier0 = SPHERES_ONLY && LDS_MODE == 2 && (int)blockIdx.x < fp.tier0_wgs && fp.state_in != nullptr;   // workgroup-uniform
    const bool tier1 = sparse && !tier0 && (int)blockIdx.x < fp.tier0_wgs + fp.tier1_wgs && fp.state_in != nullptr;
    if (tier0 || tier1) {
        __builtin_amdgcn_s_setprio(3);
        unsigned int* t0_scratch = reinterpret_cast<unsigned int*>(lds + fp.tier0_lds_offset);
        unsigned long long* t0_slots = reinterpret_cast<unsigned long long*>(t0_scratch + 4);   // [2][16]
        unsigned int* t0_leaves = t0_scratch + 4 + 64;
        int t0_n_leaves = 0, t0_parity = 0;
        if (tier0) {
            if (threadIdx.x == 0) t0_scratch[0] = 0u;
            __syncthreads();
            for (int k = (int)threadIdx.x; k < n_nodes; k += (int)blockDim.x)
                if (sc.nodes[k].prim >= 0) t0_leaves[atomicAdd(&t0_scratch[0], 1u)] = (unsigned int)k;
            __syncthreads();
            t0_n_leaves = (int)t0_scratch[0];
        }
        // every lane of the wave (tier 1) / thread of the workgroup (tier 0) carries the same pixel and computes the same
        // values; only the traversal is shared out
        for (;;) {
            uint32_t idx = 0;
            if (tier0) {
                if (threadIdx.x == 0) t0_scratch[1] = atomicAdd(fp.work_counter + 3, 1u);
                __syncthreads();
                idx = t0_scratch[1];
                __syncthreads();
                if (idx >= fp.tier0_items) break;
            } else {
                if ((threadIdx.x & 63) == 0) idx = atomicAdd(fp.work_counter + 2, 1u);
                idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
                if (idx >= fp.tier1_items) break;
                idx += fp.tier0_items;
            }
            const uint32_t pix = fp.heavy_pixels[idx];
            const int lrow = (int)(pix / (uint32_t)fp.nx), i = (int)(pix - (uint32_t)lrow * (uint32_t)fp.nx);
            const int j = local_to_global_row(fp, lrow);
            const rt_pixel_state st = fp.state_in[pix];
            rt_xorwow pg;
            pg.v0 = st.rng[0]; pg.v1 = st.rng[1]; pg.v2 = st.rng[2]; pg.v3 = st.rng[3]; pg.v4 = st.rng[4]; pg.d = st.rng[5];
            f3 pcol = mk3(st.col[0], st.col[1], st.col[2]);
            unsigned int pixel_rays = 0;
            for (int sidx = fp.sample_begin; sidx < fp.sample_end; ++sidx) {                 // main.cu:119-125
                const float u = ((float)i + rt_xorwow_uniform(pg)) / (float)fp.nx;
                const float v = ((float)j + rt_xorwow_uniform(pg)) / (float)fp.ny;
                Ray r = camera_get_ray(sd.camera, u, v, pg);
                f3 thr = mk3(1, 1, 1), rad = mk3(0, 0, 0);
                for (int depth = 0; depth < 50; ++depth) {                                   // main.cu:54-84
                    HitInfo h;
                    ++pixel_rays;
#ifdef RT_DIAG
                    const unsigned long long dg0 = __builtin_readcyclecounter();
#endif
                    const bool hit = tier0 ? trace_group(sc, r, h, t0_leaves, t0_n_leaves, t0_slots, t0_parity) : trace_wide<SPHERES_ONLY>(sc, r, h);
#ifdef RT_DIAG
                    const unsigned long long dg1 = __builtin_readcyclecounter();
                    if (threadIdx.x == 0) { diag_local[15] += 1; diag_local[14] += dg1 - dg0; }   // slots 14/15: tier loops only
#endif
                    if (!hit) { rad = fma3(thr, miss_color(fp, r), rad); break; }
                    const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, r, h);
                    f3 emitted, attenuation;
                    Ray scattered;
                    const bool go_on = shade<TEX>(sc, r, rec, pg, emitted, attenuation, scattered);

                    rad = fma3(thr, emitted, rad);
                    if (!go_on) break;
                    thr = thr * attenuation;
                    r = scattered;
                }
                pcol = pcol + rad;
            }
            if (tier0 ? threadIdx.x == 0 : (threadIdx.x & 63) == 0) {
                if (fp.state_out) {   // a middle part of a split frame: park the pixel again
                    rt_pixel_state so;
                    so.rng[0] = pg.v0; so.rng[1] = pg.v1; so.rng[2] = pg.v2; so.rng[3] = pg.v3; so.rng[4] = pg.v4; so.rng[5] = pg.d;
                    so.col[0] = pcol.x; so.col[1] = pcol.y; so.col[2] = pcol.z; so.cost = fp.state_in[pix].cost + pixel_rays;
                    fp.state_out[pix] = so;
                    atomicAdd(&fp.tile_cost[(lrow >> 3) * fp.tiles_x + (i >> 3)], pixel_rays);
                } else {
                    store_pixel(fp, i, lrow, pcol);
                }
                rays += pixel_rays;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        sparse = false;   // queue drained: this wave / workgroup becomes ordinary
    }
    // A sparse wave's few lanes are on the frame's critical path: let it win instruction-issue arbitration against the
    // three ordinary waves sharing its SIMD (priority outranks age, MI355X_MICROARCH.md "Two waves per SIMD").
    if (sparse && fp.sparse_priority > 0) {
        if (fp.sparse_priority == 1) __builtin_amdgcn_s_setprio(1);
        else if (fp.sparse_priority == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }

    for (;;) {
        DIAG_ADD(0, 1);
        // ---------------- stage A: node steps
        if (__ballot(!finite_inv && (unsigned)node < (unsigned)n_nodes) == 0ull) {
            const int trip_steps = sparse ? 2 * fp.steps_per_trip : fp.steps_per_trip;
            for (int step = 0; step < trip_steps; ++step) {
                // nobody left walking (all parked or finished): end the trip now -- this is what keeps the latency of
                // a wave's last few live lanes near one node step per step (end of frame, small multi-GPU partitions)
                if (__ballot((unsigned)node < (unsigned)n_nodes) == 0ull) break;
                DIAG_ADD(1, 1); DIAG_ADD(2, __popcll(__ballot((unsigned)node < (unsigned)n_nodes)));
                if ((unsigned)node < (unsigned)n_nodes) {
                    const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                    const bool pass = slab_test_finite(a, b, cur.o, inv, tmin, best.t);
                    const int32_t prim = __float_as_int(b.w);
                    const int skip = __float_as_int(a.w);
                    const bool at_leaf = pass && prim >= 0;
                    const int next = (pass && prim < 0) ? node + 1 : skip;
                    parked = at_leaf ? prim : parked;
                    node = at_leaf ? ~next : next;
                }
            }
        } else {   // a lane's ray has a zero direction component: the reference's own slab form for this trip
            for (int step = 0; step < fp.steps_per_trip; ++step) {
                if (__ballot((unsigned)node < (unsigned)n_nodes) == 0ull) break;
                if ((unsigned)node < (unsigned)n_nodes) {
                    const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                    const bool pass = slab_test(a, b, cur.o, inv, tmin, best.t);
                    const int32_t prim = __float_as_int(b.w);
                    const int skip = __float_as_int(a.w);
                    const bool at_leaf = pass && prim >= 0;
                    const int next = (pass && prim < 0) ? node + 1 : skip;
                    parked = at_leaf ? prim : parked;
                    node = at_leaf ? ~next : next;
                }
            }
        }
        // ---------------- stage B: leaf pass for parked lanes
        DIAG_ADD(3, __ballot(node < 0) != 0ull); DIAG_ADD(4, __popcll(__ballot(node < 0)));
        if (SPHERES_ONLY) {
            if (node < 0) {
                leaf_test<true>(sc, parked, cur, tmin, best);
                parked = -1;
                node = ~node;
            }
        } else {
            // General scenes: spheres and quads are served every trip; boxes / instances (six quad tests, a transform)
            // and media (two boundary tests, a private XORWOW, a logarithm) are long, so their lanes stay parked until a
            // ballot finds enough of them -- or nobody is left who could step.
            const int kind = node < 0 ? RT_PRIM_KIND(parked) : -1;
            if (kind == RT_PRIM_SPHERE || kind == RT_PRIM_QUAD) {
                float t;
                const bool hit = kind == RT_PRIM_SPHERE ? sphere_test(sc.spheres[RT_PRIM_INDEX(parked)], cur, tmin, best.t, t)
                                                        : quad_test(sc.quads[RT_PRIM_INDEX(parked)], cur, tmin, best.t, t);
                if (hit) { best.t = t; best.prim = parked; best.inst = -1; }
                parked = -1;
                node = ~node;
            }
            const bool nobody_steps = __ballot((unsigned)node < (unsigned)n_nodes) == 0ull;
            const int live_b = __popcll(__ballot(node != ST_DEAD));
            const unsigned long long box_mask = __ballot(kind == RT_PRIM_BOX || kind == RT_PRIM_INSTANCE);
            if (box_mask != 0ull && (nobody_steps || __popcll(box_mask) >= 1 + ((fp.box_threshold - 1) * live_b >> 6))) {
                if (kind == RT_PRIM_BOX || kind == RT_PRIM_INSTANCE) {
                    float t;
                    int32_t leaf = parked, inst = -1;
                    if (solid_test(sc, parked, cur, tmin, best.t, t, leaf, inst)) { best.t = t; best.prim = leaf; best.inst = inst; }
                    parked = -1;
                    node = ~node;
                }
            }
            const unsigned long long med_mask = __ballot(kind == RT_PRIM_MEDIUM);
            if (med_mask != 0ull && (nobody_steps || __popcll(med_mask) >= 1 + ((fp.medium_threshold - 1) * live_b >> 6))) {
                if (kind == RT_PRIM_MEDIUM) {
                    float t;
                    if (medium_test(sc, sc.media[RT_PRIM_INDEX(parked)], cur, tmin, best.t, t)) { best.t = t; best.prim = parked; best.inst = -1; }
                    parked = -1;
                    node = ~node;
                }
            }
        }
        // a finished walk that hit nothing (main.cu:57-68) needs no stage: add the background and end the path now
        if (node == ST_DONE && best.prim < 0) {
            radiance = fma3(throughput, miss_color(fp, cur), radiance);
            node = ST_NEWPATH;
        }
        const unsigned long long walking = __ballot(node < n_nodes);
        const bool force = walking == 0ull;
        const int n_done = __popcll(__ballot(node == ST_DONE));
        bool ran_stage = false;
        // The thresholds are fractions of the lanes that still have work: a wave whose lanes are running out of pixels
        // (end of the frame, or a small row partition on a multi-GPU run) must not wait for 24 lanes it no longer has.
        const int live = __popcll(__ballot(node != ST_DEAD));
        const int shade_need = 1 + ((fp.shade_threshold - 1) * live >> 6);
        const int diel_need = 1 + ((fp.diel_threshold - 1) * live >> 6);
        const int newpath_need = 1 + ((fp.newpath_threshold - 1) * live >> 6);
        const bool eager = sparse && fp.sparse_eager;     // sparse waves trade their own throughput for latency

        // ---------------- stage C: classify + resolve + diffuse/metal/isotropic scatter
        if (n_done > 0 && (n_done >= shade_need || force || eager)) {
            ran_stage = true;
            DIAG_ADD(5, 1); DIAG_ADD(6, n_done);
            DIAG_ADD(11, __popcll(__ballot(node == ST_DONE && best.prim >= 0)));
            if (node == ST_DONE) {
                {
                    const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, cur, best);
                    const rt_material m = sc.materials[rec.mat];
                    if (m.kind == RT_MAT_DIELECTRIC) {
                        if (sparse) {
                            // a sparse wave's pixels are mostly glass: scatter here instead of queueing for stage D
                            // (saves a second resolve_hit and a stage round trip on the frame's critical chain)
                            const f3 dir = dielectric_direction(cur.d, rec.n, m.ior, g);
                            ++bounce;
                            if (bounce >= 50) node = ST_NEWPATH;
                            else { cur.o = rec.p; cur.d = dir; node = ST_SETUP; }   // attenuation (1,1,1): throughput unchanged
                        } else {
                            node = ST_DIEL;
                        }
                    } else if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
                        const f3 emitted = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                        radiance = fma3(throughput, emitted, radiance);         // main.cu:71, scatter() false
                        node = ST_NEWPATH;
                    } else {
                        // lambertian / metal / isotropic: one shared rejection loop (material.cuh:12-18)
                        const f3 rs = random_in_unit_sphere(g);
                        f3 dir, attenuation;
                        bool go_on = true;
                        if (m.kind == RT_MAT_METAL) {                       // material.cuh:99-109
                            const f3 reflected = reflect(unit_vector(cur.d), rec.n);
                            dir = fma3(m.fuzz, rs, reflected);
                            attenuation = ld3(m.albedo);
                            go_on = dot(dir, rec.n) > 0.0f;
                        } else {
                            if (m.kind == RT_MAT_LAMBERTIAN) {              // material.cuh:75-86
                                const f3 target = (rec.p + rec.n) + rs;
                                dir = target - rec.p;
                            } else {                                        // isotropic, material.cuh:193-200
                                dir = rs;
                            }
                            attenuation = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                        }
                        ++bounce;
                        if (!go_on || bounce >= 50) node = ST_NEWPATH;      // main.cu:54,76-80
                        else {
                            throughput = throughput * attenuation;
                            cur.o = rec.p; cur.d = dir;                     // time carried over
                            node = ST_SETUP;
                        }
                    }
                }
            }
        }
        // ---------------- stage D: dielectric scatter (material.cuh:119-159)
        {
            const int n_diel = __popcll(__ballot(node == ST_DIEL));
            if (n_diel > 0 && (n_diel >= diel_need || force || eager)) {
                ran_stage = true;
                DIAG_ADD(7, 1); DIAG_ADD(8, n_diel);
                if (node == ST_DIEL) {
                    const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, cur, best);
                    const f3 dir = dielectric_direction(cur.d, rec.n, sc.materials[rec.mat].ior, g);
                    ++bounce;
                    if (bounce >= 50) node = ST_NEWPATH;
                    else { throughput = throughput * mk3(1.0f, 1.0f, 1.0f); cur.o = rec.p; cur.d = dir; node = ST_SETUP; }
                }
            }
        }
        // ---------------- stage E: path end -> next sample / next pixel -> camera ray (main.cu:119-132)
        {
            const int n_new = __popcll(__ballot(node == ST_NEWPATH));
            if (n_new > 0 && (n_new >= newpath_need || force || eager)) {
                ran_stage = true;
                DIAG_ADD(9, 1); DIAG_ADD(10, n_new);
                if (node == ST_NEWPATH) {
                    if (!first) { col = col + radiance; ++sample; }
                    first = false;
                    bool alive = true;
                    if (have_pixel && sample >= fp.sample_end) {
                        if (fp.state_out) {
                            // first part of a split frame: park the pixel at this sample boundary (no path is in flight
                            // here, so the XORWOW state and the colour sum are the whole state) and record what it cost
                            const unsigned int c = rays - rays_at_pixel_start;
                            const size_t at = (size_t)px_lrow * fp.nx + px_i;
                            rt_pixel_state st;
                            st.rng[0] = g.v0; st.rng[1] = g.v1; st.rng[2] = g.v2; st.rng[3] = g.v3; st.rng[4] = g.v4; st.rng[5] = g.d;
                            st.col[0] = col.x; st.col[1] = col.y; st.col[2] = col.z;
                            st.cost = c + (fp.state_in ? fp.state_in[at].cost : 0u);   // a middle part adds to what the pixel cost before
                            fp.state_out[at] = st;
                            atomicAdd(&fp.tile_cost[(px_lrow >> 3) * fp.tiles_x + (px_i >> 3)], c);
                        } else {
                            store_pixel(fp, px_i, px_lrow, col);
                        }
                        have_pixel = false;
                    }
                    while (!have_pixel && alive) {
                        bool ok;
                        if (sparse) {
                            // heavy list (sorted by descending cost): tier 1 = its first tier1_items entries
                            // tier 2 of the heavy list (tier 1 is served by the plain loop at the top of the kernel)
                            if (((threadIdx.x & 63) % (unsigned)fp.sparse_stride) != 0u) { alive = false; break; }
                            const uint32_t at = fp.tier0_items + fp.tier1_items + atomicAdd(fp.work_counter + 1, 1u);
                            if (at >= fp.heavy_items) { alive = false; break; }
                            const uint32_t pix = fp.heavy_pixels[at];
                            px_lrow = (int)(pix / (uint32_t)fp.nx); px_i = (int)(pix - (uint32_t)px_lrow * (uint32_t)fp.nx);
                            ok = true;
                        } else {
                            const uint32_t w = atomicAdd(fp.work_counter, 1u);
                            if (w >= fp.work_items) { alive = false; break; }
                            ok = work_to_pixel(fp, w, px_i, px_lrow);
                            // pixels in the heavy list belong to the sparse waves
                            if (ok && fp.heavy_items && fp.state_in[(size_t)px_lrow * fp.nx + px_i].cost >= fp.heavy_threshold) ok = false;
                        }
                        if (ok) {
                            px_j = local_to_global_row(fp, px_lrow);
                            if (fp.state_in) {   // second part of a split frame: pick the pixel up where the first part left it
                                const rt_pixel_state st = fp.state_in[(size_t)px_lrow * fp.nx + px_i];
                                g.v0 = st.rng[0]; g.v1 = st.rng[1]; g.v2 = st.rng[2]; g.v3 = st.rng[3]; g.v4 = st.rng[4]; g.d = st.rng[5];
                                col = mk3(st.col[0], st.col[1], st.col[2]);
                            } else {
                                rt_xorwow_seed(g, fp.seed_base + (uint64_t)(px_j * fp.nx + px_i));
                                col = mk3(0, 0, 0);
                            }
                            sample = fp.sample_begin; have_pixel = true;
                            rays_at_pixel_start = rays;
                        }
                    }
                    if (alive) {
                        const float u = ((float)px_i + rt_xorwow_uniform(g)) / (float)fp.nx;
                        const float v = ((float)px_j + rt_xorwow_uniform(g)) / (float)fp.ny;
                        cur = camera_get_ray(sd.camera, u, v, g);
                        throughput = mk3(1, 1, 1); radiance = mk3(0, 0, 0); bounce = 0;
                        node = ST_SETUP;
                    } else {
                        node = ST_DEAD;
                    }
                }
            }
        }
        // ---------------- stage F: per-ray setup
        if (ran_stage) {
            DIAG_ADD(12, 1); DIAG_ADD(13, __popcll(__ballot(node == ST_SETUP)));
            if (node == ST_SETUP) {
                inv = mk3(1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z);
                finite_inv = inv_is_finite(inv);
                best.t = FLT_MAX; best.prim = -1; best.inst = -1;
                node = n_nodes > 0 ? 0 : ST_DONE;
                ++rays;
            }
        } else if (force) {
            // nothing walking, nothing waiting: every lane is ST_DEAD
            if (!sparse) break;
            sparse = false;                         // heavy queue drained and our heavy pixels done: become an ordinary wave
            __builtin_amdgcn_s_setprio(0);
            node = ST_NEWPATH; first = true; have_pixel = false;
        }
    }
    unsigned long long r64 = rays;
    for (int off = 32; off > 0; off >>= 1) r64 += __shfl_down(r64, off, 64);
    if ((threadIdx.x & 63) == 0 && r64) atomicAdd(fp.ray_counter, r64);
#ifdef RT_DIAG
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 16; ++k) atomicAdd(fp.ray_counter + 1 + k, diag_local[k]);
#endif
}

// =============================================================================
// Kernel E ("wavefront"): bulk-synchronous wavefront path tracing inside one workgroup.
//
// Stage counters for kernel D (profiles/r01_diag_staged_stage_counts.txt): box
// steps are ~55 % of all issued instructions and run with 22.5 of 64 lanes,
// because a lane owns one pixel and sits idle while its ray waits for a stage.
// Here no lane owns anything.  The workgroup keeps P ray slots in LDS (ray,
// 1/d, traversal result, path throughput/radiance, and the slot's pixel:
// XORWOW state, colour sum, sample count) and compact lists of slot ids.  An
// iteration is three phases separated by __syncthreads():
//   T  lanes pop slots from READY (in a strided order, so that one wave does
//      not get 64 neighbouring pixels), walk the BVH (node steps + parked leaf
//      pass as in kernel C), push hits to DONE and misses to NEWPATH, and pop
//      again; when READY is empty and a wave runs low on walking lanes it
//      saves its rays' walk state and re-queues them for the next iteration;
//   C  lanes pop DONE slots: light -> NEWPATH; lambertian / metal / isotropic /
//      dielectric scatter -> READY(next) (NEWPATH when absorbed or at depth 50);
//   E  lanes pop NEWPATH slots: add the background for misses, accumulate,
//      next sample or next pixel, camera ray -> READY(next).
// Within a phase every list is either consumed (through an atomic cursor) or
// appended to (through an atomic count), never both, counters are reset in a
// phase that does not touch them, and a slot is touched only by the lane that
// popped it: no spin waits, nothing to deadlock on.  As in kernels B-D this
// only re-orders work: a slot's pixel draws its own XORWOW stream in the
// reference's order, so frames are bit-identical.
//
// Spheres-only scenes with inline / solid / checker textures (the headline
// random scene); other scenes use kernel D.
// =============================================================================
namespace {

enum { F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_TM, F_BEST_T, F_IX, F_IY, F_IZ, F_THR, F_RAD = F_THR + 3, F_COL = F_RAD + 3, F_COUNT = F_COL + 3 };
enum { I_NODE, I_BEST_PRIM, I_BOUNCE, I_SAMPLE, I_PX, I_RNG, I_COUNT = I_RNG + 6 };
enum { C_READY_POS, C_READY_COUNT0, C_READY_COUNT1, C_DONE_COUNT, C_DONE_POS, C_NEW_COUNT0, C_NEW_COUNT1, C_NEW_POS0, C_NEW_POS1, C_WALKING, C_NUM };

// Reserve one list entry for every lane with `want` set: one LDS atomic per wave.  Must be called with all
// 64 lanes active.  Returns this lane's index (valid only where want).
DEV int wave_reserve(unsigned int* counter, bool want) {
    const unsigned long long m = __ballot(want);
    const int cnt = __popcll(m);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && cnt) base = (int)atomicAdd(counter, (unsigned int)cnt);
    base = __shfl(base, 0, 64);
    const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    return base + below;
}

}  // namespace

#ifdef RT_DIAG
#define WF_STAMP(slot_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); wf_t[slot_] += now_ - wf_mark; wf_mark = now_; } while (0)
#else
#define WF_STAMP(slot_) do { } while (0)
#endif

template <int TEX, int LDS_MODE>
__global__ void __launch_bounds__(1024) rt_render_wavefront_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const SceneView sc = stage_scene<LDS_MODE>(sd, lds);
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const int n_nodes = sc.n_nodes;
    const float tmin = 0.001f;
    const int P = fp.wf_slots;
    const int lane = threadIdx.x & 63;

    // ---- carve the slot pool (structure of arrays) and the lists out of LDS behind the staged scene
    size_t off = 0;
    if (LDS_MODE >= 1) off += (size_t)sd.n_nodes * sizeof(rt_node);
    if (LDS_MODE >= 2) off += (size_t)sd.n_spheres * sizeof(rt_sphere);
    off = (off + 15) & ~(size_t)15;
    float* const slot_f = reinterpret_cast<float*>(lds + off); off += (size_t)P * 4 * F_COUNT;
    int32_t* const slot_i = reinterpret_cast<int32_t*>(lds + off); off += (size_t)P * 4 * I_COUNT;
#define SF(field, slot_) slot_f[(field) * P + (slot_)]
#define SI(field, slot_) slot_i[(field) * P + (slot_)]
    uint16_t* const ready_a = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const ready_b = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const done = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const new_a = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const new_b = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    off = (off + 15) & ~(size_t)15;
    unsigned int* const ctr = reinterpret_cast<unsigned int*>(lds + off);

    // every slot starts by asking for a pixel (sample = -1 marks "no pixel yet")
    for (int s = threadIdx.x; s < P; s += blockDim.x) { new_a[s] = (uint16_t)s; SI(I_SAMPLE, s) = -1; }
    if (threadIdx.x < C_NUM) ctr[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) ctr[C_NEW_COUNT0] = (unsigned int)P;
    __syncthreads();

    unsigned int rays = 0;
#ifdef RT_DIAG
    unsigned long long wf_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // T, barrier, C, barrier, E, barrier
    unsigned long long wf_mark = __builtin_amdgcn_s_memtime();
#endif
    // Every iteration retires at least one ray segment per live slot or advances a paused walk, so the frame needs
    // far fewer iterations than the cap; hitting it means a scheduling bug, reported through the error flag
    // (ray_counter[31]) instead of hanging the GPU.
    for (unsigned int iteration = 0;; ++iteration) {
        if (iteration >= fp.wf_max_iterations) { if (threadIdx.x == 0) atomicAdd(fp.ray_counter + 31, 1ull); break; }
        const int cur = (int)(iteration & 1u);
        const int n_ready = (int)ctr[C_READY_COUNT0 + cur];     // written before the previous iteration's last barrier
        if (iteration > 0 && n_ready == 0) break;                // every live slot is in READY at this point
        unsigned int* const ready_count_next = &ctr[C_READY_COUNT0 + (cur ^ 1)];
        uint16_t* const ready_cur = cur ? ready_b : ready_a;
        uint16_t* const ready_next = cur ? ready_a : ready_b;
        uint16_t* const newpath = cur ? new_b : new_a;           // NEWPATH is double-buffered by iteration parity
        unsigned int* const new_count = &ctr[C_NEW_COUNT0 + cur];
        unsigned int* const new_pos = &ctr[C_NEW_POS0 + cur];

        // =============================== T phase ===============================
        if (threadIdx.x == 0) { ctr[C_NEW_COUNT0 + (cur ^ 1)] = 0; ctr[C_NEW_POS0 + (cur ^ 1)] = 0; }   // last iteration's list is dead
        if (n_ready > 0) {
            // stride for the pop order: a prime that does not divide n_ready, so idx -> (idx * stride) % n_ready is a bijection
            const int stride = (n_ready % 37) ? 37 : ((n_ready % 41) ? 41 : 43);
            int slot = -1, node = n_nodes;
            int32_t parked = -1;
            f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(1, 1, 1);
            float tm = 0.f;
            HitInfo best; best.t = FLT_MAX; best.prim = -1; best.inst = -1;
            bool finite_inv = true, exhausted = false;
            // Re-queueing is only worth it (and only safe against livelock) when plenty of rays are around and this
            // wave has advanced its rays at least one trip since it last looked.
            const bool may_pause = 2 * n_ready >= (int)blockDim.x;
            int trips = 0;
            for (;;) {
                // -- finished walks: hits -> DONE, misses -> NEWPATH (they only need the background, added in E)
                {
                    const bool fin = slot >= 0 && node >= n_nodes;
                    if (__ballot(fin) != 0ull) {
                        const bool hit = fin && best.prim >= 0, miss = fin && best.prim < 0;
                        const int idx = wave_reserve(&ctr[C_DONE_COUNT], hit);
                        const int idx2 = wave_reserve(new_count, miss);
                        if (hit) { SF(F_BEST_T, slot) = best.t; SI(I_BEST_PRIM, slot) = best.prim; done[idx] = (uint16_t)slot; }
                        if (miss) { SI(I_BEST_PRIM, slot) = -2; newpath[idx2] = (uint16_t)slot; }   // -2: "missed, background pending"
                        if (fin) { slot = -1; ++rays; }      // one finished world->hit call (main.cu:57)
                        if (lane == 0) atomicSub(&ctr[C_WALKING], (unsigned int)__popcll(__ballot(fin)));
                    }
                }
                // -- refill idle lanes from READY
                if (!exhausted) {
                    const bool want = slot < 0;
                    const unsigned long long wm = __ballot(want);
                    if (wm != 0ull) {
                        const int idx = wave_reserve(&ctr[C_READY_POS], want);
                        const int first_idx = __shfl(idx, __ffsll((long long)wm) - 1, 64);
                        if (first_idx + __popcll(wm) >= n_ready) exhausted = true;
                        const unsigned long long got = __ballot(want && idx < n_ready);
                        if (lane == 0 && got != 0ull) atomicAdd(&ctr[C_WALKING], (unsigned int)__popcll(got));
                        if (want && idx < n_ready) {
                            slot = ready_cur[(idx * stride) % n_ready];
                            o = mk3(SF(F_OX, slot), SF(F_OY, slot), SF(F_OZ, slot));
                            d = mk3(SF(F_DX, slot), SF(F_DY, slot), SF(F_DZ, slot));
                            tm = SF(F_TM, slot);
                            node = SI(I_NODE, slot);
                            best.t = SF(F_BEST_T, slot); best.prim = SI(I_BEST_PRIM, slot);
                            inv = mk3(SF(F_IX, slot), SF(F_IY, slot), SF(F_IZ, slot));   // 1/d, stored by whoever made the ray
                            finite_inv = inv_is_finite(inv);
                            parked = -1;
                        }
                    }
                }
                const unsigned long long active = __ballot(slot >= 0);
                if (active == 0ull) break;
                // Once READY is empty the whole workgroup should stop walking at about the same time (the barrier waits
                    // for the slowest wave), so the test is on the workgroup's walking-lane count, not this wave's.
                const int wg_walking = (int)*reinterpret_cast<volatile unsigned int*>(&ctr[C_WALKING]);
                if (exhausted && may_pause && trips > 0 && wg_walking * 64 < fp.wf_pause_lanes * (int)blockDim.x) {
                    // -- READY is empty and the workgroup is running dry: save the walks, re-queue them
                    const bool have = slot >= 0;
                    if (lane == 0) atomicSub(&ctr[C_WALKING], (unsigned int)__popcll(active));
                    const int idx = wave_reserve(ready_count_next, have);
                    if (have) {
                        SI(I_NODE, slot) = node;
                        SF(F_BEST_T, slot) = best.t; SI(I_BEST_PRIM, slot) = best.prim;
                        ready_next[idx] = (uint16_t)slot;
                    }
                    break;
                }
                // -- node steps
                ++trips;
                Ray cur_ray; cur_ray.o = o; cur_ray.d = d; cur_ray.tm = tm;
                if (__ballot(!finite_inv && slot >= 0) == 0ull) {
                    for (int step = 0; step < fp.steps_per_trip; ++step) {
                        if ((unsigned)node < (unsigned)n_nodes) {
                            const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                            const bool pass = slab_test_finite(a, b, o, inv, tmin, best.t);
                            const int32_t prim = __float_as_int(b.w);
                            const int skip = __float_as_int(a.w);
                            const bool at_leaf = pass && prim >= 0;
                            const int next = (pass && prim < 0) ? node + 1 : skip;
                            parked = at_leaf ? prim : parked;
                            node = at_leaf ? ~next : next;
                        }
                    }
                } else {
                    for (int step = 0; step < fp.steps_per_trip; ++step) {
                        if ((unsigned)node < (unsigned)n_nodes) {
                            const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                            const bool pass = slab_test(a, b, o, inv, tmin, best.t);
                            const int32_t prim = __float_as_int(b.w);
                            const int skip = __float_as_int(a.w);
                            const bool at_leaf = pass && prim >= 0;
                            const int next = (pass && prim < 0) ? node + 1 : skip;
                            parked = at_leaf ? prim : parked;
                            node = at_leaf ? ~next : next;
                        }
                    }
                }
                // -- leaf pass
                if (node < 0) {
                    leaf_test<true>(sc, parked, cur_ray, tmin, best);
                    parked = -1;
                    node = ~node;
                }
            }
        }
        WF_STAMP(0);
        __syncthreads();
        WF_STAMP(1);

        // =============================== C phase ===============================
        if (threadIdx.x == 0) { ctr[C_READY_POS] = 0; ctr[C_READY_COUNT0 + cur] = 0; }   // READY(cur) was consumed in T
        {
            const int n_done = (int)ctr[C_DONE_COUNT];
            for (;;) {
                int base = 0;
                if (lane == 0) base = (int)atomicAdd(&ctr[C_DONE_POS], 64u);
                base = __shfl(base, 0, 64);
                if (base >= n_done) break;
                const bool have = base + lane < n_done;
                const int slot = have ? (int)done[base + lane] : 0;
                int dest = 0;   // 1 -> READY(next), 2 -> NEWPATH
                if (have) {
                    Ray r;
                    r.o = mk3(SF(F_OX, slot), SF(F_OY, slot), SF(F_OZ, slot));
                    r.d = mk3(SF(F_DX, slot), SF(F_DY, slot), SF(F_DZ, slot));
                    r.tm = SF(F_TM, slot);
                    HitInfo h; h.t = SF(F_BEST_T, slot); h.prim = SI(I_BEST_PRIM, slot); h.inst = -1;
                    const f3 thr = mk3(SF(F_THR, slot), SF(F_THR + 1, slot), SF(F_THR + 2, slot));
                    const HitRec rec = resolve_hit<true, false>(sc, r, h);
                    const rt_material m = sc.materials[rec.mat];
                    if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
                        const f3 emitted = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                        // main.cu:71; scatter() is false
                        SF(F_RAD, slot) = fmaf(thr.x, emitted.x, SF(F_RAD, slot)); SF(F_RAD + 1, slot) = fmaf(thr.y, emitted.y, SF(F_RAD + 1, slot));
                        SF(F_RAD + 2, slot) = fmaf(thr.z, emitted.z, SF(F_RAD + 2, slot));
                        dest = 2;
                    } else {
                        rt_xorwow g;
                        g.v0 = (uint32_t)SI(I_RNG, slot); g.v1 = (uint32_t)SI(I_RNG + 1, slot); g.v2 = (uint32_t)SI(I_RNG + 2, slot);
                        g.v3 = (uint32_t)SI(I_RNG + 3, slot); g.v4 = (uint32_t)SI(I_RNG + 4, slot); g.d = (uint32_t)SI(I_RNG + 5, slot);
                        f3 dir, attenuation;
                        bool go_on = true;
                        if (m.kind == RT_MAT_DIELECTRIC) {                  // material.cuh:119-159
                            dir = dielectric_direction(r.d, rec.n, m.ior, g);
                            attenuation = mk3(1.0f, 1.0f, 1.0f);
                        } else {
                            const f3 rs = random_in_unit_sphere(g);         // shared by lambertian / metal / isotropic
                            if (m.kind == RT_MAT_METAL) {
                                const f3 reflected = reflect(unit_vector(r.d), rec.n);
                                dir = fma3(m.fuzz, rs, reflected);
                                attenuation = ld3(m.albedo);
                                go_on = dot(dir, rec.n) > 0.0f;
                            } else {
                                if (m.kind == RT_MAT_LAMBERTIAN) {
                                    const f3 target = (rec.p + rec.n) + rs;
                                    dir = target - rec.p;
                                } else {
                                    dir = rs;
                                }
                                attenuation = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                            }
                        }
                        SI(I_RNG, slot) = (int32_t)g.v0; SI(I_RNG + 1, slot) = (int32_t)g.v1; SI(I_RNG + 2, slot) = (int32_t)g.v2;
                        SI(I_RNG + 3, slot) = (int32_t)g.v3; SI(I_RNG + 4, slot) = (int32_t)g.v4; SI(I_RNG + 5, slot) = (int32_t)g.d;
                        const int bounce = SI(I_BOUNCE, slot) + 1;
                        SI(I_BOUNCE, slot) = bounce;
                        if (!go_on || bounce >= 50) dest = 2;               // main.cu:54,76-80
                        else {
                            const f3 t2 = thr * attenuation;
                            SF(F_THR, slot) = t2.x; SF(F_THR + 1, slot) = t2.y; SF(F_THR + 2, slot) = t2.z;
                            SF(F_OX, slot) = rec.p.x; SF(F_OY, slot) = rec.p.y; SF(F_OZ, slot) = rec.p.z;
                            SF(F_DX, slot) = dir.x; SF(F_DY, slot) = dir.y; SF(F_DZ, slot) = dir.z;
                            SF(F_IX, slot) = 1.0f / dir.x; SF(F_IY, slot) = 1.0f / dir.y; SF(F_IZ, slot) = 1.0f / dir.z;
                            SI(I_NODE, slot) = 0; SF(F_BEST_T, slot) = FLT_MAX; SI(I_BEST_PRIM, slot) = -1;
                            dest = 1;
                        }
                    }
                }
                { const int idx = wave_reserve(ready_count_next, dest == 1); if (dest == 1) ready_next[idx] = (uint16_t)slot; }
                { const int idx = wave_reserve(new_count, dest == 2); if (dest == 2) newpath[idx] = (uint16_t)slot; }
            }
        }
        WF_STAMP(2);
        __syncthreads();
        WF_STAMP(3);

        // =============================== E phase ===============================
        if (threadIdx.x == 0) { ctr[C_DONE_COUNT] = 0; ctr[C_DONE_POS] = 0; }   // DONE was consumed in C
        {
            const int n_new = (int)*new_count;
            for (;;) {
                int base = 0;
                if (lane == 0) base = (int)atomicAdd(new_pos, 64u);
                base = __shfl(base, 0, 64);
                if (base >= n_new) break;
                const bool have = base + lane < n_new;
                const int slot = have ? (int)newpath[base + lane] : 0;
                bool alive = false;
                if (have) {
                    alive = true;
                    int sample = SI(I_SAMPLE, slot);
                    int px = SI(I_PX, slot);
                    f3 col = mk3(0, 0, 0);
                    rt_xorwow g = {0, 0, 0, 0, 0, 0};
                    bool have_pixel = sample >= 0;
                    if (have_pixel) {
                        f3 rad = mk3(SF(F_RAD, slot), SF(F_RAD + 1, slot), SF(F_RAD + 2, slot));
                        if (SI(I_BEST_PRIM, slot) == -2) {                                      // the path ended on a miss (main.cu:57-68)
                            Ray mr; mr.o = mk3(0, 0, 0); mr.tm = 0.f;
                            mr.d = mk3(SF(F_DX, slot), SF(F_DY, slot), SF(F_DZ, slot));
                            const f3 thr = mk3(SF(F_THR, slot), SF(F_THR + 1, slot), SF(F_THR + 2, slot));
                            rad = fma3(thr, miss_color(fp, mr), rad);                                 // radiance += throughput * bg
                        }
                        col = mk3(SF(F_COL, slot) + rad.x, SF(F_COL + 1, slot) + rad.y, SF(F_COL + 2, slot) + rad.z);   // col += color(...), main.cu:124
                        ++sample;
                        g.v0 = (uint32_t)SI(I_RNG, slot); g.v1 = (uint32_t)SI(I_RNG + 1, slot); g.v2 = (uint32_t)SI(I_RNG + 2, slot);
                        g.v3 = (uint32_t)SI(I_RNG + 3, slot); g.v4 = (uint32_t)SI(I_RNG + 4, slot); g.d = (uint32_t)SI(I_RNG + 5, slot);
                        if (sample >= fp.ns) { store_pixel(fp, px & 0xFFFF, (int)((unsigned)px >> 16), col); have_pixel = false; }
                    }
                    int px_i = px & 0xFFFF, px_lrow = (int)((unsigned)px >> 16);
                    while (!have_pixel && alive) {
                        const uint32_t w = atomicAdd(fp.work_counter, 1u);
                        if (w >= fp.work_items) { alive = false; break; }
                        if (work_to_pixel(fp, w, px_i, px_lrow)) {
                            const int px_j = local_to_global_row(fp, px_lrow);
                            rt_xorwow_seed(g, fp.seed_base + (uint64_t)(px_j * fp.nx + px_i));
                            col = mk3(0, 0, 0); sample = 0; have_pixel = true;
                            px = px_i | (px_lrow << 16);
                        }
                    }
                    if (alive) {
                        const int px_j = local_to_global_row(fp, px_lrow);
                        const float u = ((float)px_i + rt_xorwow_uniform(g)) / (float)fp.nx;
                        const float v = ((float)px_j + rt_xorwow_uniform(g)) / (float)fp.ny;
                        const Ray r = camera_get_ray(sd.camera, u, v, g);
                        SF(F_OX, slot) = r.o.x; SF(F_OY, slot) = r.o.y; SF(F_OZ, slot) = r.o.z;
                        SF(F_DX, slot) = r.d.x; SF(F_DY, slot) = r.d.y; SF(F_DZ, slot) = r.d.z;
                        SF(F_IX, slot) = 1.0f / r.d.x; SF(F_IY, slot) = 1.0f / r.d.y; SF(F_IZ, slot) = 1.0f / r.d.z;
                        SF(F_TM, slot) = r.tm;
                        SF(F_THR, slot) = 1.f; SF(F_THR + 1, slot) = 1.f; SF(F_THR + 2, slot) = 1.f;
                        SF(F_RAD, slot) = 0.f; SF(F_RAD + 1, slot) = 0.f; SF(F_RAD + 2, slot) = 0.f;
                        SF(F_COL, slot) = col.x; SF(F_COL + 1, slot) = col.y; SF(F_COL + 2, slot) = col.z;
                        SI(I_BOUNCE, slot) = 0; SI(I_SAMPLE, slot) = sample; SI(I_PX, slot) = px;
                        SI(I_RNG, slot) = (int32_t)g.v0; SI(I_RNG + 1, slot) = (int32_t)g.v1; SI(I_RNG + 2, slot) = (int32_t)g.v2;
                        SI(I_RNG + 3, slot) = (int32_t)g.v3; SI(I_RNG + 4, slot) = (int32_t)g.v4; SI(I_RNG + 5, slot) = (int32_t)g.d;
                        SI(I_NODE, slot) = 0; SF(F_BEST_T, slot) = FLT_MAX; SI(I_BEST_PRIM, slot) = -1;
                    }
                }
                { const int idx = wave_reserve(ready_count_next, alive); if (alive) ready_next[idx] = (uint16_t)slot; }
            }
        }
        WF_STAMP(4);
        __syncthreads();
        WF_STAMP(5);
    }
    unsigned long long r64 = rays;
    for (int off2 = 32; off2 > 0; off2 >>= 1) r64 += __shfl_down(r64, off2, 64);
    if (lane == 0 && r64) atomicAdd(fp.ray_counter, r64);
#ifdef RT_DIAG
    if (lane == 0) { for (int q = 0; q < 8; ++q) atomicAdd(fp.ray_counter + 1 + q, wf_t[q]); atomicAdd(fp.ray_counter + 9, 1ull); }
#endif
#undef SF
#undef SI
}

// Heavy-pixel list for the cost-aware schedule: every pixel whose prepass ray count reaches `threshold` is appended as
// (cost << 32 | pixel); the host sorts the (short) list by descending cost.
__global__ void rt_collect_heavy_kernel(const rt_pixel_state* state, unsigned int n_pixels, unsigned int threshold,
                                        unsigned long long* list, unsigned int capacity, unsigned int* count) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const unsigned int c = state[i].cost;
    if (c >= threshold) {
        const unsigned int at = atomicAdd(count, 1u);
        if (at < capacity) list[at] = ((unsigned long long)c << 32) | i;
    }
}
void rt_launch_collect_heavy(const rt_pixel_state* state, unsigned int n_pixels, unsigned int threshold, unsigned long long* list,
                             unsigned int capacity, unsigned int* count, hipStream_t st) {
    hipLaunchKernelGGL(rt_collect_heavy_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, st, state, n_pixels, threshold, list, capacity, count);
}

// ------------------------------------------------------------------ launch table
namespace {

template <bool SO, int TX, bool UV, int LM>
void launch_variant(int kernel, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (kernel == RT_KERNEL_PIXEL) {
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_pixel_kernel<SO, TX, UV, LM>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((rt_render_pixel_kernel<SO, TX, UV, LM>), grid, block, lds, st, sd, fp);
    } else if (kernel == RT_KERNEL_PERSISTENT) {
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_persistent_kernel<SO, TX, UV, LM>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((rt_render_persistent_kernel<SO, TX, UV, LM>), grid, block, lds, st, sd, fp);
    } else if (kernel == RT_KERNEL_PARKED) {
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_parked_kernel<SO, TX, UV, LM>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((rt_render_parked_kernel<SO, TX, UV, LM>), grid, block, lds, st, sd, fp);
    } else {
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_staged_kernel<SO, TX, UV, LM>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((rt_render_staged_kernel<SO, TX, UV, LM>), grid, block, lds, st, sd, fp);
    }
}

template <bool SO, int TX, bool UV>
void launch_lds(int kernel, int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (lds_mode == 3 && kernel == RT_KERNEL_STAGED) {   // only the shipping kernel carries the materials-in-LDS variant
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_staged_kernel<SO, TX, UV, 3>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((rt_render_staged_kernel<SO, TX, UV, 3>), grid, block, lds, st, sd, fp);
        return;
    }
    if (lds_mode >= 2) launch_variant<SO, TX, UV, 2>(kernel, sd, fp, grid, block, lds, st);
    else if (lds_mode == 1) launch_variant<SO, TX, UV, 1>(kernel, sd, fp, grid, block, lds, st);
    else launch_variant<SO, TX, UV, 0>(kernel, sd, fp, grid, block, lds, st);
}

}  // namespace

void rt_launch_wavefront(int lds_mode, int tex_level, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                         size_t lds_bytes, hipStream_t st) {
#define WF_LAUNCH(TX, LM)                                                                                              \
    do {                                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_wavefront_kernel<TX, LM>),                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                         \
        hipLaunchKernelGGL((rt_render_wavefront_kernel<TX, LM>), grid, block, lds_bytes, st, sd, fp);                  \
    } while (0)
    if (tex_level == 0) { if (lds_mode == 2) WF_LAUNCH(0, 2); else if (lds_mode == 1) WF_LAUNCH(0, 1); else WF_LAUNCH(0, 0); }
    else { if (lds_mode == 2) WF_LAUNCH(1, 2); else if (lds_mode == 1) WF_LAUNCH(1, 1); else WF_LAUNCH(1, 0); }
#undef WF_LAUNCH
}

// Picks the specialisation.  tex_level: 0 = every material colour is inline,
// 1 = solid + checker textures, 2 = noise / image textures too (those pull in
// Perlin noise and the sphere-uv transcendentals, which cost ~100 VGPRs, so
// scenes without them -- the headline random scene -- get leaner code).
void rt_launch_render(int kernel, int lds_mode, bool spheres_only, int tex_level, bool need_uv, const rt_scene_dev& sd,
                      const rt_frame_params& fp, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream) {
    if (spheres_only) {
        if (tex_level == 0) launch_lds<true, 0, false>(kernel, lds_mode, sd, fp, grid, block, lds_bytes, stream);
        else if (tex_level == 1) launch_lds<true, 1, false>(kernel, lds_mode, sd, fp, grid, block, lds_bytes, stream);
        else launch_lds<true, 2, true>(kernel, lds_mode, sd, fp, grid, block, lds_bytes, stream);
    } else {
        if (tex_level <= 1 && !need_uv) launch_lds<false, 1, false>(kernel, lds_mode, sd, fp, grid, block, lds_bytes, stream);
        else launch_lds<false, 2, true>(kernel, lds_mode, sd, fp, grid, block, lds_bytes, stream);
    }
}
