// rt_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain C++ restatement of the reference path tracer's hot path
// (/root/reference/src/main.cu:37-133 render_init/render/color and every
// hittable/material/texture/camera routine they reach).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library, and only as the checker.  Nothing under accelerated-ray-tracer_amd/
// includes, links or calls it.
//
// PINNING.  The reference has no tests, golden vectors or fixtures
//  and cannot be built here (it needs nvcc and cuRAND's
// curand_kernel.h, neither of which is in this image; building it against
// stand-in headers is not allowed).  What the reference does hold is output:
// its README images (images/*.png) are lossless 8-bit copies of the PPM the
// CUDA binary printed for nine scene functions of the current source at their
// own nx, ny, ns and seeds.  This oracle is pinned against those
// (tests/test_reference_images.py, tests/golden/reference_image_pins.npz;
// full frames in profiles/r01_reference_image_match.txt): it reproduces the
// pinned rows of quads, checker, earth, perlin, simple_light, Cornell and the
// headline random scene (the last three at 10000 spp) pixel for pixel, and the
// two scenes whose rays all pass a constant_medium to Monte-Carlo noise only:
// the constant_medium path (constant_medium.cuh:36-76) is PARITY UNPINNED
// against the reference -- no output the reference holds can pin it
// (profiles/r02_medium_log_ulp_experiment.txt).
// That pins the XORWOW stream (restated here from the published cuRAND
// algorithm), per-pixel seeding, draw order, scene construction, BVH rules,
// hit routines, materials, textures, camera, accumulation, gamma and the
// contraction rules below against the CUDA binary itself.  No fp32 output of
// the reference exists, so nothing is claimed below the 8-bit level.  Further
// outside pins: the counters SURVEY.md section 8 recorded from the reference's
// own code -- tests/golden/survey_pins.json, tests/test_oracle_pins.py.
//
// Floating-point contract of this oracle = the reference's real build (nvcc
// defaults: IEEE div/sqrt, -fmad=true), every rule checked against the images:
//   * IEEE-754 binary32 for + - * / sqrt.  Built with -ffp-contract=off; the
//     FMA contractions nvcc performs are written out with fmaf(): a product
//     whose only use is an add/sub is fused with it; of two products under one
//     add/sub the first is fused (dot = fma(z,z', fma(x,x', y*y'))); -a*b + c*d
//     is c*d - a*b; refract's last line fuses its second product.
//   * The camera basis, dist_to_focus and the lower_left_corner terms built
//     only from them are compile-time constants in the reference's scene
//     kernels: folded one operation at a time, no contraction (*_folded).
//   * Where the reference leaves argument evaluation order unspecified
//     (material.cuh:15, camera.cuh:11-12, main.cu:188,193) draws are taken
//     left to right.
//   * Transcendentals (powf, logf, __sinf, acos, atan2) are evaluated as the
//     correctly rounded binary32 value (double libm, then one rounding):
//     CUDA's device versions are only specified to within a few ulp of that.
//     Host-side scene construction (tanf, sinf, cosf) uses libm float
//     functions.
//
// Structure is deliberately unlike the product: a pointer-linked object tree
// walked recursively, the way the reference does it, so that the product's
// flattened/threaded traversal is checked against an independent form.

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- vec3
// vec3.cuh:8-158
struct V3 { float x, y, z; };

inline V3 v3(float a, float b, float c) { V3 r; r.x = a; r.y = b; r.z = c; return r; }
inline V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }          // vec3.cuh:57
inline V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }          // vec3.cuh:62
inline V3 vmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }          // vec3.cuh:67
inline V3 vscale(float t, V3 v) { return v3(t * v.x, t * v.y, t * v.z); }           // vec3.cuh:77,87
inline V3 vdivs(V3 v, float t) { return v3(v.x / t, v.y / t, v.z / t); }            // vec3.cuh:82
inline V3 vneg(V3 v) { return v3(-v.x, -v.y, -v.z); }
// contracted forms (see "Floating-point contract" in the header): m0 + m1 + m2 = fma(m2, fma(m0, m1))
inline float vdot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.x, b.x, a.y * b.y)); } // vec3.cuh:92
inline V3 vcross(V3 a, V3 b) {                                                       // vec3.cuh:97
    return v3(fmaf(a.y, b.z, -(a.z * b.y)), -fmaf(a.x, b.z, -(a.z * b.x)), fmaf(a.x, b.y, -(a.y * b.x)));
}
inline float vlen(V3 v) { return sqrtf(fmaf(v.z, v.z, fmaf(v.x, v.x, v.y * v.y))); } // vec3.cuh:32
inline float vsqlen(V3 v) { return fmaf(v.z, v.z, fmaf(v.x, v.x, v.y * v.y)); }      // vec3.cuh:33
inline float vlen_folded(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }   // constant-folded (uncontracted) forms
inline V3 vcross_folded(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x); }
inline V3 vfma(float t, V3 v, V3 a) { return v3(fmaf(t, v.x, a.x), fmaf(t, v.y, a.y), fmaf(t, v.z, a.z)); }      // a + t*v
inline V3 vfnma(float t, V3 v, V3 a) { return v3(fmaf(-t, v.x, a.x), fmaf(-t, v.y, a.y), fmaf(-t, v.z, a.z)); }  // a - t*v
inline V3 vunit(V3 v) { return vdivs(v, vlen(v)); }                                  // vec3.cuh:155
// operator/=(float): reciprocal formed in double, rounded once (vec3.cuh:145-153)
inline V3 vdiveq(V3 v, float t) { float k = (float)(1.0 / (double)t); return v3(v.x * k, v.y * k, v.z * k); }
inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// correctly rounded binary32 transcendentals (see header)
inline float cr_powf(float x, float y) { return (float)pow((double)x, (double)y); }
#ifdef ORC_DIAG
// Diagnostic build only (make librt_oracle_diag.so; tools/medium_ulp_experiment.py loads that library explicitly, no test
// and no bench ever does): ORC_LOG_ULP=N moves the result by one ulp for the inputs whose bit pattern is a multiple of N --
// a stand-in for a logf that is not correctly rounded in a fraction of its inputs.  The checker proper (librt_oracle.so)
// does not contain this code, so no environment variable can change it.
static const int g_log_ulp_every = [] { const char* e = getenv("ORC_LOG_ULP"); return e ? atoi(e) : 0; }();
inline float cr_logf(float x) {
    const float v = (float)log((double)x);
    if (g_log_ulp_every > 0) { uint32_t b; memcpy(&b, &x, 4); if (b % (uint32_t)g_log_ulp_every == 0u) return nextafterf(v, INFINITY); }
    return v;
}
#else
inline float cr_logf(float x) { return (float)log((double)x); }
#endif
inline float cr_sinf(float x) { return (float)sin((double)x); }
inline float cr_acosf(float x) { return (float)acos((double)x); }
inline float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }

const float PI_F = 3.141592654f;  // CUDART_PI_F

// ---------------------------------------------------------------- ray
// ray.cuh:5-21.  Time is a double; point_at narrows t to float before t*B.
struct Ray { V3 o, d; double tm; };
inline V3 ray_at(const Ray& r, double t) { float tf = (float)t; return vfma(tf, r.d, r.o); }   // A + t*B, one FMA per component

// ---------------------------------------------------------------- XORWOW
// cuRAND XORWOW, subsequence 0, offset 0 (call sites main.cu:92,104,
// constant_medium.cuh:74).  Third-party: CUDA Toolkit curand_kernel.h,
// not vendored by the reference and not pinned by any lockfile.
struct Rng { uint32_t v[5]; uint32_t d; };

inline void rng_seed(Rng& s, unsigned long long seed) {
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0;
    s.v[1] = 362436069u ^ t0;
    s.v[2] = 521288629u + t1;
    s.v[3] = 88675123u ^ t1;
    s.v[4] = 5783321u + t0;
}
inline uint32_t rng_next(Rng& s) {
    uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}
// curand_uniform: (0,1]
inline float rng_uniform(Rng& s) {
    uint32_t x = rng_next(s);
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// ---------------------------------------------------------------- aabb
// aabb.cuh:8-79
struct Box { V3 lo, hi; };
inline Box box_from(V3 a, V3 b) {                                                    // aabb.cuh:17-21
    Box r;
    r.lo = v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z));
    r.hi = v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z));
    return r;
}
inline Box box_empty() { Box r; r.lo = v3(FLT_MAX, FLT_MAX, FLT_MAX); r.hi = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX); return r; }
inline Box box_pad(const Box& b, float delta) { V3 d = v3(delta, delta, delta); return box_from(vsub(b.lo, d), vadd(b.hi, d)); }  // aabb.cuh:27
inline Box box_union(const Box& a, const Box& b) {                                   // aabb.cuh:34-43
    V3 small = v3(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z));
    V3 big = v3(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z));
    return box_from(small, big);
}
inline Box box_shift(const Box& b, V3 off) { return box_from(vadd(b.lo, off), vadd(b.hi, off)); }  // aabb.cuh:76

struct Counters {
    unsigned long long rays = 0, box_tests = 0, sphere_tests = 0, quad_tests = 0,
                       medium_calls = 0, box6_calls = 0, inst_calls = 0, samples = 0;
};
thread_local Counters* g_cnt = nullptr;
bool g_node_stats = false;   // diagnostics only: count passing box tests per bvh node (orc_node_passes)

// slab test, aabb.cuh:45-61 (three IEEE divides, ternary min/max, <= reject)
inline bool box_hit(const Box& b, const Ray& r, float tmin, float tmax) {
    if (g_cnt) g_cnt->box_tests++;
    for (int a = 0; a < 3; ++a) {
        float invD = 1.0f / axis_of(r.d, a);
        float t0 = (axis_of(b.lo, a) - axis_of(r.o, a)) * invD;
        float t1 = (axis_of(b.hi, a) - axis_of(r.o, a)) * invD;
        if (invD < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    return true;
}

// ---------------------------------------------------------------- perlin
// perlin.cuh:6-83 (hash-based gradient noise, no permutation tables)
inline uint32_t wanghash(uint32_t x) {
    x = (x ^ 61u) ^ (x >> 16); x *= 9u; x = x ^ (x >> 4); x *= 0x27d4eb2du; x = x ^ (x >> 15); return x;
}
inline uint32_t mix3(int x, int y, int z) {
    return (uint32_t)x * 73856093u ^ (uint32_t)y * 19349663u ^ (uint32_t)z * 83492791u;
}
inline float u2m11(uint32_t h) { return fmaf((float)((h >> 8) & 0x00FFFFFFu), (1.0f / 8388607.5f), -1.0f); }
inline V3 perlin_grad(int xi, int yi, int zi) {
    uint32_t h = wanghash(mix3(xi, yi, zi));
    float a = u2m11(h);
    float b = u2m11(wanghash(h));
    float c = u2m11(wanghash(h ^ 0x9e3779b9u));
    return vunit(v3(a, b, c));
}
inline float perlin_smooth(float t) { return t * t * (3.0f - 2.0f * t); }
inline float perlin_noise(V3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    V3 c[2][2][2];
    for (int di = 0; di < 2; ++di) for (int dj = 0; dj < 2; ++dj) for (int dk = 0; dk < 2; ++dk)
        c[di][dj][dk] = perlin_grad(i + di, j + dj, k + dk);
    float uu = perlin_smooth(u), vv = perlin_smooth(v), ww = perlin_smooth(w);
    float accum = 0.0f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int cc = 0; cc < 2; ++cc) {
        V3 weight = v3(u - (float)a, v - (float)b, w - (float)cc);
        float s = (a ? uu : (1.0f - uu)) * (b ? vv : (1.0f - vv)) * (cc ? ww : (1.0f - ww));
        accum = fmaf(s, vdot(c[a][b][cc], weight), accum);
    }
    return accum;
}
inline float perlin_turb(V3 p, int depth) {                                          // perlin.cuh:72-82
    float accum = 0.0f; V3 temp = p; float weight = 1.0f;
    for (int i = 0; i < depth; ++i) {
        accum = fmaf(weight, perlin_noise(temp), accum);
        weight *= 0.5f;
        temp = v3(temp.x * 2.0f, temp.y * 2.0f, temp.z * 2.0f);
    }
    return fabsf(accum);
}

// ---------------------------------------------------------------- textures
// texture.cuh:7-76
enum TexKind { TEX_SOLID, TEX_CHECKER, TEX_IMAGE, TEX_NOISE, TEX_NOODLE, TEX_FELT, TEX_UVOFF };
struct Tex {
    TexKind kind = TEX_SOLID;
    V3 color = {0, 0, 0};
    float inv_scale = 1.f; const Tex* even = nullptr; const Tex* odd = nullptr;   // checker
    const unsigned char* img = nullptr; int w = 0, h = 0;                           // image (bpp 3)
    float scale = 1.f;                                                              // noise
    float k = 3.f, A = 3.f, f = 0.6f; int octaves = 3; V3 dir = {0, 0, 1}, cN = {0, 0, 0}, cG = {0, 0, 0};   // noodle
    float m_scale = 16.f, m_amt = 0.08f, f_scale = 4.f, f_amt = 0.03f;             // felt (base colour in `color`)
    const Tex* base = nullptr; float du = 0.f, dv = 0.f;                            // uv_offset
};
inline float clamp01(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
V3 tex_value(const Tex* t, float u, float v, V3 p) {
    switch (t->kind) {
    case TEX_SOLID: return t->color;
    case TEX_CHECKER: {                                                             // texture.cuh:35-42
        int xi = (int)floorf(t->inv_scale * p.x);
        int yi = (int)floorf(t->inv_scale * p.y);
        int zi = (int)floorf(t->inv_scale * p.z);
        bool is_even = ((xi + yi + zi) & 1) == 0;
        return is_even ? tex_value(t->even, u, v, p) : tex_value(t->odd, u, v, p);
    }
    case TEX_IMAGE: {                                                               // texture.cuh:51-59
        if (!(t->img && t->w > 0 && t->h > 0)) return v3(0, 1, 1);
        u = clamp01(u); v = clamp01(v);
        int i = (int)(u * (float)t->w); if (i > t->w - 1) i = t->w - 1;
        int j = (int)((1.f - v) * (float)t->h); if (j > t->h - 1) j = t->h - 1;
        int idx = (j * t->w + i) * 3;
        const float inv255 = 1.f / 255.f;
        return v3(inv255 * (float)t->img[idx + 0], inv255 * (float)t->img[idx + 1], inv255 * (float)t->img[idx + 2]);
    }
    case TEX_NOODLE: {                                                              // texture.cuh:94-100
        float uu = vdot(p, t->dir);
        float wig = perlin_turb(vscale(t->f, p), t->octaves);
        float stripes = fabsf(cr_sinf(fmaf(t->k, uu, t->A * wig)));
        float q = clamp01((stripes - 0.75f) / (0.98f - 0.75f));                    // smoothstep, texture.cuh:78-82
        float w = q * q * (3.0f - 2.0f * q);
        return vfma(1.f - w, t->cG, vscale(w, t->cN));
    }
    case TEX_FELT: {                                                                // texture.cuh:124-147
        float m = perlin_noise(vscale(t->m_scale, p));
        float phase = fmaf(p.x, t->f_scale, 2.0f * perlin_turb(vscale(0.5f, p), 2));
        float fibers = 0.5f * (1.0f + cr_sinf(phase));
        float gain = fmaf(t->f_amt, fibers - 0.5f, fmaf(t->m_amt, m - 0.5f, 1.0f));
        gain = fminf(fmaxf(gain, 0.7f), 1.2f);
        return vscale(gain, t->color);
    }
    case TEX_UVOFF: {                                                               // texture.cuh:156-160
        float uu = u + t->du; uu -= floorf(uu);
        float vv = v + t->dv; vv = fminf(fmaxf(vv, 0.f), 1.f);
        return tex_value(t->base, uu, vv, p);
    }
    case TEX_NOISE: {                                                               // texture.cuh:67-72
        float s = cr_sinf(fmaf(t->scale, p.z, 10.0f * perlin_turb(p, 7)));
        float tt = 0.5f * (1.0f + s);
        return v3(tt, tt, tt);
    }
    }
    return v3(0, 0, 0);
}

// ---------------------------------------------------------------- materials
// material.cuh:10-201
enum MatKind { MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_LIGHT, MAT_ISOTROPIC };
struct Mat {
    MatKind kind = MAT_LAMBERTIAN;
    const Tex* tex = nullptr;   // lambertian / isotropic / light (optional)
    V3 albedo = {0, 0, 0};      // metal albedo, or light's solid colour
    float fuzz = 0.f;           // metal (clamped to <=1, material.cuh:97)
    float ior = 1.f;            // dielectric
};

struct Hit { float t; V3 p; V3 n; const Mat* mat; double u, v; };  // hittable.cuh:13-21

inline V3 random_in_unit_sphere(Rng& g) {                                           // material.cuh:12-18
    for (;;) {
        float a = 2.0f * rng_uniform(g) - 1.0f;
        float b = 2.0f * rng_uniform(g) - 1.0f;
        float c = 2.0f * rng_uniform(g) - 1.0f;
        V3 p = v3(a, b, c);
        if (vsqlen(p) < 1.0f) return p;
    }
}
inline V3 reflect(V3 v, V3 n) { return vfnma(2.0f * vdot(v, n), n, v); }      // material.cuh:20-23
inline bool refract(V3 v, V3 n, float ni_over_nt, V3& out) {                        // material.cuh:26-36
    V3 uv = vunit(v);
    float dt = vdot(uv, n);
    float disc = fmaf(-(ni_over_nt * ni_over_nt), fmaf(-dt, dt, 1.0f), 1.0f);
    if (disc > 0.0f) {
        const V3 a = vfnma(dt, n, uv); const float sq = sqrtf(disc);                 // ni*(uv - n*dt) - n*sqrt(disc)
        out = v3(fmaf(-sq, n.x, ni_over_nt * a.x), fmaf(-sq, n.y, ni_over_nt * a.y), fmaf(-sq, n.z, ni_over_nt * a.z));
        return true;
    }
    return false;
}
inline float schlick(float cosine, float ref_idx) {                                 // material.cuh:38-43
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return fmaf(1.0f - r0, cr_powf(1.0f - cosine, 5.0f), r0);
}
V3 mat_emitted(const Mat* m, float u, float v, V3 p) {                              // material.cuh:49-52,169-172
    if (m->kind != MAT_LIGHT) return v3(0.f, 0.f, 0.f);
    return m->tex ? tex_value(m->tex, u, v, p) : m->albedo;
}
bool mat_scatter(const Mat* m, const Ray& in, const Hit& rec, V3& atten, Ray& out, Rng& g) {
    switch (m->kind) {
    case MAT_LAMBERTIAN: {                                                          // material.cuh:75-86
        V3 target = vadd(vadd(rec.p, rec.n), random_in_unit_sphere(g));
        out.o = rec.p; out.d = vsub(target, rec.p); out.tm = in.tm;
        atten = m->tex ? tex_value(m->tex, (float)rec.u, (float)rec.v, rec.p) : v3(1, 1, 1);
        return true;
    }
    case MAT_METAL: {                                                               // material.cuh:99-109
        V3 refl = reflect(vunit(in.d), rec.n);
        V3 rs = random_in_unit_sphere(g);
        out.o = rec.p; out.d = vfma(m->fuzz, rs, refl); out.tm = in.tm;
        atten = m->albedo;
        return vdot(out.d, rec.n) > 0.0f;
    }
    case MAT_DIELECTRIC: {                                                          // material.cuh:119-159
        V3 outward; V3 refl = reflect(in.d, rec.n);
        float ni_over_nt; atten = v3(1.0f, 1.0f, 1.0f);
        V3 refr = v3(0, 0, 0); float reflect_prob, cosine;
        if (vdot(in.d, rec.n) > 0.0f) {
            outward = vneg(rec.n);
            ni_over_nt = m->ior;
            cosine = vdot(in.d, rec.n) / vlen(in.d);
            cosine = sqrtf(fmaxf(0.0f, fmaf(-(m->ior * m->ior), fmaf(-cosine, cosine, 1.0f), 1.0f)));
        } else {
            outward = rec.n;
            ni_over_nt = 1.0f / m->ior;
            cosine = -vdot(in.d, rec.n) / vlen(in.d);
        }
        if (refract(in.d, outward, ni_over_nt, refr)) reflect_prob = schlick(cosine, m->ior);
        else reflect_prob = 1.0f;
        out.o = rec.p; out.tm = in.tm;
        if (rng_uniform(g) < reflect_prob) out.d = refl; else out.d = refr;
        return true;
    }
    case MAT_LIGHT: return false;                                                   // material.cuh:175-178
    case MAT_ISOTROPIC: {                                                           // material.cuh:193-200
        out.o = rec.p; out.d = random_in_unit_sphere(g); out.tm = in.tm;
        atten = tex_value(m->tex, (float)rec.u, (float)rec.v, rec.p);
        return true;
    }
    }
    return false;
}

// ---------------------------------------------------------------- hittables
enum ObjKind { OBJ_SPHERE, OBJ_QUAD, OBJ_BOX6, OBJ_TRANSLATE, OBJ_ROTY, OBJ_MEDIUM, OBJ_BVH };
struct Obj {
    ObjKind kind;
    Box bbox;
    int list_index = -1;                   // creation order in the scene list (for dumps)
    // sphere (sphere.cuh:10-102): centre path c(t) = c0 + t*vel
    V3 c0, vel; float radius = 0;
    // quad (quad.cuh:11-91)
    V3 Q, eu, ev, w, normal; float D = 0;
    const Mat* mat = nullptr;
    // box6 faces, wrappers' child, medium boundary, bvh children
    const Obj* face[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const Obj* child = nullptr;
    V3 offset;                             // translate
    float sin_t = 0, cos_t = 1;            // rotate_y
    float neg_inv_density = 0;             // medium
    const Obj* left = nullptr; const Obj* right = nullptr;
    mutable unsigned long long stat_pass = 0;   // diagnostics (orc_node_passes): box tests of this bvh node that passed
};

bool obj_hit(const Obj* o, const Ray& r, float tmin, float tmax, Hit& rec);

inline void sphere_uv(V3 p, double& u, double& v) {                                 // sphere.cuh:42-49
    float theta = cr_acosf(-p.y);
    float phi = cr_atan2f(-p.z, p.x) + PI_F;
    u = (double)(phi / (2 * PI_F));
    v = (double)(theta / PI_F);
}

bool sphere_hit(const Obj* s, const Ray& r, float tmin, float tmax, Hit& rec) {      // sphere.cuh:51-89
    if (g_cnt) g_cnt->sphere_tests++;
    Ray cr; cr.o = s->c0; cr.d = s->vel; cr.tm = 0;
    V3 cc = ray_at(cr, r.tm);
    V3 oc = vsub(r.o, cc);
    float a = vdot(r.d, r.d);
    float b = vdot(oc, r.d);
    float c = fmaf(-s->radius, s->radius, vdot(oc, oc));
    float disc = fmaf(b, b, -(a * c));
    if (disc <= 0.0f) return false;
    float sq = sqrtf(disc);
    float t = (-b - sq) / a;
    if (!(t > tmin && t < tmax)) {
        t = (-b + sq) / a;
        if (!(t > tmin && t < tmax)) return false;
    }
    rec.t = t;
    rec.p = ray_at(r, (double)t);
    rec.n = vdivs(vsub(rec.p, cc), s->radius);
    sphere_uv(rec.n, rec.u, rec.v);
    rec.mat = s->mat;
    return true;
}

bool quad_hit(const Obj* q, const Ray& r, float tmin, float tmax, Hit& rec) {        // quad.cuh:60-90
    if (g_cnt) g_cnt->quad_tests++;
    const float denom = vdot(q->normal, r.d);
    if (fabsf(denom) < 1e-8f) return false;
    const float t = (q->D - vdot(q->normal, r.o)) / denom;
    if (t < tmin || t > tmax) return false;
    const V3 P = ray_at(r, (double)t);
    const V3 pl = vsub(P, q->Q);
    const float alpha = vdot(q->w, vcross(pl, q->ev));
    const float beta = vdot(q->w, vcross(q->eu, pl));
    if (alpha < 0.f || alpha > 1.f || beta < 0.f || beta > 1.f) return false;
    rec.t = t; rec.p = P; rec.u = (double)alpha; rec.v = (double)beta;
    V3 n = q->normal;
    if (vdot(n, r.d) > 0.f) n = vneg(n);
    rec.n = n; rec.mat = q->mat;
    return true;
}

bool box6_hit(const Obj* b, const Ray& r, float tmin, float tmax, Hit& rec) {        // quad.cuh:124-139
    if (g_cnt) g_cnt->box6_calls++;
    bool any = false; float closest = tmax;
    for (int i = 0; i < 6; ++i) {
        Hit tmp;
        if (obj_hit(b->face[i], r, tmin, closest, tmp)) { any = true; closest = tmp.t; rec = tmp; }
    }
    return any;
}

bool translate_hit(const Obj* t, const Ray& r, float tmin, float tmax, Hit& rec) {   // hittable.cuh:56-65
    if (g_cnt) g_cnt->inst_calls++;
    Ray moved; moved.o = vsub(r.o, t->offset); moved.d = r.d; moved.tm = r.tm;
    if (!obj_hit(t->child, moved, tmin, tmax, rec)) return false;
    rec.p = vadd(rec.p, t->offset);
    return true;
}

bool roty_hit(const Obj* ro, const Ray& r, float tmin, float tmax, Hit& rec) {       // hittable.cuh:118-145
    if (g_cnt) g_cnt->inst_calls++;
    const float c = ro->cos_t, s = ro->sin_t;
    const float ox = fmaf(c, r.o.x, -(s * r.o.z));
    const float oz = fmaf(s, r.o.x, c * r.o.z);
    const float dx = fmaf(c, r.d.x, -(s * r.d.z));
    const float dz = fmaf(s, r.d.x, c * r.d.z);
    Ray rr; rr.o = v3(ox, r.o.y, oz); rr.d = v3(dx, r.d.y, dz); rr.tm = r.tm;
    if (!obj_hit(ro->child, rr, tmin, tmax, rec)) return false;
    const float px = fmaf(c, rec.p.x, s * rec.p.z);
    const float pz = fmaf(c, rec.p.z, -(s * rec.p.x));   // -s*x + c*z is c*z - s*x to the compiler: the c*z product is the fused one
    const float nx = fmaf(c, rec.n.x, s * rec.n.z);
    const float nz = fmaf(c, rec.n.z, -(s * rec.n.x));
    rec.p = v3(px, rec.p.y, pz);
    rec.n = vunit(v3(nx, rec.n.y, nz));
    if (vdot(rec.n, r.d) > 0.f) rec.n = vneg(rec.n);
    return true;
}

inline uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// constant_medium.cuh:36-64 with the private generator of :67-76.  Inside a
// BVH the 5-argument hit is never reached (bvh.cuh:109-112 drops the rng), so
// the medium always seeds its own XORWOW from a hash of the ray.
bool medium_hit(const Obj* m, const Ray& r, float tmin, float tmax, Hit& rec) {
    if (g_cnt) g_cnt->medium_calls++;
    Rng fake;
    uint32_t seed = 1337u ^ float_bits(r.o.x) ^ float_bits(r.o.y * 3.1f) ^ float_bits(r.d.z * 5.7f);
    rng_seed(fake, (unsigned long long)seed);
    Hit rec1, rec2;
    if (!obj_hit(m->child, r, -FLT_MAX, FLT_MAX, rec1)) return false;
    if (!obj_hit(m->child, r, rec1.t + 1e-4f, FLT_MAX, rec2)) return false;
    if (rec1.t < tmin) rec1.t = tmin;
    if (rec2.t > tmax) rec2.t = tmax;
    if (rec1.t >= rec2.t) return false;
    if (rec1.t < 0) rec1.t = 0;
    const float ray_len = vlen(r.d);
    if (ray_len <= 0.0f || !std::isfinite(ray_len)) return false;
    const float distance_inside = (rec2.t - rec1.t) * ray_len;
    float U = fmaxf(1e-6f, rng_uniform(fake));
    const float hit_distance = m->neg_inv_density * cr_logf(U);
    if (hit_distance > distance_inside) return false;
    rec.t = rec1.t + hit_distance / ray_len;
    rec.p = ray_at(r, (double)rec.t);
    rec.n = v3(1, 0, 0);
    rec.u = rec.v = 0.0;
    rec.mat = m->mat;
    return true;
}

bool bvh_hit(const Obj* n, const Ray& r, float tmin, float tmax, Hit& rec) {         // bvh.cuh:95-106
    if (!box_hit(n->bbox, r, tmin, tmax)) return false;
    if (g_node_stats) __atomic_fetch_add(&n->stat_pass, 1ull, __ATOMIC_RELAXED);
    Hit lrec, rrec;
    const bool hl = n->left ? obj_hit(n->left, r, tmin, tmax, lrec) : false;
    const bool hr = n->right ? obj_hit(n->right, r, tmin, hl ? lrec.t : tmax, rrec) : false;
    if (hr) rec = rrec;
    if (hl && (!hr || lrec.t < rrec.t)) rec = lrec;
    return hl || hr;
}

bool obj_hit(const Obj* o, const Ray& r, float tmin, float tmax, Hit& rec) {
    switch (o->kind) {
    case OBJ_SPHERE: return sphere_hit(o, r, tmin, tmax, rec);
    case OBJ_QUAD: return quad_hit(o, r, tmin, tmax, rec);
    case OBJ_BOX6: return box6_hit(o, r, tmin, tmax, rec);
    case OBJ_TRANSLATE: return translate_hit(o, r, tmin, tmax, rec);
    case OBJ_ROTY: return roty_hit(o, r, tmin, tmax, rec);
    case OBJ_MEDIUM: return medium_hit(o, r, tmin, tmax, rec);
    case OBJ_BVH: return bvh_hit(o, r, tmin, tmax, rec);
    }
    return false;
}

// ---------------------------------------------------------------- camera
// camera.cuh:8-79
struct Camera {
    V3 origin, llc, horizontal, vertical, u, v, w;
    float lens_radius; double time0, time1;
};
Camera make_camera(V3 lookfrom, V3 lookat, V3 vup, float vfov, float aspect, float aperture,
                   float focus_dist, double t0, double t1) {                        // camera.cuh:59-78
    Camera c; c.time0 = t0; c.time1 = t1;
    c.lens_radius = aperture * 0.5f;
    float theta = vfov * PI_F / 180.0f;
    float half_height = tanf(theta * 0.5f);
    float half_width = aspect * half_height;
    c.origin = lookfrom;
    // lookfrom, lookat, vup, vfov and focus_dist are compile-time constants in every reference scene kernel
    // (main.cu:233-241 and the like), so the basis and the terms built only from them are what a constant folder
    // produces: one rounding per written operation, no contraction.  aspect = nx/ny is a kernel argument, so the
    // half_width term below is a run-time product and is contracted into the subtraction.
    { V3 d = vsub(lookfrom, lookat); c.w = vdivs(d, vlen_folded(d)); }
    { V3 d = vcross_folded(vup, c.w); c.u = vdivs(d, vlen_folded(d)); }
    c.v = vcross_folded(c.w, c.u);
    c.llc = vsub(vsub(vfnma(half_width * focus_dist, c.u, c.origin), vscale(half_height * focus_dist, c.v)), vscale(focus_dist, c.w));
    c.horizontal = vscale(2.0f * half_width * focus_dist, c.u);
    c.vertical = vscale(2.0f * half_height * focus_dist, c.v);
    return c;
}
inline V3 random_in_unit_disk(Rng& g) {                                              // camera.cuh:8-16
    V3 p;
    do {
        float a = rng_uniform(g);
        float b = rng_uniform(g);
        p = vsub(vscale(2.0f, v3(a, b, 0.0f)), v3(1.0f, 1.0f, 0.0f));
    } while (vdot(p, p) >= 1.0f);
    return p;
}
Ray camera_get_ray(const Camera& c, float s, float t, Rng& g) {                      // camera.cuh:35-47
    V3 rd = vscale(c.lens_radius, random_in_unit_disk(g));
    V3 offset = vfma(rd.x, c.u, vscale(rd.y, c.v));
    double tm = fma((double)rng_uniform(g), c.time1 - c.time0, c.time0);
    Ray r;
    r.o = vadd(c.origin, offset);
    r.d = vsub(vsub(vfma(t, c.vertical, vfma(s, c.horizontal, c.llc)), c.origin), offset);
    r.tm = tm;
    return r;
}

// ---------------------------------------------------------------- scene store
struct Scene {
    std::vector<std::unique_ptr<Obj>> objs;
    std::vector<std::unique_ptr<Mat>> mats;
    std::vector<std::unique_ptr<Tex>> texs;
    std::vector<unsigned char> image;
    std::vector<Obj*> list;      // the reference's d_list
    const Obj* world = nullptr;
    Camera cam;
    // what the reference's host function passes to render<<<>>>
    V3 background = {0, 0, 0}; int gradient = 0;
    int def_nx = 0, def_ny = 0, def_ns = 0; float gamma = 2.2f;

    Tex* tex_solid(V3 c) { auto t = new Tex; t->kind = TEX_SOLID; t->color = c; texs.emplace_back(t); return t; }
    Tex* tex_checker(float scale, Tex* e, Tex* o) {
        auto t = new Tex; t->kind = TEX_CHECKER; t->inv_scale = 1.f / scale; t->even = e; t->odd = o; texs.emplace_back(t); return t;
    }
    Tex* tex_noise(float scale) { auto t = new Tex; t->kind = TEX_NOISE; t->scale = scale; texs.emplace_back(t); return t; }
    Tex* tex_noodle(float k) { auto t = new Tex; t->kind = TEX_NOODLE; t->k = k; t->dir = vunit(v3(0, 0, 1));
        t->cN = v3(0.92f, 0.85f, 0.65f); t->cG = v3(0.35f, 0.20f, 0.10f); texs.emplace_back(t); return t; }
    Tex* tex_felt(V3 base, float ms, float ma, float fs, float fa) { auto t = new Tex; t->kind = TEX_FELT; t->color = base;
        t->m_scale = ms; t->m_amt = ma; t->f_scale = fs; t->f_amt = fa; texs.emplace_back(t); return t; }
    Tex* tex_uv_offset(Tex* base, float du, float dv) { auto t = new Tex; t->kind = TEX_UVOFF; t->base = base; t->du = du; t->dv = dv;
        texs.emplace_back(t); return t; }
    Tex* tex_image() {
        auto t = new Tex; t->kind = TEX_IMAGE; texs.emplace_back(t); return t;   // bound to this->image later
    }
    Mat* lambertian(V3 a) { auto m = new Mat; m->kind = MAT_LAMBERTIAN; m->tex = tex_solid(a); mats.emplace_back(m); return m; }
    Mat* lambertian(Tex* t) { auto m = new Mat; m->kind = MAT_LAMBERTIAN; m->tex = t; mats.emplace_back(m); return m; }
    Mat* metal(V3 a, float f) { auto m = new Mat; m->kind = MAT_METAL; m->albedo = a; m->fuzz = f < 1.0f ? f : 1.0f; mats.emplace_back(m); return m; }
    Mat* dielectric(float ri) { auto m = new Mat; m->kind = MAT_DIELECTRIC; m->ior = ri; mats.emplace_back(m); return m; }
    Mat* light(V3 c) { auto m = new Mat; m->kind = MAT_LIGHT; m->albedo = c; mats.emplace_back(m); return m; }
    Mat* isotropic(V3 c) { auto m = new Mat; m->kind = MAT_ISOTROPIC; m->tex = tex_solid(c); mats.emplace_back(m); return m; }

    Obj* add(Obj* o) { objs.emplace_back(o); return o; }
    Obj* sphere(V3 cen, float r, const Mat* m) {                                     // sphere.cuh:21-26
        auto o = new Obj; o->kind = OBJ_SPHERE; o->c0 = cen; o->vel = v3(0, 0, 0); o->radius = r; o->mat = m;
        V3 rv = v3(r, r, r); o->bbox = box_from(vsub(cen, rv), vadd(cen, rv));
        return add(o);
    }
    Obj* moving_sphere(V3 c1, V3 c2, float r, const Mat* m) {                        // sphere.cuh:29-38
        auto o = new Obj; o->kind = OBJ_SPHERE; o->c0 = c1; o->vel = vsub(c2, c1); o->radius = r; o->mat = m;
        V3 rv = v3(r, r, r);
        Ray cr; cr.o = o->c0; cr.d = o->vel; cr.tm = 0;
        V3 a = ray_at(cr, 0.0), b = ray_at(cr, 1.0);
        o->bbox = box_union(box_from(vsub(a, rv), vadd(a, rv)), box_from(vsub(b, rv), vadd(b, rv)));
        return add(o);
    }
    Obj* quad(V3 Q, V3 u, V3 v, const Mat* m, bool inward = false) {                 // quad.cuh:29-54
        auto o = new Obj; o->kind = OBJ_QUAD; o->Q = Q; o->eu = u; o->ev = v; o->mat = m;
        V3 n = vcross(u, v);
        o->normal = vunit(n);
        if (inward) o->normal = vneg(o->normal);
        o->D = vdot(o->normal, Q);
        o->w = vdivs(n, vdot(n, n));
        Box d1 = box_from(Q, vadd(vadd(Q, u), v));
        Box d2 = box_from(vadd(Q, u), vadd(Q, v));
        o->bbox = box_pad(box_union(d1, d2), 1e-3f);
        return add(o);
    }
    Obj* box(V3 a, V3 b, const Mat* m) {                                             // quad.cuh:145-162, 108-122
        V3 mn = v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z));
        V3 mx = v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z));
        V3 dx = v3(mx.x - mn.x, 0.f, 0.f), dy = v3(0.f, mx.y - mn.y, 0.f), dz = v3(0.f, 0.f, mx.z - mn.z);
        auto o = new Obj; o->kind = OBJ_BOX6;
        o->face[0] = quad(v3(mn.x, mn.y, mx.z), dx, dy, m);
        o->face[1] = quad(v3(mx.x, mn.y, mx.z), vneg(dz), dy, m);
        o->face[2] = quad(v3(mx.x, mn.y, mn.z), vneg(dx), dy, m);
        o->face[3] = quad(v3(mn.x, mn.y, mn.z), dz, dy, m);
        o->face[4] = quad(v3(mn.x, mx.y, mx.z), dx, vneg(dz), m);
        o->face[5] = quad(v3(mn.x, mn.y, mn.z), dx, dz, m);
        Box bb = o->face[0]->bbox;
        for (int i = 1; i < 6; ++i) {
            const Box& f = o->face[i]->bbox;
            V3 lo = v3(fminf(bb.lo.x, f.lo.x), fminf(bb.lo.y, f.lo.y), fminf(bb.lo.z, f.lo.z));
            V3 hi = v3(fmaxf(bb.hi.x, f.hi.x), fmaxf(bb.hi.y, f.hi.y), fmaxf(bb.hi.z, f.hi.z));
            bb = box_from(lo, hi);
        }
        o->bbox = bb;
        return add(o);
    }
    Obj* translate(const Obj* c, V3 d) {                                             // hittable.cuh:52-54
        auto o = new Obj; o->kind = OBJ_TRANSLATE; o->child = c; o->offset = d; o->bbox = box_shift(c->bbox, d);
        return add(o);
    }
    Obj* rotate_y(const Obj* c, float deg) {                                         // hittable.cuh:89-116
        auto o = new Obj; o->kind = OBJ_ROTY; o->child = c;
        const float rad = deg * 0.017453292519943295769f;
        o->sin_t = sinf(rad); o->cos_t = cosf(rad);
        const Box& b = c->bbox;
        V3 lo = v3(FLT_MAX, FLT_MAX, FLT_MAX), hi = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
            float x = i ? b.hi.x : b.lo.x, y = j ? b.hi.y : b.lo.y, z = k ? b.hi.z : b.lo.z;
            float nx = fmaf(o->cos_t, x, o->sin_t * z);
            float nz = fmaf(o->cos_t, z, -(o->sin_t * x));
            lo = v3(fminf(lo.x, nx), fminf(lo.y, y), fminf(lo.z, nz));
            hi = v3(fmaxf(hi.x, nx), fmaxf(hi.y, y), fmaxf(hi.z, nz));
        }
        o->bbox = box_from(lo, hi);
        return add(o);
    }
    Obj* medium(const Obj* boundary, float density, V3 albedo) {                     // constant_medium.cuh:27-28
        auto o = new Obj; o->kind = OBJ_MEDIUM; o->child = boundary; o->neg_inv_density = -1.0f / density;
        o->mat = isotropic(albedo); o->bbox = boundary->bbox;
        return add(o);
    }
    void push(Obj* o) { o->list_index = (int)list.size(); list.push_back(o); }

    // bvh.cuh:29-84: spread of box minima picks the axis, in-place selection
    // sort (strict <), split at n>>1, single objects get their own node with
    // left == right.
    const Obj* build_bvh(int start, int end) {
        auto node = new Obj; node->kind = OBJ_BVH; add(node);
        const int n = end - start;
        if (n <= 0) { node->bbox = box_empty(); return node; }
        if (n == 1) { node->left = node->right = list[start]; node->bbox = list[start]->bbox; return node; }
        float minx = 1e30f, maxx = -1e30f, miny = 1e30f, maxy = -1e30f, minz = 1e30f, maxz = -1e30f;
        for (int i = start; i < end; ++i) {
            V3 mn = list[i]->bbox.lo;
            if (mn.x < minx) minx = mn.x; if (mn.x > maxx) maxx = mn.x;
            if (mn.y < miny) miny = mn.y; if (mn.y > maxy) maxy = mn.y;
            if (mn.z < minz) minz = mn.z; if (mn.z > maxz) maxz = mn.z;
        }
        const float sx = maxx - minx, sy = maxy - miny, sz = maxz - minz;
        int axis = 0;
        if (sy > sx && sy >= sz) axis = 1;
        else if (sz > sx && sz >= sy) axis = 2;
        for (int i = start; i < end - 1; ++i) {
            int best = i;
            for (int j = i + 1; j < end; ++j)
                if (axis_of(list[j]->bbox.lo, axis) < axis_of(list[best]->bbox.lo, axis)) best = j;
            if (best != i) { Obj* t = list[i]; list[i] = list[best]; list[best] = t; }
        }
        const int mid = start + (n >> 1);
        node->left = build_bvh(start, mid);
        node->right = build_bvh(mid, end);
        node->bbox = box_union(node->left->bbox, node->right->bbox);
        return node;
    }
    void finish() { world = build_bvh(0, (int)list.size()); }
};

// ---------------------------------------------------------------- scenes
// "two_spheres": BASELINE config 1 as SURVEY.md 8(d) defines it (not in the
// reference): book chapter-8 scene through the reference's classes.
void scene_two_spheres(Scene& S, int nx, int ny) {
    S.push(S.sphere(v3(0, 0, -1), 0.5f, S.lambertian(v3(0.5f, 0.5f, 0.5f))));
    S.push(S.sphere(v3(0, -100.5f, -1), 100.f, S.lambertian(v3(0.5f, 0.5f, 0.5f))));
    S.finish();
    (void)nx; (void)ny;
    S.cam = make_camera(v3(0, 0, 0), v3(0, 0, -1), v3(0, 1, 0), 90.f, 2.0f, 0.0f, 1.0f, 0.0, 0.0);
    S.gradient = 1; S.def_nx = 200; S.def_ny = 100; S.def_ns = 1;
}

// "degenerate": not a reference scene.  vfov = 0 collapses the viewport, so every primary ray is exactly
// (0,0,-f) from (0,0,5): 1/d is +-inf on two axes and sphere boxes are placed with faces through the ray origin's
// x/y, which makes aabb::hit (aabb.cuh:45-61) take its 0*inf = NaN comparisons.  Used to pin the NaN semantics.
void scene_degenerate(Scene& S, int nx, int ny) {
    S.push(S.sphere(v3(0.5f, 0.0f, 0.0f), 0.5f, S.lambertian(v3(0.8f, 0.3f, 0.3f))));
    S.push(S.sphere(v3(-0.5f, 0.5f, 1.0f), 0.5f, S.metal(v3(0.8f, 0.8f, 0.8f), 0.3f)));
    S.push(S.sphere(v3(0.0f, -0.5f, 2.0f), 0.5f, S.dielectric(1.5f)));
    S.push(S.sphere(v3(0.0f, 0.0f, -1.0f), 1.0f, S.lambertian(v3(0.3f, 0.8f, 0.3f))));
    S.push(S.sphere(v3(0.0f, -101.0f, 0.0f), 100.0f, S.lambertian(v3(0.5f, 0.5f, 0.5f))));
    S.finish();
    S.cam = make_camera(v3(0, 0, 5), v3(0, 0, 0), v3(0, 1, 0), 0.0f, (float)nx / (float)ny, 0.0f, 5.0f, 0.0, 0.0);
    S.gradient = 1; S.def_nx = 32; S.def_ny = 16; S.def_ns = 8;
}

inline V3 pick_ut_color(float r) {                                                   // main.cu:149-158
    if (r < 0.25f) return v3(1.0f, 1.0f, 1.0f);
    else if (r < 0.50f) return v3(1.0f, 0.51f, 0.0f);
    else if (r < 0.75f) return v3(0.60f, 0.60f, 0.60f);
    else return v3(0.0f, 0.0f, 0.0f);
}

// main.cu:160-244 create_world_bouncing; host side main.cu:654-744
void scene_bouncing(Scene& S, int nx, int ny) {
    Rng g; rng_seed(g, 1984ULL);                                                    // main.cu:92
    const V3 orange = v3(1.0f, 0.51f, 0.0f);
    Tex* checker = S.tex_checker(0.64f, S.tex_solid(v3(1.0f, 1.0f, 1.0f)), S.tex_solid(orange));
    S.push(S.sphere(v3(0.0f, -1000.0f, -1.0f), 1000.0f, S.lambertian(checker)));
    for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
        float choose = rng_uniform(g);
        float cx = fmaf(0.9f, rng_uniform(g), (float)a);
        float cz = fmaf(0.9f, rng_uniform(g), (float)b);
        V3 center = v3(cx, 0.2f, cz);
        if (choose < 0.8f) {
            float vy = 0.5f * rng_uniform(g);
            float vz = 0.25f * (rng_uniform(g) - 0.5f);
            V3 center2 = vadd(center, v3(0.0f, vy, vz));
            if (rng_uniform(g) < 0.10f) {
                S.push(S.moving_sphere(center, center2, 0.2f, S.light(vscale(4.0f, orange))));
            } else {
                V3 alb = pick_ut_color(rng_uniform(g));
                S.push(S.moving_sphere(center, center2, 0.2f, S.lambertian(alb)));
            }
        } else if (choose < 0.95f) {
            V3 alb = pick_ut_color(rng_uniform(g));
            if (alb.x + alb.y + alb.z < 1e-5f) alb = v3(0.15f, 0.15f, 0.15f);
            float fuzz = 0.5f * rng_uniform(g);
            S.push(S.sphere(center, 0.2f, S.metal(alb, fuzz)));
        } else {
            S.push(S.sphere(center, 0.2f, S.dielectric(1.5f)));
        }
    }
    S.push(S.sphere(v3(0.0f, 1.0f, 0.0f), 1.0f, S.dielectric(1.5f)));
    S.push(S.sphere(v3(-4.0f, 1.0f, 0.0f), 1.0f, S.lambertian(v3(0.4f, 0.2f, 0.1f))));
    S.push(S.sphere(v3(4.0f, 1.0f, 0.0f), 1.0f, S.metal(v3(0.7f, 0.6f, 0.5f), 0.0f)));
    S.finish();
    V3 from = v3(13.0f, 2.0f, 3.0f), at = v3(0, 0, 0);
    S.cam = make_camera(from, at, v3(0, 1, 0), 30.0f, (float)nx / (float)ny, 0.1f, vlen_folded(vsub(from, at)), 0.0, 1.0);
    S.gradient = 0; S.def_nx = 1200; S.def_ny = 600; S.def_ns = 10000;
}

// Book-1 "random_scene" with the book's material rules (SURVEY.md 8(d) config
// 2b; not present at the reference's HEAD): grey ground, static spheres,
// gradient sky.  Same loop shape and draw order as main.cu:185-222.
void scene_book1(Scene& S, int nx, int ny) {
    Rng g; rng_seed(g, 1984ULL);
    S.push(S.sphere(v3(0.0f, -1000.0f, -1.0f), 1000.0f, S.lambertian(v3(0.5f, 0.5f, 0.5f))));
    for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
        float choose = rng_uniform(g);
        float cx = (float)a + rng_uniform(g);
        float cz = (float)b + rng_uniform(g);
        V3 center = v3(cx, 0.2f, cz);
        if (choose < 0.8f) {
            float r1 = rng_uniform(g); float r2 = rng_uniform(g); float cr = r1 * r2;
            float g1 = rng_uniform(g); float g2 = rng_uniform(g); float cg = g1 * g2;
            float b1 = rng_uniform(g); float b2 = rng_uniform(g); float cb = b1 * b2;
            S.push(S.sphere(center, 0.2f, S.lambertian(v3(cr, cg, cb))));
        } else if (choose < 0.95f) {
            float cr = 0.5f * (1.0f + rng_uniform(g));
            float cg = 0.5f * (1.0f + rng_uniform(g));
            float cb = 0.5f * (1.0f + rng_uniform(g));
            float fuzz = 0.5f * rng_uniform(g);
            S.push(S.sphere(center, 0.2f, S.metal(v3(cr, cg, cb), fuzz)));
        } else {
            S.push(S.sphere(center, 0.2f, S.dielectric(1.5f)));
        }
    }
    S.push(S.sphere(v3(0.0f, 1.0f, 0.0f), 1.0f, S.dielectric(1.5f)));
    S.push(S.sphere(v3(-4.0f, 1.0f, 0.0f), 1.0f, S.lambertian(v3(0.4f, 0.2f, 0.1f))));
    S.push(S.sphere(v3(4.0f, 1.0f, 0.0f), 1.0f, S.metal(v3(0.7f, 0.6f, 0.5f), 0.0f)));
    S.finish();
    S.cam = make_camera(v3(13.0f, 2.0f, 3.0f), v3(0, 0, 0), v3(0, 1, 0), 20.0f, (float)nx / (float)ny, 0.1f, 10.0f, 0.0, 0.0);
    S.gradient = 1; S.def_nx = 1200; S.def_ny = 800; S.def_ns = 100;
}

// main.cu:402-450 create_world_cornell; host main.cu:1072-1127
void scene_cornell(Scene& S, int nx, int ny) {
    Mat* red = S.lambertian(v3(.65f, .05f, .05f));
    Mat* blue = S.lambertian(v3(.15f, .15f, .75f));
    Mat* white = S.lambertian(v3(.73f, .73f, .73f));
    Mat* lamp = S.light(v3(15.f, 15.f, 15.f));
    S.push(S.quad(v3(0, 0, 0), v3(0, 555, 0), v3(0, 0, 555), blue, true));
    S.push(S.quad(v3(555, 0, 555), v3(0, 555, 0), v3(0, 0, -555), red, true));
    S.push(S.quad(v3(0, 0, 0), v3(555, 0, 0), v3(0, 0, 555), white, true));
    S.push(S.quad(v3(0, 555, 555), v3(555, 0, 0), v3(0, 0, -555), white, true));
    S.push(S.quad(v3(555, 0, 555), v3(-555, 0, 0), v3(0, 555, 0), white, true));
    S.push(S.quad(v3(213, 554, 227), v3(130, 0, 0), v3(0, 0, 105), lamp, true));
    Obj* shortb = S.box(v3(0, 0, 0), v3(165, 165, 165), white);
    Obj* tallb = S.box(v3(0, 0, 0), v3(165, 330, 165), white);
    S.push(S.translate(S.rotate_y(shortb, -18.f), v3(130.f, 0.f, 65.f)));
    S.push(S.translate(S.rotate_y(tallb, 15.f), v3(265.f, 0.f, 295.f)));
    Mat* glass = S.dielectric(1.5f);
    S.push(S.sphere(v3(278.f, 335.f, 150.f), 60.f, glass));
    S.push(S.sphere(v3(278.f, 335.f, 150.f), -59.0f, glass));
    S.finish();
    V3 from = v3(278, 278, -800), at = v3(278, 278, 0);
    S.cam = make_camera(from, at, v3(0, 1, 0), 40.0f, (float)nx / (float)ny, 0.0f, vlen_folded(vsub(from, at)), 0.0, 1.0);
    S.gradient = 0; S.def_nx = 600; S.def_ny = 600; S.def_ns = 10000;
}

// main.cu:452-486 create_world_cornell_smoke; host main.cu:1129-1176
void scene_cornell_smoke(Scene& S, int nx, int ny) {
    Mat* red = S.lambertian(v3(.65f, .05f, .05f));
    Mat* white = S.lambertian(v3(.73f, .73f, .73f));
    Mat* green = S.lambertian(v3(.12f, .45f, .15f));
    Mat* lamp = S.light(v3(7.f, 7.f, 7.f));
    S.push(S.quad(v3(555, 0, 0), v3(0, 555, 0), v3(0, 0, 555), green, true));
    S.push(S.quad(v3(0, 0, 0), v3(0, 555, 0), v3(0, 0, 555), red, true));
    S.push(S.quad(v3(0, 555, 0), v3(555, 0, 0), v3(0, 0, 555), white, true));
    S.push(S.quad(v3(0, 0, 0), v3(555, 0, 0), v3(0, 0, 555), white, true));
    S.push(S.quad(v3(0, 0, 555), v3(555, 0, 0), v3(0, 555, 0), white, true));
    S.push(S.quad(v3(113, 554, 127), v3(330, 0, 0), v3(0, 0, 305), lamp, true));
    Obj* b1 = S.box(v3(0, 0, 0), v3(165, 330, 165), white);
    b1 = S.translate(S.rotate_y(b1, 15.f), v3(265.f, 0.f, 295.f));
    Obj* b2 = S.box(v3(0, 0, 0), v3(165, 165, 165), white);
    b2 = S.translate(S.rotate_y(b2, -18.f), v3(130.f, 0.f, 65.f));
    S.push(S.medium(b1, 0.01f, v3(0.5f, 0.5f, 0.5f)));
    S.push(S.medium(b2, 0.01f, v3(1, 1, 1)));
    S.finish();
    V3 from = v3(278, 278, -800), at = v3(278, 278, 0);
    S.cam = make_camera(from, at, v3(0, 1, 0), 40.0f, (float)nx / (float)ny, 0.0f, vlen_folded(vsub(from, at)), 0.0, 1.0);
    S.gradient = 0; S.def_nx = 600; S.def_ny = 600; S.def_ns = 1000;
}

inline V3 cube_point(int seed) {                                                     // util.cuh:3-11
    uint32_t s = 1103515245u * (uint32_t)(seed + 1) + 12345u;
    float out[3];
    for (int k = 0; k < 3; ++k) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        out[k] = (float)(s & 0xFFFFFFu) * (1.0f / 16777216.0f);
    }
    return v3(out[0], out[1], out[2]);
}

// main.cu:498-562 create_world_final; host main.cu:1178-1237
void scene_final(Scene& S, int nx, int ny) {
    Mat* white = S.lambertian(v3(.73f, .73f, .73f));
    Mat* ground = S.lambertian(v3(0.48f, 0.83f, 0.53f));
    Mat* lamp = S.light(v3(7, 7, 7));
    for (int ix = 0; ix < 20; ++ix) for (int iz = 0; iz < 20; ++iz) {
        float w = 100.0f;
        float x0 = -1000.0f + (float)ix * w;
        float z0 = -1000.0f + (float)iz * w;
        float y1 = 1.0f + 100.0f * (float)((ix * 13 + iz * 37) % 100) / 100.0f;
        S.push(S.box(v3(x0, 0, z0), v3(x0 + w, y1, z0 + w), ground));
    }
    S.push(S.quad(v3(123, 554, 147), v3(300, 0, 0), v3(0, 0, 265), lamp, true));
    V3 c1 = v3(400, 400, 200), c2 = vadd(c1, v3(30, 0, 0));
    S.push(S.moving_sphere(c1, c2, 50.f, S.lambertian(v3(0.7f, 0.3f, 0.1f))));
    S.push(S.sphere(v3(260, 150, 45), 50.f, S.dielectric(1.5f)));
    S.push(S.sphere(v3(0, 150, 145), 50.f, S.metal(v3(0.8f, 0.8f, 0.9f), 1.0f)));
    S.push(S.sphere(v3(360, 150, 145), 70.f, S.dielectric(1.5f)));
    S.push(S.medium(S.sphere(v3(360, 150, 145), 70.f, S.dielectric(1.5f)), 0.2f, v3(0.2f, 0.4f, 0.9f)));
    S.push(S.medium(S.sphere(v3(0, 0, 0), 5000.f, S.dielectric(1.5f)), 0.0001f, v3(1, 1, 1)));
    Tex* earth = S.tex_image();
    S.push(S.sphere(v3(400, 200, 400), 100.f, S.lambertian(earth)));
    S.push(S.sphere(v3(220, 280, 300), 80.f, S.lambertian(S.tex_noise(0.2f))));
    for (int j = 0; j < 1000; ++j) {
        V3 p = vscale(165.0f, cube_point(j));
        float r = 15.0f * 0.017453292519943295f;                                    // main.cu:489-496
        float c = cosf(r), s = sinf(r);
        p = v3(fmaf(c, p.x, s * p.z), p.y, fmaf(c, p.z, -(s * p.x)));
        p = vadd(p, v3(-100, 270, 395));
        S.push(S.sphere(p, 10.0f, white));
    }
    S.finish();
    V3 from = v3(478, 278, -600), at = v3(278, 278, 0);
    S.cam = make_camera(from, at, v3(0, 1, 0), 40.0f, (float)nx / (float)ny, 0.0f, vlen_folded(vsub(from, at)), 0.0, 1.0);
    S.gradient = 0; S.def_nx = 800; S.def_ny = 800; S.def_ns = 10000;
}

// main.cu:246-280 create_world_checker; host main.cu:746-800
void scene_checker(Scene& S, int nx, int ny) {
    Tex* chk = S.tex_checker(0.32f, S.tex_solid(v3(0.2f, 0.3f, 0.1f)), S.tex_solid(v3(0.9f, 0.9f, 0.9f)));
    Mat* lam = S.lambertian(chk);
    S.push(S.sphere(v3(0, -10, 0), 10.0f, lam));
    S.push(S.sphere(v3(0, 10, 0), 10.0f, lam));
    S.finish();
    S.cam = make_camera(v3(13.0f, 2.0f, 3.0f), v3(0, 0, 0), v3(0, 1, 0), 20.0f, (float)nx / (float)ny, 0.0f, 10.0f, 0.0, 1.0);
    S.gradient = 1; S.def_nx = 1200; S.def_ny = 600; S.def_ns = 500;
}
// main.cu:282-308 create_world_earth; host main.cu:802-880
void scene_earth(Scene& S, int nx, int ny) {
    S.push(S.sphere(v3(0, 0, 0), 2.0f, S.lambertian(S.tex_image())));
    S.finish();
    S.cam = make_camera(v3(0.0f, 0.0f, 12.0f), v3(0, 0, 0), v3(0, 1, 0), 20.0f, (float)nx / (float)ny, 0.0f, 12.0f, 0.0, 1.0);
    S.gradient = 1; S.def_nx = 1200; S.def_ny = 600; S.def_ns = 500;
}
// main.cu:310-329 create_world_perlin (scale 4.0, main.cu:903); host main.cu:882-937
void scene_perlin(Scene& S, int nx, int ny) {
    Mat* lam = S.lambertian(S.tex_noise(4.0f));
    S.push(S.sphere(v3(0, -1000, 0), 1000.f, lam));
    S.push(S.sphere(v3(0, 2, 0), 2.f, lam));
    S.finish();
    S.cam = make_camera(v3(13, 2, 3), v3(0, 0, 0), v3(0, 1, 0), 20.0f, (float)nx / (float)ny, 0.0f, 10.0f, 0.0, 1.0);
    S.gradient = 1; S.def_nx = 1200; S.def_ny = 600; S.def_ns = 500;
}
// main.cu:331-358 create_world_quads; host main.cu:939-993
void scene_quads(Scene& S, int nx, int ny) {
    S.push(S.quad(v3(-3, -2, 5), v3(0, 0, -4), v3(0, 4, 0), S.lambertian(v3(1.0f, 0.2f, 0.2f))));
    S.push(S.quad(v3(-2, -2, 0), v3(4, 0, 0), v3(0, 4, 0), S.lambertian(v3(0.2f, 1.0f, 0.2f))));
    S.push(S.quad(v3(3, -2, 1), v3(0, 0, 4), v3(0, 4, 0), S.lambertian(v3(0.2f, 0.2f, 1.0f))));
    S.push(S.quad(v3(-2, 3, 1), v3(4, 0, 0), v3(0, 0, 4), S.lambertian(v3(1.0f, 0.5f, 0.0f))));
    S.push(S.quad(v3(-2, -3, 5), v3(4, 0, 0), v3(0, 0, -4), S.lambertian(v3(0.2f, 0.8f, 0.8f))));
    S.finish();
    S.cam = make_camera(v3(0, 0, 9), v3(0, 0, 0), v3(0, 1, 0), 80.0f, (float)nx / (float)ny, 0.0f, 10.0f, 0.0, 1.0);
    S.gradient = 1; S.def_nx = 1200; S.def_ny = 600; S.def_ns = 500;
}

// main.cu:360-400 create_world_simple_light; host main.cu:995-1070
void scene_simple_light(Scene& S, int nx, int ny) {
    S.push(S.sphere(v3(0, -1000, 0), 1000.f, S.lambertian(S.tex_felt(v3(0.06f, 0.36f, 0.18f), 16.0f, 0.08f, 4.0f, 0.03f))));
    Tex* ball = S.tex_uv_offset(S.tex_image(), 60.0f / 360.0f, 0.f);
    const V3 C = v3(0, 2, 0); const float R = 2.0f;
    S.push(S.sphere(C, R, S.lambertian(ball)));
    S.push(S.sphere(C, R + 0.02f, S.dielectric(1.5f)));
    S.push(S.sphere(v3(0, 7, 0), 2.f, S.light(v3(4, 4, 4))));
    S.push(S.quad(v3(3, 1, -2), v3(2, 0, 0), v3(0, 2, 0), S.light(v3(4, 4, 4))));
    S.finish();
    V3 from = v3(26, 3, 6), at = v3(0, 2, 0);
    S.cam = make_camera(from, at, v3(0, 1, 0), 20.0f, (float)nx / (float)ny, 0.0f, vlen_folded(vsub(from, at)), 0.0, 1.0);
    S.gradient = 0; S.def_nx = 1200; S.def_ny = 600; S.def_ns = 10000;
}

// main.cu:564-635 create_world_original; host main.cu:1239-1305 (the scene the reference's main() renders)
void scene_original(Scene& S, int nx, int ny) {
    Mat* white = S.lambertian(v3(.73f, .73f, .73f));
    Mat* ground = S.lambertian(v3(0.88f, 0.50f, 0.76f));
    Mat* lamp = S.light(v3(7, 7, 7));
    for (int ix = 0; ix < 20; ++ix) for (int iz = 0; iz < 20; ++iz) {
        float w = 100.0f;
        float x0 = -1000.0f + (float)ix * w;
        float z0 = -1000.0f + (float)iz * w;
        float y1 = 1.0f + 100.0f * (float)((ix * 13 + iz * 37) % 100) / 100.0f;
        S.push(S.box(v3(x0, 0, z0), v3(x0 + w, y1, z0 + w), ground));
    }
    S.push(S.quad(v3(123, 554, 147), v3(300, 0, 0), v3(0, 0, 265), lamp, true));
    V3 c1 = v3(400, 400, 200), c2 = vadd(c1, v3(30, 0, 0));
    S.push(S.moving_sphere(c1, c2, 50.f, S.lambertian(v3(0.0488f, 0.0148f, 0.0171f))));
    S.push(S.sphere(v3(260, 150, 45), 50.f, S.dielectric(1.5f)));
    S.push(S.sphere(v3(0, 150, 145), 50.f, S.metal(v3(0.6387f, 0.3605f, 0.8826f), 1.0f)));
    S.push(S.sphere(v3(360.f, 150.f, 145.f), 70.f, S.lambertian(S.tex_image())));
    S.push(S.sphere(v3(360, 150, 145), 70.f + 0.5f, S.dielectric(1.5f)));
    S.push(S.medium(S.sphere(v3(0, 0, 0), 5000.f, S.dielectric(1.5f)), 0.0001f, v3(1, 1, 1)));
    S.push(S.sphere(v3(400, 200, 400), 100.f, S.metal(v3(0.23f, 0.24f, 0.85f), 0.02f)));
    S.push(S.sphere(v3(220, 280, 300), 80.f, S.lambertian(S.tex_noodle(0.2f))));
    for (int j = 0; j < 1000; ++j) {
        V3 p = vscale(165.0f, cube_point(j));
        float r = 15.0f * 0.017453292519943295f;
        float c = cosf(r), s = sinf(r);
        p = v3(fmaf(c, p.x, s * p.z), p.y, fmaf(c, p.z, -(s * p.x)));
        p = vadd(p, v3(-100, 270, 395));
        S.push(S.sphere(p, 10.0f, white));
    }
    S.finish();
    V3 from = v3(478, 278, -600), at = v3(278, 278, 0);
    S.cam = make_camera(from, at, v3(0, 1, 0), 40.0f, (float)nx / (float)ny, 0.0f, vlen_folded(vsub(from, at)), 0.0, 1.0);
    S.gradient = 0; S.background = v3(0.043f, 0.030f, 0.094f); S.def_nx = 800; S.def_ny = 800; S.def_ns = 10000;
}

// ---------------------------------------------------------------- render
inline float apply_gamma(float c, float gamma) {                                    // main.cu:37-42
    if (gamma == 1.0f) return c;
    float inv = 1.0f / gamma;
    return cr_powf(fmaxf(c, 0.0f), inv);
}

// Diagnostics for tools/order_experiment.py (never a parity test, never changes a result): every g_ray_stride-th ray of a
// single-threaded render is appended as (origin, direction, time, t of its closest hit or FLT_MAX), 8 floats.
unsigned long long g_ray_stride = 0, g_ray_seen = 0;
std::vector<float>* g_ray_out = nullptr;

V3 path_color(const Scene& S, const Ray& r0, V3 background, bool gradient, Rng& g) { // main.cu:44-87
    Ray cur = r0;
    V3 throughput = v3(1, 1, 1), radiance = v3(0, 0, 0);
    for (int bounce = 0; bounce < 50; ++bounce) {
        Hit rec;
        if (g_cnt) g_cnt->rays++;
        const bool hit_any = obj_hit(S.world, cur, 0.001f, FLT_MAX, rec);
        if (g_ray_out && g_ray_stride && (g_ray_seen++ % g_ray_stride) == 0) {
            const float v[8] = {cur.o.x, cur.o.y, cur.o.z, cur.d.x, cur.d.y, cur.d.z, (float)cur.tm, hit_any ? (float)rec.t : FLT_MAX};
            g_ray_out->insert(g_ray_out->end(), v, v + 8);
        }
        if (!hit_any) {
            V3 bg = background;
            if (gradient) {
                V3 ud = vunit(cur.d);
                float t = 0.5f * (ud.y + 1.0f);
                bg = v3(fmaf(t, 0.5f, 1.0f - t), fmaf(t, 0.7f, 1.0f - t), (1.0f - t) + t);   // (1-t)*1 folds to (1-t), t*1 to t
            }
            radiance = v3(fmaf(throughput.x, bg.x, radiance.x), fmaf(throughput.y, bg.y, radiance.y), fmaf(throughput.z, bg.z, radiance.z));
            break;
        }
        const V3 em = mat_emitted(rec.mat, (float)rec.u, (float)rec.v, rec.p);
        radiance = v3(fmaf(throughput.x, em.x, radiance.x), fmaf(throughput.y, em.y, radiance.y), fmaf(throughput.z, em.z, radiance.z));
        Ray scattered; V3 atten;
        if (!mat_scatter(rec.mat, cur, rec, atten, scattered, g)) break;
        throughput = vmul(throughput, atten);
        cur = scattered;
    }
    return radiance;
}

V3 render_pixel(const Scene& S, int i, int j, int nx, int ny, int ns, float gamma, V3 bg, int gradient,
                unsigned long long seed_base) {                                      // main.cu:96-133
    Rng g; rng_seed(g, seed_base + (unsigned long long)(j * nx + i));
    V3 col = v3(0, 0, 0);
    for (int s = 0; s < ns; ++s) {
        float u = ((float)i + rng_uniform(g)) / (float)nx;
        float v = ((float)j + rng_uniform(g)) / (float)ny;
        Ray r = camera_get_ray(S.cam, u, v, g);
        col = vadd(col, path_color(S, r, bg, gradient != 0, g));
        if (g_cnt) g_cnt->samples++;
    }
    col = vdiveq(col, (float)ns);
    col.x = apply_gamma(col.x, gamma); col.y = apply_gamma(col.y, gamma); col.z = apply_gamma(col.z, gamma);
    return col;
}

std::vector<std::unique_ptr<Scene>> g_scenes;

void dump_nodes_rec(const Obj* n, std::vector<float>& out) {
    // DFS pre-order; per node: lo[3], hi[3], leaf list_index or -1, 0
    bool leaf = (n->left == n->right);
    out.push_back(n->bbox.lo.x); out.push_back(n->bbox.lo.y); out.push_back(n->bbox.lo.z);
    out.push_back(n->bbox.hi.x); out.push_back(n->bbox.hi.y); out.push_back(n->bbox.hi.z);
    out.push_back(leaf && n->left ? (float)n->left->list_index : -1.0f);
    out.push_back(0.0f);
    if (!leaf) { dump_nodes_rec(n->left, out); dump_nodes_rec(n->right, out); }
}

}  // namespace

// ------------------------------------------------------------------ C entry points (ctypes)
extern "C" {

// Build a named scene; image (RGB8, may be null) is used by "final".  Returns a handle >= 0 or -1.
int orc_scene_create(const char* name, int nx, int ny, const unsigned char* img, int iw, int ih) {
    auto S = std::make_unique<Scene>();
    std::string n(name);
    if (img && iw > 0 && ih > 0) S->image.assign(img, img + (size_t)iw * ih * 3);
    if (n == "two_spheres") scene_two_spheres(*S, nx, ny);
    else if (n == "degenerate") scene_degenerate(*S, nx, ny);
    else if (n == "bouncing") scene_bouncing(*S, nx, ny);
    else if (n == "book1") scene_book1(*S, nx, ny);
    else if (n == "cornell") scene_cornell(*S, nx, ny);
    else if (n == "cornell_smoke") scene_cornell_smoke(*S, nx, ny);
    else if (n == "final") scene_final(*S, nx, ny);
    else if (n == "simple_light") scene_simple_light(*S, nx, ny);
    else if (n == "checker") scene_checker(*S, nx, ny);
    else if (n == "earth") scene_earth(*S, nx, ny);
    else if (n == "perlin") scene_perlin(*S, nx, ny);
    else if (n == "quads") scene_quads(*S, nx, ny);
    else if (n == "original") scene_original(*S, nx, ny);
    else return -1;
    for (auto& t : S->texs) if (t->kind == TEX_IMAGE && !S->image.empty()) { t->img = S->image.data(); t->w = iw; t->h = ih; }
    g_scenes.push_back(std::move(S));
    return (int)g_scenes.size() - 1;
}

// defaults the reference's host function uses: out[0..6] = nx, ny, ns, gradient, bg r,g,b ; returns gamma
float orc_scene_defaults(int h, float* out) {
    const Scene& S = *g_scenes[h];
    out[0] = (float)S.def_nx; out[1] = (float)S.def_ny; out[2] = (float)S.def_ns; out[3] = (float)S.gradient;
    out[4] = S.background.x; out[5] = S.background.y; out[6] = S.background.z;
    return S.gamma;
}

// Render rows [row0,row1) of an nx*ny frame into fb (full frame layout, index (j*nx+i)*3, row 0 = bottom).
// counters (may be null): rays, box_tests, sphere_tests, quad_tests, medium_calls, box6_calls, inst_calls, samples.
void orc_render(int h, float* fb, int nx, int ny, int ns, float gamma, const float* bg, int gradient,
                unsigned long long seed_base, int row0, int row1, unsigned long long* counters, int nthreads) {
    const Scene& S = *g_scenes[h];
    V3 b = v3(bg[0], bg[1], bg[2]);
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> cnt((size_t)nthreads);
    auto work = [&](int tid) {
        g_cnt = counters ? &cnt[(size_t)tid] : nullptr;
        for (int j = row0 + tid; j < row1; j += nthreads)
            for (int i = 0; i < nx; ++i) {
                V3 c = render_pixel(S, i, j, nx, ny, ns, gamma, b, gradient, seed_base);
                float* p = fb + ((size_t)j * nx + i) * 3;
                p[0] = c.x; p[1] = c.y; p[2] = c.z;
            }
        g_cnt = nullptr;
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (counters) {
        for (int k = 0; k < 8; ++k) counters[k] = 0;
        for (auto& c : cnt) {
            counters[0] += c.rays; counters[1] += c.box_tests; counters[2] += c.sphere_tests; counters[3] += c.quad_tests;
            counters[4] += c.medium_calls; counters[5] += c.box6_calls; counters[6] += c.inst_calls; counters[7] += c.samples;
        }
    }
}

// Rays traced (world->hit calls) by each pixel of row j: the per-pixel cost that bounds strong scaling.
void orc_row_pixel_rays(int h, int nx, int ny, int ns, const float* bg, int gradient, unsigned long long seed_base, int j,
                        unsigned long long* out_nx) {
    const Scene& S = *g_scenes[h];
    V3 b = v3(bg[0], bg[1], bg[2]);
    Counters c;
    g_cnt = &c;
    for (int i = 0; i < nx; ++i) {
        const unsigned long long before = c.rays;
        (void)render_pixel(S, i, j, nx, ny, ns, 1.0f, b, gradient, seed_base);
        out_nx[i] = c.rays - before;
    }
    g_cnt = nullptr;
}

// XORWOW known-answer helper: state after init (v0..v4,d) and the first n uniforms / raw words.
void orc_xorwow(unsigned long long seed, int n, unsigned int* state6, float* uniforms, unsigned int* raw) {
    Rng g; rng_seed(g, seed);
    if (state6) { for (int k = 0; k < 5; ++k) state6[k] = g.v[k]; state6[5] = g.d; }
    Rng g2 = g;
    for (int k = 0; k < n; ++k) {
        if (uniforms) uniforms[k] = rng_uniform(g);
        if (raw) raw[k] = rng_next(g2);
    }
}

// BVH in DFS pre-order, 8 floats per node (lo, hi, leaf list_index or -1, 0). Returns node count.
int orc_dump_nodes(int h, float* out, int cap_nodes) {
    std::vector<float> v; dump_nodes_rec(g_scenes[h]->world, v);
    int n = (int)(v.size() / 8);
    if (out) for (int k = 0; k < n && k < cap_nodes; ++k) memcpy(out + 8 * k, v.data() + 8 * k, 32);
    return n;
}

// Per-node count of box tests that passed since the last reset (tools/, and tests/test_gpu_parity.py's check that the
// device's calibration pass counts what the oracle counts; it never decides a pixel), in
// the DFS pre-order of orc_dump_nodes.  enable != 0 switches the counting on and clears the counters.
static void node_pass_rec(const Obj* n, std::vector<unsigned long long>* out, bool clear) {
    if (out) out->push_back(n->stat_pass);
    if (clear) n->stat_pass = 0;
    if (n->left != n->right) { node_pass_rec(n->left, out, clear); node_pass_rec(n->right, out, clear); }
}
void orc_node_stats_enable(int h, int enable) { g_node_stats = enable != 0; node_pass_rec(g_scenes[h]->world, nullptr, true); }
int orc_node_passes(int h, unsigned long long* out, int cap) {
    std::vector<unsigned long long> v; node_pass_rec(g_scenes[h]->world, &v, false);
    for (int k = 0; k < (int)v.size() && k < cap; ++k) out[k] = v[k];
    return (int)v.size();
}

// scene census: out[0..] = list size, spheres, moving spheres, quads(in list), boxes, instances(translate), media,
// lambertian, metal, dielectric, light (materials of list spheres only), bvh depth
// Diagnostics (tools/order_experiment.py): render rows [row0, row1) at ns spp on ONE thread and return up to `cap` rays,
// every stride-th one, as 8 floats each (origin, direction, time, closest t or FLT_MAX).  Returns the number of rays written.
int orc_ray_sample(int h, int nx, int ny, int ns, int row0, int row1, unsigned long long stride, float* out, int cap) {
    std::vector<float> rays;
    g_ray_out = &rays; g_ray_stride = stride ? stride : 1; g_ray_seen = 0;
    const Scene& S = *g_scenes[h];
    for (int j = row0; j < row1; ++j)
        for (int i = 0; i < nx; ++i) (void)render_pixel(S, i, j, nx, ny, ns, 1.0f, S.background, S.gradient, 1984ull);
    g_ray_out = nullptr; g_ray_stride = 0;
    const int n = (int)(rays.size() / 8);
    const int m = n < cap ? n : cap;
    if (out) memcpy(out, rays.data(), (size_t)m * 8 * sizeof(float));
    return m;
}

void orc_scene_census(int h, int* out) {
    const Scene& S = *g_scenes[h];
    for (int k = 0; k < 12; ++k) out[k] = 0;
    out[0] = (int)S.list.size();
    for (const Obj* o : S.list) {
        if (o->kind == OBJ_SPHERE) {
            out[1]++; if (o->vel.x != 0 || o->vel.y != 0 || o->vel.z != 0) out[2]++;
            switch (o->mat->kind) { case MAT_LAMBERTIAN: out[7]++; break; case MAT_METAL: out[8]++; break;
                case MAT_DIELECTRIC: out[9]++; break; case MAT_LIGHT: out[10]++; break; default: break; }
        } else if (o->kind == OBJ_QUAD) out[3]++;
        else if (o->kind == OBJ_BOX6) out[4]++;
        else if (o->kind == OBJ_TRANSLATE) out[5]++;
        else if (o->kind == OBJ_MEDIUM) out[6]++;
    }
    struct D { static int depth(const Obj* n) { if (n->kind != OBJ_BVH || n->left == n->right) return 1;
        int a = depth(n->left), b = depth(n->right); return 1 + (a > b ? a : b); } };
    out[11] = D::depth(S.world);
}

// camera POD for cross-checks: origin, llc, horizontal, vertical, u, v (18 floats), lens_radius, time0, time1
void orc_camera(int h, float* out21) {
    const Camera& c = g_scenes[h]->cam;
    const V3 vs[6] = {c.origin, c.llc, c.horizontal, c.vertical, c.u, c.v};
    for (int k = 0; k < 6; ++k) { out21[3 * k] = vs[k].x; out21[3 * k + 1] = vs[k].y; out21[3 * k + 2] = vs[k].z; }
    out21[18] = c.lens_radius; out21[19] = (float)c.time0; out21[20] = (float)c.time1;
}

// single-function vectors
float orc_perlin_noise(float x, float y, float z) { return perlin_noise(v3(x, y, z)); }
float orc_perlin_turb(float x, float y, float z, int depth) { return perlin_turb(v3(x, y, z), depth); }

}  // extern "C"
