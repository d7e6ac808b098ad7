// rtw.h -- host-side mirror of the reference's hittable / material / texture /
// camera interface (namespace rtw).
//
// In the reference these classes live on the device heap: single-thread
// kernels `new` them (src/main.cu:160-635) and the render kernel walks the
// pointer graph through virtual calls.  Here they are HOST objects that only
// record the scene: same class names, same constructor signatures, same
// derived quantities (bounding boxes, quad plane constants, rotation sin/cos,
// BVH topology) computed with the same fp32 arithmetic, and a flattener
// (rtw::flatten) that turns the graph into the plain arrays of
// include/rt_abi.h for the HIP kernels.  There is deliberately no hit() /
// scatter() here: the only render path of the product is the GPU one.
//
// Build with -ffp-contract=off.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rt_abi.h"

namespace rtw {

// ------------------------------------------------------------------ vec3
// src/vec3.cuh:8-158.  Only what scene construction needs.
struct vec3 {
    float e[3];
    vec3() : e{0.f, 0.f, 0.f} {}
    vec3(float a, float b, float c) : e{a, b, c} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float operator[](int i) const { return e[i]; }
    float& operator[](int i) { return e[i]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    // Contracted the way nvcc's default -fmad=true contracts the reference's device code (DESIGN.md "numerical
    // contract"): m0 + m1 + m2 = fma(m2, fma(m0, m1)).  The *_folded forms are the uncontracted per-operation values
    // a compiler's constant folder produces; the reference's camera arguments are compile-time constants.
    float length() const { return sqrtf(fmaf(e[2], e[2], fmaf(e[0], e[0], e[1] * e[1]))); }
    float squared_length() const { return fmaf(e[2], e[2], fmaf(e[0], e[0], e[1] * e[1])); }
    float length_folded() const { return sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]); }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline vec3 operator*(float t, const vec3& v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3& v, float t) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator/(const vec3& v, float t) { return vec3(v.e[0] / t, v.e[1] / t, v.e[2] / t); }
inline float dot(const vec3& a, const vec3& b) { return fmaf(a.e[2], b.e[2], fmaf(a.e[0], b.e[0], a.e[1] * b.e[1])); }
inline vec3 cross(const vec3& a, const vec3& b) {
    return vec3(fmaf(a.e[1], b.e[2], -(a.e[2] * b.e[1])), -fmaf(a.e[0], b.e[2], -(a.e[2] * b.e[0])), fmaf(a.e[0], b.e[1], -(a.e[1] * b.e[0])));
}
inline vec3 cross_folded(const vec3& a, const vec3& b) {
    return vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], -(a.e[0] * b.e[2] - a.e[2] * b.e[0]), a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline vec3 fma3(float t, const vec3& v, const vec3& a) { return vec3(fmaf(t, v.e[0], a.e[0]), fmaf(t, v.e[1], a.e[1]), fmaf(t, v.e[2], a.e[2])); }   // a + t*v
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }

// ------------------------------------------------------------------ aabb
// src/aabb.cuh:8-79 (construction helpers only; the slab test is device code)
struct aabb {
    vec3 minimum, maximum;
    aabb() : minimum(FLT_MAX, FLT_MAX, FLT_MAX), maximum(-FLT_MAX, -FLT_MAX, -FLT_MAX) {}
    aabb(const vec3& a, const vec3& b)
        : minimum(fminf(a.x(), b.x()), fminf(a.y(), b.y()), fminf(a.z(), b.z())),
          maximum(fmaxf(a.x(), b.x()), fmaxf(a.y(), b.y()), fmaxf(a.z(), b.z())) {}
    const vec3& min() const { return minimum; }
    const vec3& max() const { return maximum; }
    aabb pad(float delta) const { vec3 d(delta, delta, delta); return aabb(minimum - d, maximum + d); }
    static aabb surrounding_box(const aabb& b0, const aabb& b1) {
        vec3 lo(fminf(b0.minimum.x(), b1.minimum.x()), fminf(b0.minimum.y(), b1.minimum.y()), fminf(b0.minimum.z(), b1.minimum.z()));
        vec3 hi(fmaxf(b0.maximum.x(), b1.maximum.x()), fmaxf(b0.maximum.y(), b1.maximum.y()), fmaxf(b0.maximum.z(), b1.maximum.z()));
        return aabb(lo, hi);
    }
};
inline aabb operator+(const aabb& b, const vec3& off) { return aabb(b.minimum + off, b.maximum + off); }

// ------------------------------------------------------------------ ownership
// The reference's ownership rules (leaves own materials, materials own
// textures, make_box shares one material between six owning quads, ...) end
// in leaks and double frees.  Here every scene object
// registers with the arena that is active when it is constructed; the arena
// frees them all.  Callers keep writing `new sphere(...)`.
class object;
class arena {
public:
    arena() {}
    ~arena();
    arena(const arena&) = delete;
    arena& operator=(const arena&) = delete;
    static arena* current();
    void adopt(object* o) { owned_.push_back(o); }
    // objects constructed while a scope is alive belong to its arena
    struct scope {
        explicit scope(arena& a);
        ~scope();
        arena* prev;
    };
private:
    std::vector<object*> owned_;
};
class object {
public:
    object() { if (arena* a = arena::current()) a->adopt(this); }
    virtual ~object() {}
};

// ------------------------------------------------------------------ textures
// src/texture.cuh:7-76 and src/image_io.h:5-47
struct DeviceImage {           // host pixels here; the library uploads them with the scene
    const unsigned char* data = nullptr;
    int width = 0, height = 0, bpp = 3;
    bool valid() const { return data && width > 0 && height > 0 && bpp >= 3; }
};

class texture : public object {
public:
    virtual int tex_kind() const = 0;   // RT_TEX_*
};
class solid_color : public texture {
public:
    vec3 albedo;
    solid_color() {}
    solid_color(const vec3& a) : albedo(a) {}
    int tex_kind() const override { return RT_TEX_SOLID; }
};
class checker_texture : public texture {
public:
    float inv_scale = 1.f;
    texture* even = nullptr;
    texture* odd = nullptr;
    checker_texture(float scale, texture* e, texture* o) : inv_scale(1.f / scale), even(e), odd(o) {}
    int tex_kind() const override { return RT_TEX_CHECKER; }
};
class image_texture : public texture {
public:
    DeviceImage img;
    image_texture() {}
    image_texture(const DeviceImage& d) : img(d) {}
    int tex_kind() const override { return RT_TEX_IMAGE; }
};
class noise_texture : public texture {
public:
    float scale;
    explicit noise_texture(float s) : scale(s) {}
    int tex_kind() const override { return RT_TEX_NOISE; }
};

// src/texture.cuh:84-103
class noodle_texture : public texture {
public:
    float k, A, f;
    int octaves;
    vec3 d, cN, cG;
    noodle_texture(float stripes_k = 3.0f, float wiggle_amp = 3.0f, float wiggle_freq = 0.6f, int oct = 3,
                   vec3 dir = vec3(0, 0, 1), vec3 noodle = vec3(0.92f, 0.85f, 0.65f), vec3 gap = vec3(0.35f, 0.20f, 0.10f))
        : k(stripes_k), A(wiggle_amp), f(wiggle_freq), octaves(oct), d(unit_vector(dir)), cN(noodle), cG(gap) {}
    int tex_kind() const override { return RT_TEX_NOODLE; }
};
// src/texture.cuh:109-148
class felt_texture : public texture {
public:
    vec3 base_col;
    float m_scale, m_amt, f_scale, f_amt;
    felt_texture(const vec3& base = vec3(0.06f, 0.36f, 0.18f), float mottling_scale = 16.0f, float mottling_amt = 0.08f,
                 float fiber_scale = 4.0f, float fiber_amt = 0.03f)
        : base_col(base), m_scale(mottling_scale), m_amt(mottling_amt), f_scale(fiber_scale), f_amt(fiber_amt) {}
    int tex_kind() const override { return RT_TEX_FELT; }
};
// src/texture.cuh:151-164
class uv_offset_texture : public texture {
public:
    texture* base_;
    float du, dv;
    uv_offset_texture(texture* base, float u_offset_turns, float v_offset = 0.f) : base_(base), du(u_offset_turns), dv(v_offset) {}
    int tex_kind() const override { return RT_TEX_UV_OFFSET; }
};

// ------------------------------------------------------------------ materials
// src/material.cuh:46-201
class material : public object {
public:
    virtual int mat_kind() const = 0;   // RT_MAT_*
};
class lambertian : public material {
public:
    texture* tex;
    lambertian(const vec3& albedo) : tex(new solid_color(albedo)) {}
    lambertian(texture* t) : tex(t) {}
    int mat_kind() const override { return RT_MAT_LAMBERTIAN; }
};
class metal : public material {
public:
    vec3 albedo;
    float fuzz;
    metal(const vec3& a, float f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {}
    int mat_kind() const override { return RT_MAT_METAL; }
};
class dielectric : public material {
public:
    float ref_idx;
    dielectric(float ri) : ref_idx(ri) {}
    int mat_kind() const override { return RT_MAT_DIELECTRIC; }
};
class diffuse_light : public material {
public:
    texture* tex;   // optional
    vec3 solid;     // used when tex == nullptr
    diffuse_light(texture* t) : tex(t) {}
    diffuse_light(const vec3& c) : tex(nullptr), solid(c) {}
    int mat_kind() const override { return RT_MAT_DIFFUSE_LIGHT; }
};
class isotropic : public material {
public:
    texture* tex;
    isotropic(texture* t) : tex(t) {}
    isotropic(const vec3& c) : tex(new solid_color(c)) {}
    int mat_kind() const override { return RT_MAT_ISOTROPIC; }
};

// ------------------------------------------------------------------ hittables
// src/hittable.cuh:9-178
enum HKind : int { HK_Sphere = 0, HK_Quad = 1, HK_BVH = 2, HK_Composite = 3 };

class hittable : public object {
public:
    virtual aabb bounding_box() const = 0;
    virtual HKind kind() const = 0;
};

// src/sphere.cuh:10-102
class sphere : public hittable {
public:
    vec3 center0, velocity;   // c(t) = center0 + t*velocity
    float radius;
    material* mat_ptr;
    aabb bbox;
    sphere(vec3 cen, float r, material* m, bool owns = true);
    sphere(vec3 cen1, vec3 cen2, float r, material* m);
    aabb bounding_box() const override { return bbox; }
    HKind kind() const override { return HK_Sphere; }
};

// src/quad.cuh:11-91
class quad : public hittable {
public:
    vec3 Q, u, v, w, normal;
    float D;
    aabb bbox;
    material* mat_ptr;
    bool inward;
    quad(const vec3& Q_, const vec3& u_, const vec3& v_, material* m, bool inward_ = false, bool owns_ = true);
    aabb bounding_box() const override { return bbox; }
    HKind kind() const override { return HK_Quad; }
};

// src/quad.cuh:94-143
class compound6 : public hittable {
public:
    hittable* faces[6];
    aabb box;
    compound6(hittable* f0, hittable* f1, hittable* f2, hittable* f3, hittable* f4, hittable* f5);
    aabb bounding_box() const override { return box; }
    HKind kind() const override { return HK_Composite; }
};
hittable* make_box(const vec3& a, const vec3& b, material* mat);   // src/quad.cuh:145-162

// src/hittable.cuh:40-69
class translate : public hittable {
public:
    hittable* obj;
    vec3 offset;
    aabb box;
    translate(hittable* p, const vec3& d) : obj(p), offset(d) { box = obj->bounding_box() + d; }
    aabb bounding_box() const override { return box; }
    HKind kind() const override { return obj->kind(); }
};
// src/hittable.cuh:77-149
class rotate_y : public hittable {
public:
    hittable* obj;
    float sin_t, cos_t;
    aabb box;
    rotate_y(hittable* p, float angle_degrees);
    aabb bounding_box() const override { return box; }
    HKind kind() const override { return obj->kind(); }
};
// src/hittable.cuh:154-178
class with_material : public hittable {
public:
    hittable* obj;
    material* mat;
    aabb box;
    with_material(hittable* p, material* m) : obj(p), mat(m) { box = obj->bounding_box(); }
    aabb bounding_box() const override { return box; }
    HKind kind() const override { return obj->kind(); }
};

// src/constant_medium.cuh:16-80
class constant_medium : public hittable {
public:
    hittable* boundary;
    float neg_inv_density;
    material* phase_function;
    constant_medium(hittable* b, float density, texture* tex)
        : boundary(b), neg_inv_density(-1.0f / density), phase_function(new isotropic(tex)) {}
    constant_medium(hittable* b, float density, const vec3& albedo)
        : boundary(b), neg_inv_density(-1.0f / density), phase_function(new isotropic(new solid_color(albedo))) {}
    aabb bounding_box() const override { return boundary->bounding_box(); }
    HKind kind() const override { return HK_Composite; }
};

// src/bvh.cuh:9-116.  The constructor reorders objects[start,end) in place,
// exactly as the reference's does.
class bvh_node : public hittable {
public:
    hittable* left = nullptr;
    hittable* right = nullptr;
    aabb box;
    bvh_node() {}
    bvh_node(hittable** objects, int start, int end);
    aabb bounding_box() const override { return box; }
    HKind kind() const override { return HK_BVH; }
};

// ------------------------------------------------------------------ camera
// src/camera.cuh:18-79
class camera : public object {
public:
    camera(vec3 lookfrom, vec3 lookat, vec3 vup, float vfov, float aspect, float aperture, float focus_dist)
        : time0(0.0), time1(0.0) { init(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist); }
    camera(vec3 lookfrom, vec3 lookat, vec3 vup, float vfov, float aspect, float aperture, float focus_dist,
           double t0, double t1)
        : time0(t0), time1(t1) { init(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist); }
    vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float lens_radius;
    double time0, time1;
private:
    void init(vec3 lookfrom, vec3 lookat, vec3 vup, float vfov, float aspect, float aperture, float focus_dist);
};

// ------------------------------------------------------------------ flattening
// Owns the arrays an rt_scene_desc points into.
struct flat_scene {
    std::vector<rt_node> nodes;
    std::vector<rt_sphere> spheres;
    std::vector<rt_quad> quads;
    std::vector<rt_box> boxes;
    std::vector<rt_instance> instances;
    std::vector<rt_medium> media;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<uint8_t> images;
    std::vector<int32_t> leaf_order;   // per node: index of the object in the caller's list, or -1
    rt_camera camera;
    rt_scene_desc desc() const;
};

// Turns a bvh_node-rooted world + camera into flat arrays.  `list`/`count`
// (optional) is the caller's object list in creation order; when given,
// leaf_order maps each leaf node to its position there.  Returns RT_OK, or
// RT_ERR_UNSUPPORTED with a message in `err` for nestings the kernels do not
// implement.
rt_status flatten(const hittable* world, const camera& cam, flat_scene& out, std::string& err,
                  hittable* const* creation_order = nullptr, int count = 0);

// util.cuh:3-11
vec3 random_in_unit_cube(int seed);

}  // namespace rtw
