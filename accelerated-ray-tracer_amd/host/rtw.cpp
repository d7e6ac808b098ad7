// rtw.cpp -- constructors of the host scene classes, the BVH builder and the
// flattener.  See rtw.h.  Build with -ffp-contract=off.
#include "rtw.h"

namespace rtw {

// ------------------------------------------------------------------ arena
static thread_local arena* g_arena = nullptr;
arena::scope::scope(arena& a) : prev(g_arena) { g_arena = &a; }
arena::scope::~scope() { g_arena = prev; }
arena::~arena() {
    // objects may have registered children constructed inside their own
    // constructors (lambertian(vec3) makes a solid_color); plain reverse
    // deletion is safe because no destructor touches another object.
    for (size_t i = owned_.size(); i-- > 0;) delete owned_[i];
}
arena* arena::current() { return g_arena; }

// ------------------------------------------------------------------ sphere
// src/sphere.cuh:21-26: static sphere, box = centre -/+ radius
sphere::sphere(vec3 cen, float r, material* m, bool /*owns*/)
    : center0(cen), velocity(0.f, 0.f, 0.f), radius(r), mat_ptr(m) {
    vec3 rvec(radius, radius, radius);
    bbox = aabb(cen - rvec, cen + rvec);
}
// src/sphere.cuh:29-38: moving sphere, box = union of the boxes at t=0 and t=1
sphere::sphere(vec3 cen1, vec3 cen2, float r, material* m)
    : center0(cen1), velocity(cen2 - cen1), radius(r), mat_ptr(m) {
    vec3 rvec(radius, radius, radius);
    vec3 at0 = center0 + 0.0f * velocity;   // ray::point_at_parameter(0.0)
    vec3 at1 = center0 + 1.0f * velocity;   // ray::point_at_parameter(1.0)
    bbox = aabb::surrounding_box(aabb(at0 - rvec, at0 + rvec), aabb(at1 - rvec, at1 + rvec));
}

// ------------------------------------------------------------------ quad
// src/quad.cuh:29-54
quad::quad(const vec3& Q_, const vec3& u_, const vec3& v_, material* m, bool inward_, bool /*owns_*/)
    : Q(Q_), u(u_), v(v_), mat_ptr(m), inward(inward_) {
    vec3 n = cross(u, v);
    normal = unit_vector(n);
    if (inward) normal = -normal;
    D = dot(normal, Q);
    w = n / dot(n, n);
    aabb diag1(Q, Q + u + v);
    aabb diag2(Q + u, Q + v);
    bbox = aabb::surrounding_box(diag1, diag2).pad(1e-3f);
}

// src/quad.cuh:108-122
compound6::compound6(hittable* f0, hittable* f1, hittable* f2, hittable* f3, hittable* f4, hittable* f5) {
    faces[0] = f0; faces[1] = f1; faces[2] = f2; faces[3] = f3; faces[4] = f4; faces[5] = f5;
    box = faces[0]->bounding_box();
    for (int i = 1; i < 6; ++i) {
        const aabb b = faces[i]->bounding_box();
        const vec3 lo(fminf(box.minimum.x(), b.minimum.x()), fminf(box.minimum.y(), b.minimum.y()), fminf(box.minimum.z(), b.minimum.z()));
        const vec3 hi(fmaxf(box.maximum.x(), b.maximum.x()), fmaxf(box.maximum.y(), b.maximum.y()), fmaxf(box.maximum.z(), b.maximum.z()));
        box = aabb(lo, hi);
    }
}

// src/quad.cuh:145-162.  Face order front(+Z), right(+X), back(-Z), left(-X), top(+Y), bottom(-Y).
hittable* make_box(const vec3& a, const vec3& b, material* mat) {
    const vec3 lo(fminf(a.x(), b.x()), fminf(a.y(), b.y()), fminf(a.z(), b.z()));
    const vec3 hi(fmaxf(a.x(), b.x()), fmaxf(a.y(), b.y()), fmaxf(a.z(), b.z()));
    const vec3 ex(hi.x() - lo.x(), 0.f, 0.f);
    const vec3 ey(0.f, hi.y() - lo.y(), 0.f);
    const vec3 ez(0.f, 0.f, hi.z() - lo.z());
    hittable* f[6];
    f[0] = new quad(vec3(lo.x(), lo.y(), hi.z()), ex, ey, mat);
    f[1] = new quad(vec3(hi.x(), lo.y(), hi.z()), -ez, ey, mat);
    f[2] = new quad(vec3(hi.x(), lo.y(), lo.z()), -ex, ey, mat);
    f[3] = new quad(vec3(lo.x(), lo.y(), lo.z()), ez, ey, mat);
    f[4] = new quad(vec3(lo.x(), hi.y(), hi.z()), ex, -ez, mat);
    f[5] = new quad(vec3(lo.x(), lo.y(), lo.z()), ex, ez, mat);
    return new compound6(f[0], f[1], f[2], f[3], f[4], f[5]);
}

// src/hittable.cuh:89-116
rotate_y::rotate_y(hittable* p, float angle_degrees) : obj(p) {
    const float rad = angle_degrees * 0.017453292519943295769f;
    sin_t = sinf(rad);
    cos_t = cosf(rad);
    const aabb b = obj->bounding_box();
    vec3 lo(FLT_MAX, FLT_MAX, FLT_MAX), hi(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                const float x = i ? b.maximum.x() : b.minimum.x();
                const float y = j ? b.maximum.y() : b.minimum.y();
                const float z = k ? b.maximum.z() : b.minimum.z();
                const float rx = fmaf(cos_t, x, sin_t * z);
                const float rz = fmaf(cos_t, z, -(sin_t * x));   // -s*x + c*z = c*z - s*x: the c*z product is the fused one
                lo = vec3(fminf(lo.x(), rx), fminf(lo.y(), y), fminf(lo.z(), rz));
                hi = vec3(fmaxf(hi.x(), rx), fmaxf(hi.y(), y), fmaxf(hi.z(), rz));
            }
    box = aabb(lo, hi);
}

// ------------------------------------------------------------------ bvh_node
// src/bvh.cuh:29-84: split axis = largest spread of box minima; in-place
// selection sort with strict <; split at n>>1; a single object gets a node of
// its own with left == right.
static inline float axis_min(const hittable* h, int axis) {
    const aabb b = h->bounding_box();
    return axis == 0 ? b.minimum.x() : (axis == 1 ? b.minimum.y() : b.minimum.z());
}
bvh_node::bvh_node(hittable** objects, int start, int end) {
    const int n = end - start;
    if (n <= 0) { left = right = nullptr; box = aabb(); return; }
    if (n == 1) { left = right = objects[start]; box = left->bounding_box(); return; }

    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = start; i < end; ++i) {
        const vec3 mn = objects[i]->bounding_box().min();
        for (int a = 0; a < 3; ++a) {
            if (mn[a] < lo[a]) lo[a] = mn[a];
            if (mn[a] > hi[a]) hi[a] = mn[a];
        }
    }
    const float sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
    int axis = 0;
    if (sy > sx && sy >= sz) axis = 1;
    else if (sz > sx && sz >= sy) axis = 2;

    for (int i = start; i < end - 1; ++i) {
        int best = i;
        for (int j = i + 1; j < end; ++j)
            if (axis_min(objects[j], axis) < axis_min(objects[best], axis)) best = j;
        if (best != i) { hittable* t = objects[i]; objects[i] = objects[best]; objects[best] = t; }
    }
    const int mid = start + (n >> 1);
    left = new bvh_node(objects, start, mid);
    right = new bvh_node(objects, mid, end);
    box = aabb::surrounding_box(left->bounding_box(), right->bounding_box());
}

// ------------------------------------------------------------------ camera
// src/camera.cuh:59-78
void camera::init(vec3 lookfrom, vec3 lookat, vec3 vup, float vfov, float aspect, float aperture, float focus_dist) {
    lens_radius = aperture * 0.5f;
    const float theta = vfov * 3.141592654f / 180.0f;
    const float half_height = tanf(theta * 0.5f);
    const float half_width = aspect * half_height;
    origin = lookfrom;
    // lookfrom, lookat, vup, vfov and focus_dist are compile-time constants in every reference scene kernel, so the
    // basis and the terms built only from them are folded per operation (no contraction); aspect (nx/ny) is a kernel
    // argument, so the half_width term is a run-time product contracted into the subtraction.
    const vec3 dw = lookfrom - lookat;
    w = dw / dw.length_folded();
    const vec3 du = cross_folded(vup, w);
    u = du / du.length_folded();
    v = cross_folded(w, u);
    lower_left_corner = fma3(-(half_width * focus_dist), u, origin) - half_height * focus_dist * v - focus_dist * w;
    horizontal = 2.0f * half_width * focus_dist * u;
    vertical = 2.0f * half_height * focus_dist * v;
}

// src/util.cuh:3-11
vec3 random_in_unit_cube(int seed) {
    uint32_t s = 1103515245u * (uint32_t)(seed + 1) + 12345u;
    float c[3];
    for (int k = 0; k < 3; ++k) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        c[k] = (float)(s & 0xFFFFFFu) * (1.0f / 16777216.0f);
    }
    return vec3(c[0], c[1], c[2]);
}

// ------------------------------------------------------------------ flatten
rt_scene_desc flat_scene::desc() const {
    rt_scene_desc d;
    memset(&d, 0, sizeof(d));
    d.nodes = nodes.data(); d.n_nodes = (int32_t)nodes.size();
    d.spheres = spheres.data(); d.n_spheres = (int32_t)spheres.size();
    d.quads = quads.data(); d.n_quads = (int32_t)quads.size();
    d.boxes = boxes.data(); d.n_boxes = (int32_t)boxes.size();
    d.instances = instances.data(); d.n_instances = (int32_t)instances.size();
    d.media = media.data(); d.n_media = (int32_t)media.size();
    d.materials = materials.data(); d.n_materials = (int32_t)materials.size();
    d.textures = textures.data(); d.n_textures = (int32_t)textures.size();
    d.images = images.data(); d.image_bytes = images.size();
    d.camera = camera;
    return d;
}

namespace {

inline void put3(float* dst, const vec3& v) { dst[0] = v.x(); dst[1] = v.y(); dst[2] = v.z(); }

struct flattener {
    flat_scene& out;
    std::string& err;
    std::unordered_map<const texture*, int> tex_ids;
    std::unordered_map<const material*, int> mat_ids;
    std::unordered_map<const hittable*, int32_t> prim_refs;
    std::unordered_map<const unsigned char*, int> image_offsets;
    std::unordered_map<const hittable*, int> order;
    bool ok = true;

    flattener(flat_scene& o, std::string& e) : out(o), err(e) {}

    void fail(const std::string& why) { if (ok) err = why; ok = false; }

    int texture_id(const texture* t) {
        auto it = tex_ids.find(t);
        if (it != tex_ids.end()) return it->second;
        rt_texture r; memset(&r, 0, sizeof(r));
        r.kind = t->tex_kind();
        if (auto s = dynamic_cast<const solid_color*>(t)) {
            put3(r.color, s->albedo);
        } else if (auto c = dynamic_cast<const checker_texture*>(t)) {
            r.scale = c->inv_scale;
            r.a = texture_id(c->even);
            r.b = texture_id(c->odd);
        } else if (auto im = dynamic_cast<const image_texture*>(t)) {
            if (im->img.valid()) {
                auto f = image_offsets.find(im->img.data);
                int off;
                if (f != image_offsets.end()) off = f->second;
                else {
                    off = (int)out.images.size();
                    const unsigned char* px = im->img.data;
                    // the kernel reads RGB8 with stride 3 (bpp is forced to 3 by image_io.h:26)
                    if (im->img.bpp == 3) out.images.insert(out.images.end(), px, px + (size_t)im->img.width * im->img.height * 3);
                    else for (size_t k = 0; k < (size_t)im->img.width * im->img.height; ++k)
                        for (int ch = 0; ch < 3; ++ch) out.images.push_back(px[k * im->img.bpp + ch]);
                    while (out.images.size() % 16) out.images.push_back(0);
                    image_offsets[im->img.data] = off;
                }
                r.a = off; r.b = im->img.width; r.c = im->img.height;
            } else {
                r.a = -1; r.b = 0; r.c = 0;   // invalid image: value() returns (0,1,1), texture.cuh:52
            }
        } else if (auto nz = dynamic_cast<const noise_texture*>(t)) {
            r.scale = nz->scale;
        } else if (auto nd = dynamic_cast<const noodle_texture*>(t)) {
            r.scale = nd->k; r.p[6] = nd->A; r.p[7] = nd->f; r.a = nd->octaves;
            put3(r.color, nd->cN); put3(r.p, nd->cG); put3(r.p + 3, nd->d);
        } else if (auto fe = dynamic_cast<const felt_texture*>(t)) {
            put3(r.color, fe->base_col); r.scale = fe->m_scale; r.p[0] = fe->m_amt; r.p[1] = fe->f_scale; r.p[2] = fe->f_amt;
        } else if (auto uo = dynamic_cast<const uv_offset_texture*>(t)) {
            r.a = texture_id(uo->base_); r.scale = uo->du; r.p[0] = uo->dv;
        } else {
            fail("texture kind not supported by the render kernels");
        }
        int id = (int)out.textures.size();
        out.textures.push_back(r);
        tex_ids[t] = id;
        return id;
    }

    // solid colours are folded into the material record (tex = -1)
    void bind_texture(rt_material& r, const texture* t) {
        if (!t) return;
        if (auto s = dynamic_cast<const solid_color*>(t)) { put3(r.albedo, s->albedo); r.tex = -1; }
        else r.tex = texture_id(t);
    }

    int material_id(const material* m) {
        auto it = mat_ids.find(m);
        if (it != mat_ids.end()) return it->second;
        rt_material r; memset(&r, 0, sizeof(r));
        r.kind = m->mat_kind(); r.tex = -1;
        if (auto l = dynamic_cast<const lambertian*>(m)) {
            r.albedo[0] = r.albedo[1] = r.albedo[2] = 1.f;   // tex == nullptr -> vec3(1,1,1), material.cuh:84
            bind_texture(r, l->tex);
        } else if (auto me = dynamic_cast<const metal*>(m)) {
            put3(r.albedo, me->albedo); r.fuzz = me->fuzz;
        } else if (auto d = dynamic_cast<const dielectric*>(m)) {
            r.ior = d->ref_idx;
        } else if (auto dl = dynamic_cast<const diffuse_light*>(m)) {
            put3(r.albedo, dl->solid);
            bind_texture(r, dl->tex);
        } else if (auto iso = dynamic_cast<const isotropic*>(m)) {
            bind_texture(r, iso->tex);
        } else {
            fail("material kind not supported by the render kernels");
        }
        int id = (int)out.materials.size();
        out.materials.push_back(r);
        mat_ids[m] = id;
        return id;
    }

    int32_t push_quad(const quad* q, const material* override_mat) {
        rt_quad r; memset(&r, 0, sizeof(r));
        put3(r.Q, q->Q); r.D = q->D; put3(r.u, q->u); put3(r.v, q->v); put3(r.w, q->w); put3(r.n, q->normal);
        r.mat = material_id(override_mat ? override_mat : q->mat_ptr);
        out.quads.push_back(r);
        return (int32_t)out.quads.size() - 1;
    }

    // sphere / quad / compound6, optionally with the material replaced (with_material)
    int32_t simple_ref(const hittable* h, const material* override_mat) {
        if (!override_mat) { auto it = prim_refs.find(h); if (it != prim_refs.end()) return it->second; }
        int32_t ref = -1;
        if (auto s = dynamic_cast<const sphere*>(h)) {
            rt_sphere r; memset(&r, 0, sizeof(r));
            put3(r.c0, s->center0); r.radius = s->radius; put3(r.vel, s->velocity);
            r.mat = material_id(override_mat ? override_mat : s->mat_ptr);
            out.spheres.push_back(r);
            ref = RT_PRIM_REF(RT_PRIM_SPHERE, out.spheres.size() - 1);
        } else if (auto q = dynamic_cast<const quad*>(h)) {
            ref = RT_PRIM_REF(RT_PRIM_QUAD, push_quad(q, override_mat));
        } else if (auto c = dynamic_cast<const compound6*>(h)) {
            rt_box b; b.first_quad = (int32_t)out.quads.size();
            for (int i = 0; i < 6; ++i) {
                auto fq = dynamic_cast<const quad*>(c->faces[i]);
                if (!fq) { fail("compound6 face is not a quad"); return -1; }
                push_quad(fq, override_mat);
            }
            out.boxes.push_back(b);
            ref = RT_PRIM_REF(RT_PRIM_BOX, out.boxes.size() - 1);
        } else if (auto wm = dynamic_cast<const with_material*>(h)) {
            return simple_ref(wm->obj, override_mat ? override_mat : wm->mat);   // outermost override wins (hittable.cuh:170)
        } else {
            return -1;
        }
        if (!override_mat) prim_refs[h] = ref;
        return ref;
    }

    // translate(rotate_y(x)), translate(x), rotate_y(x) with x simple
    int32_t instance_ref(const hittable* h) {
        auto it = prim_refs.find(h);
        if (it != prim_refs.end()) return it->second;
        rt_instance r; memset(&r, 0, sizeof(r));
        r.cos_t = 1.f;
        const hittable* inner = h;
        if (auto t = dynamic_cast<const translate*>(inner)) {
            r.flags |= RT_INST_TRANSLATE; put3(r.offset, t->offset); inner = t->obj;
        }
        if (auto ro = dynamic_cast<const rotate_y*>(inner)) {
            r.flags |= RT_INST_ROTATE_Y; r.sin_t = ro->sin_t; r.cos_t = ro->cos_t; inner = ro->obj;
        }
        if (!r.flags) return -1;
        r.child = simple_ref(inner, nullptr);
        if (r.child < 0) { fail("instance child must be a sphere, quad or box (deeper wrapper chains are not implemented)"); return -1; }
        out.instances.push_back(r);
        int32_t ref = RT_PRIM_REF(RT_PRIM_INSTANCE, out.instances.size() - 1);
        prim_refs[h] = ref;
        return ref;
    }

    int32_t leaf_ref(const hittable* h) {
        int32_t ref = simple_ref(h, nullptr);
        if (ref >= 0 || !ok) return ref;
        ref = instance_ref(h);
        if (ref >= 0 || !ok) return ref;
        if (auto m = dynamic_cast<const constant_medium*>(h)) {
            auto it = prim_refs.find(h);
            if (it != prim_refs.end()) return it->second;
            rt_medium r; memset(&r, 0, sizeof(r));
            r.boundary = simple_ref(m->boundary, nullptr);
            if (r.boundary < 0 && ok) r.boundary = instance_ref(m->boundary);
            if (r.boundary < 0) { fail("constant_medium boundary must be a sphere, quad, box or an instance of one"); return -1; }
            r.neg_inv_density = m->neg_inv_density;
            r.mat = material_id(m->phase_function);
            out.media.push_back(r);
            ref = RT_PRIM_REF(RT_PRIM_MEDIUM, out.media.size() - 1);
            prim_refs[h] = ref;
            return ref;
        }
        fail("object kind not supported as a BVH leaf");
        return -1;
    }

    // depth-first pre-order emission with skip links
    void emit(const hittable* h) {
        if (!ok) return;
        auto node = dynamic_cast<const bvh_node*>(h);
        if (!node) { fail("BVH child is not a bvh_node (build the world with bvh_node(list, 0, n))"); return; }
        const int idx = (int)out.nodes.size();
        rt_node r; memset(&r, 0, sizeof(r));
        const aabb b = node->bounding_box();
        put3(r.bmin, b.minimum); put3(r.bmax, b.maximum);
        r.prim = -1; r.skip = idx + 1;
        out.nodes.push_back(r);
        out.leaf_order.push_back(-1);
        if (node->left == node->right) {
            if (node->left) {   // the reference's single-object node: object tested (twice) after the box
                int32_t ref = leaf_ref(node->left);
                if (!ok) return;
                out.nodes[idx].prim = ref;
                auto o = order.find(node->left);
                out.leaf_order[idx] = o == order.end() ? -1 : o->second;
            }
            return;   // empty node (n <= 0): box is empty, nothing below
        }
        emit(node->left);
        emit(node->right);
        out.nodes[idx].skip = (int32_t)out.nodes.size();
    }
};

}  // namespace

rt_status flatten(const hittable* world, const camera& cam, flat_scene& out, std::string& err,
                  hittable* const* creation_order, int count) {
    out = flat_scene();
    flattener f(out, err);
    for (int i = 0; i < count; ++i) f.order[creation_order[i]] = i;
    if (!world) { err = "null world"; return RT_ERR_INVALID; }
    f.emit(world);
    if (!f.ok) return RT_ERR_UNSUPPORTED;
    rt_camera& c = out.camera;
    memset(&c, 0, sizeof(c));
    put3(c.origin, cam.origin); put3(c.lower_left_corner, cam.lower_left_corner);
    put3(c.horizontal, cam.horizontal); put3(c.vertical, cam.vertical);
    put3(c.u, cam.u); put3(c.v, cam.v);
    c.lens_radius = cam.lens_radius; c.time0 = cam.time0; c.time1 = cam.time1;
    return RT_OK;
}

}  // namespace rtw
