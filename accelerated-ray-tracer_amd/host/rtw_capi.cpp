// rtw_capi.cpp -- C entry points of the host scene library (librtw_host.so) for
// tests, bench.py and any non-C++ caller: build a named reference scene on the
// host, flatten it, and hand out the rt_scene_desc the render library takes.
// Pure host code: no HIP here.
#include <cstdio>
#include <memory>
#include <string>

#include "rtw_scenes.h"

namespace {
struct handle {
    std::unique_ptr<rtw::built_scene> scene;
    rtw::flat_scene flat;
    rt_scene_desc desc;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

const char* rtw_last_error(void) { return g_err.c_str(); }

int rtw_scene_count(void) { int n = 0; rtw::scene_names(&n); return n; }
const char* rtw_scene_name(int i) { int n = 0; const char* const* v = rtw::scene_names(&n); return (i >= 0 && i < n) ? v[i] : nullptr; }

// nx/ny <= 0: the reference host function's frame size.  rgb may be null.
void* rtw_scene_build(const char* name, int nx, int ny, const unsigned char* rgb, int w, int h) {
    g_err.clear();
    std::unique_ptr<handle> hd(new handle);
    hd->scene = rtw::build_scene(name ? name : "", nx, ny, rgb, w, h, g_err);
    if (!hd->scene) return nullptr;
    rt_status st = rtw::flatten(hd->scene->world, *hd->scene->cam, hd->flat, g_err,
                                hd->scene->created.data(), (int)hd->scene->created.size());
    if (st != RT_OK) return nullptr;
    hd->desc = hd->flat.desc();
    return hd.release();
}

void rtw_scene_free(void* p) { delete static_cast<handle*>(p); }

const rt_scene_desc* rtw_scene_desc(void* p) { return p ? &static_cast<handle*>(p)->desc : nullptr; }

// out[0..3] = nx, ny, ns, use_gradient_bg ; bg[0..2] ; returns gamma
float rtw_scene_defaults(void* p, int* out4, float* bg3, int* ppm_double_scale) {
    const rtw::built_scene& s = *static_cast<handle*>(p)->scene;
    out4[0] = s.nx; out4[1] = s.ny; out4[2] = s.ns; out4[3] = s.use_gradient_bg;
    bg3[0] = s.background.x(); bg3[1] = s.background.y(); bg3[2] = s.background.z();
    if (ppm_double_scale) *ppm_double_scale = s.ppm_double_scale ? 1 : 0;
    return s.gamma;
}

// per BVH node (depth-first order): creation index of the leaf's object, or -1
int rtw_scene_leaf_order(void* p, int* out, int cap) {
    const rtw::flat_scene& f = static_cast<handle*>(p)->flat;
    int n = (int)f.leaf_order.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = f.leaf_order[i];
    return n;
}

// the reference's output stage: ASCII P3 to `path` ("-" = stdout)
// flags: bit 0 = the double 255.99 of bouncing_spheres (main.cu:725), bit 1 = binary P6 instead of ASCII P3
int rtw_write_ppm(const char* path, const float* fb, int nx, int ny, int flags) {
    FILE* f = (path && std::string(path) != "-") ? fopen(path, "wb") : stdout;
    if (!f) return -1;
    if (flags & 2) rtw::write_ppm_p6(f, fb, nx, ny, (flags & 1) != 0);
    else rtw::write_ppm_p3(f, fb, nx, ny, (flags & 1) != 0);
    if (f != stdout) fclose(f); else fflush(f);
    return 0;
}

int rtw_load_ppm(const char* path, unsigned char* out, int cap, int* w, int* h) {
    std::vector<unsigned char> px;
    if (!rtw::load_ppm(path, px, *w, *h)) return -1;
    if (out && (int)px.size() <= cap) memcpy(out, px.data(), px.size());
    return (int)px.size();
}

}  // extern "C"
