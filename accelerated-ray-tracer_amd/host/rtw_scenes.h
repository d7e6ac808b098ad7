// rtw_scenes.h -- the reference's scene functions as host-side builders.
//
// Each reference scene is a pair: a single-thread create_world_* kernel
// (src/main.cu:160-635) and a host function that fixes nx, ny, ns, gamma,
// background and the gradient flag and launches render (src/main.cu:654-1305).
// A built_scene carries both halves.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "rtw.h"

namespace rtw {

struct built_scene {
    arena mem;                       // owns every object below
    std::vector<hittable*> created;  // the reference's d_list in creation order
    hittable* world = nullptr;       // bvh_node root
    camera* cam = nullptr;
    // what the reference's host function passes to render<<<>>>
    int nx = 0, ny = 0, ns = 0;
    float gamma = 2.2f;
    vec3 background;
    int use_gradient_bg = 0;
    bool ppm_double_scale = false;   // bouncing_spheres() scales by double 255.99 (main.cu:722), the rest by 255.99f
    std::vector<unsigned char> image_pixels;   // texture pixels kept alive for image_texture
    int image_w = 0, image_h = 0;
};

// Names: two_spheres, bouncing (alias random_scene), book1, checker, earth,
// perlin, quads, cornell, cornell_smoke, final, simple_light, original.  nx/ny <= 0 pick the reference
// host function's size.  `rgb` (optional, RGB8 w*h*3) feeds image textures
// (earth, final); without it they render the reference's invalid-image colour.
std::unique_ptr<built_scene> build_scene(const std::string& name, int nx, int ny,
                                         const unsigned char* rgb, int w, int h, std::string& err);

const char* const* scene_names(int* count);

// Binary PPM (P6) / ASCII PPM (P3) loader for texture pixels; returns false on failure.
bool load_ppm(const std::string& path, std::vector<unsigned char>& rgb, int& w, int& h);

// The reference's output stage (main.cu:715-727): ASCII P3, rows ny-1..0, int(255.99*c), no clamp.
void write_ppm_p3(FILE* f, const float* fb, int nx, int ny, bool double_scale);
void write_ppm_p6(FILE* f, const float* fb, int nx, int ny, bool double_scale);   // binary, clamped to 0..255

}  // namespace rtw
