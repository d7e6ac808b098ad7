// rtw_scenes.cpp -- scene builders (host side).  See rtw_scenes.h.
// World-builder random draws use the cuRAND-compatible XORWOW seeded with 1984
// (rand_init, src/main.cu:89-94).  Where the reference leaves the order of two
// draws in one expression unspecified they are taken left to right, in
// separate statements.
#include "rtw_scenes.h"

#include <cstdio>

#include "../csrc/rt_xorwow.h"

namespace rtw {
namespace {

struct world_rng {
    rt_xorwow s;
    world_rng() { rt_xorwow_seed(s, 1984ull); }
    float operator()() { return rt_xorwow_uniform(s); }
};

void finish(built_scene& sc, std::vector<hittable*>& list, camera* cam) {
    sc.created = list;                                  // creation order, before the builder sorts in place
    sc.world = new bvh_node(list.data(), 0, (int)list.size());
    sc.cam = cam;
}

float aspect_of(int nx, int ny) { return float(nx) / float(ny); }

// ---- BASELINE config 1; not a reference scene ----
void two_spheres(built_scene& sc) {
    std::vector<hittable*> objs;
    objs.push_back(new sphere(vec3(0, 0, -1), 0.5f, new lambertian(vec3(0.5f, 0.5f, 0.5f))));
    objs.push_back(new sphere(vec3(0, -100.5f, -1), 100.f, new lambertian(vec3(0.5f, 0.5f, 0.5f))));
    finish(sc, objs, new camera(vec3(0, 0, 0), vec3(0, 0, -1), vec3(0, 1, 0), 90.f, 2.0f, 0.0f, 1.0f));
    sc.use_gradient_bg = 1;
}

// ---- test scene (not in the reference): vfov 0 makes every primary ray exactly (0,0,-f), so 1/d is infinite on
// two axes, and the spheres are placed so that box faces pass through the ray origin's x and y.
void degenerate_axes(built_scene& sc) {
    std::vector<hittable*> objs;
    objs.push_back(new sphere(vec3(0.5f, 0.0f, 0.0f), 0.5f, new lambertian(vec3(0.8f, 0.3f, 0.3f))));
    objs.push_back(new sphere(vec3(-0.5f, 0.5f, 1.0f), 0.5f, new metal(vec3(0.8f, 0.8f, 0.8f), 0.3f)));
    objs.push_back(new sphere(vec3(0.0f, -0.5f, 2.0f), 0.5f, new dielectric(1.5f)));
    objs.push_back(new sphere(vec3(0.0f, 0.0f, -1.0f), 1.0f, new lambertian(vec3(0.3f, 0.8f, 0.3f))));
    objs.push_back(new sphere(vec3(0.0f, -101.0f, 0.0f), 100.0f, new lambertian(vec3(0.5f, 0.5f, 0.5f))));
    finish(sc, objs, new camera(vec3(0, 0, 5), vec3(0, 0, 0), vec3(0, 1, 0), 0.0f, aspect_of(sc.nx, sc.ny), 0.0f, 5.0f));
    sc.use_gradient_bg = 1;
}

vec3 ut_palette(float r) {   // main.cu:149-158
    if (r < 0.25f) return vec3(1.0f, 1.0f, 1.0f);
    if (r < 0.50f) return vec3(1.0f, 0.51f, 0.0f);
    if (r < 0.75f) return vec3(0.60f, 0.60f, 0.60f);
    return vec3(0.0f, 0.0f, 0.0f);
}

// ---- main.cu:160-244 + 654-744 ----
void bouncing_spheres(built_scene& sc) {
    world_rng rnd;
    std::vector<hittable*> objs;
    const vec3 orange(1.0f, 0.51f, 0.0f);
    texture* ground_tex = new checker_texture(0.64f, new solid_color(vec3(1.0f, 1.0f, 1.0f)), new solid_color(orange));
    objs.push_back(new sphere(vec3(0.0f, -1000.0f, -1.0f), 1000.0f, new lambertian(ground_tex)));

    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const float pick = rnd();
            const float px = fmaf(0.9f, rnd(), (float)a);   // a + 0.9f*RND, contracted
            const float pz = fmaf(0.9f, rnd(), (float)b);
            const vec3 at(px, 0.2f, pz);
            if (pick < 0.8f) {
                const float vy = 0.5f * rnd();
                const float vz = 0.25f * (rnd() - 0.5f);
                const vec3 at_close = at + vec3(0.0f, vy, vz);
                if (rnd() < 0.10f) {
                    objs.push_back(new sphere(at, at_close, 0.2f, new diffuse_light(4.0f * orange)));
                } else {
                    const vec3 tint = ut_palette(rnd());
                    objs.push_back(new sphere(at, at_close, 0.2f, new lambertian(tint)));
                }
            } else if (pick < 0.95f) {
                vec3 tint = ut_palette(rnd());
                if (tint.x() + tint.y() + tint.z() < 1e-5f) tint = vec3(0.15f, 0.15f, 0.15f);
                const float fuzz = 0.5f * rnd();
                objs.push_back(new sphere(at, 0.2f, new metal(tint, fuzz)));
            } else {
                objs.push_back(new sphere(at, 0.2f, new dielectric(1.5f)));
            }
        }
    }
    objs.push_back(new sphere(vec3(0.0f, 1.0f, 0.0f), 1.0f, new dielectric(1.5f)));
    objs.push_back(new sphere(vec3(-4.0f, 1.0f, 0.0f), 1.0f, new lambertian(vec3(0.4f, 0.2f, 0.1f))));
    objs.push_back(new sphere(vec3(4.0f, 1.0f, 0.0f), 1.0f, new metal(vec3(0.7f, 0.6f, 0.5f), 0.0f)));

    const vec3 eye(13.0f, 2.0f, 3.0f), target(0.0f, 0.0f, 0.0f);
    finish(sc, objs, new camera(eye, target, vec3(0.0f, 1.0f, 0.0f), 30.0f, aspect_of(sc.nx, sc.ny), 0.1f,
                                (eye - target).length_folded(), 0.0, 1.0));
    sc.ppm_double_scale = true;
}

// ---- Book-1 random_scene with the book's materials ----
void book1_random_scene(built_scene& sc) {
    world_rng rnd;
    std::vector<hittable*> objs;
    objs.push_back(new sphere(vec3(0.0f, -1000.0f, -1.0f), 1000.0f, new lambertian(vec3(0.5f, 0.5f, 0.5f))));
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const float pick = rnd();
            const float px = a + rnd();
            const float pz = b + rnd();
            const vec3 at(px, 0.2f, pz);
            if (pick < 0.8f) {
                float c[3];
                for (int k = 0; k < 3; ++k) { const float p = rnd(); const float q = rnd(); c[k] = p * q; }
                objs.push_back(new sphere(at, 0.2f, new lambertian(vec3(c[0], c[1], c[2]))));
            } else if (pick < 0.95f) {
                float c[3];
                for (int k = 0; k < 3; ++k) c[k] = 0.5f * (1.0f + rnd());
                const float fuzz = 0.5f * rnd();
                objs.push_back(new sphere(at, 0.2f, new metal(vec3(c[0], c[1], c[2]), fuzz)));
            } else {
                objs.push_back(new sphere(at, 0.2f, new dielectric(1.5f)));
            }
        }
    }
    objs.push_back(new sphere(vec3(0.0f, 1.0f, 0.0f), 1.0f, new dielectric(1.5f)));
    objs.push_back(new sphere(vec3(-4.0f, 1.0f, 0.0f), 1.0f, new lambertian(vec3(0.4f, 0.2f, 0.1f))));
    objs.push_back(new sphere(vec3(4.0f, 1.0f, 0.0f), 1.0f, new metal(vec3(0.7f, 0.6f, 0.5f), 0.0f)));
    finish(sc, objs, new camera(vec3(13.0f, 2.0f, 3.0f), vec3(0, 0, 0), vec3(0, 1, 0), 20.0f, aspect_of(sc.nx, sc.ny), 0.1f, 10.0f));
    sc.use_gradient_bg = 1;
}

// ---- main.cu:246-280 + 746-800 ----
void checkered_spheres(built_scene& sc) {
    std::vector<hittable*> objs;
    texture* chk = new checker_texture(0.32f, new solid_color(vec3(0.2f, 0.3f, 0.1f)), new solid_color(vec3(0.9f, 0.9f, 0.9f)));
    material* lam = new lambertian(chk);
    objs.push_back(new sphere(vec3(0, -10, 0), 10.0f, lam));
    objs.push_back(new sphere(vec3(0, 10, 0), 10.0f, lam));
    finish(sc, objs, new camera(vec3(13.0f, 2.0f, 3.0f), vec3(0, 0, 0), vec3(0, 1, 0), 20.0f, aspect_of(sc.nx, sc.ny), 0.0f, 10.0f, 0.0, 1.0));
    sc.use_gradient_bg = 1;
}

DeviceImage image_view(const built_scene& sc) {
    DeviceImage d;
    if (!sc.image_pixels.empty()) { d.data = sc.image_pixels.data(); d.width = sc.image_w; d.height = sc.image_h; d.bpp = 3; }
    return d;
}

// ---- main.cu:282-308 + 802-880 ----
void earth(built_scene& sc) {
    std::vector<hittable*> objs;
    objs.push_back(new sphere(vec3(0, 0, 0), 2.0f, new lambertian(new image_texture(image_view(sc)))));
    finish(sc, objs, new camera(vec3(0.0f, 0.0f, 12.0f), vec3(0, 0, 0), vec3(0, 1, 0), 20.0f, aspect_of(sc.nx, sc.ny), 0.0f, 12.0f, 0.0, 1.0));
    sc.use_gradient_bg = 1;
}

// ---- main.cu:310-329 + 882-937 (scale 4.0 at main.cu:903) ----
void perlin_spheres(built_scene& sc) {
    std::vector<hittable*> objs;
    material* lam = new lambertian(new noise_texture(4.0f));
    objs.push_back(new sphere(vec3(0, -1000, 0), 1000.f, lam));
    objs.push_back(new sphere(vec3(0, 2, 0), 2.f, lam));
    finish(sc, objs, new camera(vec3(13, 2, 3), vec3(0, 0, 0), vec3(0, 1, 0), 20.0f, aspect_of(sc.nx, sc.ny), 0.0f, 10.0f, 0.0, 1.0));
    sc.use_gradient_bg = 1;
}

// ---- main.cu:331-358 + 939-993 ----
void quads_scene(built_scene& sc) {
    std::vector<hittable*> objs;
    objs.push_back(new quad(vec3(-3, -2, 5), vec3(0, 0, -4), vec3(0, 4, 0), new lambertian(vec3(1.0f, 0.2f, 0.2f))));
    objs.push_back(new quad(vec3(-2, -2, 0), vec3(4, 0, 0), vec3(0, 4, 0), new lambertian(vec3(0.2f, 1.0f, 0.2f))));
    objs.push_back(new quad(vec3(3, -2, 1), vec3(0, 0, 4), vec3(0, 4, 0), new lambertian(vec3(0.2f, 0.2f, 1.0f))));
    objs.push_back(new quad(vec3(-2, 3, 1), vec3(4, 0, 0), vec3(0, 0, 4), new lambertian(vec3(1.0f, 0.5f, 0.0f))));
    objs.push_back(new quad(vec3(-2, -3, 5), vec3(4, 0, 0), vec3(0, 0, -4), new lambertian(vec3(0.2f, 0.8f, 0.8f))));
    finish(sc, objs, new camera(vec3(0, 0, 9), vec3(0, 0, 0), vec3(0, 1, 0), 80.0f, aspect_of(sc.nx, sc.ny), 0.0f, 10.0f, 0.0, 1.0));
    sc.use_gradient_bg = 1;
}

camera* cornell_camera(const built_scene& sc) {
    const vec3 eye(278, 278, -800), target(278, 278, 0);
    return new camera(eye, target, vec3(0, 1, 0), 40.0f, aspect_of(sc.nx, sc.ny), 0.0f, (eye - target).length_folded(), 0.0, 1.0);
}

// ---- main.cu:402-450 + 1072-1127 ----
void cornell_box(built_scene& sc) {
    std::vector<hittable*> objs;
    material* red = new lambertian(vec3(.65f, .05f, .05f));
    material* blue = new lambertian(vec3(.15f, .15f, .75f));
    material* white = new lambertian(vec3(.73f, .73f, .73f));
    material* lamp = new diffuse_light(vec3(15.f, 15.f, 15.f));
    objs.push_back(new quad(vec3(0, 0, 0), vec3(0, 555, 0), vec3(0, 0, 555), blue, true));
    objs.push_back(new quad(vec3(555, 0, 555), vec3(0, 555, 0), vec3(0, 0, -555), red, true));
    objs.push_back(new quad(vec3(0, 0, 0), vec3(555, 0, 0), vec3(0, 0, 555), white, true));
    objs.push_back(new quad(vec3(0, 555, 555), vec3(555, 0, 0), vec3(0, 0, -555), white, true));
    objs.push_back(new quad(vec3(555, 0, 555), vec3(-555, 0, 0), vec3(0, 555, 0), white, true));
    objs.push_back(new quad(vec3(213, 554, 227), vec3(130, 0, 0), vec3(0, 0, 105), lamp, true));
    hittable* cube = make_box(vec3(0, 0, 0), vec3(165, 165, 165), white);
    hittable* tower = make_box(vec3(0, 0, 0), vec3(165, 330, 165), white);
    objs.push_back(new translate(new rotate_y(cube, -18.f), vec3(130.f, 0.f, 65.f)));
    objs.push_back(new translate(new rotate_y(tower, 15.f), vec3(265.f, 0.f, 295.f)));
    material* glass = new dielectric(1.5f);
    objs.push_back(new sphere(vec3(278.f, 335.f, 150.f), 60.f, glass));
    objs.push_back(new sphere(vec3(278.f, 335.f, 150.f), -59.0f, glass));   // hollow bubble
    finish(sc, objs, cornell_camera(sc));
}

// ---- main.cu:452-486 + 1129-1176 ----
void cornell_smoke(built_scene& sc) {
    std::vector<hittable*> objs;
    material* red = new lambertian(vec3(.65f, .05f, .05f));
    material* white = new lambertian(vec3(.73f, .73f, .73f));
    material* green = new lambertian(vec3(.12f, .45f, .15f));
    material* lamp = new diffuse_light(vec3(7.f, 7.f, 7.f));
    objs.push_back(new quad(vec3(555, 0, 0), vec3(0, 555, 0), vec3(0, 0, 555), green, true));
    objs.push_back(new quad(vec3(0, 0, 0), vec3(0, 555, 0), vec3(0, 0, 555), red, true));
    objs.push_back(new quad(vec3(0, 555, 0), vec3(555, 0, 0), vec3(0, 0, 555), white, true));
    objs.push_back(new quad(vec3(0, 0, 0), vec3(555, 0, 0), vec3(0, 0, 555), white, true));
    objs.push_back(new quad(vec3(0, 0, 555), vec3(555, 0, 0), vec3(0, 555, 0), white, true));
    objs.push_back(new quad(vec3(113, 554, 127), vec3(330, 0, 0), vec3(0, 0, 305), lamp, true));
    hittable* tower = make_box(vec3(0, 0, 0), vec3(165, 330, 165), white);
    tower = new translate(new rotate_y(tower, 15.f), vec3(265.f, 0.f, 295.f));
    hittable* cube = make_box(vec3(0, 0, 0), vec3(165, 165, 165), white);
    cube = new translate(new rotate_y(cube, -18.f), vec3(130.f, 0.f, 65.f));
    objs.push_back(new constant_medium(tower, 0.01f, vec3(0.5f, 0.5f, 0.5f)));
    objs.push_back(new constant_medium(cube, 0.01f, vec3(1, 1, 1)));
    finish(sc, objs, cornell_camera(sc));
}

// ---- main.cu:498-562 + 1178-1237 ----
void final_scene(built_scene& sc) {
    std::vector<hittable*> objs;
    material* white = new lambertian(vec3(.73f, .73f, .73f));
    material* grass = new lambertian(vec3(0.48f, 0.83f, 0.53f));
    material* lamp = new diffuse_light(vec3(7, 7, 7));
    for (int ix = 0; ix < 20; ++ix) {
        for (int iz = 0; iz < 20; ++iz) {
            const float side = 100.0f;
            const float x0 = -1000.0f + ix * side;
            const float z0 = -1000.0f + iz * side;
            const float top = 1.0f + 100.0f * ((ix * 13 + iz * 37) % 100) / 100.0f;
            objs.push_back(make_box(vec3(x0, 0, z0), vec3(x0 + side, top, z0 + side), grass));
        }
    }
    objs.push_back(new quad(vec3(123, 554, 147), vec3(300, 0, 0), vec3(0, 0, 265), lamp, true));
    const vec3 from(400, 400, 200);
    objs.push_back(new sphere(from, from + vec3(30, 0, 0), 50.f, new lambertian(vec3(0.7f, 0.3f, 0.1f))));
    objs.push_back(new sphere(vec3(260, 150, 45), 50.f, new dielectric(1.5f)));
    objs.push_back(new sphere(vec3(0, 150, 145), 50.f, new metal(vec3(0.8f, 0.8f, 0.9f), 1.0f)));
    objs.push_back(new sphere(vec3(360, 150, 145), 70.f, new dielectric(1.5f)));
    objs.push_back(new constant_medium(new sphere(vec3(360, 150, 145), 70.f, new dielectric(1.5f)), 0.2f, vec3(0.2f, 0.4f, 0.9f)));
    objs.push_back(new constant_medium(new sphere(vec3(0, 0, 0), 5000.f, new dielectric(1.5f)), 0.0001f, vec3(1, 1, 1)));
    objs.push_back(new sphere(vec3(400, 200, 400), 100.f, new lambertian(new image_texture(image_view(sc)))));
    objs.push_back(new sphere(vec3(220, 280, 300), 80.f, new lambertian(new noise_texture(0.2f))));
    const float turn = 15.0f * 0.017453292519943295f;   // deg2rad, main.cu:489
    const float ct = cosf(turn), st = sinf(turn);
    for (int j = 0; j < 1000; ++j) {
        const vec3 q = random_in_unit_cube(j) * 165.0f;
        const vec3 turned(fmaf(ct, q.x(), st * q.z()), q.y(), fmaf(ct, q.z(), -(st * q.x())));   // rotate_y_deg, main.cu:491-496
        objs.push_back(new sphere(turned + vec3(-100, 270, 395), 10.0f, white));
    }
    const vec3 eye(478, 278, -600), target(278, 278, 0);
    finish(sc, objs, new camera(eye, target, vec3(0, 1, 0), 40.0f, aspect_of(sc.nx, sc.ny), 0.0f, (eye - target).length_folded(), 0.0, 1.0));
}

// ---- main.cu:360-400 + 995-1070 ----
void simple_light(built_scene& sc) {
    std::vector<hittable*> objs;
    objs.push_back(new sphere(vec3(0, -1000, 0), 1000.f, new lambertian(new felt_texture(vec3(0.06f, 0.36f, 0.18f), 16.0f, 0.08f, 4.0f, 0.03f))));
    texture* decal = new uv_offset_texture(new image_texture(image_view(sc)), 60.0f / 360.0f);
    const vec3 centre(0, 2, 0);
    const float radius = 2.0f;
    objs.push_back(new sphere(centre, radius, new lambertian(decal)));
    objs.push_back(new sphere(centre, radius + 0.02f, new dielectric(1.5f)));          // clear coat
    objs.push_back(new sphere(vec3(0, 7, 0), 2.f, new diffuse_light(vec3(4, 4, 4))));
    objs.push_back(new quad(vec3(3, 1, -2), vec3(2, 0, 0), vec3(0, 2, 0), new diffuse_light(vec3(4, 4, 4))));
    const vec3 eye(26, 3, 6), target(0, 2, 0);
    finish(sc, objs, new camera(eye, target, vec3(0, 1, 0), 20.0f, aspect_of(sc.nx, sc.ny), 0.0f, (eye - target).length_folded(), 0.0, 1.0));
}

// ---- main.cu:564-635 + 1239-1305: the scene the reference's main() actually renders (case 10) ----
void original_scene(built_scene& sc) {
    std::vector<hittable*> objs;
    material* white = new lambertian(vec3(.73f, .73f, .73f));
    material* pink = new lambertian(vec3(0.88f, 0.50f, 0.76f));
    material* lamp = new diffuse_light(vec3(7, 7, 7));
    for (int ix = 0; ix < 20; ++ix) {
        for (int iz = 0; iz < 20; ++iz) {
            const float side = 100.0f;
            const float x0 = -1000.0f + ix * side;
            const float z0 = -1000.0f + iz * side;
            const float top = 1.0f + 100.0f * ((ix * 13 + iz * 37) % 100) / 100.0f;
            objs.push_back(make_box(vec3(x0, 0, z0), vec3(x0 + side, top, z0 + side), pink));
        }
    }
    objs.push_back(new quad(vec3(123, 554, 147), vec3(300, 0, 0), vec3(0, 0, 265), lamp, true));
    const vec3 from(400, 400, 200);
    objs.push_back(new sphere(from, from + vec3(30, 0, 0), 50.f, new lambertian(vec3(0.0488f, 0.0148f, 0.0171f))));
    objs.push_back(new sphere(vec3(260, 150, 45), 50.f, new dielectric(1.5f)));
    objs.push_back(new sphere(vec3(0, 150, 145), 50.f, new metal(vec3(0.6387f, 0.3605f, 0.8826f), 1.0f)));
    objs.push_back(new sphere(vec3(360.f, 150.f, 145.f), 70.f, new lambertian(new image_texture(image_view(sc)))));   // 8-ball
    objs.push_back(new sphere(vec3(360, 150, 145), 70.f + 0.5f, new dielectric(1.5f)));
    objs.push_back(new constant_medium(new sphere(vec3(0, 0, 0), 5000.f, new dielectric(1.5f)), 0.0001f, vec3(1, 1, 1)));
    objs.push_back(new sphere(vec3(400, 200, 400), 100.f, new metal(vec3(0.23f, 0.24f, 0.85f), 0.02f)));
    objs.push_back(new sphere(vec3(220, 280, 300), 80.f, new lambertian(new noodle_texture(0.2f))));
    const float turn = 15.0f * 0.017453292519943295f;
    const float ct = cosf(turn), st = sinf(turn);
    for (int j = 0; j < 1000; ++j) {
        const vec3 q = random_in_unit_cube(j) * 165.0f;
        const vec3 turned(fmaf(ct, q.x(), st * q.z()), q.y(), fmaf(ct, q.z(), -(st * q.x())));
        objs.push_back(new sphere(turned + vec3(-100, 270, 395), 10.0f, white));
    }
    const vec3 eye(478, 278, -600), target(278, 278, 0);
    finish(sc, objs, new camera(eye, target, vec3(0, 1, 0), 40.0f, aspect_of(sc.nx, sc.ny), 0.0f, (eye - target).length_folded(), 0.0, 1.0));
    sc.background = vec3(0.043f, 0.030f, 0.094f);      // main.cu:1276
}

struct entry {
    const char* name;
    void (*build)(built_scene&);
    int nx, ny, ns;   // the reference host function's frame
};
const entry k_scenes[] = {
    {"two_spheres", two_spheres, 200, 100, 1},
    {"degenerate", degenerate_axes, 32, 16, 8},
    {"bouncing", bouncing_spheres, 1200, 600, 10000},
    {"random_scene", bouncing_spheres, 1200, 800, 500},   // the headline frame on the reference's random scene
    {"book1", book1_random_scene, 1200, 800, 100},
    {"checker", checkered_spheres, 1200, 600, 500},
    {"earth", earth, 1200, 600, 500},
    {"perlin", perlin_spheres, 1200, 600, 500},
    {"quads", quads_scene, 1200, 600, 500},
    {"cornell", cornell_box, 600, 600, 10000},
    {"cornell_smoke", cornell_smoke, 600, 600, 1000},
    {"final", final_scene, 800, 800, 10000},
    {"simple_light", simple_light, 1200, 600, 10000},
    {"original", original_scene, 800, 800, 10000},
};
const int k_num_scenes = (int)(sizeof(k_scenes) / sizeof(k_scenes[0]));

}  // namespace

const char* const* scene_names(int* count) {
    static const char* names[k_num_scenes];
    for (int i = 0; i < k_num_scenes; ++i) names[i] = k_scenes[i].name;
    if (count) *count = k_num_scenes;
    return names;
}

std::unique_ptr<built_scene> build_scene(const std::string& name, int nx, int ny, const unsigned char* rgb, int w, int h,
                                         std::string& err) {
    for (const entry& e : k_scenes) {
        if (name != e.name) continue;
        std::unique_ptr<built_scene> sc(new built_scene);
        sc->nx = nx > 0 ? nx : e.nx;
        sc->ny = ny > 0 ? ny : e.ny;
        sc->ns = e.ns;
        if (rgb && w > 0 && h > 0) { sc->image_pixels.assign(rgb, rgb + (size_t)w * h * 3); sc->image_w = w; sc->image_h = h; }
        arena::scope guard(sc->mem);
        e.build(*sc);
        return sc;
    }
    err = "unknown scene '" + name + "'";
    return nullptr;
}

// ------------------------------------------------------------------ PPM in/out
static bool ppm_token(FILE* f, int& v) {
    int c = fgetc(f);
    for (;;) {
        while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        break;
    }
    if (c < '0' || c > '9') return false;
    v = 0;
    while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = fgetc(f); }
    return true;   // the single whitespace after the token has been consumed
}

bool load_ppm(const std::string& path, std::vector<unsigned char>& rgb, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    int c0 = fgetc(f), c1 = fgetc(f), maxv = 0;
    if (c0 == 'P' && (c1 == '6' || c1 == '3') && ppm_token(f, w) && ppm_token(f, h) && ppm_token(f, maxv) && w > 0 && h > 0 && maxv == 255) {
        rgb.resize((size_t)w * h * 3);
        if (c1 == '6') ok = fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
        else {
            ok = true;
            for (size_t k = 0; k < rgb.size() && ok; ++k) { int v; ok = ppm_token(f, v); rgb[k] = (unsigned char)v; }
        }
    }
    fclose(f);
    return ok;
}

void write_ppm_p3(FILE* f, const float* fb, int nx, int ny, bool double_scale) {
    fprintf(f, "P3\n%d %d\n255\n", nx, ny);
    std::string buf;
    buf.reserve(1 << 20);
    char tmp[64];
    for (int j = ny - 1; j >= 0; --j) {
        for (int i = 0; i < nx; ++i) {
            const float* p = fb + ((size_t)j * nx + i) * 3;
            int r, g, b;
            if (double_scale) { r = int(255.99 * p[0]); g = int(255.99 * p[1]); b = int(255.99 * p[2]); }
            else { r = int(255.99f * p[0]); g = int(255.99f * p[1]); b = int(255.99f * p[2]); }
            int n = snprintf(tmp, sizeof(tmp), "%d %d %d\n", r, g, b);
            buf.append(tmp, (size_t)n);
            if (buf.size() > (1 << 20) - 64) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
        }
    }
    fwrite(buf.data(), 1, buf.size(), f);
}

// Binary PPM (optional): the same quantisation, int(255.99 * c), one byte per channel -- so values the P3 form
// prints above 255 (emitters: the reference does not clamp, main.cu:715-727) are clamped to 255 here, and negatives to 0.
void write_ppm_p6(FILE* f, const float* fb, int nx, int ny, bool double_scale) {
    fprintf(f, "P6\n%d %d\n255\n", nx, ny);
    std::vector<unsigned char> row((size_t)nx * 3);
    for (int j = ny - 1; j >= 0; --j) {
        for (int i = 0; i < nx * 3; ++i) {
            const float c = fb[(size_t)j * nx * 3 + i];
            const int v = double_scale ? int(255.99 * c) : int(255.99f * c);
            row[(size_t)i] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        fwrite(row.data(), 1, row.size(), f);
    }
}

}  // namespace rtw
