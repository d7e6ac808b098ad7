// main.cpp -- drop-in for the reference's src/main.cu host side: renders one of
// the reference's scenes on an MI355X through the C ABI and writes an ASCII
// PPM (P3) to stdout, progress and timing to stderr (main.cu:668-669,712,
// 715-727).  The reference selects the scene with an integer literal in
// `switch (10)` (main.cu:1309); here it is a flag, and with no flags the
// program renders what the reference's main() does not fall through to:
// `--scene bouncing` is case 1, `--scene final` case 9.
//
//   rayTracer [--scene NAME] [--nx W --ny H] [--ns SPP] [--seed S]
//             [--texture file.ppm] [--device N] [--gpus N] [--p6] [--progressive K] [--list]
//
// --gpus N (N > 1) spreads the frame over the first N GPUs of the node: interleaved 4-row tiles, one scene replica
// per device, one RCCL gather to device 0 (rt_multi_*, include/rt_abi.h).  The PPM is byte-identical for every N by construction (global per-pixel seeds,
// no cross-device rays); verified on one GPU for N = 1 and for every rank's share, not yet on N > 1 hardware.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rtw_scenes.h"

// checkCudaErrors (main.cu:23-35): message to stderr, exit code 99
static void check(rt_status st, const char* what) {
    if (st == RT_OK) return;
    fprintf(stderr, "%s failed: %s -- %s\n", what, rt_strerror(st), rt_last_error_detail());
    exit(99);
}

int main(int argc, char** argv) {
    std::string scene_name = "bouncing", texture_path;
    int nx = 0, ny = 0, ns = 0, device = 0, gpus = 1, progressive = 0;
    bool p6 = false;
    unsigned long long seed = 1984ull;
    for (int a = 1; a < argc; ++a) {
        std::string k = argv[a];
        auto val = [&]() -> const char* { if (a + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(2); } return argv[++a]; };
        if (k == "--scene") scene_name = val();
        else if (k == "--nx") nx = atoi(val());
        else if (k == "--ny") ny = atoi(val());
        else if (k == "--ns") ns = atoi(val());
        else if (k == "--seed") seed = strtoull(val(), nullptr, 10);
        else if (k == "--texture") texture_path = val();
        else if (k == "--device") device = atoi(val());
        else if (k == "--gpus") gpus = atoi(val());
        else if (k == "--p6") p6 = true;                        // binary PPM (clamped); the default is the reference's ASCII P3
        else if (k == "--progressive") progressive = atoi(val());   // render in windows of K samples (rt_render_window): same pixels, a frame after each
        else if (k == "--list") { int n = 0; const char* const* v = rtw::scene_names(&n); for (int i = 0; i < n; ++i) printf("%s\n", v[i]); return 0; }
        else { fprintf(stderr, "unknown argument %s\n", k.c_str()); return 2; }
    }

    std::vector<unsigned char> tex;
    int tw = 0, th = 0;
    if (!texture_path.empty() && !rtw::load_ppm(texture_path, tex, tw, th)) {
        fprintf(stderr, "could not read texture '%s' (binary or ASCII PPM expected)\n", texture_path.c_str());
        return 1;   // the reference returns 1 when its texture fails to load (main.cu:817-820)
    }
    std::string err;
    auto scene = rtw::build_scene(scene_name, nx, ny, tex.empty() ? nullptr : tex.data(), tw, th, err);
    if (!scene) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
    if (ns > 0) scene->ns = ns;

    rtw::flat_scene flat;
    rt_status st = rtw::flatten(scene->world, *scene->cam, flat, err, scene->created.data(), (int)scene->created.size());
    if (st != RT_OK) { fprintf(stderr, "flatten: %s\n", err.c_str()); return 2; }
    const rt_scene_desc desc = flat.desc();

    fprintf(stderr, "Rendering a %dx%d image in 8x8 blocks.\n", scene->nx, scene->ny);
    rt_frame_desc f;
    memset(&f, 0, sizeof(f));
    f.nx = scene->nx; f.ny = scene->ny; f.ns = scene->ns; f.gamma = scene->gamma;
    f.background[0] = scene->background.x(); f.background[1] = scene->background.y(); f.background[2] = scene->background.z();
    f.use_gradient_bg = scene->use_gradient_bg;
    f.seed_base = seed;
    f.tile_rows = scene->ny; f.tile_first = 0; f.tile_stride = 1;

    std::vector<float> fb((size_t)scene->nx * scene->ny * 3);
    rt_stats stats;
    rt_scene* dev_scene = nullptr;
    rt_multi* multi = nullptr;
    if (gpus > 1) {
        check(rt_init_devices(gpus), "rt_init_devices");
        check(rt_multi_create(&desc, gpus, &multi), "rt_multi_create");
        check(rt_multi_render(multi, &f, fb.data(), /*fb_on_device=*/0, /*tile_rows=*/4, &stats), "rt_multi_render");
    } else {
        check(rt_init(device), "rt_init");
        check(rt_scene_create(&desc, &dev_scene), "rt_scene_create");
        if (progressive > 0) {
            // progressive accumulation: the per-pixel XORWOW state and colour sum are carried from window to window (the
            // reference writes its curandState back for exactly this, main.cu:126); the last window's frame is the one-shot frame
            void* state = nullptr;
            check(rt_progressive_state_create(dev_scene, &f, &state), "rt_progressive_state_create");
            rt_stats part;
            memset(&stats, 0, sizeof(stats));
            for (int begin = 0; begin < scene->ns; begin += progressive) {
                const int end = begin + progressive < scene->ns ? begin + progressive : scene->ns;
                check(rt_render_window(dev_scene, &f, fb.data(), 0, state, begin, end, nullptr, 1, &part), "rt_render_window");
                stats.rays += part.rays; stats.ms_render += part.ms_render;
                fprintf(stderr, "samples [%d, %d): %.3f ms\n", begin, end, part.ms_render);
            }
            check(rt_progressive_state_destroy(dev_scene, state), "rt_progressive_state_destroy");
        } else {
            check(rt_render(dev_scene, &f, fb.data(), /*fb_on_device=*/0, /*stream=*/nullptr, /*blocking=*/1, &stats), "rt_render");
        }
    }
    fprintf(stderr, "took %g seconds.\n", stats.ms_render * 1e-3);
    fprintf(stderr, "{\"scene\": \"%s\", \"nx\": %d, \"ny\": %d, \"ns\": %d, \"gpus\": %d, \"rays\": %llu, \"ms_render\": %.3f, \"mrays_per_s\": %.1f}\n",
            scene_name.c_str(), scene->nx, scene->ny, scene->ns, gpus, (unsigned long long)stats.rays, stats.ms_render,
            stats.ms_render > 0 ? (double)stats.rays / (stats.ms_render * 1e3) : 0.0);

    if (p6) rtw::write_ppm_p6(stdout, fb.data(), scene->nx, scene->ny, scene->ppm_double_scale);
    else rtw::write_ppm_p3(stdout, fb.data(), scene->nx, scene->ny, scene->ppm_double_scale);

    if (multi) check(rt_multi_destroy(multi), "rt_multi_destroy");
    if (dev_scene) check(rt_scene_destroy(dev_scene), "rt_scene_destroy");
    check(rt_shutdown(), "rt_shutdown");
    return 0;
}
