// rt_abi.hip -- implementation of include/rt_abi.h: device bookkeeping, scene
// validation + upload, the walk-array planner, tier data and cost prior, frame
// launch, statistics.  The kernels are in rt_kernel_pixel.hip, rt_staged_*.hip
// (rt_kernel_staged.h), rt_tier_*.hip (rt_kernel_tier.h) and rt_rank.hip.
// There is no CPU render path here or anywhere else in the
// library: without a HIP device every entry point fails with
// RT_ERR_NO_DEVICE / RT_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <utility>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_device.h"

namespace {

// Devices this process has initialised (rt_init / rt_init_devices).  A scene remembers the device it was created on and
// every entry point that touches it makes that device current first, so a caller may drive several devices from one
// thread (rt_multi_*) or change the current device between calls.
struct device_info {
    bool ready = false;
    int num_cu = 256;
    size_t lds_per_cu = 160 * 1024;
};
enum { RT_MAX_DEVICES = 16 };
device_info g_devices[RT_MAX_DEVICES];
int g_device = -1;              // device new scenes are created on (the last rt_init)
int g_last_hip_error = 0;
std::string g_detail;

// Tuning knobs (rt_set_option); rt_reset_options() restores exactly these defaults, which are what ships.
struct rt_options {
    int kernel = RT_KERNEL_STAGED;
    int lds_mode = -1;          // -1 = choose from the scene size
    int steps_per_trip = 12;
    int shade_threshold = 0;     // lanes with a finished walk that trigger stage C; 0 = 32, or 16 where the launch has less than 2.75 pixels per
                                 // resident lane (a share of a frame: waiting for a stage's quorum costs chain time there; 1/8: 50.0 -> 47.6 ms, 1/4: 62.1 -> 60.2)
    int wg_per_cu = 0;          // workgroups per CU of the staged kernel; 0 = per kernel family (2 x 512 lean, 3 x 256 otherwise)
    int threads = 0;            // workgroup size of the staged kernel; 0 = what the kernel family's register budget calls for
    int leaf_threshold = 8;     // lanes with an object test due that trigger the leaf pass (they keep walking meanwhile)
    int diel_threshold = 2;
    int box_threshold = 8;
    int medium_threshold = 16;
    int newpath_threshold = 0;   // lanes waiting for a new path before stage E runs; 0 = by kernel family: 24 spheres-only, 8 general (measured: Book-2 final 382 -> 364 ms with 8, Book-1 32.2 -> 35.1)
    int sparse_stride = 8;      // lanes per pixel in sparse waves (64 / live lanes); 0 disables sparse waves
    int split_samples = 16;      // samples per pixel rendered before pixels are ranked by measured cost (round 3: 16 and no presplit -- two parts; was 32 after a first look at 8)
    int tier_auto = 1;           // size the tiers from the share of the frame this call renders (see rank_pixels); 0 = the knobs as set
    int tier_kernel = 1;         // tier 1 of the list goes to the tier kernel (rt_kernel_tier.h) on a side stream; 0 = no tier 1
    int prior = 1;               // the first part of a split frame is already ranked: on the cost prior of the calibration frame
    int resplit_samples = 0;     // a second ranking: samples [split, resplit) run with tiers ranked on `split` samples, the rest ranked on `resplit` (0 = off)
    int presplit_samples = 0;    // a first, shorter look: samples [presplit, split) already run with tiers ranked on it (0 = off).  Off since the
                                 // cost prior ranks the first part: [0,16) + [16,ns) is as fast or faster than [0,8) + [8,32) + [32,ns) on every
                                 // BASELINE frame and 10 % faster at 100 spp (profiles/r03_two_parts.log)
    int tier1_factor_x10 = 45;   // tier 1 = heavy pixels costing >= this/10 x the mean
    int tier1_pixels = 1536;     // heavy pixels served one per wave at a time (tier 1)
    int cost_smooth_percent = 0;  // ranking: a pixel's cost estimate is at least this share of its dearest 4-neighbour's (0 = own cost only)
    int tier1_depth = 3;         // ... each wave taking about this many of them, one after the other
    int heavy_factor_x10 = 20;   // a pixel is listed ("heavy") when its cost so far is >= this/10 x the mean ...
    int sparse_factor_x10 = 40;  // ... and goes to a sparse wave (tier 2) from this/10 x the mean; below, ordinary lanes take it first (tier 3)
    int heavy_max_tiles = 0;     // 0 = as many as the sparse workgroups hold at once
    int semi_stride = -1;        // lanes per pixel in the workgroups serving tier 3 (0 = ordinary lanes take tier 3 first); -1 = by kernel family:
                                 // lean 1 -- whole waves of tier-3 pixels, which with semi_priority 1 measured 100.2 -> 94.3 ms on the headline
                                 // (profiles/r03_priorities_whole.log) --, the others 0 (Book-2 final 352 -> 388 ms with 1, profiles/r03_general_defaults.log)
    int sparse_priority = 3;
    int tier_priority = 1;       // s_setprio level of the tier kernel's waves (1 instead of 3: 80.6 -> 76.3 ms on a half, 70.2 -> 62.7 on a quarter of
                                 // the headline frame, nothing on the whole frame or an eighth: profiles/r03_share_sweep_pass2.log)
    int semi_priority = 1;       // s_setprio level of the semi workgroups' waves (tier 3 on workgroups of its own)
    int handoff = 1;             // tail hand-off (rt_device.h): the main kernel's last pixels are finished by a launch of the tier kernel after it
    int handoff_scan = 1;        // ... also in scenes scanned in lockstep (lds_mode 4), which have no tier 1
    int handoff_poll_us = 1000;  // ... looked for this often by each wave that has run out of queued work
    int handoff_pixels = -1;     // ... when at most this many are in flight and the tile queue is dry; -1 = auto (render_impl)
    int sparse_eager = 0;
    int sparse_work_percent = 5;  // tiers 0-2 hold at most this share of the frame's work (rays so far); dearer-than-average pixels beyond it go to tier 3
    int sparse_wg_percent = 35;   // at most this share of the workgroups starts in sparse mode
    int bvh_collapse = 3;        // walk array (rt_scene_create): 0 = the reference's tree as is, 1 = interior nodes that do not pay
                                 // removed, decided from box surface areas, 2 = decided from pass counts measured on a small frame,
                                 // 3 = as 2, and the leaves regrouped by surface-area cost if that predicts fewer box tests
    int scan_nodes = 24;         // scenes whose walk array has at most this many nodes are scanned in lockstep (lds_mode 4); 0 = never
    int multi_force_rccl = 0;    // rt_multi_render: go through the RCCL gather even with one device (tests the path on a one-GPU box)
    int lpt = 1;                 // cost prepass + longest-first tile order (staged kernel, ns >= 2 * split_samples)
};
rt_options g_opt;

// the reference's checkCudaErrors (main.cu:23-35) records "<code> at file:line 'expr'"; it then
// exits with 99, which a library must not do, so the status is returned instead.
#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            g_last_hip_error = (int)e_;                                               \
            char buf_[512];                                                           \
            snprintf(buf_, sizeof(buf_), "HIP error = %u at %s:%d '%s' (%s)", (unsigned)e_, __FILE__, __LINE__, #expr, \
                     hipGetErrorString(e_));                                          \
            g_detail = buf_;                                                          \
            return RT_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

rt_status invalid(const char* why) { g_detail = why; return RT_ERR_INVALID; }

template <class T>
rt_status upload(const T* src, size_t count, const T** dst) {
    *dst = nullptr;
    // always allocate at least one element so kernels never see a null array base
    size_t bytes = (count ? count : 1) * sizeof(T);
    bytes = (bytes + 63) & ~(size_t)63;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, bytes));
    HIPCHK(hipMemset(p, 0, bytes));
    if (count) HIPCHK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return RT_OK;
}

}  // namespace

rt_status rt_internal_scene_create_on(int device, const rt_scene_desc* d, rt_scene** out);

struct rt_scene {
    int device = 0;             // the device every allocation below lives on
    rt_scene_dev dev;
    std::vector<void*> allocs;
    bool spheres_only = false, need_uv = false;
    int tex_level = 0;
    size_t node_bytes = 0, sphere_bytes = 0, shade_bytes = 0;   // node_bytes: the walk array; shade_bytes: materials + textures
    double walk_tests_before = 0, walk_tests_after = 0;          // expected box tests per calibration ray, reference tree / walk array
    // per-frame resources
    unsigned long long* d_ray_counter = nullptr;
    unsigned int* d_work_counter = nullptr;
    float* d_fb = nullptr;
    size_t d_fb_floats = 0;
    unsigned int* d_tile_cost = nullptr;   // cost prepass: rays per 8x8 tile
    unsigned int* d_tile_order = nullptr;  // tiles in descending cost
    rt_pixel_state* d_state = nullptr;     // split frames: pixels parked between the two parts
    unsigned long long* d_heavy_list = nullptr;   // (cost << 32 | pixel), unsorted, from rt_collect_heavy_kernel
    unsigned int* d_heavy_pixels = nullptr;       // heavy pixels, dearest first
    size_t tile_capacity = 0, pixel_capacity = 0;
    unsigned long long* d_handoff = nullptr;      // tail hand-off queue: (sample << 32 | pixel), one entry per resident lane at most
    size_t handoff_capacity = 0;
    rt_rank_info* d_rank = nullptr;               // tier sizes of the next ranked launch (written and read on the device only)
    unsigned int* d_cal_cost = nullptr;           // cost prior: rays per pixel of the calibration frame (cal_nx x cal_ny at 4 spp)
    int cal_nx = 0, cal_ny = 0;
    hipStream_t tier_stream = nullptr;            // the tier kernel's stream (forked from / joined to the caller's stream by events)
    hipEvent_t ev_fork[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ranked_frame = false;                    // the pending frame used the cost-aware schedule
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool frame_pending = false;
    hipStream_t pending_stream = nullptr;
    rt_stats pending_stats;
};

// Progressive accumulation (rt_render_window): the parked pixels of one frame description between windows.
struct rt_progressive {
    rt_scene* scene = nullptr;
    rt_pixel_state* d_state = nullptr;
    size_t pixels = 0;
    rt_frame_desc frame;         // the description it was made for (its partition must not change between windows)
    int32_t next_sample = 0;     // the sample_begin the next window must have
};

namespace {

// picks the kernel family; LDS_MODE and texture level select the instantiation inside (rt_staged_*.hip)
hipError_t launch_render(int kernel, int lds_mode, const rt_scene* s, const rt_frame_params& fp, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream) {
    if (kernel == RT_KERNEL_PIXEL) return rt_launch_pixel(s->spheres_only, s->tex_level, s->need_uv, s->dev, fp, grid, block, stream);
    if (s->spheres_only) {
        if (s->tex_level <= 1) return rt_launch_staged_spheres(s->tex_level, lds_mode, s->dev, fp, grid, block, lds_bytes, stream);
        return rt_launch_staged_spheres_tex(lds_mode, s->dev, fp, grid, block, lds_bytes, stream);
    }
    if (s->tex_level <= 1 && !s->need_uv) return rt_launch_staged_general(lds_mode, s->dev, fp, grid, block, lds_bytes, stream);
    return rt_launch_staged_general_tex(lds_mode, s->dev, fp, grid, block, lds_bytes, stream);
}

bool simple_ref_ok(const rt_scene_desc* d, int32_t ref) {
    if (ref < 0) return false;
    const int k = RT_PRIM_KIND(ref), i = RT_PRIM_INDEX(ref);
    if (k == RT_PRIM_SPHERE) return i < d->n_spheres;
    if (k == RT_PRIM_QUAD) return i < d->n_quads;
    if (k == RT_PRIM_BOX) return i < d->n_boxes;
    return false;
}
bool solid_ref_ok(const rt_scene_desc* d, int32_t ref) {
    if (ref < 0) return false;
    if (RT_PRIM_KIND(ref) == RT_PRIM_INSTANCE) return RT_PRIM_INDEX(ref) < d->n_instances;
    return simple_ref_ok(d, ref);
}

// Every index a kernel will follow is checked here, on the host, so that a
// malformed description is an error code and never a GPU fault.
rt_status validate(const rt_scene_desc* d, bool& spheres_only, int& tex_level, bool& need_uv) {
    if (!d) return invalid("null scene description");
    if (d->n_nodes < 0 || d->n_spheres < 0 || d->n_quads < 0 || d->n_boxes < 0 || d->n_instances < 0 || d->n_media < 0 ||
        d->n_materials < 0 || d->n_textures < 0)
        return invalid("negative count");
    if ((d->n_nodes && !d->nodes) || (d->n_spheres && !d->spheres) || (d->n_quads && !d->quads) || (d->n_boxes && !d->boxes) ||
        (d->n_instances && !d->instances) || (d->n_media && !d->media) || (d->n_materials && !d->materials) ||
        (d->n_textures && !d->textures) || (d->image_bytes && !d->images))
        return invalid("null array with non-zero count");
    if (d->n_nodes >= (1 << 28) || d->n_spheres >= (1 << 28) || d->n_quads >= (1 << 28)) return invalid("scene too large");
    spheres_only = true;
    for (int i = 0; i < d->n_nodes; ++i) {
        const rt_node& n = d->nodes[i];
        if (n.skip <= i || n.skip > d->n_nodes) return invalid("node skip link does not move forward");
        if (n.prim >= 0) {
            const int k = RT_PRIM_KIND(n.prim), idx = RT_PRIM_INDEX(n.prim);
            if (k != RT_PRIM_SPHERE) spheres_only = false;
            if (k == RT_PRIM_MEDIUM) { if (idx >= d->n_media) return invalid("medium index out of range"); }
            else if (!solid_ref_ok(d, n.prim)) return invalid("leaf primitive reference out of range");
        }
    }
    for (int i = 0; i < d->n_spheres; ++i)
        if (d->spheres[i].mat < 0 || d->spheres[i].mat >= d->n_materials) return invalid("sphere material out of range");
    for (int i = 0; i < d->n_quads; ++i)
        if (d->quads[i].mat < 0 || d->quads[i].mat >= d->n_materials) return invalid("quad material out of range");
    for (int i = 0; i < d->n_boxes; ++i)
        if (d->boxes[i].first_quad < 0 || d->boxes[i].first_quad + 6 > d->n_quads) return invalid("box faces out of range");
    for (int i = 0; i < d->n_instances; ++i)
        if (!simple_ref_ok(d, d->instances[i].child)) return invalid("instance child must be a sphere, quad or box");
    for (int i = 0; i < d->n_media; ++i) {
        if (!solid_ref_ok(d, d->media[i].boundary)) return invalid("medium boundary must be a sphere, quad, box or instance");
        if (d->media[i].mat < 0 || d->media[i].mat >= d->n_materials) return invalid("medium material out of range");
    }
    tex_level = 0; need_uv = false;
    for (int i = 0; i < d->n_materials; ++i) {
        const rt_material& m = d->materials[i];
        if (m.kind < RT_MAT_LAMBERTIAN || m.kind > RT_MAT_ISOTROPIC) return invalid("unknown material kind");
        if (m.tex >= d->n_textures) return invalid("material texture out of range");
        if (m.tex >= 0 && tex_level < 1) tex_level = 1;
    }
    for (int i = 0; i < d->n_textures; ++i) {
        const rt_texture& t = d->textures[i];
        if (t.kind == RT_TEX_CHECKER) {
            // even/odd must point at later-or-earlier non-self entries that terminate: forbid checker children
            if (t.a < 0 || t.a >= d->n_textures || t.b < 0 || t.b >= d->n_textures) return invalid("checker child out of range");
            if (d->textures[t.a].kind == RT_TEX_CHECKER || d->textures[t.b].kind == RT_TEX_CHECKER ||
                d->textures[t.a].kind == RT_TEX_UV_OFFSET || d->textures[t.b].kind == RT_TEX_UV_OFFSET)
                return invalid("checker children must be plain textures");
        } else if (t.kind == RT_TEX_IMAGE) {
            need_uv = true; tex_level = 2;
            if (t.a >= 0) {
                if (t.b <= 0 || t.c <= 0) return invalid("image texture with non-positive size");
                if ((size_t)t.a + (size_t)t.b * t.c * 3 > d->image_bytes) return invalid("image texture outside the image pool");
            }
        } else if (t.kind == RT_TEX_NOISE || t.kind == RT_TEX_NOODLE || t.kind == RT_TEX_FELT) {
            tex_level = 2;
            if (t.kind == RT_TEX_NOODLE && (t.a < 0 || t.a > 16)) return invalid("noodle texture octaves out of range");
        } else if (t.kind == RT_TEX_UV_OFFSET) {
            tex_level = 2; need_uv = true;
            if (t.a < 0 || t.a >= d->n_textures) return invalid("uv_offset child out of range");
            if (d->textures[t.a].kind == RT_TEX_UV_OFFSET || d->textures[t.a].kind == RT_TEX_CHECKER)
                return invalid("uv_offset may wrap image, solid, noise, noodle or felt textures only");
        } else if (t.kind != RT_TEX_SOLID) {
            return invalid("unknown texture kind");
        }
    }
    if (d->n_quads || d->n_boxes || d->n_instances || d->n_media) spheres_only = spheres_only && false;
    return RT_OK;
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------------------------
// "Collapse": which interior nodes of the reference's tree are worth testing.
//
// bvh_node::hit (bvh.cuh:95-106) reaches an object iff every ancestor box and the object's own box (its single-object
// node, bvh.cuh:38-43) pass against the closest hit so far.  The slab test (aabb.cuh:45-61) is monotone in the box (a
// box containing another is entered no later and left no earlier, rounding included: the same subtract-multiply on
// ordered operands; a NaN from 0 * inf leaves a limit unconstrained for parent and child alike) and in the limit, and
// limits only shrink during a walk.  So "the own box passes at the moment of the object test" already implies every
// ancestor passed earlier: the interior boxes never change a result, they only save work -- and only if they fail
// often enough.  Dropping interior node X (its children take its place in the depth-first array) removes one test per
// visit of X and adds one failing test per direct child whenever X would have failed; with v = visits and p = passes
// that is a saving iff p / v > 1 - 1 / k.  A median-split tree has many such nodes (the random scene: 40.1 -> 29.4 box
// tests per ray, Cornell 14.6 -> 8.6, Book-2 final 55.1 -> 35.6).  The choice is an exact tree DP over (node, nearest
// kept ancestor) on pass counts -- measured by kernel 0 on a small frame of the scene's own camera (bvh_collapse = 2),
// or taken proportional to box surface area (1).  Order of the leaves, every leaf node and every box value are the
// reference's; frames are bit-identical with the option on or off (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------------------------
struct collapse_plan {
    std::vector<char> keep;     // per reference node
    double tests_before = 0, tests_after = 0;   // expected box tests per calibration ray
};

// children of interior node i in the depth-first array: i + 1, then each next sibling at the previous one's skip link
template <class F> void for_children(const rt_node* nodes, int i, F f) {
    for (int c = i + 1; c < nodes[i].skip; c = nodes[c].skip) f(c);
}

bool plan_collapse(const rt_node* nodes, int n, const std::vector<double>& pass, double root_visits, collapse_plan& plan) {
    plan.keep.assign((size_t)n, 1);
    if (n < 3 || root_visits <= 0) return false;
    // depth of every node (= length of its ancestor chain); parents precede children in the array
    std::vector<int> depth((size_t)n, 0), parent((size_t)n, -1);
    int max_depth = 0;
    for (int i = 0; i < n; ++i)
        if (nodes[i].prim < 0) for_children(nodes, i, [&](int c) { depth[c] = depth[i] + 1; parent[c] = i; if (depth[c] > max_depth) max_depth = depth[c]; });
    if (max_depth > 60) return false;   // a degenerate (list-like) tree: leave it alone
    // the argument above needs a real tree (children tile their parent's range of the array) whose boxes contain their
    // children's; a description that is anything else is walked as given
    for (int i = 0; i < n; ++i) {
        if (i > 0 && parent[i] < 0) return false;
        if (nodes[i].prim >= 0) { if (nodes[i].skip != i + 1) return false; continue; }
        int c = i + 1, kids = 0;
        for (; c < nodes[i].skip; c = nodes[c].skip) {
            ++kids;
            for (int a = 0; a < 3; ++a)
                if (!(nodes[c].bmin[a] >= nodes[i].bmin[a] && nodes[c].bmax[a] <= nodes[i].bmax[a])) return false;
        }
        if (c != nodes[i].skip || kids == 0) return false;
    }
    // cost[i][d]: fewest tests in the subtree of i per calibration run when the nearest kept ancestor is the d-th entry of
    // i's ancestor chain (0 = "above the root": root_visits, 1.. = ancestors from the root down); visits of a node = passes
    // of its nearest kept ancestor (a fixed-order walk has no other way to skip it)
    std::vector<std::vector<double>> cost((size_t)n);
    std::vector<int> chain;   // scratch
    auto visits = [&](int i, int d) -> double {   // d-th entry of i's chain
        if (d == 0) return root_visits;
        int a = i;
        for (int up = depth[i] - d + 1; up > 0; --up) a = parent[a];
        return pass[a];
    };
    for (int i = n - 1; i >= 0; --i) {   // children have larger indices: reverse order is post-order enough
        const int D = depth[i] + 1;
        cost[i].assign((size_t)D, 0.0);
        if (nodes[i].prim >= 0) { for (int d = 0; d < D; ++d) cost[i][d] = visits(i, d); continue; }
        for (int d = 0; d < D; ++d) {
            double keep = visits(i, d), drop = 0.0;
            for_children(nodes, i, [&](int c) { keep += cost[c][(size_t)D]; drop += cost[c][(size_t)d]; });
            cost[i][d] = keep <= drop ? keep : drop;
        }
    }
    // decide top-down
    std::vector<int> nearest((size_t)n, 0);
    double before = 0.0;
    for (int i = 0; i < n; ++i) before += (i == 0) ? root_visits : pass[parent[i]];
    for (int i = 0; i < n; ++i) {
        const int d = nearest[i];
        bool keep = true;
        if (nodes[i].prim < 0) {
            double k = visits(i, d), drop = 0.0;
            const int D = depth[i] + 1;
            for_children(nodes, i, [&](int c) { k += cost[c][(size_t)D]; drop += cost[c][(size_t)d]; });
            keep = k <= drop;
            for_children(nodes, i, [&](int c) { nearest[c] = keep ? D : d; });
        }
        plan.keep[i] = keep ? 1 : 0;
    }
    plan.tests_before = before / root_visits;
    plan.tests_after = cost[0][0] / root_visits;
    return true;
}

// the walk array: kept nodes in the reference's depth-first order, skip links re-pointed at the next kept node
std::vector<rt_node> build_walk_array(const rt_node* nodes, int n, const std::vector<char>& keep) {
    std::vector<int> kept_before((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) kept_before[i + 1] = kept_before[i] + (keep[i] ? 1 : 0);
    std::vector<rt_node> walk;
    walk.reserve((size_t)kept_before[n]);
    for (int i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        rt_node w = nodes[i];
        w.skip = kept_before[nodes[i].skip];
        walk.push_back(w);
    }
    return walk;
}

// ---------------------------------------------------------------------------------------------------------------
// "Regroup": another hierarchy over the same leaves.
//
// By the argument above the walk returns what the reference's returns as long as (1) the leaves -- the reference's
// single-object nodes, with their boxes -- come in the reference's depth-first order and (2) every interior box contains
// the boxes below it: then "a leaf's own box passes" still implies that everything above it passed, and a subtree is
// only ever skipped when none of its leaves could have passed.  Which contiguous runs of leaves are grouped is free.  The
// reference groups by halving (bvh.cuh:25-36), which puts a huge object (the ground sphere: a 2000-unit box) into half of
// the top of the tree: every box above it is as large as it is and always passes.  regroup_leaves() builds a binary tree
// over the same leaf sequence, interior boxes = the exact union (float min / max) of their leaves' boxes, in one of two
// ways: top-down -- split [a, b) at the k that minimises area(a..k) * (k - a) + area(k..b) * (b - k) -- or bottom-up --
// keep merging the two neighbouring groups with the smallest union, "smallest" by surface area (method 1) or by how many
// of ~4000 rays sampled from the calibration pass meet it (method 2: the scene as this camera's paths see it).  Each goes
// through the same calibration pass and collapse DP as the reference's tree, and whichever walk array predicts the fewest
// box tests per ray is used (bvh_collapse = 3).
// ---------------------------------------------------------------------------------------------------------------
std::vector<rt_node> regroup_leaves(const rt_node* nodes, int n, int method, const std::vector<float>* sample = nullptr) {
    std::vector<rt_node> leaves;
    for (int i = 0; i < n; ++i) if (nodes[i].prim >= 0) leaves.push_back(nodes[i]);
    const int m = (int)leaves.size();
    std::vector<rt_node> out;
    if (m == 0) return out;
    out.reserve((size_t)(2 * m));
    struct box { float lo[3], hi[3]; };
    auto grow = [](box& b, const rt_node& l) { for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], l.bmin[a]); b.hi[a] = fmaxf(b.hi[a], l.bmax[a]); } };
    auto area = [](const box& b) -> double {
        const double x = fmax(0.0, (double)b.hi[0] - b.lo[0]), y = fmax(0.0, (double)b.hi[1] - b.lo[1]), z = fmax(0.0, (double)b.hi[2] - b.lo[2]);
        return 2.0 * (x * y + y * z + x * z);
    };
    const box empty = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    if (method >= 1) {
        // bottom-up: merge, again and again, the two neighbouring groups whose union has the smallest surface area (a heap
        // of neighbour pairs with lazy deletion); the huge leaf is merged last and ends up directly under the root.
        // method 2: "smallest" by the share of the calibration pass's sampled rays that pass the union box (each with the
        // limit it ended with) -- the scene as this camera's paths see it -- with the surface area as the tie-break
        const size_t n_rays = (method == 2 && sample) ? sample->size() / 7 : 0;
        // the sample as seven arrays (origin, 1 / direction, limit) so that the loop below vectorises
        std::vector<float> q[7];
        for (int a = 0; a < 7; ++a) q[a].resize(n_rays);
        for (size_t r = 0; r < n_rays; ++r) {
            for (int a = 0; a < 3; ++a) { q[a][r] = (*sample)[r * 7 + a]; q[3 + a][r] = 1.0f / (*sample)[r * 7 + 3 + a]; }
            q[6][r] = (*sample)[r * 7 + 6];
        }
        auto seen_by = [&](const box& b) -> double {
            unsigned int hits = 0;
            for (size_t r = 0; r < n_rays; ++r) {
                float t0 = 0.001f, t1 = q[6][r];
                for (int a = 0; a < 3; ++a) {
                    const float x = (b.lo[a] - q[a][r]) * q[3 + a][r], y = (b.hi[a] - q[a][r]) * q[3 + a][r];
                    const float lo_t = x < y ? x : y, hi_t = x < y ? y : x;
                    t0 = lo_t > t0 ? lo_t : t0;
                    t1 = hi_t < t1 ? hi_t : t1;
                }
                hits += (t1 <= t0) ? 0u : 1u;
            }
            return (double)hits;
        };
        struct group { box b; int left, right, leaf, prev, next, count; bool alive; };
        std::vector<group> g((size_t)m);
        for (int i = 0; i < m; ++i) {
            g[(size_t)i].b = empty; grow(g[(size_t)i].b, leaves[(size_t)i]);
            g[(size_t)i].left = g[(size_t)i].right = -1; g[(size_t)i].leaf = i; g[(size_t)i].prev = i - 1; g[(size_t)i].next = i + 1 < m ? i + 1 : -1;
            g[(size_t)i].count = 1; g[(size_t)i].alive = true;
        }
        auto unite = [](const box& x, const box& y) { box u = x; for (int a = 0; a < 3; ++a) { u.lo[a] = fminf(u.lo[a], y.lo[a]); u.hi[a] = fmaxf(u.hi[a], y.hi[a]); } return u; };
        struct pair_key { double area; int i, j; };
        auto worse = [](const pair_key& x, const pair_key& y) { return x.area > y.area || (x.area == y.area && x.i > y.i); };
        std::vector<pair_key> heap;
        auto push = [&](int i, int j) {
            const box ub = unite(g[(size_t)i].b, g[(size_t)j].b);
            double ar = area(ub);
            if (!std::isfinite(ar)) ar = DBL_MAX;
            else if (n_rays) ar = seen_by(ub) + ar / (ar + 1.0);      // rays first, area (squashed below one ray) second
            heap.push_back({ar, i, j});
            std::push_heap(heap.begin(), heap.end(), worse);
        };
        for (int i = 0; i + 1 < m; ++i) push(i, i + 1);
        int root = m - 1;
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), worse);
            const pair_key k = heap.back();
            heap.pop_back();
            if (!g[(size_t)k.i].alive || !g[(size_t)k.j].alive) continue;
            group u;
            u.b = unite(g[(size_t)k.i].b, g[(size_t)k.j].b); u.left = k.i; u.right = k.j; u.leaf = -1;
            u.prev = g[(size_t)k.i].prev; u.next = g[(size_t)k.j].next; u.count = g[(size_t)k.i].count + g[(size_t)k.j].count; u.alive = true;
            g[(size_t)k.i].alive = g[(size_t)k.j].alive = false;
            const int id = (int)g.size();
            g.push_back(u);
            root = id;
            if (u.prev >= 0) { g[(size_t)u.prev].next = id; push(u.prev, id); }
            if (u.next >= 0) { g[(size_t)u.next].prev = id; push(id, u.next); }
        }
        // emit depth-first
        std::vector<int> todo2;
        todo2.push_back(root);
        std::vector<std::pair<int, int>> open2;
        while (!todo2.empty()) {
            const int id = todo2.back();
            todo2.pop_back();
            if (g[(size_t)id].leaf >= 0) {
                rt_node l = leaves[(size_t)g[(size_t)id].leaf];
                l.skip = (int32_t)out.size() + 1;
                out.push_back(l);
                for (auto& o : open2) --o.second;
                while (!open2.empty() && open2.back().second == 0) { out[(size_t)open2.back().first].skip = (int32_t)out.size(); open2.pop_back(); }
                continue;
            }
            rt_node nd;
            for (int a = 0; a < 3; ++a) { nd.bmin[a] = g[(size_t)id].b.lo[a]; nd.bmax[a] = g[(size_t)id].b.hi[a]; }
            nd.prim = -1; nd.skip = 0;
            open2.push_back({(int)out.size(), g[(size_t)id].count});
            out.push_back(nd);
            todo2.push_back(g[(size_t)id].right);
            todo2.push_back(g[(size_t)id].left);
        }
        return out;
    }
    std::vector<double> right_area((size_t)m + 1);
    // explicit stack: (a, b) ranges in depth-first order; the interior node's skip link is patched when its range ends
    struct item { int a, b; };
    std::vector<item> todo;
    todo.push_back({0, m});
    std::vector<std::pair<int, int>> open;   // (index of an interior node, number of leaves still to be emitted under it)
    auto leaf_done = [&]() {
        // every enclosing interior node has one leaf fewer to wait for; those that are complete end here
        for (auto& o : open) --o.second;
        while (!open.empty() && open.back().second == 0) { out[(size_t)open.back().first].skip = (int32_t)out.size(); open.pop_back(); }
    };
    while (!todo.empty()) {
        const item it = todo.back();
        todo.pop_back();
        if (it.b - it.a == 1) {
            rt_node l = leaves[(size_t)it.a];
            l.skip = (int32_t)out.size() + 1;
            out.push_back(l);
            leaf_done();
            continue;
        }
        box all = empty;
        for (int k = it.a; k < it.b; ++k) grow(all, leaves[(size_t)k]);
        // areas of the right parts [k, b), then sweep the left part
        box r = empty;
        for (int k = it.b - 1; k > it.a; --k) { grow(r, leaves[(size_t)k]); right_area[(size_t)k] = area(r); }
        box l = empty;
        int best_k = (it.a + it.b) / 2;
        double best = -1.0;
        for (int k = it.a + 1; k < it.b; ++k) {
            grow(l, leaves[(size_t)k - 1]);
            const double c = area(l) * (double)(k - it.a) + right_area[(size_t)k] * (double)(it.b - k);
            if (std::isfinite(c) && (best < 0.0 || c < best)) { best = c; best_k = k; }
        }
        rt_node nd;
        for (int a = 0; a < 3; ++a) { nd.bmin[a] = all.lo[a]; nd.bmax[a] = all.hi[a]; }
        nd.prim = -1; nd.skip = 0;
        open.push_back({(int)out.size(), it.b - it.a});
        out.push_back(nd);
        todo.push_back({best_k, it.b});      // right part second
        todo.push_back({it.a, best_k});      // left part first
    }
    return out;
}

}  // namespace

extern "C" {

namespace {
rt_status init_device(int device_ordinal) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_last_hip_error = (int)e;
        g_detail = "no HIP device visible (the render path has no CPU fallback)";
        return RT_ERR_NO_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= count || device_ordinal >= RT_MAX_DEVICES) return invalid("device ordinal out of range");
    HIPCHK(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_ordinal));
    if (!strstr(prop.gcnArchName, "gfx950")) {
        g_detail = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
        return RT_ERR_NO_DEVICE;
    }
    device_info& d = g_devices[device_ordinal];
    d.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    d.lds_per_cu = prop.maxSharedMemoryPerMultiProcessor > 0 ? (size_t)prop.maxSharedMemoryPerMultiProcessor : (size_t)160 * 1024;
    d.ready = true;
    return RT_OK;
}
// makes the scene's device current (a no-op when it already is)
rt_status use_device(int device) {
    if (device < 0 || device >= RT_MAX_DEVICES || !g_devices[device].ready) { g_detail = "rt_init has not succeeded for this device"; return RT_ERR_NO_DEVICE; }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) HIPCHK(hipSetDevice(device));
    return RT_OK;
}
}  // namespace

rt_status rt_init(int device_ordinal) {
    const rt_status st = init_device(device_ordinal);
    if (st == RT_OK) g_device = device_ordinal;
    return st;
}

rt_status rt_shutdown(void) {
    for (device_info& d : g_devices) d.ready = false;
    g_device = -1;
    return RT_OK;
}

rt_status rt_reset_options(void) {
    g_opt = rt_options();
    return RT_OK;
}

const char* rt_strerror(rt_status s) {
    switch (s) {
    case RT_OK: return "ok";
    case RT_ERR_INVALID: return "invalid argument or malformed scene description";
    case RT_ERR_NO_DEVICE: return "no gfx950 HIP device";
    case RT_ERR_HIP: return "HIP runtime error";
    case RT_ERR_UNSUPPORTED: return "unsupported scene construct";
    default: return "unknown status";
    }
}
int rt_last_hip_error(void) { return g_last_hip_error; }
const char* rt_last_error_detail(void) { return g_detail.c_str(); }

rt_status rt_set_option(const char* key, int value) {
    if (!key) return invalid("null option key");
    const std::string k(key);
    if (k == "kernel") { if (value != RT_KERNEL_PIXEL && value != RT_KERNEL_STAGED) return invalid("kernel: 0 (pixel) or 3 (staged)"); g_opt.kernel = value; }
    else if (k == "leaf_threshold") { if (value < 1 || value > 64) return invalid("leaf_threshold: 1..64"); g_opt.leaf_threshold = value; }
    else if (k == "box_threshold") { if (value < 1 || value > 64) return invalid("box_threshold: 1..64"); g_opt.box_threshold = value; }
    else if (k == "medium_threshold") { if (value < 1 || value > 64) return invalid("medium_threshold: 1..64"); g_opt.medium_threshold = value; }
    else if (k == "diel_threshold") { if (value < 1 || value > 64) return invalid("diel_threshold: 1..64"); g_opt.diel_threshold = value; }
    else if (k == "newpath_threshold") { if (value < 0 || value > 64) return invalid("newpath_threshold: 0 (by kernel family) or 1..64"); g_opt.newpath_threshold = value; }
    else if (k == "sparse_stride") { if (value != 0 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64) return invalid("sparse_stride: 0, 2, 4, ... 64"); g_opt.sparse_stride = value; }
    else if (k == "sparse_factor_x10") { if (value < 10 || value > 1000) return invalid("sparse_factor_x10: 10..1000"); g_opt.sparse_factor_x10 = value; }
    else if (k == "heavy_factor_x10") { if (value < 10 || value > 1000) return invalid("heavy_factor_x10: 10..1000"); g_opt.heavy_factor_x10 = value; }
    else if (k == "tier_auto") { if (value < 0 || value > 1) return invalid("tier_auto: 0 or 1"); g_opt.tier_auto = value; }
    else if (k == "tier_kernel") { if (value < 0 || value > 1) return invalid("tier_kernel: 0 or 1"); g_opt.tier_kernel = value; }
    else if (k == "handoff") { if (value < 0 || value > 1) return invalid("handoff: 0 or 1"); g_opt.handoff = value; }
    else if (k == "handoff_scan") { if (value < 0 || value > 1) return invalid("handoff_scan: 0 or 1"); g_opt.handoff_scan = value; }
    else if (k == "handoff_poll_us") { if (value < 1 || value > 1000000) return invalid("handoff_poll_us: 1..1000000"); g_opt.handoff_poll_us = value; }
    else if (k == "handoff_pixels") { if (value < -1 || value > (1 << 24)) return invalid("handoff_pixels: -1 (auto) or 0..16777216"); g_opt.handoff_pixels = value; }
    else if (k == "prior") { if (value < 0 || value > 1) return invalid("prior: 0 or 1"); g_opt.prior = value; }
    else if (k == "presplit_samples") { if (value < 0 || value > 4096) return invalid("presplit_samples: 0..4096"); g_opt.presplit_samples = value; }
    else if (k == "resplit_samples") { if (value < 0 || value > 65536) return invalid("resplit_samples: 0..65536"); g_opt.resplit_samples = value; }
    else if (k == "split_samples") { if (value < 1 || value > 4096) return invalid("split_samples: 1..4096"); g_opt.split_samples = value; }
    else if (k == "tier1_factor_x10") { if (value < 10 || value > 10000) return invalid("tier1_factor_x10: 10..10000"); g_opt.tier1_factor_x10 = value; }
    else if (k == "cost_smooth_percent") { if (value < 0 || value > 100) return invalid("cost_smooth_percent: 0..100"); g_opt.cost_smooth_percent = value; }
    else if (k == "tier1_depth") { if (value < 1 || value > 64) return invalid("tier1_depth: 1..64"); g_opt.tier1_depth = value; }
    else if (k == "tier1_pixels") { if (value < 0 || value > 65536) return invalid("tier1_pixels: 0..65536"); g_opt.tier1_pixels = value; }
    else if (k == "semi_stride") { if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return invalid("semi_stride: -1 (by kernel family), 0, 1, 2, 4 or 8"); g_opt.semi_stride = value; }
    else if (k == "sparse_eager") { if (value < 0 || value > 1) return invalid("sparse_eager: 0 or 1"); g_opt.sparse_eager = value; }
    else if (k == "tier_priority") { if (value < 0 || value > 3) return invalid("tier_priority: 0..3"); g_opt.tier_priority = value; }
    else if (k == "semi_priority") { if (value < 0 || value > 3) return invalid("semi_priority: 0..3"); g_opt.semi_priority = value; }
    else if (k == "sparse_priority") { if (value < 0 || value > 3) return invalid("sparse_priority: 0..3"); g_opt.sparse_priority = value; }
    else if (k == "sparse_work_percent") { if (value < 0 || value > 100) return invalid("sparse_work_percent: 0..100"); g_opt.sparse_work_percent = value; }
    else if (k == "sparse_wg_percent") { if (value < 1 || value > 100) return invalid("sparse_wg_percent: 1..100"); g_opt.sparse_wg_percent = value; }
    else if (k == "heavy_max_tiles") { if (value < 0 || value > 4096) return invalid("heavy_max_tiles: 0..4096"); g_opt.heavy_max_tiles = value; }
    else if (k == "bvh_collapse") { if (value < 0 || value > 3) return invalid("bvh_collapse: 0..3 (read by rt_scene_create)"); g_opt.bvh_collapse = value; }
    else if (k == "scan_nodes") { if (value < 0 || value > 64) return invalid("scan_nodes: 0..64"); g_opt.scan_nodes = value; }
    else if (k == "multi_force_rccl") { if (value < 0 || value > 1) return invalid("multi_force_rccl: 0 or 1"); g_opt.multi_force_rccl = value; }
    else if (k == "lpt") { if (value < 0 || value > 1) return invalid("lpt: 0 or 1"); g_opt.lpt = value; }
    else if (k == "threads") { if (value != 0 && (value < 64 || value > 768 || (value % 64))) return invalid("threads: 0 (per kernel family) or a multiple of 64 up to 768 (512 for the lean spheres-only kernels)"); g_opt.threads = value; }
    else if (k == "lds_mode") { if (value < -1 || value > 4) return invalid("lds_mode: -1..4"); g_opt.lds_mode = value; }
    else if (k == "steps_per_trip") { if (value < 1 || value > 64) return invalid("steps_per_trip: 1..64"); g_opt.steps_per_trip = value; }
    else if (k == "shade_threshold") { if (value < 0 || value > 64) return invalid("shade_threshold: 0 (by the launch's load) or 1..64"); g_opt.shade_threshold = value; }
    else if (k == "wg_per_cu") { if (value < 0 || value > 8) return invalid("wg_per_cu: 0 (per kernel family) .. 8"); g_opt.wg_per_cu = value; }
    else return invalid("unknown option");
    return RT_OK;
}

rt_status rt_scene_destroy(rt_scene* s) {
    if (!s) return RT_OK;
    (void)use_device(s->device);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->d_ray_counter) (void)hipFree(s->d_ray_counter);
    if (s->d_work_counter) (void)hipFree(s->d_work_counter);
    if (s->d_fb) (void)hipFree(s->d_fb);
    if (s->d_tile_cost) (void)hipFree(s->d_tile_cost);
    if (s->d_tile_order) (void)hipFree(s->d_tile_order);
    if (s->d_state) (void)hipFree(s->d_state);
    if (s->d_heavy_list) (void)hipFree(s->d_heavy_list);
    if (s->d_heavy_pixels) (void)hipFree(s->d_heavy_pixels);
    if (s->d_rank) (void)hipFree(s->d_rank);
    if (s->d_handoff) (void)hipFree(s->d_handoff);
    if (s->d_cal_cost) (void)hipFree(s->d_cal_cost);
    if (s->tier_stream) (void)hipStreamDestroy(s->tier_stream);
    for (int k = 0; k < 4; ++k) { if (s->ev_fork[k]) (void)hipEventDestroy(s->ev_fork[k]); if (s->ev_join[k]) (void)hipEventDestroy(s->ev_join[k]); }
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
    delete s;
    return RT_OK;
}

namespace {
// Pass counts of the reference's nodes on a small frame through the scene's own camera (kernel 0 with its counters on).
// keep_cost: the pass's per-pixel ray counts stay in the scene as the cost prior of ranked first parts (rt_prior_kernel)
rt_status measure_pass_counts(rt_scene* s, const rt_node* d_tree, int n_tree, std::vector<double>& pass, double& rays, std::vector<float>* sample = nullptr, uint32_t sample_stride = 0, bool keep_cost = false) {
    const rt_camera& c = s->dev.camera;
    const double hw = sqrt((double)c.horizontal[0] * c.horizontal[0] + (double)c.horizontal[1] * c.horizontal[1] + (double)c.horizontal[2] * c.horizontal[2]);
    const double vh = sqrt((double)c.vertical[0] * c.vertical[0] + (double)c.vertical[1] * c.vertical[1] + (double)c.vertical[2] * c.vertical[2]);
    double aspect = (hw > 0 && vh > 0) ? hw / vh : 1.0;
    if (!(aspect > 0.05 && aspect < 20.0)) aspect = 1.0;
    int nx = aspect >= 1.0 ? 256 : (int)(256 * aspect + 0.5), ny = aspect >= 1.0 ? (int)(256 / aspect + 0.5) : 256;
    if (nx < 8) nx = 8;
    if (ny < 8) ny = 8;
    const int n = n_tree;
    rt_scene_dev dev = s->dev;
    dev.nodes_ref = d_tree; dev.n_nodes_ref = n_tree;          // the tree kernel 0 walks in this pass
    unsigned int* d_pass = nullptr;
    unsigned int* d_cost = nullptr;
    float* d_fb = nullptr;
    HIPCHK(hipMalloc((void**)&d_pass, (size_t)n * sizeof(unsigned int)));
    hipError_t e = hipMalloc((void**)&d_fb, (size_t)nx * ny * 3 * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(d_pass); HIPCHK(e); }
    if (keep_cost && hipMalloc((void**)&d_cost, (size_t)nx * ny * sizeof(unsigned int)) != hipSuccess) d_cost = nullptr;   // no prior then
    rt_frame_params fp;
    memset(&fp, 0, sizeof(fp));
    fp.fb = d_fb; fp.ray_counter = s->d_ray_counter; fp.work_counter = s->d_work_counter; fp.node_pass = d_pass; fp.pixel_cost = d_cost;
    fp.node_pass_lds = (size_t)n * sizeof(unsigned int) <= 48u * 1024u ? 1 : 0;
    fp.seed_base = 1984; fp.nx = nx; fp.ny = ny; fp.ns = 4; fp.gamma = 1.0f;
    fp.tile_rows = ny; fp.tile_first = 0; fp.tile_stride = 1; fp.local_rows = ny;
    fp.tiles_x = (nx + 7) / 8;
    fp.work_items = (uint32_t)fp.tiles_x * (uint32_t)((ny + 7) / 8) * 64u;
    fp.sample_begin = 0; fp.sample_end = fp.ns;
    rt_status st = RT_OK;
    std::vector<unsigned int> h((size_t)n);
    unsigned long long r = 0;
    enum { SAMPLE_CAP = 8192 };
    float* d_sample = nullptr;
    if (sample && sample_stride > 0) {
        // every sample_stride-th ray of the pass (the caller knows how many there will be and keeps that below the
        // capacity, so the SET of sampled rays does not depend on the order the lanes get there)
        if (hipMalloc((void**)&d_sample, (size_t)SAMPLE_CAP * 7 * sizeof(float)) == hipSuccess) {
            fp.ray_sample = d_sample; fp.ray_sample_cap = SAMPLE_CAP; fp.ray_sample_stride = sample_stride;
        }
    }
    do {
        if ((e = hipMemset(d_pass, 0, (size_t)n * sizeof(unsigned int))) != hipSuccess) break;
        if ((e = hipMemset(s->d_ray_counter, 0, RT_COUNTER_BYTES)) != hipSuccess) break;
        if ((e = rt_launch_pixel(s->spheres_only, s->tex_level, s->need_uv, dev, fp, dim3((fp.work_items + 255u) / 256u), dim3(256), nullptr)) != hipSuccess) break;
        if ((e = hipDeviceSynchronize()) != hipSuccess) break;
        if ((e = hipMemcpy(h.data(), d_pass, (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(&r, s->d_ray_counter, sizeof(r), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if (d_sample) {
            unsigned long long taken = 0;
            if ((e = hipMemcpy(&taken, s->d_ray_counter + 2, sizeof(taken), hipMemcpyDeviceToHost)) != hipSuccess) break;
            if (taken > SAMPLE_CAP) taken = SAMPLE_CAP;
            sample->resize((size_t)taken * 7);
            if (taken && (e = hipMemcpy(sample->data(), d_sample, (size_t)taken * 7 * sizeof(float), hipMemcpyDeviceToHost)) != hipSuccess) break;
        }
    } while (0);
    (void)hipFree(d_pass); (void)hipFree(d_fb);
    if (d_sample) (void)hipFree(d_sample);
    if (d_cost) {
        if (e == hipSuccess) { if (s->d_cal_cost) (void)hipFree(s->d_cal_cost); s->d_cal_cost = d_cost; s->cal_nx = nx; s->cal_ny = ny; }
        else (void)hipFree(d_cost);
    }
    if (e != hipSuccess) { g_last_hip_error = (int)e; g_detail = std::string("calibration pass: ") + hipGetErrorString(e); st = RT_ERR_HIP; }
    pass.assign((size_t)n, 0.0);
    for (int i = 0; i < n; ++i) pass[i] = (double)h[i];
    rays = (double)r;
    return st;
}

// a node array in the device's link encoding (rt_device.h, RT_NODE_SKIP)
void device_nodes(const rt_node* nodes, size_t n, std::vector<rt_node>& out) {
    out.assign(nodes, nodes + n);
    for (size_t i = 0; i < n; ++i) {
        out[i].skip = ~out[i].skip;
        if (out[i].prim < 0) out[i].prim = ~(int32_t)(i + 1);
    }
}

// Tier data (rt_kernel_tier.h): the scene's leaves -- the reference's single-object nodes, in its depth-first order, which
// every walk array keeps -- as two float4 arrays padded to whole 64-leaf slots, and per slot the union of its boxes.  A scene
// with more than 64 slots (one lane tests one slot's union) or more than two constant_medium leaves gets none: it renders
// without a tier kernel.
rt_status build_tier_data(rt_scene* s, const rt_scene_desc* d) {
    s->dev.leaf_lo = s->dev.leaf_hi = nullptr; s->dev.slot_ranges = nullptr;
    s->dev.n_leaves = s->dev.n_slots = s->dev.n_media_leaves = 0;
    s->dev.media_ord[0] = s->dev.media_ord[1] = 0x7FFFFFFF;
    std::vector<const rt_node*> leaves;
    for (int i = 0; i < d->n_nodes; ++i) if (d->nodes[i].prim >= 0) leaves.push_back(&d->nodes[i]);
    const int m = (int)leaves.size();
    if (m == 0 || m > 64 * 64) return RT_OK;
    int media = 0, media_ord[2] = {0x7FFFFFFF, 0x7FFFFFFF};
    for (int q = 0; q < m; ++q)
        if (RT_PRIM_KIND(leaves[q]->prim) == RT_PRIM_MEDIUM) { if (media < 2) media_ord[media] = q; ++media; }
    if (media > 2) return RT_OK;
    const int slots = (m + 63) / 64;
    std::vector<float> lo((size_t)slots * 64 * 4), hi((size_t)slots * 64 * 4), ranges((size_t)slots * 8, 0.0f);
    for (int k = 0; k < slots; ++k) {
        float rl[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rh[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int l = 0; l < 64; ++l) {
            const int q = k * 64 + l;
            float* a = &lo[(size_t)q * 4];
            float* b = &hi[(size_t)q * 4];
            // lo.w = what a lane tests: the leaf's sphere or quad, for a box (compound6, quad.cuh:124-139) its FIRST face, for an
            // instance the same of its child; a medium stays itself.  hi.w = instance index + 1 (0 = none) | bit 30: "six faces
            // from lo.w on" (rt_kernel_tier.h shares those out over six lanes)
            int32_t prim = -1, xw = 0;
            if (q < m) {
                for (int c = 0; c < 3; ++c) { a[c] = leaves[q]->bmin[c]; b[c] = leaves[q]->bmax[c]; rl[c] = fminf(rl[c], a[c]); rh[c] = fmaxf(rh[c], b[c]); }
                prim = leaves[q]->prim;
                if (RT_PRIM_KIND(prim) == RT_PRIM_INSTANCE) {
                    const int idx = RT_PRIM_INDEX(prim);
                    if (idx >= d->n_instances || idx >= (1 << 29)) return RT_OK;   // (rt_scene_create has validated the refs; no tier data otherwise)
                    xw = idx + 1;
                    prim = d->instances[idx].child;
                }
                if (RT_PRIM_KIND(prim) == RT_PRIM_BOX) {
                    const int idx = RT_PRIM_INDEX(prim);
                    if (idx >= d->n_boxes) return RT_OK;
                    prim = RT_PRIM_REF(RT_PRIM_QUAD, d->boxes[idx].first_quad & 0x3FFFFFFF);
                    xw |= 1 << 30;
                }
            } else {
                for (int c = 0; c < 3; ++c) { a[c] = 0.0f; b[c] = 0.0f; }
            }
            memcpy(&a[3], &prim, 4);
            memcpy(&b[3], &xw, 4);
        }
        for (int c = 0; c < 3; ++c) { ranges[(size_t)k * 8 + c] = rl[c]; ranges[(size_t)k * 8 + 3 + c] = rh[c]; }
    }
    const float* d_lo = nullptr; const float* d_hi = nullptr; const float* d_ranges = nullptr;
    rt_status st = upload(lo.data(), lo.size(), &d_lo);
    if (st != RT_OK) return st;
    s->allocs.push_back(const_cast<float*>(d_lo));
    st = upload(hi.data(), hi.size(), &d_hi);
    if (st != RT_OK) return st;
    s->allocs.push_back(const_cast<float*>(d_hi));
    st = upload(ranges.data(), ranges.size(), &d_ranges);
    if (st != RT_OK) return st;
    s->allocs.push_back(const_cast<float*>(d_ranges));
    s->dev.leaf_lo = reinterpret_cast<const float4*>(d_lo); s->dev.leaf_hi = reinterpret_cast<const float4*>(d_hi); s->dev.slot_ranges = d_ranges;
    s->dev.n_leaves = m; s->dev.n_slots = slots; s->dev.n_media_leaves = media;
    s->dev.media_ord[0] = media_ord[0]; s->dev.media_ord[1] = media_ord[1];
    return RT_OK;
}

// A walk array this small is not walked but scanned in lockstep (lds_mode 4): every lane of a wave steps through every node
// whatever its own boxes said, so an interior node can only cost.  The leaves alone, in their order, are the array then (the
// Cornell box: 11 -> 10 nodes).  Returns whether `walk` was replaced.
bool leaves_only_if_scanned(const rt_node* src, int m, std::vector<rt_node>& walk) {
    if (g_opt.scan_nodes <= 0 || m > g_opt.scan_nodes) return false;
    std::vector<rt_node> leaves;
    for (int i = 0; i < m; ++i)
        if (src[i].prim >= 0) { rt_node w = src[i]; w.skip = (int32_t)leaves.size() + 1; leaves.push_back(w); }
    if (leaves.empty() || (int)leaves.size() >= m) return false;
    walk.swap(leaves);
    return true;
}

// builds the walk array (see "Collapse" above) and points dev.nodes at it
rt_status build_walk(rt_scene* s, const rt_scene_desc* d) {
    const int n = d->n_nodes;
    s->walk_tests_before = s->walk_tests_after = 0.0;
    if (g_opt.bvh_collapse == 0 || n < 3) return RT_OK;
    std::vector<double> pass;
    std::vector<float> ray_sample;
    double root_visits = 0.0;
    if (g_opt.bvh_collapse >= 2) {
        const rt_status st = measure_pass_counts(s, s->dev.nodes_ref, n, pass, root_visits, nullptr, 0, /*keep_cost=*/true);
        if (st != RT_OK) return st;
    }
    const bool measured = root_visits > 0.0;
    if (!measured) {   // by surface area: a ray that passes a box passes a box inside it about in proportion to the areas
        pass.assign((size_t)n, 0.0);
        for (int i = 0; i < n; ++i) {
            const double ex = fmax(0.0, (double)d->nodes[i].bmax[0] - d->nodes[i].bmin[0]), ey = fmax(0.0, (double)d->nodes[i].bmax[1] - d->nodes[i].bmin[1]),
                         ez = fmax(0.0, (double)d->nodes[i].bmax[2] - d->nodes[i].bmin[2]);
            pass[i] = 2.0 * (ex * ey + ey * ez + ex * ez);
        }
        root_visits = pass[0];
    }
    collapse_plan plan;
    const bool planned = plan_collapse(d->nodes, n, pass, root_visits, plan);
    std::vector<rt_node> walk;
    if (planned) walk = build_walk_array(d->nodes, n, plan.keep);
    bool changed = planned && (int)walk.size() != n;
    // the regrouped hierarchy over the same leaves ("Regroup" above), through the same calibration pass and DP; it needs
    // the reference's own tree to have passed the checks (a real tree, boxes containing their children's)
    if (g_opt.bvh_collapse >= 3 && measured && planned) {
        // bounds on the planning time of very large scenes (the reference's have at most 1410 leaves): the top-down split is
        // quadratic in the worst case, the sampled-ray key costs leaves x rays
        int n_leaves = 0;
        for (int i = 0; i < n; ++i) n_leaves += d->nodes[i].prim >= 0 ? 1 : 0;
        for (int method = 0; method < 3; ++method) {
            if (method == 0 && n_leaves > 65536) continue;
            if (method == 2 && (ray_sample.size() < 7 * 256 || n_leaves > 8192)) continue;
            const std::vector<rt_node> tree = regroup_leaves(d->nodes, n, method, &ray_sample);
            const int m = (int)tree.size();
            if (m < 3) continue;
            std::vector<rt_node> enc;
            device_nodes(tree.data(), tree.size(), enc);
            const rt_node* d_tree = nullptr;
            rt_status st = upload(enc.data(), enc.size(), &d_tree);
            if (st != RT_OK) return st;
            std::vector<double> pass2;
            double rays2 = 0.0;
            // the first of these passes also keeps ~4000 of its rays for method 2 (same frame, same rays in every pass)
            st = measure_pass_counts(s, d_tree, m, pass2, rays2, method == 0 ? &ray_sample : nullptr, (uint32_t)(root_visits / 4096.0) + 1u);
            (void)hipFree(const_cast<rt_node*>(d_tree));
            if (st != RT_OK) return st;
            collapse_plan plan2;
            const bool ok2 = rays2 == root_visits && plan_collapse(tree.data(), m, pass2, rays2, plan2);
            if (ok2 && plan2.tests_after < plan.tests_after) {
                walk = build_walk_array(tree.data(), m, plan2.keep);
                plan.tests_after = plan2.tests_after;            // "before" stays the reference tree's figure
                changed = true;
            }
        }
    }
    {   // (the estimates stay the plan's: per-lane box tests of the array a walk would use)
        std::vector<rt_node> cur = changed ? walk : std::vector<rt_node>(d->nodes, d->nodes + n);
        if (leaves_only_if_scanned(cur.data(), (int)cur.size(), walk)) changed = true;
    }
    if (!changed) return RT_OK;
    const rt_node* d_walk = nullptr;
    std::vector<rt_node> enc;
    device_nodes(walk.data(), walk.size(), enc);
    const rt_status st = upload(enc.data(), enc.size(), &d_walk);
    if (st != RT_OK) return st;
    s->allocs.push_back(const_cast<void*>(static_cast<const void*>(d_walk)));
    s->dev.nodes = d_walk; s->dev.n_nodes = (int32_t)walk.size();
    s->walk_tests_before = plan.tests_before; s->walk_tests_after = plan.tests_after;
    return RT_OK;
}
}  // namespace

rt_status rt_scene_create(const rt_scene_desc* d, rt_scene** out) {
    if (!out) return invalid("null output pointer");
    *out = nullptr;
    if (g_device < 0) { g_detail = "rt_init has not succeeded"; return RT_ERR_NO_DEVICE; }
    return rt_internal_scene_create_on(g_device, d, out);
}

}  // extern "C"

// ---- internals shared with rt_multi.hip
rt_status rt_internal_init_device(int device_ordinal) {
    const rt_status st = init_device(device_ordinal);
    if (st == RT_OK) g_device = device_ordinal;
    return st;
}
void rt_internal_set_error(rt_status st, int hip_error, const std::string& detail) { (void)st; g_last_hip_error = hip_error; g_detail = detail; }
int rt_internal_option(const char* key) { return std::string(key) == "multi_force_rccl" ? g_opt.multi_force_rccl : 0; }

rt_status rt_internal_scene_create_on(int device, const rt_scene_desc* d, rt_scene** out) {
    if (!out) return invalid("null output pointer");
    *out = nullptr;
    { const rt_status ud = use_device(device); if (ud != RT_OK) return ud; }
    bool so = false, uv = false;
    int tx = 0;
    rt_status st = validate(d, so, tx, uv);
    if (st != RT_OK) return st;

#define UPSRC_spheres spheres
#define UPSRC_quads quads
#define UPSRC_boxes boxes
#define UPSRC_instances instances
#define UPSRC_media media
#define UPSRC_materials materials
#define UPSRC_textures textures
#define UPSRC_images images
    rt_scene* s = new rt_scene;
    s->device = device;
    memset(&s->dev, 0, sizeof(s->dev));
    memset(&s->pending_stats, 0, sizeof(s->pending_stats));
    s->spheres_only = so; s->tex_level = tx; s->need_uv = uv;
#define UP(field, count)                                                             \
    do {                                                                             \
        st = upload(d->UPSRC_##field, (size_t)(count), &s->dev.field);               \
        if (st != RT_OK) { rt_scene_destroy(s); return st; }                         \
        s->allocs.push_back(const_cast<void*>(static_cast<const void*>(s->dev.field))); \
    } while (0)
    {
        std::vector<rt_node> enc;
        device_nodes(d->nodes, (size_t)d->n_nodes, enc);
        st = upload(enc.data(), enc.size(), &s->dev.nodes_ref);
        if (st != RT_OK) { rt_scene_destroy(s); return st; }
        s->allocs.push_back(const_cast<void*>(static_cast<const void*>(s->dev.nodes_ref)));
    }
    UP(spheres, d->n_spheres);
    {   // quads and boxes: mark the axis-aligned ones (quad_test_axis, rt_device_funcs.h).  A quad qualifies when its unit
        // normal is exactly +-e_C and u, v, w each have exactly one non-zero component on the fitting axes; a box when its
        // six faces do, with normals along z, x, z, x, y, y (make_box's order, quad.cuh:145-162).
        std::vector<rt_quad> quads(d->quads, d->quads + d->n_quads);
        auto axis_of = [](const rt_quad& q) -> int {
            for (int c = 0; c < 3; ++c) {
                const int a = (c + 1) % 3, b = (c + 2) % 3;
                if (!((q.n[c] == 1.0f || q.n[c] == -1.0f) && q.n[a] == 0.0f && q.n[b] == 0.0f)) continue;
                if (!(q.w[a] == 0.0f && q.w[b] == 0.0f && q.w[c] != 0.0f && q.u[c] == 0.0f && q.v[c] == 0.0f)) continue;
                const bool u_on_a = q.u[a] != 0.0f && q.u[b] == 0.0f, u_on_b = q.u[b] != 0.0f && q.u[a] == 0.0f;
                const bool v_on_a = q.v[a] != 0.0f && q.v[b] == 0.0f, v_on_b = q.v[b] != 0.0f && q.v[a] == 0.0f;
                if ((u_on_a && v_on_b) || (u_on_b && v_on_a)) return c;
            }
            return -1;
        };
        for (rt_quad& q : quads) { const int c = axis_of(q); const int32_t code = c >= 0 ? 1 + c : 0; memcpy(&q.pad0, &code, 4); }
        std::vector<rt_box> boxes(d->boxes, d->boxes + d->n_boxes);
        static const int face_axis[6] = {2, 0, 2, 0, 1, 1};
        for (rt_box& b : boxes) {
            bool canonical = true;
            for (int f = 0; f < 6 && canonical; ++f) canonical = axis_of(quads[(size_t)b.first_quad + f]) == face_axis[f];
            if (canonical) b.first_quad |= 0x40000000;
        }
        st = upload(quads.data(), quads.size(), &s->dev.quads);
        if (st != RT_OK) { rt_scene_destroy(s); return st; }
        s->allocs.push_back(const_cast<void*>(static_cast<const void*>(s->dev.quads)));
        st = upload(boxes.data(), boxes.size(), &s->dev.boxes);
        if (st != RT_OK) { rt_scene_destroy(s); return st; }
        s->allocs.push_back(const_cast<void*>(static_cast<const void*>(s->dev.boxes)));
    }
    UP(instances, d->n_instances);
    UP(media, d->n_media);
    {   // materials: `pad` tells resolve_hit() whether the material's texture reads the hit's (u, v)
        std::vector<rt_material> mats(d->materials, d->materials + d->n_materials);
        auto reads_uv = [&](int tex) -> bool {
            if (tex < 0) return false;
            const rt_texture& t = d->textures[tex];
            if (t.kind == RT_TEX_IMAGE || t.kind == RT_TEX_UV_OFFSET) return true;
            if (t.kind == RT_TEX_CHECKER) {
                const int ka = d->textures[t.a].kind, kb = d->textures[t.b].kind;   // validate(): checker children are plain textures
                return ka == RT_TEX_IMAGE || kb == RT_TEX_IMAGE;
            }
            return false;
        };
        for (rt_material& m : mats) m.pad = reads_uv(m.tex) ? 1.0f : 0.0f;
        st = upload(mats.data(), mats.size(), &s->dev.materials);
        if (st != RT_OK) { rt_scene_destroy(s); return st; }
        s->allocs.push_back(const_cast<void*>(static_cast<const void*>(s->dev.materials)));
    }
    UP(textures, d->n_textures);
    UP(images, d->image_bytes);
#undef UP
    s->dev.n_nodes_ref = d->n_nodes;
    for (int c = 0; c < 3; ++c) {   // the scene's coordinate bound per axis (regrouped interior boxes are unions of these: the same bound)
        float b = 0.0f;
        for (int i = 0; i < d->n_nodes; ++i) b = fmaxf(b, fmaxf(fabsf(d->nodes[i].bmin[c]), fabsf(d->nodes[i].bmax[c])));
        s->dev.bound[c] = b;
    }
    s->dev.bound_pad = 0.0f;
    s->dev.nodes = s->dev.nodes_ref; s->dev.n_nodes = d->n_nodes;     // until the walk array is built below
    s->dev.n_spheres = d->n_spheres;
    s->dev.n_materials = d->n_materials; s->dev.n_textures = d->n_textures;
    s->shade_bytes = (size_t)d->n_materials * sizeof(rt_material) + (size_t)d->n_textures * sizeof(rt_texture);
    s->dev.camera = d->camera;
    s->node_bytes = (size_t)d->n_nodes * sizeof(rt_node);
    s->sphere_bytes = (size_t)d->n_spheres * sizeof(rt_sphere);
    hipError_t e;
    if ((e = hipMalloc((void**)&s->d_ray_counter, RT_COUNTER_BYTES)) != hipSuccess || (e = hipMalloc((void**)&s->d_work_counter, RT_WORK_COUNTER_BYTES)) != hipSuccess ||
        (e = hipEventCreate(&s->ev_start)) != hipSuccess || (e = hipEventCreate(&s->ev_stop)) != hipSuccess) {
        g_last_hip_error = (int)e; g_detail = "allocating per-frame resources failed";
        rt_scene_destroy(s);
        return RT_ERR_HIP;
    }
    {   // the tier kernel's stream and the events that fork it from / join it to the caller's stream, one pair per ranked part
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        e = hipStreamCreateWithPriority(&s->tier_stream, hipStreamNonBlocking, prio_hi);
        for (int k = 0; k < 4 && e == hipSuccess; ++k) {
            e = hipEventCreateWithFlags(&s->ev_fork[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) { g_last_hip_error = (int)e; g_detail = "creating the tier stream failed"; rt_scene_destroy(s); return RT_ERR_HIP; }
    }
    st = build_tier_data(s, d);
    if (st != RT_OK) { rt_scene_destroy(s); return st; }
    st = build_walk(s, d);
    if (st != RT_OK) { rt_scene_destroy(s); return st; }
    s->node_bytes = (size_t)s->dev.n_nodes * sizeof(rt_node);
    *out = s;
    return RT_OK;
}

extern "C" {

// Host-only: the walk array rt_scene_create would build for `nodes` given per-node pass counts (`pass`, one per node;
// null = proportional to box surface area) and `root_visits` rays.  Writes at most `cap` nodes to `out`, returns the
// walk array's size through `n_out` and the expected box tests per ray before / after.  No device involved.
rt_status rt_plan_walk_array(const rt_node* nodes, int32_t n, const double* pass, double root_visits, rt_node* out, int32_t cap,
                             int32_t* n_out, double* tests_before, double* tests_after) {
    if (!nodes || n <= 0 || !n_out) return invalid("rt_plan_walk_array: bad argument");
    for (int i = 0; i < n; ++i) if (nodes[i].skip <= i || nodes[i].skip > n) return invalid("node skip link does not move forward");
    std::vector<double> p((size_t)n, 0.0);
    if (pass) p.assign(pass, pass + n);
    else {
        for (int i = 0; i < n; ++i) {
            const double ex = fmax(0.0, (double)nodes[i].bmax[0] - nodes[i].bmin[0]), ey = fmax(0.0, (double)nodes[i].bmax[1] - nodes[i].bmin[1]),
                         ez = fmax(0.0, (double)nodes[i].bmax[2] - nodes[i].bmin[2]);
            p[i] = 2.0 * (ex * ey + ey * ez + ex * ez);
        }
        root_visits = p[0];
    }
    collapse_plan plan;
    std::vector<rt_node> walk(nodes, nodes + n);
    if (plan_collapse(nodes, n, p, root_visits, plan)) walk = build_walk_array(nodes, n, plan.keep);
    *n_out = (int32_t)walk.size();
    if (tests_before) *tests_before = plan.tests_before;
    if (tests_after) *tests_after = plan.tests_after;
    if (out) for (int i = 0; i < (int)walk.size() && i < cap; ++i) out[i] = walk[i];
    return RT_OK;
}

rt_status rt_regroup_leaves(const rt_node* nodes, int32_t n, int32_t method, rt_node* out, int32_t cap, int32_t* n_out) {
    if (!nodes || n <= 0 || !n_out || method < 0 || method > 1) return invalid("rt_regroup_leaves: bad argument");
    const std::vector<rt_node> tree = regroup_leaves(nodes, n, method);
    *n_out = (int32_t)tree.size();
    if (out) for (int i = 0; i < (int)tree.size() && i < cap; ++i) out[i] = tree[i];
    return RT_OK;
}

rt_status rt_scene_walk_info(const rt_scene* s, int32_t* nodes_reference, int32_t* nodes_walked, double* tests_before, double* tests_after) {
    if (!s) return invalid("null scene");
    if (nodes_reference) *nodes_reference = s->dev.n_nodes_ref;
    if (nodes_walked) *nodes_walked = s->dev.n_nodes;
    if (tests_before) *tests_before = s->walk_tests_before;
    if (tests_after) *tests_after = s->walk_tests_after;
    return RT_OK;
}

int32_t rt_frame_local_rows(const rt_frame_desc* f) {
    if (!f || f->ny <= 0 || f->tile_rows <= 0 || f->tile_stride <= 0 || f->tile_first < 0) return -1;
    const int n_tiles = (f->ny + f->tile_rows - 1) / f->tile_rows;
    int rows = 0;
    for (int t = f->tile_first; t < n_tiles; t += f->tile_stride) {
        const int r0 = t * f->tile_rows;
        const int r1 = r0 + f->tile_rows < f->ny ? r0 + f->tile_rows : f->ny;
        rows += r1 - r0;
    }
    return rows;
}

int32_t rt_local_to_global_row(const rt_frame_desc* f, int32_t local_row) {
    if (!f || f->tile_rows <= 0) return -1;
    const int t = local_row / f->tile_rows;
    return (f->tile_first + t * f->tile_stride) * f->tile_rows + (local_row - t * f->tile_rows);
}

rt_status rt_frame_finish(rt_scene* s, rt_stats* stats) {
    if (!s) return invalid("null scene");
    if (!s->frame_pending) { if (stats) *stats = s->pending_stats; return RT_OK; }
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    HIPCHK(hipEventSynchronize(s->ev_stop));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
    unsigned long long rays = 0;
    HIPCHK(hipMemcpy(&rays, s->d_ray_counter, sizeof(rays), hipMemcpyDeviceToHost));
    s->pending_stats.ms_render = (double)ms;
    s->pending_stats.rays = rays;
    if (s->ranked_frame && s->d_rank) {   // how many pixels the last ranking listed as heavy (diagnostics)
        rt_rank_info inf;
        HIPCHK(hipMemcpy(&inf, s->d_rank, sizeof(inf), hipMemcpyDeviceToHost));
        s->pending_stats.reserved = (int32_t)inf.heavy_items;
    }
    s->frame_pending = false;
    if (stats) *stats = s->pending_stats;
    return RT_OK;
}

// Tail hand-off of the last frame (diagnostics, every build): [0] pixels the main kernel handed to the tail launches, summed over
// the frame's parts, [1] the samples those pixels still had to go.
rt_status rt_debug_handoff(rt_scene* s, unsigned long long* out2) {
    if (!s || !out2) return invalid("null argument");
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    HIPCHK(hipMemcpy(out2, s->d_ray_counter + 28, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}

// Diagnostic builds (-DRT_DIAG) leave per-stage execution counts behind the ray counter; 16 values.
rt_status rt_debug_counters(rt_scene* s, unsigned long long* out16) {
    if (!s || !out16) return invalid("null argument");
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    HIPCHK(hipMemcpy(out16, s->d_ray_counter + 1, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}
// Diagnostic builds: cycles the waves of the frame spent per part of the staged kernel's loop, summed over waves
// ([0] box steps, [1] object tests, [2] stage C, [3] D, [4] E, [6] stage gating, [7] stage F + loop; then, first wave of each
// tier workgroup: [8] resolve + shade in the tier loops, [9] the whole tier loop; shader-clock cycles).
rt_status rt_debug_stage_cycles(rt_scene* s, unsigned long long* out10) {
    if (!s || !out10) return invalid("null argument");
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    HIPCHK(hipMemcpy(out10, s->d_ray_counter + 17, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}
// Diagnostic builds: when the waves of the LAST launch ended, as two histograms of RT_DIAG_BINS bins of 1 ms after the first
// wave's start (ordinary waves, then waves that started in sparse / tier mode).
rt_status rt_debug_wave_ends(rt_scene* s, unsigned long long* out, int n) {
    if (!s || !out || n < 0 || n > 2 * RT_DIAG_BINS) return invalid("bad argument");
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    HIPCHK(hipMemcpy(out, s->d_ray_counter + RT_DIAG_HIST_SLOT, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}
// ... and per wave of the last launch, two words about the lane that ran out of work last: (done_us << 32 | fetch_us of its last
// pixel), (source queue << 60 | started sparse << 59 | local pixel id)
rt_status rt_debug_wave_last(rt_scene* s, unsigned long long* out, int n_waves) {
    if (!s || !out || n_waves < 0 || n_waves > RT_DIAG_MAX_WAVES) return invalid("bad argument");
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    HIPCHK(hipMemcpy(out, s->d_ray_counter + RT_DIAG_WAVE_SLOT, (size_t)2 * n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}

static rt_status render_impl(rt_scene* s, const rt_frame_desc* f, float* fb, int fb_on_device, void* stream_v, int blocking, rt_stats* stats,
                             rt_progressive* win, int32_t win_begin, int32_t win_end);

rt_status rt_render(rt_scene* s, const rt_frame_desc* f, float* fb, int fb_on_device, void* stream_v, int blocking, rt_stats* stats) {
    return render_impl(s, f, fb, fb_on_device, stream_v, blocking, stats, nullptr, 0, 0);
}

rt_status rt_progressive_state_create(rt_scene* s, const rt_frame_desc* f, void** state) {
    if (!s || !f || !state) return invalid("null argument");
    *state = nullptr;
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    if (f->nx <= 0 || f->ny <= 0 || (long long)f->nx * f->ny >= (1ll << 31)) return invalid("bad frame size");
    const int local_rows = rt_frame_local_rows(f);
    if (local_rows < 0) return invalid("bad row partition");
    rt_progressive* p = new rt_progressive;
    p->scene = s; p->frame = *f; p->pixels = (size_t)local_rows * (size_t)f->nx; p->next_sample = 0;
    const hipError_t e = hipMalloc((void**)&p->d_state, (p->pixels ? p->pixels : 1) * sizeof(rt_pixel_state));
    if (e != hipSuccess) { delete p; g_last_hip_error = (int)e; g_detail = "allocating the progressive state failed"; return RT_ERR_HIP; }
    *state = p;
    return RT_OK;
}

rt_status rt_progressive_state_destroy(rt_scene* s, void* state) {
    if (!state) return RT_OK;
    rt_progressive* p = static_cast<rt_progressive*>(state);
    if (s) (void)use_device(s->device);
    if (p->d_state) (void)hipFree(p->d_state);
    delete p;
    return RT_OK;
}

rt_status rt_render_window(rt_scene* s, const rt_frame_desc* f, float* fb, int fb_on_device, void* state, int32_t sample_begin, int32_t sample_end,
                           void* stream_v, int blocking, rt_stats* stats) {
    if (!s || !f || !fb || !state) return invalid("null argument");
    rt_progressive* p = static_cast<rt_progressive*>(state);
    if (p->scene != s) return invalid("rt_render_window: the state belongs to another scene");
    if (f->nx != p->frame.nx || f->ny != p->frame.ny || f->tile_rows != p->frame.tile_rows || f->tile_first != p->frame.tile_first ||
        f->tile_stride != p->frame.tile_stride || f->seed_base != p->frame.seed_base)
        return invalid("rt_render_window: the frame description differs from the one the state was created for");
    if (sample_begin != p->next_sample || sample_end <= sample_begin) return invalid("rt_render_window: windows must follow each other (sample_begin = the previous sample_end, 0 first)");
    const rt_status st = render_impl(s, f, fb, fb_on_device, stream_v, blocking, stats, p, sample_begin, sample_end);
    if (st == RT_OK) p->next_sample = sample_end;
    return st;
}

static rt_status render_impl(rt_scene* s, const rt_frame_desc* f, float* fb, int fb_on_device, void* stream_v, int blocking, rt_stats* stats,
                             rt_progressive* win, int32_t win_begin, int32_t win_end) {
    if (!s || !f || !fb) return invalid("null argument");
    { const rt_status ud = use_device(s->device); if (ud != RT_OK) return ud; }
    const int g_num_cu = g_devices[s->device].num_cu;
    const size_t g_lds_per_cu = g_devices[s->device].lds_per_cu;
    const int frame_ns = win ? win_end : f->ns;       // a progressive window averages over the samples rendered so far
    if (f->nx <= 0 || f->ny <= 0 || frame_ns <= 0) return invalid("nx, ny and ns must be positive");
    if ((long long)f->nx * f->ny >= (1ll << 31)) return invalid("frame too large");
    const int local_rows = rt_frame_local_rows(f);
    if (local_rows < 0) return invalid("bad row partition");
    if (s->frame_pending) { rt_status st = rt_frame_finish(s, nullptr); if (st != RT_OK) return st; }
    hipStream_t stream = static_cast<hipStream_t>(stream_v);

    rt_stats out;
    memset(&out, 0, sizeof(out));
    out.local_rows = local_rows;
    out.samples = (uint64_t)local_rows * f->nx * (uint64_t)(win ? win_end - win_begin : f->ns);
    if (local_rows == 0) { s->pending_stats = out; if (stats) *stats = out; return RT_OK; }

    rt_frame_params fp;
    memset(&fp, 0, sizeof(fp));
    const size_t floats = (size_t)local_rows * f->nx * 3;
    if (fb_on_device) fp.fb = fb;
    else {
        if (s->d_fb_floats < floats) {
            if (s->d_fb) (void)hipFree(s->d_fb);
            s->d_fb = nullptr; s->d_fb_floats = 0;
            HIPCHK(hipMalloc((void**)&s->d_fb, floats * sizeof(float)));
            s->d_fb_floats = floats;
        }
        fp.fb = s->d_fb;
    }
    fp.ray_counter = s->d_ray_counter;
    fp.work_counter = s->d_work_counter;
    fp.seed_base = f->seed_base;
    fp.nx = f->nx; fp.ny = f->ny; fp.ns = frame_ns; fp.gamma = f->gamma;
    fp.background[0] = f->background[0]; fp.background[1] = f->background[1]; fp.background[2] = f->background[2];
    fp.use_gradient_bg = f->use_gradient_bg;
    fp.tile_rows = f->tile_rows; fp.tile_first = f->tile_first; fp.tile_stride = f->tile_stride;
    fp.local_rows = local_rows;
    fp.tiles_x = (f->nx + 7) / 8;
    const int tiles_y = (local_rows + 7) / 8;
    if ((long long)fp.tiles_x * tiles_y * 64 >= (1ll << 31)) return invalid("frame too large");
    fp.work_items = (uint32_t)fp.tiles_x * (uint32_t)tiles_y * 64u;
    fp.sparse_priority = g_opt.sparse_priority; fp.sparse_eager = g_opt.sparse_eager; fp.semi_priority = g_opt.semi_priority; fp.tier_priority = g_opt.tier_priority;
    fp.steps_per_trip = g_opt.steps_per_trip;
    fp.leaf_threshold = g_opt.leaf_threshold;
    fp.diel_threshold = g_opt.diel_threshold;
    fp.box_threshold = g_opt.box_threshold; fp.medium_threshold = g_opt.medium_threshold;

    // LDS residency: nodes + spheres in every workgroup of a CU if they fit that many times (2 workgroups for the lean
    // spheres-only kernels, 3 otherwise: see the workgroup shapes below), else once (one big workgroup per CU), else nodes only
    const int kernel = win ? RT_KERNEL_STAGED : g_opt.kernel;   // (kernel 0 renders whole pixels only)
    int lds_mode = g_opt.lds_mode;
    const bool lean_family = s->spheres_only && s->tex_level < 2;
    const size_t budget1 = g_lds_per_cu - 2048;   // one workgroup per CU
    if (lds_mode < 0) {
        // nodes + spheres where they fit a CU at all (several workgroups each with its own image, or one big workgroup
        // sharing one: see the workgroup shapes below), else nodes only, else everything through L1 / L2
        // (lds_mode 3 -- materials and textures in LDS too -- is selectable but measured no faster: profiles/r01_sweep34)
        if (s->node_bytes + s->sphere_bytes <= budget1) lds_mode = 2;
        else if (s->node_bytes <= budget1) lds_mode = 1;
        else lds_mode = 0;
    }
    if (g_opt.lds_mode < 0 && kernel == RT_KERNEL_STAGED && g_opt.scan_nodes > 0 && s->dev.n_nodes <= g_opt.scan_nodes) lds_mode = 4;   // lockstep scan of a tiny scene
    if (lds_mode >= 3 && kernel != RT_KERNEL_STAGED) lds_mode = 2;
    if (kernel == RT_KERNEL_PIXEL) lds_mode = 0;   // the cross-check kernel reads the scene through L1/L2
    size_t lds_bytes = 0;
    if (lds_mode >= 1 && lds_mode <= 3) lds_bytes += s->node_bytes;
    if (lds_mode >= 2 && lds_mode <= 3) lds_bytes += s->sphere_bytes;
    if (lds_mode == 3) lds_bytes += s->shade_bytes;
    if (lds_bytes > budget1) return invalid("requested lds_mode does not fit the CU's LDS");
    dim3 grid, block;
    int per_cu_resident = 1;   // workgroups of this launch that can be resident on one CU (persistent kernels)
    if (kernel == RT_KERNEL_PIXEL) {
        block = dim3(256);
        grid = dim3((fp.work_items + 255u) / 256u);
    } else {
        // workgroup shape per kernel family (register budgets: launch bounds, rt_device.h): the lean spheres-only
        // kernels 2 x 512 threads per CU (4 waves per SIMD), the others 3 x 256 (3 waves per SIMD; workgroups of four
        // waves, one per SIMD -- 2 x 384 threads is the same occupancy and measured 1.4x slower on the Cornell box)
        const bool lean = lean_family;
        const int fam_max_threads = lean ? RT_LEAN_MAX_THREADS : RT_HEAVY_MAX_THREADS;
        const int lds_fit = lds_bytes ? (int)(g_lds_per_cu / (lds_bytes + 512)) : 8;
        int per_cu = g_opt.wg_per_cu > 0 ? g_opt.wg_per_cu : (lean ? 2 : 3);
        int threads = g_opt.threads > 0 ? g_opt.threads : (lean ? 512 : 256);
        if (g_opt.wg_per_cu <= 0 && g_opt.threads <= 0 && lds_fit < per_cu) {
            // the scene's LDS image does not fit that many times: one workgroup with all the CU's waves shares one image
            per_cu = 1; threads = fam_max_threads;
        }
        if (lds_fit < per_cu) per_cu = lds_fit < 1 ? 1 : lds_fit;
        if (threads > fam_max_threads) threads = fam_max_threads;
        if (threads < 64) threads = 64;
        block = dim3((unsigned)threads);
        unsigned want = (unsigned)(g_num_cu * per_cu);
        const unsigned need = (fp.work_items + (unsigned)threads - 1) / (unsigned)threads;
        grid = dim3(want < need ? want : need);
        per_cu_resident = per_cu;
    }
    {   // stage quorums.  In a launch that leaves the machine mostly empty (a small share of a frame) a lane waiting for 32 others
        // to finish their walks is waiting on the frame's critical path: the quorums are halved there.
        const double pixels_per_lane = (double)local_rows * (double)f->nx / ((double)g_num_cu * (double)per_cu_resident * (double)block.x);
        // (lean family: every share of the BASELINE frames -- a whole frame is 3.3 - 3.7; the Cornell box's 1/8 share is slower with them: 172 -> 179 ms)
        const bool latency_regime = kernel == RT_KERNEL_STAGED && lean_family && pixels_per_lane < 2.75;
        fp.shade_threshold = g_opt.shade_threshold > 0 ? g_opt.shade_threshold : (latency_regime ? 16 : 32);
        fp.newpath_threshold = g_opt.newpath_threshold > 0 ? g_opt.newpath_threshold : (s->spheres_only ? (latency_regime ? 12 : 24) : 8);
    }
    // ---- the tier kernel of ranked launches (rt_kernel_tier.h): its LDS image and where its workgroups find room.
    // Lean family: the main kernel's 4 x 104 registers per SIMD leave 96 free, so ONE tier workgroup (four waves, one per
    // SIMD, 76 VGPRs) is resident on a CU beside a full main grid if the LDS left over holds its image.  Other families:
    // no register room beside a full main grid; the ranking makes main workgroups leave (main_skip_wgs) and a tier workgroup
    // has what one of them had.  The image always holds the leaf arrays; spheres, materials and textures too where all of them fit.
    bool tier_possible = false;
    size_t tier_lds = 0;
    unsigned tier_grid = 0;
    int tier_waves_per_main_wg = 0;
    fp.tier_lds_scene = 0;
    // (not for scenes scanned in lockstep, lds_mode 4: a handful of leaves, every pixel about as dear as the next -- the Cornell
    // box's 1/8 share measured 182 ms without it and 199 ms with it, profiles/r03_general_defaults.log)
    bool tier_fits = false;      // the tier kernel's image fits: enough for the tail launches (tail hand-off) even where tier 1 is not used
    if (kernel == RT_KERNEL_STAGED && (g_opt.tier_kernel || g_opt.handoff) && s->dev.leaf_lo != nullptr) {
        size_t budget;
        if (lean_family) {
            const size_t used = (size_t)per_cu_resident * (lds_bytes + 512);
            budget = g_lds_per_cu > used + 1024 ? g_lds_per_cu - used - 1024 : 0;
        } else {
            // the slot of one main workgroup, shared by the tier workgroups it holds: as many as it has groups of four waves
            const unsigned tier_wgs_per_slot = block.x / RT_TIER_THREADS > 0 ? block.x / RT_TIER_THREADS : 1u;
            budget = (g_lds_per_cu / (size_t)per_cu_resident - 1024) / tier_wgs_per_slot;
            tier_waves_per_main_wg = (int)(tier_wgs_per_slot * (RT_TIER_THREADS / 64));
        }
        const int ns_ = s->dev.n_slots, nsph = s->dev.n_spheres, nm = s->dev.n_materials, nt = s->dev.n_textures;
        if (rt_tier_lds_bytes(ns_, nsph, nm, nt, false) <= budget) {
            tier_fits = true;
            tier_possible = g_opt.tier_kernel && lds_mode != 4 && g_opt.tier1_pixels > 0;
            if (rt_tier_lds_bytes(ns_, nsph, nm, nt, true) <= budget) fp.tier_lds_scene = 1;
            tier_lds = rt_tier_lds_bytes(ns_, nsph, nm, nt, fp.tier_lds_scene != 0);
            // the tier kernel's grid is fixed before the ranking has sized the tier: what can be resident beside the main
            // grid (one workgroup per CU) and as much again queued behind it; workgroups beyond the tier's size leave at once
            tier_grid = lean_family ? (unsigned)(2 * g_num_cu) : (unsigned)(g_num_cu * per_cu_resident) * (unsigned)(tier_waves_per_main_wg / (RT_TIER_THREADS / 64)) / 2u;
            if (tier_grid < 1u) tier_grid = 1u;
        }
    }
    // the tail launch has the machine to itself: as many tier workgroups per CU as registers (launch bounds: 4 resp. 3 waves per SIMD,
    // a workgroup is one wave per SIMD) and LDS hold
    unsigned tail_grid = 0;
    if (tier_fits && g_opt.handoff && (lds_mode != 4 || g_opt.handoff_scan)) {
        const unsigned by_lds = (unsigned)(g_lds_per_cu / (tier_lds + 512));
        const unsigned by_regs = lean_family ? 4u : 3u;
        tail_grid = (unsigned)g_num_cu * (by_lds < by_regs ? by_lds : by_regs);
        if (tail_grid < 1u) tail_grid = 1u;
    }
    out.kernel_variant = kernel * 1000 + lds_mode * 100 + s->tex_level * 10 + (s->spheres_only ? 1 : 0);
    out.workgroups = (int)grid.x; out.threads_per_group = (int)block.x; out.lds_bytes = (int)lds_bytes;

    HIPCHK(hipEventRecord(s->ev_start, stream));
    // ---- cost-aware schedule (staged kernel): the frame is split at sample boundaries.
    // A pixel's samples are one sequential chain (one XORWOW stream) and the dearest pixels of a frame trace ~10x the
    // mean number of rays, so the frame time is bounded by a few pixels' chains, not by throughput
    // (tools/critical_chain.py: the worst 8 rows alone take as long as the whole frame).  Every part but the last parks
    // each pixel at its end (XORWOW state, colour sum, rays traced: at a sample boundary no path is in flight, so that is
    // the whole state), and every part is RANKED on what is known about the pixels' costs when it starts -- the
    // calibration frame's prior for the first part, measured rays for the later ones:
    //   * 8x8 tiles are served in descending cost (longest first);
    //   * pixels far above the mean go to a short list, dearest first: tier 1 to the tier kernel (rt_kernel_tier.h, one
    //     pixel per wave, on a stream of its own beside the main kernel), tier 2 to "sparse" main workgroups with a few live
    //     lanes per wave at raised priority -- a lane's rays advance about twice as fast in a wave with few live lanes --,
    //     tier 3 to ordinary lanes before any tile; ordinary waves skip listed pixels.
    //   * the end of every part is handed over (rt_device.h, "tail hand-off"): when the tile queue is dry and only a few thousand
    //     pixels are still in flight, the main kernel's lanes park them at their next sample boundary and a second launch of
    //     the tier kernel, right behind the main kernel on the same stream, finishes them one per wave -- the last pixels of a
    //     part are its cheapest ones started last, each still a chain of hundreds of rays at an ordinary lane's pace.
    // The ranking runs on the device (rt_rank.hip) and leaves the tier sizes in device memory, so the whole frame is
    // enqueued without a host round trip.  Scheduling only: every sample of every pixel is rendered exactly once, in
    // its pixel's stream order; frames are bit-identical with and without it (tests sweep the knobs).
    fp.tile_order = nullptr; fp.tile_cost = nullptr; fp.state_out = nullptr; fp.state_in = nullptr; fp.heavy_pixels = nullptr; fp.rank = nullptr;
    fp.sample_begin = 0; fp.sample_end = frame_ns; fp.fresh = 0; fp.store_parked = 0;
    const size_t n_tiles = (size_t)fp.tiles_x * (size_t)tiles_y;
    const size_t n_pixels = (size_t)local_rows * (size_t)f->nx;
    enum { RT_HEAVY_CAP = 262144 };
    s->ranked_frame = false;
    HIPCHK(hipMemsetAsync(s->d_ray_counter, 0, 256, stream));
    int part_index = 0;
    // one part of the frame: the tier kernel (ranked parts of scenes that have tier data) on its own stream, forked from
    // and joined to the caller's stream by events, and the main kernel
    auto launch_part = [&](const rt_frame_params& q, dim3 grid_q, bool ranked) -> rt_status {
        HIPCHK(hipMemsetAsync(s->d_work_counter, 0, RT_WORK_COUNTER_BYTES, stream));
        const bool tiers = ranked && tier_possible;
        const int pi = part_index++ & 3;
        if (tiers) {
            HIPCHK(hipEventRecord(s->ev_fork[pi], stream));
            HIPCHK(hipStreamWaitEvent(s->tier_stream, s->ev_fork[pi], 0));
            HIPCHK(s->spheres_only ? rt_launch_tier_spheres(s->tex_level, s->dev, q, dim3(tier_grid), tier_lds, s->tier_stream)
                                   : rt_launch_tier_general(s->tex_level, s->need_uv, s->dev, q, dim3(tier_grid), tier_lds, s->tier_stream));
            HIPCHK(hipEventRecord(s->ev_join[pi], s->tier_stream));
        }
        HIPCHK(launch_render(kernel, lds_mode, s, q, grid_q, block, lds_bytes, stream));
        if (q.handoff_queue) {
            // the tail: the pixels the main kernel handed off, one per wave, on whatever the tier kernel -- which may still be
            // running on its own stream: other pixels, other queue head -- leaves free of the machine
            rt_frame_params t = q;
            t.tail_mode = 1;
            HIPCHK(s->spheres_only ? rt_launch_tier_spheres(s->tex_level, s->dev, t, dim3(tail_grid), tier_lds, stream)
                                   : rt_launch_tier_general(s->tex_level, s->need_uv, s->dev, t, dim3(tail_grid), tier_lds, stream));
        }
        if (tiers) HIPCHK(hipStreamWaitEvent(stream, s->ev_join[pi], 0));
        return RT_OK;
    };
    if (win) {
        // a progressive window: one launch, every pixel resumed from (first window: started in) and parked into the caller's state,
        // and written to fb as the average so far
        fp.state_in = win_begin > 0 ? win->d_state : nullptr; fp.state_out = win->d_state;
        fp.sample_begin = win_begin; fp.sample_end = win_end; fp.store_parked = 1;
    } else if (g_opt.lpt && kernel == RT_KERNEL_STAGED && f->ns >= 2 * g_opt.split_samples && n_tiles >= 64 && n_pixels < (1ull << 31)) {
        if (s->tile_capacity < n_tiles || s->pixel_capacity < n_pixels || !s->d_rank) {
            for (void* p : {(void*)s->d_tile_cost, (void*)s->d_tile_order, (void*)s->d_state, (void*)s->d_heavy_list, (void*)s->d_heavy_pixels, (void*)s->d_rank})
                if (p) (void)hipFree(p);
            s->d_tile_cost = s->d_tile_order = s->d_heavy_pixels = nullptr; s->d_state = nullptr; s->d_heavy_list = nullptr; s->d_rank = nullptr;
            s->tile_capacity = s->pixel_capacity = 0;
            HIPCHK(hipMalloc((void**)&s->d_tile_cost, n_tiles * sizeof(unsigned int)));
            HIPCHK(hipMalloc((void**)&s->d_tile_order, n_tiles * sizeof(unsigned int)));
            HIPCHK(hipMalloc((void**)&s->d_state, n_pixels * sizeof(rt_pixel_state)));
            HIPCHK(hipMalloc((void**)&s->d_heavy_list, (size_t)RT_HEAVY_CAP * sizeof(unsigned long long)));
            HIPCHK(hipMalloc((void**)&s->d_heavy_pixels, (size_t)RT_HEAVY_CAP * sizeof(unsigned int)));
            HIPCHK(hipMalloc((void**)&s->d_rank, sizeof(rt_rank_info)));
            s->tile_capacity = n_tiles; s->pixel_capacity = n_pixels;
        }
        if (tail_grid > 0) {
            // tail hand-off (rt_device.h): a lane hands off at most one pixel per launch, so the queue holds one entry per resident lane
            const size_t lanes = (size_t)g_num_cu * (size_t)per_cu_resident * (size_t)block.x;
            if (s->handoff_capacity < lanes) {
                if (s->d_handoff) (void)hipFree(s->d_handoff);
                s->d_handoff = nullptr; s->handoff_capacity = 0;
                HIPCHK(hipMalloc((void**)&s->d_handoff, lanes * sizeof(unsigned long long)));
                s->handoff_capacity = lanes;
            }
            fp.handoff_queue = s->d_handoff; fp.handoff_cap = (uint32_t)s->handoff_capacity; fp.handoff_state = s->d_state;
            fp.handoff_poll_ticks = g_opt.handoff_poll_us * 100;
            // auto: six pixels per wave of the tail launch (the headline frame: 18432 of 960000; 8192 .. 32768 measure the same,
            // profiles/r03_handoff.log), and never more than 1/8 of the pixels (small frames and shares: the Cornell box's
            // 1/8 share, 45000 pixels, 178 ms without, 163 ms at 4096 - 8192, 180 ms at 16384)
            const size_t tail_waves = (size_t)tail_grid * (RT_TIER_THREADS / 64);
            const size_t cap_px = n_pixels / 8;
            fp.handoff_pixels = g_opt.handoff_pixels >= 0 ? g_opt.handoff_pixels : (int32_t)(6 * tail_waves < cap_px ? 6 * tail_waves : cap_px);
        }
        // One ranking: three small kernels order the tiles, list the heavy pixels and size the tiers for the launch
        // described by `q` (which resumes every pixel from d_state).  The grids are fixed here, before the sizes are known:
        // the workgroups the ordinary queue needs plus the most the sparse tier may take; a workgroup that finds both its
        // queues empty leaves at once.  `total` = the device counter holding the sum of the costs ranked on.
        const unsigned max_grid = (unsigned)(g_num_cu * per_cu_resident);
        auto rank_pixels = [&](rt_frame_params& q, dim3& grid_q, const unsigned long long* total) -> rt_status {
            // Effective tier sizes by the share of the frame this call renders (1/N in an N-GPU run): the fewer pixels a
            // rank has per lane, the more of them can afford a wave of their own.  Measured on rank-local renders of the
            // headline frame (tools/partition_time.py).
            int e_tier1_pixels = g_opt.tier1_pixels, e_tier1_factor = g_opt.tier1_factor_x10, e_tier1_depth = g_opt.tier1_depth,
                e_heavy_factor = g_opt.heavy_factor_x10, e_sparse_factor = g_opt.sparse_factor_x10, e_sparse_percent = g_opt.sparse_wg_percent,
                e_work_percent = g_opt.sparse_work_percent;
            if (g_opt.tier_auto) {
                // keyed by pixels per resident lane (the 1200x800 frame: 3.7 whole, 1.8 / 0.9 / 0.5 for a half, a quarter,
                // an eighth; a quarter of 1920x1080 is 2.0): what matters is how empty the machine is, not the fraction
                const double per_lane = (double)n_pixels / ((double)max_grid * (double)block.x);
                if (per_lane > 2.75) {
                    // whole frames.  Lean family: the defaults.  The others: a tier wave is a main workgroup's slot taken away and
                    // their dear pixels are many and alike (Book-2 final: the fog ball), so only the very dearest get one
                    // (Book-2 final 800x800 @ 200: 352 ms with the lean sizes, 342 ms with these, profiles/r03_general_defaults.log)
                    if (!lean_family) { e_tier1_factor = 70; e_tier1_pixels = 256; e_tier1_depth = 1; }
                }
                // shares, lean family: re-fitted in round 3 with the tier kernel beside the main kernel (tools/share_sweep.py on rank 0
                // of the 1200x800 and 1920x1080 frames, profiles/r03_share_sweep_pass*.log; slowest-rank tables in DESIGN.md section 6)
                // -- and again with the tail hand-off, which takes over what the largest tiers were there for (rank 0 of 8: 48.1 ms with
                // round 3's first fit 16384 / 1.5x, 40.6 ms with the quarter's sizes; rank 0 of 2: 66.2 -> 62.6 ms, profiles/r03_handoff_shares.log)
                else if (lean_family) {
                    if (per_lane > 1.375) { e_tier1_pixels = 1536; e_tier1_factor = 40; e_tier1_depth = 3; e_heavy_factor = 20; e_sparse_factor = 40; e_sparse_percent = 80; e_work_percent = 5; }
                    else { e_tier1_pixels = 8192; e_tier1_factor = 20; e_tier1_depth = 4; e_heavy_factor = 15; e_sparse_factor = 20; e_sparse_percent = 80; e_work_percent = 40; }
                }
                // shares, other families: round 2's sizes (Book-2 final's 1/8 share: 216 ms with these, 236 with the lean family's,
                // 252 without a tier kernel, profiles/r03_share_sweep_final_eighth.log)
                else if (per_lane > 1.375) { e_tier1_pixels = 4096; e_tier1_factor = 30; e_tier1_depth = 4; e_heavy_factor = 20; e_sparse_factor = 30; e_sparse_percent = 80; }
                else if (per_lane > 0.6875) { e_tier1_pixels = 4096; e_tier1_factor = 30; e_tier1_depth = 4; e_heavy_factor = 20; e_sparse_factor = 30; e_sparse_percent = 80; e_work_percent = 20; }
                else { e_tier1_pixels = 8192; e_tier1_factor = 20; e_tier1_depth = 4; e_heavy_factor = 15; e_sparse_factor = 15; e_sparse_percent = 80; e_work_percent = 40; }
            }
            rt_rank_params rp;
            memset(&rp, 0, sizeof(rp));
            rp.state = s->d_state; rp.tile_cost = s->d_tile_cost; rp.tile_order = s->d_tile_order; rp.ray_counter = total;
            rp.heavy_list = s->d_heavy_list; rp.heavy_pixels = s->d_heavy_pixels; rp.info = s->d_rank;
            rp.n_pixels = (uint32_t)n_pixels; rp.n_tiles = (uint32_t)n_tiles; rp.heavy_cap = RT_HEAVY_CAP;
            rp.max_grid = max_grid; rp.waves_per_wg = block.x / 64u;
            rp.normal_need = (uint32_t)((q.work_items + block.x - 1) / block.x);
            rp.sparse_stride = (g_opt.sparse_stride > 0 && block.x >= 64) ? g_opt.sparse_stride : 0;
            rp.semi_stride = g_opt.semi_stride >= 0 ? g_opt.semi_stride : (lean_family ? 1 : 0);
            rp.sparse_percent = e_sparse_percent;
            rp.sparse_work_percent = e_work_percent;
            rp.tier_possible = tier_possible ? 1 : 0;
            rp.tier1_pixels = e_tier1_pixels; rp.tier1_depth = e_tier1_depth;
            rp.tier_wgs_cap = (int32_t)tier_grid; rp.tier_waves_per_main_wg = tier_waves_per_main_wg;
            rp.nx = f->nx; rp.smooth_percent = g_opt.cost_smooth_percent;
            if (e_sparse_factor < e_heavy_factor) e_sparse_factor = e_heavy_factor;
            rp.heavy_factor = (float)e_heavy_factor / 10.0f; rp.sparse_factor = (float)e_sparse_factor / 10.0f; rp.tier1_factor = (float)e_tier1_factor / 10.0f;
            HIPCHK(rt_launch_rank(rp, stream));
            q.tile_order = s->d_tile_order; q.heavy_pixels = s->d_heavy_pixels; q.rank = s->d_rank;
            unsigned total_wgs = rp.normal_need + (rp.sparse_stride > 0 ? max_grid * (unsigned)e_sparse_percent / 100u : 0u);
            if (tier_waves_per_main_wg > 0 && tier_possible) total_wgs = max_grid;      // (workgroups that make room for the tier kernel are part of the grid)
            if (total_wgs > max_grid) total_wgs = max_grid;
            if (total_wgs < 1u) total_wgs = 1u;
            grid_q = dim3(total_wgs);
            return RT_OK;
        };
        // ---- part 1: samples [0, S_a); S_a = presplit_samples, or S0.  Nothing has been measured yet; with the cost prior
        // (rt_prior_kernel: the calibration frame's rays per pixel, scaled to this frame) the part is ranked all the same,
        // so that the dearest chains start on tier waves at sample 0 instead of running at an ordinary lane's pace.
        const int first_end = (g_opt.presplit_samples > 0 && g_opt.presplit_samples < g_opt.split_samples) ? g_opt.presplit_samples : g_opt.split_samples;
        HIPCHK(hipMemsetAsync(s->d_tile_cost, 0, n_tiles * sizeof(unsigned int), stream));
        rt_frame_params p1 = fp;
        dim3 grid1 = grid;
        p1.sample_end = first_end; p1.state_out = s->d_state; p1.tile_cost = s->d_tile_cost;
        bool ranked1 = false;
        if (g_opt.prior && s->d_cal_cost && s->cal_nx > 0 && s->cal_ny > 0) {
            rt_prior_params pp;
            memset(&pp, 0, sizeof(pp));
            pp.state = s->d_state; pp.tile_cost = s->d_tile_cost; pp.total = s->d_ray_counter + 3;
            pp.cal_cost = s->d_cal_cost; pp.cal_nx = s->cal_nx; pp.cal_ny = s->cal_ny;
            pp.nx = f->nx; pp.ny = f->ny; pp.local_rows = local_rows; pp.tiles_x = fp.tiles_x;
            pp.tile_rows = f->tile_rows; pp.tile_first = f->tile_first; pp.tile_stride = f->tile_stride;
            HIPCHK(rt_launch_prior(pp, stream));
            p1.state_in = s->d_state; p1.fresh = 1;
            const rt_status st1 = rank_pixels(p1, grid1, s->d_ray_counter + 3);
            if (st1 != RT_OK) return st1;
            HIPCHK(hipMemsetAsync(s->d_tile_cost, 0, n_tiles * sizeof(unsigned int), stream));   // from here on: measured rays
            ranked1 = true;
        }
        { const rt_status st1 = launch_part(p1, grid1, ranked1); if (st1 != RT_OK) return st1; }
        // ---- part 1b: samples [S_a, S0), with tiers ranked on the first S_a samples.
        if (first_end < g_opt.split_samples) {
            rt_frame_params p2 = fp;
            dim3 grid2 = grid;
            p2.sample_begin = first_end; p2.sample_end = g_opt.split_samples;
            p2.state_in = s->d_state; p2.state_out = s->d_state; p2.tile_cost = s->d_tile_cost;
            rt_status st2 = rank_pixels(p2, grid2, s->d_ray_counter);
            if (st2 != RT_OK) return st2;
            st2 = launch_part(p2, grid2, true);
            if (st2 != RT_OK) return st2;
        }
        // ---- part 1c (optional): samples [S0, S1), ranked on the first S0 samples; the last part is then ranked again on
        // S1 samples.  A pixel's cost over 32 samples is a noisy estimate of its cost over 500 (paths through glass are
        // heavy-tailed): pixels that look cheap and are not start late and end the frame (tools/diag_wave_ends.py).
        int last_begin = g_opt.split_samples;
        if (g_opt.resplit_samples > g_opt.split_samples && f->ns >= 2 * g_opt.resplit_samples) {
            rt_frame_params p3 = fp;
            dim3 grid3 = grid;
            p3.sample_begin = g_opt.split_samples; p3.sample_end = g_opt.resplit_samples;
            p3.state_in = s->d_state; p3.state_out = s->d_state; p3.tile_cost = s->d_tile_cost;
            rt_status st2 = rank_pixels(p3, grid3, s->d_ray_counter);
            if (st2 != RT_OK) return st2;
            st2 = launch_part(p3, grid3, true);
            if (st2 != RT_OK) return st2;
            last_begin = g_opt.resplit_samples;
        }
        // ---- last part: samples [S1 or S0, ns), ranked on everything rendered so far
        fp.state_in = s->d_state; fp.sample_begin = last_begin;
        const rt_status st3 = rank_pixels(fp, grid, s->d_ray_counter);
        if (st3 != RT_OK) return st3;
        out.workgroups = (int)grid.x;
        s->ranked_frame = true;
    }
    out.reserved = 0;
#ifdef RT_DIAG
    // wave-end histograms of the frame's last launch only (rt_debug_wave_ends)
    HIPCHK(hipMemsetAsync(s->d_ray_counter + RT_DIAG_T0_SLOT, 0xFF, 8, stream));
    HIPCHK(hipMemsetAsync(s->d_ray_counter + RT_DIAG_HIST_SLOT, 0, (size_t)(2 * RT_DIAG_BINS + 7 + 2 * RT_DIAG_MAX_WAVES) * 8, stream));
#endif
    { const rt_status stl = launch_part(fp, grid, s->ranked_frame); if (stl != RT_OK) return stl; }
    HIPCHK(hipEventRecord(s->ev_stop, stream));
    s->frame_pending = true;
    s->pending_stream = stream;
    s->pending_stats = out;
    if (!fb_on_device) {
        HIPCHK(hipMemcpyAsync(fb, s->d_fb, floats * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        blocking = 1;
    }
    if (blocking) return rt_frame_finish(s, stats);
    if (stats) *stats = out;
    return RT_OK;
}

}  // extern "C"
