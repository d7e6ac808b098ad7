// rt_device.h -- kernel argument blocks shared by the kernel translation units (rt_kernel_pixel.hip, rt_staged_*.hip,
// rt_tier_*.hip, rt_rank.hip) and rt_abi.hip (the C-ABI implementation).  Internal to librt_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"

#define RT_PERSISTENT_THREADS 512   // default workgroup size of the staged kernel
// Register budgets of the staged kernel families, as launch bounds (threads per workgroup the code may be launched
// with, waves per SIMD it must leave room for).  "Lean" = spheres-only scenes without procedural / image textures (the
// headline kernel): 97 VGPRs (allocated in eights: 104), no scratch (profiles/r03_kernel_resources.md) -- the double-precision
// transcendentals are out of line and the one-pixel-per-wave loops live in their own kernel (rt_kernel_tier.h).  It is launched
// 2 x 512 threads per CU = 4 waves per SIMD: 4 x 104 registers leave 96 of a SIMD's 512 for one co-resident wave of the tier
// kernel (76, allocated 80).  104 is therefore this kernel's ceiling, not the 128 the launch bounds would allow.
// Everything else (quads / boxes / media, Perlin / image textures) is budgeted 168 VGPRs: 3 waves per SIMD.
#ifndef RT_LEAN_MIN_WAVES
#define RT_LEAN_MIN_WAVES 4
#endif
#ifndef RT_LEAN_MAX_THREADS
#define RT_LEAN_MAX_THREADS 512
#endif
#define RT_HEAVY_MAX_THREADS 768
#define RT_HEAVY_MIN_WAVES 3

// the tier kernel (rt_kernel_tier.h): workgroups of four waves, one per SIMD; the lean family's must fit beside the main kernel's
// 4 x 104 registers per SIMD (they take 76), the others get 168 (their workgroups take the slots main workgroups vacate: main_skip_wgs)
#define RT_TIER_THREADS 256
// kernel ids keep their round-1 numbers (1 = persistent, 2 = parked, 4 = wavefront were experiments; removed)
// the statistics block behind rt_frame_params.ray_counter (64-bit words): [0] rays, [1..16] stage counters and [32..] the
// wave-end histograms of diagnostic builds (-DRT_DIAG), [3] the sum of the cost prior (rt_prior_kernel)
enum { RT_DIAG_BINS = 192, RT_DIAG_T0_SLOT = 32, RT_DIAG_HIST_SLOT = 33, RT_DIAG_WAVE_SLOT = 33 + 2 * 192 + 7, RT_DIAG_MAX_WAVES = 8192,
       RT_COUNTER_BYTES = (33 + 2 * 192 + 7 + 2 * 8192) * 8 };   // + per wave: two words about the lane that finished last
enum { RT_KERNEL_PIXEL = 0, RT_KERNEL_STAGED = 3 };

// A pixel parked at a sample boundary (split frames): XORWOW state, colour sum, rays traced so far.
struct rt_pixel_state {
    uint32_t rng[6];
    float col[3];
    uint32_t cost;               // rays so far; bit 31 = "on the heavy list of the coming launch" (set by the ranking)
};

// What the ranking kernels (rt_rank.hip) leave for the next render launch of a split frame: how many parked pixels are
// "heavy", how the heavy list is cut into tiers and how many workgroups serve each.  Lives in device memory; the render
// kernel reads it at its start, the host never does (a frame is one stream enqueue, no round trip).
struct rt_rank_info {
    uint32_t heavy_items;        // entries of heavy_pixels (dearest first); 0 = no list
    uint32_t heavy_threshold;    // pixels whose parked cost is >= this are in the list (ordinary waves skip them)
    uint32_t tier1_items;        // leading entries served one per WAVE by the tier kernel (tier 1)
    uint32_t tier2_items;        // following entries served by sparse waves (tier 2); the REST of the list (tier 3) is taken by
                                 // ordinary lanes before anything else, so that every dear pixel's chain starts at once
    int32_t tier1_wgs;           // workgroups of the TIER kernel that have work (the rest of its grid leaves at once)
    int32_t main_skip_wgs;       // workgroups [0, main_skip_wgs) of the MAIN kernel leave at once: their slots are the tier
                                 // kernel's (kernel families whose register budget leaves no room beside a full main grid)
    int32_t sparse_wgs;          // main workgroups [main_skip_wgs, main_skip_wgs + sparse_wgs) start in sparse mode (tier 2)
    int32_t sparse_stride;       // sparse waves: every sparse_stride-th lane takes a pixel
    int32_t semi_wgs;            // the next semi_wgs workgroups serve tier 3 with every semi_stride-th lane (0: ordinary lanes take tier 3)
    int32_t semi_stride;
    uint32_t threshold1;         // cost that qualifies for tier 1 (set with heavy_threshold by the first ranking kernel)
    uint32_t threshold2;         // cost that qualifies for the sparse waves (tier 2)
    uint32_t collected;          // entries rt_collect_heavy_kernel appended (may exceed the capacity: then there is no list)
};

// constants of one ranking (host-filled kernel argument)
struct rt_rank_params {
    rt_pixel_state* state;               // (the ranking marks listed pixels in bit 31 of their cost)
    const unsigned int* tile_cost;
    unsigned int* tile_order;
    const unsigned long long* ray_counter;
    unsigned long long* heavy_list;      // scratch: (cost << 32 | pixel), unsorted
    unsigned int* heavy_pixels;          // out: pixel ids, dearest first
    rt_rank_info* info;
    uint32_t n_pixels, n_tiles, heavy_cap;
    uint32_t max_grid, waves_per_wg, normal_need;
    int32_t sparse_stride;               // 0 = no heavy list at all
    int32_t semi_stride;                 // lanes per pixel in the workgroups that serve tier 3 (0 = ordinary lanes take tier 3 first)
    int32_t sparse_percent;              // at most this share of max_grid starts in sparse mode
    int32_t sparse_work_percent;         // ... and tiers 0-2 together hold at most this share of the frame's rays so far
    int32_t tier_possible;               // the scene has tier data (leaf arrays, rt_scene_dev) and the tier kernel is launched
    int32_t tier1_pixels;                // cap on tier 1
    int32_t tier1_depth;                 // pixels a tier-1 wave is meant to take, one after the other
    int32_t tier_wgs_cap;                // the tier kernel's grid (fixed on the host before the sizes are known)
    int32_t tier_waves_per_main_wg;      // 0: tier workgroups fit beside a full main grid; else: tier waves that fit into the
                                         // slot of one main workgroup (main_skip_wgs = tier waves / this, rounded up)
    int32_t nx, smooth_percent;          // cost estimate of a pixel = max(own, smooth_percent % of its dearest 4-neighbour's); nx = pixels per local row
    float heavy_factor, sparse_factor, tier1_factor;   // cost thresholds as multiples of the mean cost per pixel:
                                         // >= heavy: in the list at all; >= sparse: tier 2; >= tier1: tier 1
};

// Device encoding of a node's two links (both node arrays below; rt_abi.hip device_nodes()): `skip` holds ~skip and
// `prim` holds ~(own index + 1) at an interior node (still < 0) and the object id (>= 0) at a leaf.  The walk's "where
// next" is then ~((pass && prim < 0) ? prim : skip) -- one select, one NOT -- and the staged kernel's stopped state
// ~skip is the stored word itself.
#define RT_NODE_SKIP(stored) (~(stored))

// device-resident scene: the rt_scene_desc arrays after upload
struct rt_scene_dev {
    const rt_node* nodes;        // the walk array: the reference's tree in depth-first order, interior nodes that do not pay removed (rt_abi.hip, "collapse")
    const rt_node* nodes_ref;    // the reference's tree, every node (kernel 0 and the calibration pass walk this one)
    const rt_sphere* spheres;
    const rt_quad* quads;
    const rt_box* boxes;
    const rt_instance* instances;
    const rt_medium* media;
    const rt_material* materials;
    const rt_texture* textures;
    const uint8_t* images;
    int32_t n_nodes, n_nodes_ref, n_spheres, n_materials, n_textures;
    float bound[3];              // every box coordinate of the scene lies within +-bound[axis] (the walk loop's widened box test)
    float bound_pad;
    // tier data (rt_kernel_tier.h; built by rt_scene_create from the walk array, null when the scene has none): the leaves
    // of the walk array in depth-first order as two float4 arrays padded to a multiple of 64 -- leaf_lo[q] = (bmin, prim as
    // int bits; -1 in the padding), leaf_hi[q] = (bmax, 0) --, per 64 leaves the union of their boxes (slot_ranges: 8 floats
    // each: lo, hi, 0, 0), and the ordinals of the (at most two) constant_medium leaves
    const float4* leaf_lo;
    const float4* leaf_hi;
    const float* slot_ranges;
    int32_t n_leaves, n_slots;
    int32_t n_media_leaves;
    int32_t media_ord[2];
    rt_camera camera;
};

enum { RT_WC_DEAD = 64,        // work_counter[]: lanes of the main launch that have run out of work
       RT_WC_PUSHED = 65,      // entries pushed to handoff_queue
       RT_WC_TAIL_HEAD = 66,   // the tail launch's queue head
       RT_WORK_COUNTER_BYTES = 512 };
struct rt_frame_params {
    float* fb;                            // compact local rows, nx*3 floats each
    unsigned long long* ray_counter;      // += rays traced
    unsigned int* work_counter;           // persistent kernel's pixel queue head
    unsigned int* pixel_cost;             // calibration pass (kernel 0 only): rays traced per pixel, nx * local_rows; null otherwise
    unsigned int* node_pass;              // calibration pass (kernel 0 only): += 1 per box test of nodes_ref[i] that passed; null otherwise
    int32_t node_pass_lds;                // ... collected in LDS per workgroup (n_nodes_ref x 4 B of dynamic LDS) and flushed at its end
    float* ray_sample;                    // calibration pass: every ray_sample_stride-th ray as (origin, direction, t of its hit or FLT_MAX), 7 floats each
    uint32_t ray_sample_cap, ray_sample_stride;   // ... up to this many; the count is kept in ray_counter[2]
    const unsigned int* tile_order;       // optional: 8x8 tiles in descending cost (LPT order); null = natural order
    unsigned int* tile_cost;              // first part of a split frame: rays per 8x8 tile
    rt_pixel_state* state_out;            // first part of a split frame: where pixels are parked (the frame is not written)
    const rt_pixel_state* state_in;       // second part: the parked pixels (null = pixels start from their seed)
    int32_t sample_begin, sample_end;     // samples [sample_begin, sample_end) are rendered by this launch; ns is the frame's total
    const unsigned int* heavy_pixels;     // ranked launches: local pixel ids (lrow * nx + i) of the heavy pixels, dearest first
    const rt_rank_info* rank;             // ranked launches: tier sizes left by the ranking kernels; null = no heavy list, no tiers
    int32_t store_parked;                 // progressive windows (rt_render_window): a pixel parked at the window's end is ALSO written to fb,
                                          // averaged over the fp.ns samples so far
    int32_t fresh;                        // ranked FIRST part (tiers from the cost prior): state_in only carries the prior's cost and
                                          // list flag; pixels start from their seed with an empty colour sum
    int32_t tier_lds_scene;               // tier kernel: its LDS image holds spheres, materials and textures besides the leaf arrays
    // Tail hand-off (rt_abi.hip, "tail"): once the tile queue is dry and at most handoff_pixels pixels are still in flight on the
    // main kernel's lanes (= lanes that still have work), every lane parks its pixel at its next sample boundary into handoff_state[pixel] and pushes
    // (sample << 32 | pixel) to handoff_queue; the tier kernel, launched again AFTER the main kernel (tail_mode), finishes
    // them one pixel per wave.  Its counters live in a cache line of their own, away from the queue heads at work_counter[0..4]: with
    // the polled word in the queue heads' line every pixel fetch of the frame slowed down (headline 96 -> 118 ms).
    unsigned long long* handoff_queue;    // null = no hand-off
    uint32_t handoff_cap;                 // its entries
    rt_pixel_state* handoff_state;
    int32_t handoff_pixels;
    int32_t handoff_poll_ticks;           // ... looked for this often (10 ns ticks of s_memrealtime)
    int32_t tail_mode;                    // tier kernel: serve handoff_queue instead of tier 1 of the heavy list
    uint64_t seed_base;
    int32_t nx, ny, ns;
    float gamma;
    float background[3];
    int32_t use_gradient_bg;
    int32_t tile_rows, tile_first, tile_stride;
    int32_t local_rows;                   // rows this call renders
    int32_t tiles_x;                      // 8x8 pixel tiles per local row band
    uint32_t work_items;                  // tiles_x * tiles_y * 64
    int32_t sparse_eager;                 // staged kernel: sparse waves run every stage as soon as one lane needs it
    int32_t sparse_priority;              // staged kernel: s_setprio level of sparse waves (0 = leave alone)
    int32_t tier_priority;                // tier kernel: s_setprio level of its waves
    int32_t semi_priority;                // staged kernel: s_setprio level of the waves serving tier 3 on workgroups of their own
    int32_t steps_per_trip;               // persistent kernel: node visits between ballots
    int32_t shade_threshold;              // persistent kernel: waiting lanes that trigger shading
    int32_t leaf_threshold;               // staged kernel: lanes with an object test due that trigger the leaf pass
    int32_t box_threshold, medium_threshold;   // staged kernel, general scenes: parked box/instance and medium lanes that trigger their leaf tests
    int32_t diel_threshold;               // staged kernel: dielectric hits that trigger their stage
    int32_t newpath_threshold;            // staged kernel: ended paths that trigger the new-path stage
};

// the three ranking kernels of one ranked part, enqueued on `st` (no host synchronisation)
hipError_t rt_launch_rank(const rt_rank_params& rp, hipStream_t st);
// the cost prior of a ranked first part (rt_rank.hip): d_state[p].cost and tile_cost from the calibration frame's per-pixel rays
struct rt_prior_params {
    rt_pixel_state* state;
    unsigned int* tile_cost;
    unsigned long long* total;            // += sum of the priors (what the ranking kernels read as "rays so far")
    const unsigned int* cal_cost;         // rays per calibration pixel, cal_nx x cal_ny, row 0 = bottom
    int32_t cal_nx, cal_ny;
    int32_t nx, ny, local_rows, tiles_x;
    int32_t tile_rows, tile_first, tile_stride;
};
hipError_t rt_launch_prior(const rt_prior_params& pp, hipStream_t st);
// bytes of the tier kernel's LDS image (rt_kernel_tier.h stages exactly this): leaf arrays + slot unions, then -- where the
// launch plan (rt_abi.hip) says so -- spheres, materials and textures
static inline size_t rt_tier_lds_bytes(int n_slots, int n_spheres, int n_materials, int n_textures, bool scene) {
    size_t b = (size_t)n_slots * 64 * 32 + (size_t)n_slots * 32;
    if (scene) b += (size_t)n_spheres * sizeof(rt_sphere) + (size_t)n_materials * sizeof(rt_material) + (size_t)n_textures * sizeof(rt_texture);
    return b;
}

// Launchers, one per translation unit (each returns the launch's hipError_t, including a failed
// hipFuncSetAttribute(MaxDynamicSharedMemorySize)).  tex_level: 0 = every material colour is inline, 1 = solid +
// checker textures, 2 = noise / image / noodle / felt / uv-offset textures too (Perlin noise and the sphere-uv
// transcendentals cost ~100 VGPRs, so scenes without them -- the headline random scene -- get leaner code).
hipError_t rt_launch_pixel(bool spheres_only, int tex_level, bool need_uv, const rt_scene_dev& sd, const rt_frame_params& fp,
                           dim3 grid, dim3 block, hipStream_t stream);
hipError_t rt_launch_staged_spheres(int tex_level, int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid,
                                    dim3 block, size_t lds, hipStream_t st);
hipError_t rt_launch_staged_spheres_tex(int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                                        size_t lds, hipStream_t st);
hipError_t rt_launch_staged_general(int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                                    size_t lds, hipStream_t st);
hipError_t rt_launch_staged_general_tex(int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                                        size_t lds, hipStream_t st);
// the tier kernel of a ranked launch (rt_tier_*.hip); *vgprs_out (optional) = the instantiation's register count
hipError_t rt_launch_tier_spheres(int tex_level, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st);
hipError_t rt_launch_tier_general(int tex_level, bool need_uv, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st);
