// rt_device.h -- kernel argument blocks shared by rt_device.hip (kernels) and
// rt_abi.hip (the C-ABI implementation).  Internal to librt_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"

#define RT_PERSISTENT_THREADS 512   // default workgroup size of the staged kernel
// Register budgets of the staged kernel families, as launch bounds (threads per workgroup the code may be launched
// with, waves per SIMD it must leave room for).  "Lean" = spheres-only scenes without procedural / image textures (the
// headline kernel): 94 VGPRs once the double-precision transcendentals are out of line, but more than 4 waves per SIMD
// only slow the dearest pixels' chains down (profiles/r02g_occupancy_sweep.log: 2 x 640 threads 131 ms, 2 x 512 110 ms).
// Everything else (quads / boxes / media, Perlin / image textures) fits 168 VGPRs without spilling: 3 waves per SIMD.
#ifndef RT_LEAN_MIN_WAVES
#define RT_LEAN_MIN_WAVES 4
#endif
#define RT_LEAN_MAX_THREADS 512
#define RT_HEAVY_MAX_THREADS 768
#define RT_HEAVY_MIN_WAVES 3

// kernel ids keep their round-1 numbers (1 = persistent and 2 = parked were the steps between 0 and 3; removed)
// the statistics block behind rt_frame_params.ray_counter (64-bit words): [0] rays, [1..16] stage counters and [32..] the
// wave-end histograms of diagnostic builds (-DRT_DIAG), [31] the wavefront kernel's iteration-cap flag
enum { RT_DIAG_BINS = 192, RT_DIAG_T0_SLOT = 32, RT_DIAG_HIST_SLOT = 33, RT_DIAG_WAVE_SLOT = 33 + 2 * 192 + 7, RT_DIAG_MAX_WAVES = 8192,
       RT_COUNTER_BYTES = (33 + 2 * 192 + 7 + 2 * 8192) * 8 };   // + per wave: two words about the lane that finished last
enum { RT_KERNEL_PIXEL = 0, RT_KERNEL_STAGED = 3, RT_KERNEL_WAVEFRONT = 4 };

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
    uint32_t tier0_items;        // leading entries served one per WORKGROUP (tier 0)
    uint32_t tier1_items;        // following entries served one per WAVE (tier 1)
    uint32_t tier2_items;        // following entries served by sparse waves (tier 2); the REST of the list (tier 3) is taken by
                                 // ordinary lanes before anything else, so that every dear pixel's chain starts at once
    int32_t tier0_wgs, tier1_wgs;   // workgroups [0, tier0_wgs) are tier 0, the next tier1_wgs tier 1
    int32_t sparse_wgs;          // workgroups [0, sparse_wgs) start in sparse mode (tiers 0, 1, 2)
    int32_t sparse_stride;       // sparse waves: every sparse_stride-th lane takes a pixel
    int32_t semi_wgs;            // the next semi_wgs workgroups serve tier 3 with every semi_stride-th lane (0: ordinary lanes take tier 3)
    int32_t semi_stride;
    uint32_t threshold0, threshold1;   // costs that qualify for tier 0 / tier 1 (set with heavy_threshold by the first ranking kernel)
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
    int32_t tier0_possible;              // tier 0 needs a spheres-only scene resident in LDS
    int32_t tier0_pixels, tier1_pixels;  // caps on the tier sizes
    int32_t tier1_depth;                 // pixels a tier-1 wave is meant to take, one after the other
    int32_t nx, smooth_percent;          // cost estimate of a pixel = max(own, smooth_percent % of its dearest 4-neighbour's); nx = pixels per local row
    float heavy_factor, sparse_factor, tier1_factor, tier0_factor;   // cost thresholds as multiples of the mean cost per pixel:
                                         // >= heavy: in the list at all; >= sparse: tier 2; >= tier1 / tier0: those tiers
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
    rt_camera camera;
};

struct rt_frame_params {
    float* fb;                            // compact local rows, nx*3 floats each
    unsigned long long* ray_counter;      // += rays traced
    unsigned int* work_counter;           // persistent kernel's pixel queue head
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
    uint32_t tier0_lds_offset;            // tier 0: byte offset of the workgroup's scratch (leaf list, reduction slots) in dynamic LDS
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
    int32_t steps_per_trip;               // persistent kernel: node visits between ballots
    int32_t shade_threshold;              // persistent kernel: waiting lanes that trigger shading
    int32_t leaf_threshold;               // staged kernel: lanes with an object test due that trigger the leaf pass
    int32_t box_threshold, medium_threshold;   // staged kernel, general scenes: parked box/instance and medium lanes that trigger their leaf tests
    int32_t diel_threshold;               // staged kernel: dielectric hits that trigger their stage
    int32_t newpath_threshold;            // staged kernel: ended paths that trigger the new-path stage
    int32_t wf_slots;                     // wavefront kernel: ray slots per workgroup (multiple of 64)
    uint32_t wf_max_iterations;           // wavefront kernel: safety cap on workgroup iterations
    int32_t wf_pause_lanes;               // wavefront kernel: a wave with fewer walking lanes re-queues them once READY is empty
};

// the three ranking kernels of one ranked part, enqueued on `st` (no host synchronisation)
hipError_t rt_launch_rank(const rt_rank_params& rp, hipStream_t st);
#define RT_WF_BYTES_PER_SLOT (31 * 4 + 5 * 2)   /* 20 float + 11 int arrays, 5 u16 lists */

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
hipError_t rt_launch_wavefront(int lds_mode, int tex_level, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                               size_t lds_bytes, hipStream_t stream);
