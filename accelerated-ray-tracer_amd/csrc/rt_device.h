// rt_device.h -- kernel argument blocks shared by rt_device.hip (kernels) and
// rt_abi.hip (the C-ABI implementation).  Internal to librt_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"

#define RT_PERSISTENT_THREADS 512
#ifndef RT_PARKED_MIN_WAVES
#define RT_PARKED_MIN_WAVES 4   // waves per SIMD the lean staged kernel (spheres-only, solid / checker colours) is register-limited to allow
#endif

// kernel ids keep their round-1 numbers (1 = persistent and 2 = parked were the steps between 0 and 3; removed)
enum { RT_KERNEL_PIXEL = 0, RT_KERNEL_STAGED = 3, RT_KERNEL_WAVEFRONT = 4 };

// A pixel parked at a sample boundary (split frames): XORWOW state, colour sum, rays traced so far.
struct rt_pixel_state {
    uint32_t rng[6];
    float col[3];
    uint32_t cost;
};

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
    const unsigned int* tile_order;       // optional: 8x8 tiles in descending cost (LPT order); null = natural order
    unsigned int* tile_cost;              // first part of a split frame: rays per 8x8 tile
    rt_pixel_state* state_out;            // first part of a split frame: where pixels are parked (the frame is not written)
    const rt_pixel_state* state_in;       // second part: the parked pixels (null = pixels start from their seed)
    int32_t sample_begin, sample_end;     // samples [sample_begin, sample_end) are rendered by this launch; ns is the frame's total
    const unsigned int* heavy_pixels;     // second part: local pixel ids (lrow * nx + i) of the heavy pixels, dearest first
    uint32_t heavy_threshold;             // pixels whose parked cost is >= this are in heavy_pixels
    uint32_t tier1_items;                 // leading heavy_pixels entries served by tier-1 sparse workgroups
    int32_t tier1_wgs, tier1_stride;      // tier 1: the first tier1_wgs workgroups, one pixel per tier1_stride lanes
    uint32_t tier0_items;                 // leading heavy_pixels entries served by tier-0 workgroups (one pixel per workgroup); tier 1 follows
    int32_t tier0_wgs;                    // tier 0: the first tier0_wgs workgroups; tier 1: the next tier1_wgs
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
    uint32_t heavy_items;                 // staged kernel: entries of heavy_pixels (served by sparse workgroups); 0 = none
    int32_t sparse_wgs;                   // staged kernel: workgroups that start in sparse mode
    int32_t sparse_eager;                 // staged kernel: sparse waves run every stage as soon as one lane needs it
    int32_t sparse_priority;              // staged kernel: s_setprio level of sparse waves (0 = leave alone)
    int32_t sparse_stride;                // staged kernel: in sparse mode only every sparse_stride-th lane takes a pixel
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

void rt_launch_collect_heavy(const rt_pixel_state* state, unsigned int n_pixels, unsigned int threshold, unsigned long long* list,
                             unsigned int capacity, unsigned int* count, hipStream_t st);
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
