/* rt_abi.h -- C ABI of the MI355X render path (librt_mi355x.so).
 *
 * This is the drop-in boundary for the reference's one hot path: the kernel
 * launches its host scene functions make (src/main.cu:685-732 and the same
 * pattern in every scene function).  The reference has no FFI layer; what a
 * maintainer would bind is exactly this set of entry points, each of which
 * replaces a group of reference launches/runtime calls:
 *
 *   rt_init / rt_shutdown       cudaDeviceSetLimit x2 (main.cu:665-666), cudaDeviceReset (main.cu:743)
 *   rt_scene_create             cudaMalloc(d_list|d_world|d_camera) + create_world_*<<<1,1>>>
 *                               (main.cu:688-697); the device-heap object graph becomes flat arrays
 *   rt_render                   cudaMallocManaged(fb) + cudaMalloc(d_rand_state) + render_init<<<>>> +
 *                               render<<<>>> + both syncs (main.cu:676-680, 702-709)
 *   rt_scene_destroy            free_world<<<1,1>>> + cudaFree x5 (main.cu:732-740)
 *   rt_strerror / rt_last_hip_error   checkCudaErrors (main.cu:23-35), minus the exit(99)
 *
 * Plain C, plain pointers and sizes.  No C++ types, no torch types.
 * Caller owns every rt_scene_desc array and the framebuffer; the library owns
 * all device memory behind rt_scene*.  Not re-entrant per rt_scene*.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int rt_status;
enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,      /* bad argument / malformed scene description */
    RT_ERR_NO_DEVICE = 2,    /* no gfx950 device visible */
    RT_ERR_HIP = 3,          /* a HIP runtime call failed; see rt_last_hip_error() */
    RT_ERR_UNSUPPORTED = 4   /* scene uses a nesting the kernels do not implement */
};

/* ---- flattened scene (what create_world_* builds with device-side new) ---- */

/* primitive reference = (kind << 28) | index into that kind's array */
enum { RT_PRIM_SPHERE = 0, RT_PRIM_QUAD = 1, RT_PRIM_BOX = 2, RT_PRIM_INSTANCE = 3, RT_PRIM_MEDIUM = 4 };
#define RT_PRIM_REF(kind, index) ((int32_t)(((uint32_t)(kind) << 28) | (uint32_t)(index)))
#define RT_PRIM_KIND(ref) ((int)(((uint32_t)(ref)) >> 28))
#define RT_PRIM_INDEX(ref) ((int)(((uint32_t)(ref)) & 0x0FFFFFFFu))

/* bvh_node (bvh.cuh:9-116) flattened in depth-first pre-order ("threaded"):
 * a node whose box is hit continues at index+1 (its left child) unless it is a
 * leaf; a node whose box is missed, and a finished leaf, continue at `skip`.
 * prim < 0: internal node.  Every object sits in its own leaf node whose box
 * is the object's box (the reference's n==1 node, bvh.cuh:38-43). */
typedef struct rt_node {
    float bmin[3];
    int32_t skip;
    float bmax[3];
    int32_t prim;
} rt_node; /* 32 B */

/* sphere (sphere.cuh:10-102): centre c(t) = c0 + t*vel, vel = 0 when static */
typedef struct rt_sphere {
    float c0[3];
    float radius;
    float vel[3];
    int32_t mat;
} rt_sphere; /* 32 B */

/* quad (quad.cuh:11-91) with its constructor-derived fields precomputed */
typedef struct rt_quad {
    float Q[3];
    float D;
    float u[3];
    int32_t mat;
    float v[3];
    float pad0;
    float w[3];
    float pad1;
    float n[3];
    float pad2;
} rt_quad; /* 80 B */

/* compound6 (quad.cuh:94-143): six consecutive quads, scan order = array order */
typedef struct rt_box {
    int32_t first_quad;
} rt_box;

/* translate(rotate_y(child)) (hittable.cuh:40-149); either half may be absent */
enum { RT_INST_ROTATE_Y = 1, RT_INST_TRANSLATE = 2 };
typedef struct rt_instance {
    float sin_t, cos_t;
    float offset[3];
    int32_t child;  /* prim ref: sphere, quad or box */
    int32_t flags;
    int32_t pad;
} rt_instance; /* 32 B */

/* constant_medium (constant_medium.cuh:16-80) */
typedef struct rt_medium {
    int32_t boundary; /* prim ref: sphere, quad, box or instance */
    float neg_inv_density;
    int32_t mat;      /* isotropic phase function */
    int32_t pad;
} rt_medium; /* 16 B */

enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_DIFFUSE_LIGHT = 3, RT_MAT_ISOTROPIC = 4 };
/* material.cuh:62-201.  tex < 0: `albedo` is the (solid) colour. */
typedef struct rt_material {
    int32_t kind;
    int32_t tex;
    float fuzz; /* metal, already clamped to <= 1 (material.cuh:97) */
    float ior;  /* dielectric */
    float albedo[3];
    float pad;
} rt_material; /* 32 B */

enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2, RT_TEX_NOISE = 3, RT_TEX_NOODLE = 4, RT_TEX_FELT = 5, RT_TEX_UV_OFFSET = 6 };
/* texture.cuh:16-164.
 * checker: a/b = even/odd texture index, scale = 1/scale.
 * image: a = byte offset into `images`, b = width, c = height (RGB8).
 * noise: scale.
 * noodle (texture.cuh:84-103): scale = k, p[6] = A, p[7] = f, a = octaves, p[3..5] = unit direction,
 *   color = noodle colour, p[0..2] = gap colour.
 * felt (texture.cuh:109-148): color = base, scale = mottling scale, p[0] = mottling amount, p[1] = fibre scale,
 *   p[2] = fibre amount.
 * uv_offset (texture.cuh:151-164): a = wrapped texture, scale = du (turns), p[0] = dv. */
typedef struct rt_texture {
    int32_t kind;
    int32_t a, b;
    float scale;
    float color[3];
    int32_t c;
    float p[8];
} rt_texture; /* 64 B */

/* camera (camera.cuh:18-79) after init() */
typedef struct rt_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float lens_radius;
    float pad;
    double time0, time1;
} rt_camera;

typedef struct rt_scene_desc {
    const rt_node* nodes;         int32_t n_nodes;
    const rt_sphere* spheres;     int32_t n_spheres;
    const rt_quad* quads;         int32_t n_quads;
    const rt_box* boxes;          int32_t n_boxes;
    const rt_instance* instances; int32_t n_instances;
    const rt_medium* media;       int32_t n_media;
    const rt_material* materials; int32_t n_materials;
    const rt_texture* textures;   int32_t n_textures;
    const uint8_t* images;        size_t image_bytes;
    rt_camera camera;
} rt_scene_desc;

/* ---- one frame (the arguments of render<<<>>>, main.cu:107-109) ---- */
typedef struct rt_frame_desc {
    int32_t nx, ny;          /* full image size; pixel_index = j*nx + i, row 0 = bottom (main.cu:115) */
    int32_t ns;              /* samples per pixel */
    float gamma;             /* 1.0 = identity (main.cu:39) */
    float background[3];
    int32_t use_gradient_bg;
    uint64_t seed_base;      /* per-pixel seed = seed_base + pixel_index (main.cu:104: 1984) */
    /* Row partition for multi-GPU runs: the image is cut into tiles of
     * `tile_rows` rows; this call renders tiles tile_first, tile_first +
     * tile_stride, ...  The output buffer is compact: local row k holds global
     * row rt_local_to_global_row(k).  Whole frame: tile_rows = ny,
     * tile_first = 0, tile_stride = 1. */
    int32_t tile_rows, tile_first, tile_stride;
    int32_t reserved;
} rt_frame_desc;

typedef struct rt_stats {
    uint64_t rays;           /* world->hit calls from color() (main.cu:57) */
    uint64_t samples;        /* primary rays */
    double ms_render;        /* device time of the render kernel, HIP events on the launch stream */
    int32_t local_rows;      /* rows written by this call */
    int32_t kernel_variant;  /* which specialisation ran */
    int32_t workgroups, threads_per_group, lds_bytes, reserved;
} rt_stats;

typedef struct rt_scene rt_scene;

rt_status rt_init(int device_ordinal);
rt_status rt_shutdown(void);
const char* rt_strerror(rt_status s);
int rt_last_hip_error(void);            /* hipError_t of the last failing call, 0 if none */
const char* rt_last_error_detail(void); /* "file:line 'expr'" of the last failure, like main.cu:28-29 */

rt_status rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
rt_status rt_scene_destroy(rt_scene* scene);

/* The traversal array behind a scene.  rt_scene_create keeps the reference's depth-first tree and, beside it, the array
 * the render kernels walk: the same leaves in the same order, with the interior nodes whose box test does not pay
 * removed (option "bvh_collapse", read at creation; results are bit-identical either way).  Reports the two
 * node counts and the expected box tests per ray before / after on the calibration frame (0 when nothing was removed). */
rt_status rt_scene_walk_info(const rt_scene* scene, int32_t* nodes_reference, int32_t* nodes_walked,
                             double* tests_before, double* tests_after);
/* The planner behind it, host only (no device needed): the walk array for `nodes` given per-node counts of passing box
 * tests (`pass`, null = proportional to box surface area) out of `root_visits` rays.  out/cap may be null/0. */
rt_status rt_plan_walk_array(const rt_node* nodes, int32_t n, const double* pass, double root_visits, rt_node* out, int32_t cap,
                             int32_t* n_out, double* tests_before, double* tests_after);

/* Another hierarchy over the same leaves, host only: the leaves of `nodes` (its single-object nodes, boxes and order
 * untouched) under a binary tree whose interior boxes are the union of their leaves' boxes; method 0 = top-down by
 * surface-area cost, 1 = bottom-up (merge the neighbouring groups with the smallest union).  Any such tree gives the
 * reference's results; rt_scene_create (bvh_collapse = 3) measures both beside the reference's tree and
 * walks whichever needs the fewest box tests.  out holds up to cap nodes; *n_out = 2 * leaves - 1. */
rt_status rt_regroup_leaves(const rt_node* nodes, int32_t n, int32_t method, rt_node* out, int32_t cap, int32_t* n_out);

/* Number of rows a frame description assigns to this call, and the mapping
 * from a compact local row to its global row. */
int32_t rt_frame_local_rows(const rt_frame_desc* f);
int32_t rt_local_to_global_row(const rt_frame_desc* f, int32_t local_row);

/* render_init + render (main.cu:96-133) for the rows this call owns.
 * fb: float RGB, rt_frame_local_rows(f) * nx * 3 elements.  fb_on_device != 0:
 * fb is device memory and the kernel writes it directly; otherwise it is host
 * memory and the library copies the rows back.  stream: a hipStream_t (0 =
 * default stream).  The call returns after the frame is complete when
 * `blocking` != 0; otherwise work is only enqueued on `stream` and
 * stats->ms_render / rays are valid after rt_frame_finish(). */
rt_status rt_render(rt_scene* scene, const rt_frame_desc* f, float* fb, int fb_on_device,
                    void* stream, int blocking, rt_stats* stats);
rt_status rt_frame_finish(rt_scene* scene, rt_stats* stats);

/* ---- progressive accumulation (the reference writes every pixel's curandState back at the end of render(),
 * main.cu:126, which is what would allow it and what nothing in the reference uses) ----
 * rt_render_window renders samples [sample_begin, sample_end) of every pixel the frame description assigns to the call,
 * continuing from `state` -- device memory made by rt_progressive_state_create for that frame description, opaque to the
 * caller: per pixel the XORWOW state, the colour sum and the rays so far -- and writes to fb the frame averaged over the
 * sample_end samples rendered so far, gamma applied (f->ns is ignored).  sample_begin must be 0 on the first call for a
 * state and the previous call's sample_end afterwards; anything else is RT_ERR_INVALID.  A sequence of windows gives
 * pixels bit-identical to one rt_render with ns = the last sample_end (tests/test_gpu_parity.py); stats->rays counts the
 * window's rays.  Windows run as one launch each (no cost-aware split: a window is usually short). */
rt_status rt_progressive_state_create(rt_scene* scene, const rt_frame_desc* f, void** state);
rt_status rt_progressive_state_destroy(rt_scene* scene, void* state);
rt_status rt_render_window(rt_scene* scene, const rt_frame_desc* f, float* fb, int fb_on_device, void* state,
                           int32_t sample_begin, int32_t sample_end, void* stream, int blocking, rt_stats* stats);

/* ---- several GPUs of one node from one host thread ----
 * The reference is single-GPU (one render<<<>>> launch, main.cu:707); these entry points are what its host function
 * would call to spread that launch over the N GPUs of a node: rt_init_devices(N) replaces rt_init, rt_multi_create /
 * rt_multi_render / rt_multi_destroy replace rt_scene_create / rt_render / rt_scene_destroy.  The frame is cut into
 * tiles of `tile_rows` rows dealt round-robin to the devices (tile t -> device t % N); every device renders its rows
 * with its own replica of the scene, one ncclGather (rccl.h:745) over xGMI brings the compact row buffers to device 0
 * and a small kernel puts them into the reference's frame layout.  `f` describes the WHOLE frame (its tile_* fields
 * are ignored); fb receives nx*ny*3 floats (host memory, or memory of device 0 when fb_on_device != 0).  The call
 * returns when the frame is complete.  stats: rays / samples summed over the devices, ms_render = host wall time of
 * the frame (render on every device + gather + reassembly + copy; buffer allocation and communicator set-up of a first
 * call excluded), reserved = the slowest device's own render time in microseconds.  By construction pixels are
 * bit-identical to rt_render's on one device -- no ray crosses a device and the per-pixel seed is seed_base + the
 * global pixel index; on one GPU that is tested for every rank's share and for the reassembly at world sizes 2..8
 * (tests/test_gpu_parity.py), with N > 1 devices it is UNVERIFIED ON HARDWARE so far.  On any failure after the frames
 * were enqueued every device is drained (streams synchronised, pending frames closed) before the error is returned. */
typedef struct rt_multi rt_multi;
rt_status rt_init_devices(int n_gpus);
rt_status rt_multi_create(const rt_scene_desc* desc, int n_gpus, rt_multi** out);
rt_status rt_multi_render(rt_multi* m, const rt_frame_desc* f, float* fb, int fb_on_device, int tile_rows, rt_stats* stats);
rt_status rt_multi_destroy(rt_multi* m);
int32_t rt_multi_device_count(const rt_multi* m);
/* the row partition rt_multi_render uses: which device renders global row j and at which row of its compact buffer
 * (the inverse of rt_local_to_global_row for tile_first = device, tile_stride = n_gpus) */
rt_status rt_multi_row_owner(int32_t global_row, int32_t tile_rows, int32_t n_gpus, int32_t* device, int32_t* local_row);
/* diagnostics of the multi-GPU path (tests): load RCCL exactly as rt_multi_render would (`library_name` = that name only,
 * null = the usual search) -- a missing library is RT_ERR_HIP with the loader's message, no device needed; and the
 * reassembly step on caller-supplied device buffers, staging[world][max_rows][nx*3] -> frame[ny][nx*3], synchronous. */
rt_status rt_multi_probe_rccl(const char* library_name);
rt_status rt_multi_debug_uninterleave(const float* staging, float* frame, int32_t nx, int32_t ny, int32_t tile_rows, int32_t world, int32_t max_rows);

/* Tuning knobs (for A/B measurements; defaults are what ships).  Unknown keys
 * return RT_ERR_INVALID.  The knobs are process-wide; rt_reset_options()
 * restores every one of them to the shipped default.  None changes a pixel. */
rt_status rt_set_option(const char* key, int value);
rt_status rt_reset_options(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
