// rt_multi.hip -- several GPUs of one node behind the C ABI (include/rt_abi.h, rt_multi_*): one host thread, one
// scene replica and one stream per device, the frame cut into interleaved row tiles, one RCCL gather
// of the compact per-device row buffers to device 0 over xGMI, and a small kernel that puts the rows back into the
// reference's frame layout (pixel_index = j*nx + i, row 0 = bottom, main.cu:115).  No ray crosses a device and the
// per-pixel seed is seed_base + GLOBAL pixel index, so the frame is bit-identical to the one-GPU frame.
//
// RCCL is loaded with dlopen, and only when a gather is needed (n_gpus > 1, or the "multi_force_rccl" option on a
// one-GPU box): a process that also runs PyTorch (bench.py) already carries torch's own copy of the library, and
// resolving ncclGather through the global symbol table would mix the two.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_device.h"

// internals of rt_abi.hip this file builds on
rt_status rt_internal_init_device(int device_ordinal);
rt_status rt_internal_scene_create_on(int device, const rt_scene_desc* d, rt_scene** out);
void rt_internal_set_error(rt_status st, int hip_error, const std::string& detail);
int rt_internal_option(const char* key);

namespace {

// the handful of RCCL entry points used, with the types of <rccl/rccl.h> (ncclComm_t is an opaque pointer,
// ncclFloat = 7, ncclSuccess = 0; rccl.h:36,466,52)
typedef void* nccl_comm;
typedef int (*fn_CommInitAll)(nccl_comm*, int, const int*);
typedef int (*fn_CommDestroy)(nccl_comm);
typedef int (*fn_GroupStart)();
typedef int (*fn_GroupEnd)();
typedef int (*fn_Gather)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);   // rccl.h:745
typedef const char* (*fn_GetErrorString)(int);
enum { NCCL_FLOAT = 7 };

struct rccl_api {
    void* handle = nullptr;
    fn_CommInitAll CommInitAll = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_GroupStart GroupStart = nullptr;
    fn_GroupEnd GroupEnd = nullptr;
    fn_Gather Gather = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
};

// `only` (tests): try exactly this name instead of the usual ones
bool load_rccl(rccl_api& r, std::string& why, const char* only = nullptr) {
    if (r.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    std::string last;
    for (const char* n : names) {
        if (only) n = only;
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
        const char* e = dlerror();   // (returns the message once and clears it)
        last = e ? e : "?";
        if (only) break;
    }
    if (!r.handle) { why = std::string("cannot load librccl.so: ") + last; return false; }
    r.CommInitAll = (fn_CommInitAll)dlsym(r.handle, "ncclCommInitAll");
    r.CommDestroy = (fn_CommDestroy)dlsym(r.handle, "ncclCommDestroy");
    r.GroupStart = (fn_GroupStart)dlsym(r.handle, "ncclGroupStart");
    r.GroupEnd = (fn_GroupEnd)dlsym(r.handle, "ncclGroupEnd");
    r.Gather = (fn_Gather)dlsym(r.handle, "ncclGather");
    r.GetErrorString = (fn_GetErrorString)dlsym(r.handle, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Gather) {
        why = "librccl.so lacks ncclCommInitAll / ncclGather";
        dlclose(r.handle); r.handle = nullptr;
        return false;
    }
    return true;
}

// which device renders global row j, and where in that device's compact buffer (the inverse of rt_local_to_global_row)
__host__ __device__ inline void row_owner(int j, int tile_rows, int world, int& rank, int& local_row) {
    const int tile = j / tile_rows;
    rank = tile % world;
    local_row = (tile / world) * tile_rows + (j - tile * tile_rows);
}

// staging[rank][local_row][nx*3] -> frame[global_row][nx*3]; one thread per float (a row is nx*3 floats)
__global__ void rt_uninterleave_kernel(const float* staging, float* frame, int nx3, int ny, int tile_rows, int world, int max_rows) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)ny * nx3;
    if (idx >= total) return;
    const int j = (int)(idx / nx3), k = (int)(idx - (long long)j * nx3);
    int rank, local_row;
    row_owner(j, tile_rows, world, rank, local_row);
    frame[idx] = staging[((size_t)rank * max_rows + local_row) * nx3 + k];
}

}  // namespace

struct rt_multi {
    int n = 0;
    std::vector<rt_scene*> scenes;
    std::vector<hipStream_t> streams;
    std::vector<float*> d_local;        // per device: compact rows of this device, max_rows * nx * 3 floats
    float* d_staging = nullptr;         // device 0: the gather's receive buffer, n * max_rows * nx * 3
    float* d_frame = nullptr;           // device 0: the assembled frame
    size_t local_floats = 0, frame_floats = 0, staging_floats = 0;   // capacities of d_local[*], d_frame, d_staging
    rccl_api rccl;
    std::vector<nccl_comm> comms;
    bool comms_ready = false;
};

#define MHIP(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            char b_[512];                                                                                  \
            snprintf(b_, sizeof(b_), "HIP error = %u at %s:%d '%s' (%s)", (unsigned)e_, __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            rt_internal_set_error(RT_ERR_HIP, (int)e_, b_);                                                \
            return RT_ERR_HIP;                                                                             \
        }                                                                                                  \
    } while (0)

extern "C" {

rt_status rt_init_devices(int n_gpus) {
    if (n_gpus < 1) { rt_internal_set_error(RT_ERR_INVALID, 0, "rt_init_devices: n_gpus must be >= 1"); return RT_ERR_INVALID; }
    for (int d = n_gpus - 1; d >= 0; --d) {   // device 0 last: it stays the device new single-GPU scenes are created on
        const rt_status st = rt_internal_init_device(d);
        if (st != RT_OK) return st;
    }
    return RT_OK;
}

rt_status rt_multi_destroy(rt_multi* m) {
    if (!m) return RT_OK;
    for (int d = 0; d < (int)m->scenes.size(); ++d) {
        if (hipSetDevice(d) != hipSuccess) continue;
        if (d < (int)m->comms.size() && m->comms_ready && m->comms[d]) (void)m->rccl.CommDestroy(m->comms[d]);
        if (d < (int)m->d_local.size() && m->d_local[d]) (void)hipFree(m->d_local[d]);
        if (d < (int)m->streams.size() && m->streams[d]) (void)hipStreamDestroy(m->streams[d]);
        if (m->scenes[d]) (void)rt_scene_destroy(m->scenes[d]);
    }
    if (hipSetDevice(0) == hipSuccess) {
        if (m->d_staging) (void)hipFree(m->d_staging);
        if (m->d_frame) (void)hipFree(m->d_frame);
    }
    if (m->rccl.handle) dlclose(m->rccl.handle);
    delete m;
    return RT_OK;
}

rt_status rt_multi_create(const rt_scene_desc* desc, int n_gpus, rt_multi** out) {
    if (!out) { rt_internal_set_error(RT_ERR_INVALID, 0, "null output pointer"); return RT_ERR_INVALID; }
    *out = nullptr;
    if (n_gpus < 1 || n_gpus > 16) { rt_internal_set_error(RT_ERR_INVALID, 0, "rt_multi_create: n_gpus must be 1..16"); return RT_ERR_INVALID; }
    rt_multi* m = new rt_multi;
    m->n = n_gpus;
    m->scenes.assign((size_t)n_gpus, nullptr);
    m->streams.assign((size_t)n_gpus, nullptr);
    m->d_local.assign((size_t)n_gpus, nullptr);
    for (int d = 0; d < n_gpus; ++d) {
        const rt_status st = rt_internal_scene_create_on(d, desc, &m->scenes[d]);   // also makes device d current
        if (st != RT_OK) { rt_multi_destroy(m); return st; }
        hipError_t e = hipStreamCreateWithFlags(&m->streams[d], hipStreamNonBlocking);
        if (e != hipSuccess) { rt_internal_set_error(RT_ERR_HIP, (int)e, "rt_multi_create: hipStreamCreate failed"); rt_multi_destroy(m); return RT_ERR_HIP; }
    }
    *out = m;
    return RT_OK;
}

int32_t rt_multi_device_count(const rt_multi* m) { return m ? m->n : 0; }

rt_status rt_multi_row_owner(int32_t global_row, int32_t tile_rows, int32_t n_gpus, int32_t* device, int32_t* local_row) {
    if (global_row < 0 || tile_rows <= 0 || n_gpus <= 0 || !device || !local_row) { rt_internal_set_error(RT_ERR_INVALID, 0, "rt_multi_row_owner: bad argument"); return RT_ERR_INVALID; }
    int r, l;
    row_owner(global_row, tile_rows, n_gpus, r, l);
    *device = r; *local_row = l;
    return RT_OK;
}

// After a failure once frames are enqueued: wait for every device and close every scene's pending frame, so that the
// replicas are usable again (and nothing still writes d_local) when the error status reaches the caller.
static void drain_devices(rt_multi* m) {
    for (int d = 0; d < m->n; ++d) {
        if (hipSetDevice(d) != hipSuccess) continue;
        if (m->streams[d]) (void)hipStreamSynchronize(m->streams[d]);
        if (m->scenes[d]) (void)rt_frame_finish(m->scenes[d], nullptr);
    }
    (void)hipSetDevice(0);
}

// buffers and communicators of a frame shape (no-ops once they are large enough / exist)
static rt_status multi_setup(rt_multi* m, size_t local_floats, size_t frame_floats, bool gather) {
    const int n = m->n;
    if (m->local_floats < local_floats) {
        for (int d = 0; d < n; ++d) {
            MHIP(hipSetDevice(d));
            if (m->d_local[d]) (void)hipFree(m->d_local[d]);
            m->d_local[d] = nullptr;
        }
        m->local_floats = 0;
        for (int d = 0; d < n; ++d) {
            MHIP(hipSetDevice(d));
            MHIP(hipMalloc((void**)&m->d_local[d], (local_floats ? local_floats : 1) * sizeof(float)));
        }
        m->local_floats = local_floats;
    }
    MHIP(hipSetDevice(0));
    if (m->frame_floats < frame_floats) {
        if (m->d_frame) (void)hipFree(m->d_frame);
        m->d_frame = nullptr; m->frame_floats = 0;
        MHIP(hipMalloc((void**)&m->d_frame, frame_floats * sizeof(float)));
        m->frame_floats = frame_floats;
    }
    // the gather writes n * local_floats floats of THIS frame: its capacity is tracked on its own (a frame rendered
    // without the gather allocates none)
    const size_t staging_need = gather ? (size_t)n * (local_floats ? local_floats : 1) : 0;
    if (m->staging_floats < staging_need) {
        if (m->d_staging) (void)hipFree(m->d_staging);
        m->d_staging = nullptr; m->staging_floats = 0;
        MHIP(hipMalloc((void**)&m->d_staging, staging_need * sizeof(float)));
        m->staging_floats = staging_need;
    }
    if (gather && !m->comms_ready) {
        std::string why;
        if (!load_rccl(m->rccl, why)) { rt_internal_set_error(RT_ERR_HIP, 0, why); return RT_ERR_HIP; }
        m->comms.assign((size_t)n, nullptr);
        std::vector<int> devs((size_t)n);
        for (int d = 0; d < n; ++d) devs[d] = d;
        const int rc = m->rccl.CommInitAll(m->comms.data(), n, devs.data());    // rccl.h:236
        if (rc != 0) { rt_internal_set_error(RT_ERR_HIP, rc, std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "failed")); return RT_ERR_HIP; }
        m->comms_ready = true;
    }
    return RT_OK;
}

rt_status rt_multi_render(rt_multi* m, const rt_frame_desc* whole, float* fb, int fb_on_device, int tile_rows, rt_stats* stats) {
    if (!m || !whole || !fb) { rt_internal_set_error(RT_ERR_INVALID, 0, "null argument"); return RT_ERR_INVALID; }
    if (whole->nx <= 0 || whole->ny <= 0 || whole->ns <= 0) { rt_internal_set_error(RT_ERR_INVALID, 0, "nx, ny and ns must be positive"); return RT_ERR_INVALID; }
    const int n = m->n, nx = whole->nx, ny = whole->ny;
    const bool force_rccl = rt_internal_option("multi_force_rccl") != 0;
    const bool gather = n > 1 || force_rccl;
    if (tile_rows <= 0) tile_rows = 4;
    if (!gather) tile_rows = ny;

    // rows per device and the common (padded) buffer size of the gather
    std::vector<rt_frame_desc> f((size_t)n, *whole);
    int max_rows = 0;
    for (int d = 0; d < n; ++d) {
        f[d].tile_rows = tile_rows; f[d].tile_first = d; f[d].tile_stride = n;
        const int rows = rt_frame_local_rows(&f[d]);
        if (rows < 0) { rt_internal_set_error(RT_ERR_INVALID, 0, "bad row partition"); return RT_ERR_INVALID; }
        if (rows > max_rows) max_rows = rows;
    }
    const size_t local_floats = (size_t)max_rows * nx * 3, frame_floats = (size_t)ny * nx * 3;
    {   // one-time setup (allocations, dlopen, ncclCommInitAll: seconds on a first call) stays out of the frame's time
        const rt_status st = multi_setup(m, local_floats, frame_floats, gather);
        if (st != RT_OK) return st;
    }
    const auto t0 = std::chrono::steady_clock::now();
    // from here on frames are in flight: an error drains every device before it is returned
    rt_status failed = RT_OK;
#define MTRY(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess && failed == RT_OK) {                                                         \
            char b_[512];                                                                                  \
            snprintf(b_, sizeof(b_), "HIP error = %u at %s:%d '%s' (%s)", (unsigned)e_, __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            rt_internal_set_error(RT_ERR_HIP, (int)e_, b_);                                                \
            failed = RT_ERR_HIP;                                                                           \
        }                                                                                                  \
    } while (0)

    // ---- every device renders its rows (enqueue only: the devices run concurrently)
    for (int d = 0; d < n && failed == RT_OK; ++d) {
        float* dst = gather ? m->d_local[d] : m->d_frame;
        failed = rt_render(m->scenes[d], &f[d], dst, /*fb_on_device=*/1, m->streams[d], /*blocking=*/0, nullptr);
    }
    // ---- one gather to device 0 over xGMI, then the rows go to their places
    if (gather && failed == RT_OK) {
        int rc = m->rccl.GroupStart();
        for (int d = 0; d < n && rc == 0 && failed == RT_OK; ++d) {
            MTRY(hipSetDevice(d));
            if (failed == RT_OK) rc = m->rccl.Gather(m->d_local[d], d == 0 ? m->d_staging : nullptr, local_floats, NCCL_FLOAT, 0, m->comms[d], m->streams[d]);
        }
        const int rc2 = m->rccl.GroupEnd();   // always closes the group that GroupStart opened
        if (rc == 0) rc = rc2;
        if (rc != 0 && failed == RT_OK) {
            rt_internal_set_error(RT_ERR_HIP, rc, std::string("ncclGather: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "failed"));
            failed = RT_ERR_HIP;
        }
        if (failed == RT_OK) {
            MTRY(hipSetDevice(0));
            const long long total = (long long)ny * nx * 3;
            hipLaunchKernelGGL(rt_uninterleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m->streams[0], m->d_staging, m->d_frame, nx * 3, ny, tile_rows, n, max_rows);
            MTRY(hipGetLastError());
        }
    }
    if (failed == RT_OK) {
        MTRY(hipSetDevice(0));
        MTRY(hipMemcpyAsync(fb, m->d_frame, frame_floats * sizeof(float), fb_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, m->streams[0]));
    }
    if (failed != RT_OK) { drain_devices(m); return failed; }
    // ---- wait for every device, add up the statistics
    rt_stats total_stats;
    memset(&total_stats, 0, sizeof(total_stats));
    double slowest = 0.0;
    for (int d = n - 1; d >= 0; --d) {
        MTRY(hipSetDevice(d));
        MTRY(hipStreamSynchronize(m->streams[d]));
        rt_stats st;
        memset(&st, 0, sizeof(st));
        const rt_status rs = rt_frame_finish(m->scenes[d], &st);
        if (rs != RT_OK && failed == RT_OK) failed = rs;
        total_stats.rays += st.rays; total_stats.samples += st.samples; total_stats.local_rows += st.local_rows;
        total_stats.workgroups += st.workgroups;
        if (st.ms_render > slowest) slowest = st.ms_render;
        if (d == 0) { total_stats.kernel_variant = st.kernel_variant; total_stats.threads_per_group = st.threads_per_group; total_stats.lds_bytes = st.lds_bytes; }
    }
#undef MTRY
    if (failed != RT_OK) { drain_devices(m); return failed; }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // ms_render: host wall time of the whole multi-device frame (render on every device + gather + reassembly + the copy
    // into fb; one-time setup excluded); the slowest device's own render time is in `reserved` (microseconds)
    total_stats.ms_render = wall_ms;
    total_stats.reserved = (int32_t)(slowest * 1000.0);
    if (stats) *stats = total_stats;
    return RT_OK;
}

// ---- test hooks (declared in include/rt_abi.h under "diagnostics")
// Tries to load RCCL the way rt_multi_render does -- from `library_name` only when one is given.  No device involved:
// a missing library is an error status with the loader's message, never a crash.
rt_status rt_multi_probe_rccl(const char* library_name) {
    rccl_api r;
    std::string why;
    if (!load_rccl(r, why, library_name)) { rt_internal_set_error(RT_ERR_HIP, 0, why); return RT_ERR_HIP; }
    dlclose(r.handle);
    return RT_OK;
}
// The reassembly step of rt_multi_render on caller-supplied DEVICE buffers: staging[world][max_rows][nx*3] -> frame[ny][nx*3]
// on the current device's default stream (synchronous).  Lets a one-GPU box check the index arithmetic for any world size.
rt_status rt_multi_debug_uninterleave(const float* staging, float* frame, int32_t nx, int32_t ny, int32_t tile_rows, int32_t world, int32_t max_rows) {
    if (!staging || !frame || nx <= 0 || ny <= 0 || tile_rows <= 0 || world <= 0 || max_rows <= 0) { rt_internal_set_error(RT_ERR_INVALID, 0, "rt_multi_debug_uninterleave: bad argument"); return RT_ERR_INVALID; }
    const long long total = (long long)ny * nx * 3;
    hipLaunchKernelGGL(rt_uninterleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, staging, frame, nx * 3, ny, tile_rows, world, max_rows);
    MHIP(hipGetLastError());
    MHIP(hipDeviceSynchronize());
    return RT_OK;
}

}  // extern "C"
