// rt_kernel_pixel.hip -- kernel 0 ("pixel"), see below.
#include "rt_device_funcs.h"

// =============================================================================
// Kernel A ("pixel"): the reference's own loop nest, one lane per pixel, one
// 8x8 tile per wave.  Kept as the simple form the persistent kernel is checked
// against on the GPU, and as the A/B baseline for the scheduling work.
// =============================================================================
template <bool SPHERES_ONLY, int TEX, bool NEED_UV, int LDS_MODE>
__global__ void __launch_bounds__(256) rt_render_pixel_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    SceneView sc = stage_scene<LDS_MODE>(sd, lds);
    sc.nodes = sd.nodes_ref; sc.n_nodes = sd.n_nodes_ref;   // every node of the reference's tree, straight from memory

    // calibration pass (rt_abi.hip, "collapse"): the per-node pass counts are collected in LDS and added to memory once
    // per workgroup -- a scene of 19 nodes would otherwise funnel every lane's atomics into 19 addresses
    unsigned int* node_pass = fp.node_pass;
    unsigned int* lds_pass = reinterpret_cast<unsigned int*>(lds);
    const bool pass_in_lds = fp.node_pass != nullptr && fp.node_pass_lds != 0;     // grid-uniform
    if (pass_in_lds) {
        for (int k = (int)threadIdx.x; k < sd.n_nodes_ref; k += (int)blockDim.x) lds_pass[k] = 0u;
        __syncthreads();
        node_pass = lds_pass;
    }
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    int i, lrow;
    unsigned long long rays = 0;
    if (w < fp.work_items && work_to_pixel(fp, w, i, lrow)) {
        const int j = local_to_global_row(fp, lrow);
        rt_xorwow g;
        rt_xorwow_seed(g, fp.seed_base + (uint64_t)(j * fp.nx + i));   // render_init, main.cu:101-104
        f3 col = mk3(0, 0, 0);
        for (int s = 0; s < fp.ns; ++s) {
            const float u = ((float)i + rt_xorwow_uniform(g)) / (float)fp.nx;
            const float v = ((float)j + rt_xorwow_uniform(g)) / (float)fp.ny;
            Ray cur = camera_get_ray(sd.camera, u, v, g);
            f3 throughput = mk3(1, 1, 1), radiance = mk3(0, 0, 0);
            for (int bounce = 0; bounce < 50; ++bounce) {
                HitInfo h;
                ++rays;
                const bool hit_something = trace<SPHERES_ONLY>(sc, cur, h, node_pass);
                if (fp.ray_sample && (rays + (unsigned long long)w * 7ull) % fp.ray_sample_stride == 0ull) {
                    // a sample of the pass's rays for the regrouping's view-dependent cost (rt_abi.hip, "Regroup")
                    const unsigned long long at = atomicAdd(fp.ray_counter + 2, 1ull);
                    if (at < fp.ray_sample_cap) {
                        float* o = fp.ray_sample + at * 7ull;
                        o[0] = cur.o.x; o[1] = cur.o.y; o[2] = cur.o.z; o[3] = cur.d.x; o[4] = cur.d.y; o[5] = cur.d.z;
                        o[6] = hit_something ? h.t : FLT_MAX;
                    }
                }
                if (!hit_something) {
                    radiance = fma3(throughput, miss_color(fp, cur), radiance);
                    break;
                }
                const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, cur, h);
                f3 emitted, attenuation;
                Ray scattered;
                const bool go_on = shade<TEX>(sc, cur, rec, g, emitted, attenuation, scattered);
                radiance = fma3(throughput, emitted, radiance);
                if (!go_on) break;
                throughput = throughput * attenuation;
                cur = scattered;
            }
            col = col + radiance;
        }
        store_pixel(fp, i, lrow, col);
        if (fp.pixel_cost) fp.pixel_cost[(size_t)lrow * fp.nx + i] = (unsigned int)rays;   // calibration pass: this pixel's rays (the cost prior)
    }
    if (pass_in_lds) {
        __syncthreads();
        for (int k = (int)threadIdx.x; k < sd.n_nodes_ref; k += (int)blockDim.x)
            if (lds_pass[k] != 0u) atomicAdd(&fp.node_pass[k], lds_pass[k]);
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(fp.ray_counter, rays);
}

// Kernel 0 is the GPU-side cross-check of the staged kernel (tests) and the A/B baseline; it reads the scene through
// L1/L2 (no LDS staging), which is all its role needs.
namespace {
template <bool SO, int TX, bool UV>
hipError_t launch_pixel(const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block, hipStream_t st) {
    const size_t lds = (fp.node_pass != nullptr && fp.node_pass_lds != 0) ? (size_t)sd.n_nodes_ref * sizeof(unsigned int) : 0;
    hipLaunchKernelGGL((rt_render_pixel_kernel<SO, TX, UV, 0>), grid, block, lds, st, sd, fp);
    return hipGetLastError();
}
}  // namespace

hipError_t rt_launch_pixel(bool spheres_only, int tex_level, bool need_uv, const rt_scene_dev& sd, const rt_frame_params& fp,
                           dim3 grid, dim3 block, hipStream_t stream) {
    if (spheres_only) {
        if (tex_level == 0) return launch_pixel<true, 0, false>(sd, fp, grid, block, stream);
        if (tex_level == 1) return launch_pixel<true, 1, false>(sd, fp, grid, block, stream);
        return launch_pixel<true, 2, true>(sd, fp, grid, block, stream);
    }
    if (tex_level <= 1 && !need_uv) return launch_pixel<false, 1, false>(sd, fp, grid, block, stream);
    return launch_pixel<false, 2, true>(sd, fp, grid, block, stream);
}
