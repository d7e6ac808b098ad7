// rt_tier_spheres.hip -- the tier kernel (rt_kernel_tier.h) for spheres-only scenes: texture levels 0 / 1 (the lean
// family: <= 128 VGPRs, co-resident with the main kernel) and 2 (Perlin / image textures).
#include "rt_kernel_tier.h"

hipError_t rt_launch_tier_spheres(int tex_level, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st) {
    if (tex_level == 0) return rt_launch_tier_one<true, 0, false>(sd, fp, grid, lds, st);
    if (tex_level == 1) return rt_launch_tier_one<true, 1, false>(sd, fp, grid, lds, st);
    return rt_launch_tier_one<true, 2, true>(sd, fp, grid, lds, st);
}
