// rt_tier_general.hip -- the tier kernel (rt_kernel_tier.h) for scenes with quads, boxes, instances or media.
#include "rt_kernel_tier.h"

hipError_t rt_launch_tier_general(int tex_level, bool need_uv, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st) {
    if (tex_level <= 1 && !need_uv) return rt_launch_tier_one<false, 1, false>(sd, fp, grid, lds, st);
    return rt_launch_tier_one<false, 2, true>(sd, fp, grid, lds, st);
}
