// rt_group_spheres.hip -- the group kernel (rt_kernel_group.h) for spheres-only scenes, 8 or 16 lanes per pixel.
#include "rt_kernel_group.h"

hipError_t rt_launch_group_spheres(int tex_level, int lanes, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st) {
    if (tex_level == 0) return rt_launch_group_one<0, false>(lanes, sd, fp, grid, lds, st);
    return rt_launch_group_one<1, false>(lanes, sd, fp, grid, lds, st);
}
