// rt_staged_spheres.hip -- staged kernel, spheres-only scenes with inline / solid / checker colours (the headline
// random scene: <true, 1, false>).
#include "rt_kernel_staged.h"

hipError_t rt_launch_staged_spheres(int tex_level, int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid,
                                    dim3 block, size_t lds, hipStream_t st) {
    if (tex_level == 0) return rt_launch_staged_family<true, 0, false>(lds_mode, sd, fp, grid, block, lds, st);
    return rt_launch_staged_family<true, 1, false>(lds_mode, sd, fp, grid, block, lds, st);
}
