// rt_staged_spheres_tex.hip -- staged kernel, spheres-only scenes with noise / image / noodle / felt textures.
#include "rt_kernel_staged.h"

hipError_t rt_launch_staged_spheres_tex(int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                                        size_t lds, hipStream_t st) {
    return rt_launch_staged_family<true, 2, true>(lds_mode, sd, fp, grid, block, lds, st);
}
