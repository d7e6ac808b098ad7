// rt_staged_general_tex.hip -- staged kernel, general scenes with every texture kind (Book-2 final, original).
#include "rt_kernel_staged.h"

hipError_t rt_launch_staged_general_tex(int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                                        size_t lds, hipStream_t st) {
    return rt_launch_staged_family<false, 2, true>(lds_mode, sd, fp, grid, block, lds, st);
}
