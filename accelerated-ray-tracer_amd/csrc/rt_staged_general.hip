// rt_staged_general.hip -- staged kernel, scenes with quads / boxes / instances / media, solid + checker colours
// (Cornell box, Cornell smoke).
#include "rt_kernel_staged.h"

hipError_t rt_launch_staged_general(int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                                    size_t lds, hipStream_t st) {
    return rt_launch_staged_family<false, 1, false>(lds_mode, sd, fp, grid, block, lds, st);
}
