// rt_rank.hip -- ranking kernels of the cost-aware schedule.
#include <hip/hip_runtime.h>
#include "rt_device.h"

// Heavy-pixel list for the cost-aware schedule: every pixel whose prepass ray count reaches `threshold` is appended as
// (cost << 32 | pixel); the host sorts the (short) list by descending cost.
__global__ void rt_collect_heavy_kernel(const rt_pixel_state* state, unsigned int n_pixels, unsigned int threshold,
                                        unsigned long long* list, unsigned int capacity, unsigned int* count) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const unsigned int c = state[i].cost;
    if (c >= threshold) {
        const unsigned int at = atomicAdd(count, 1u);
        if (at < capacity) list[at] = ((unsigned long long)c << 32) | i;
    }
}
void rt_launch_collect_heavy(const rt_pixel_state* state, unsigned int n_pixels, unsigned int threshold, unsigned long long* list,
                             unsigned int capacity, unsigned int* count, hipStream_t st) {
    hipLaunchKernelGGL(rt_collect_heavy_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, st, state, n_pixels, threshold, list, capacity, count);
}
