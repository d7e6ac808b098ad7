// Does a wave64 VALU op cost less when only lanes 0..31 (or 0..15) are active?  (EXEC-mask half skipping)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k(float* out, int iters, int active_lanes) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;
    if ((threadIdx.x & 63) < active_lanes) {
        for (int i = 0; i < iters; ++i) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24);
    for (int lanes : {64, 48, 32, 16, 1}) {
        const int threads = 1024, blocks = 256, iters = 2000;   // 4 waves per SIMD
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 10, lanes); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, lanes);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double per_simd = (double)iters * 64 * (blocks * threads / 64) / 1024;
        printf("active lanes %2d: %.3f ms -> %.2f cycles per wave-instruction per SIMD\n", lanes, ms, ms * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
