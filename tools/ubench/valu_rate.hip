// Microbenchmark: issue rate of plain vs packed fp32 VALU ops on gfx950 at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
        if (OP == 2) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
        if (OP == 4) { REP16(asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_min3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 5) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 6) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_rcp_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 7) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_lshlrev_b32 %1, 3, %1\n v_add_u32 %2, %2, %4\n v_xad_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y;
}
template <int OP> void run(const char* name, float* d) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int threads = 256 * waves_per_simd;   // one block per CU: 4 SIMDs x waves
        int blocks = 256;
        int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (threads <= 1024) { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 10); }
        else { threads = 1024; blocks = 256 * waves_per_simd / 4; hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 10); }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_wave = (double)iters * 64;
        double waves_total = (double)blocks * threads / 64;
        double per_simd = instr_per_wave * waves_total / 1024;
        printf("%-28s waves/SIMD %d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / per_simd);
    }
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_add/v_mul_f32", d); run<3>("v_pk_add/v_pk_mul_f32", d);
    run<4>("v_min/max/min3/max3_f32", d); run<5>("v_cmp+v_cndmask", d); run<6>("v_rcp/v_sqrt", d); run<7>("int xor/shl/add/xor3", d);
    return 0;
}
