// Microbenchmark: does a small kernel on a second stream start while a persistent kernel that fills the launch grid's
// share of every CU is still resident?  (Design question behind csrc/rt_kernel_tier.hip: the tier kernel of a ranked
// launch runs on a side stream next to the main render kernel.)
// Kernel A imitates the main render kernel: `a_wgs` workgroups of `a_threads` threads, `a_lds` bytes of dynamic LDS, at
// least `A_VGPRS` registers, each spinning for `a_ms`.  Kernel B imitates the tier kernel: `b_wgs` x 256 threads, 128
// registers, 20 KB LDS, spinning 1 ms.  Reported: when B's first and last workgroup STARTED relative to A's first start
// and A's last end.  Build: hipcc --offload-arch=gfx950 -O3 concurrent_kernels.hip -o concurrent_kernels.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int VGPRS>
__device__ __forceinline__ void touch_regs() {
    if (VGPRS >= 168) asm volatile("v_mov_b32 v166, 0" ::: "v166");
    else if (VGPRS >= 128) asm volatile("v_mov_b32 v126, 0" ::: "v126");
    else if (VGPRS >= 96) asm volatile("v_mov_b32 v94, 0" ::: "v94");
}

template <int VGPRS, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) spin_kernel(unsigned long long* start, unsigned long long* end, unsigned long long ticks) {
    extern __shared__ unsigned char lds[];
    touch_regs<VGPRS>();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; lds[0] = 1; }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int AV, int AT, int AW>
int run_case(const char* name, int a_wgs, int a_threads, int a_lds, int b_wgs, bool high_priority, bool with_event, bool b_first) {
    unsigned long long *sa, *ea, *sb, *eb;
    CHK(hipMalloc(&sa, a_wgs * 8)); CHK(hipMalloc(&ea, a_wgs * 8)); CHK(hipMalloc(&sb, b_wgs * 8)); CHK(hipMalloc(&eb, b_wgs * 8));
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (high_priority) CHK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi)); else CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev; CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (a_lds > 65536) CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel<AV, AT, AW>), hipFuncAttributeMaxDynamicSharedMemorySize, a_lds));
    const unsigned long long a_ticks = 2000000ull /* 20 ms */, b_ticks = 100000ull /* 1 ms */;
    for (int rep = 0; rep < 3; ++rep) {
        if (with_event) {
            hipLaunchKernelGGL((spin_kernel<0, 256, 1>), dim3(1), dim3(64), 64, s1, sb, eb, 100ull);   // stands for the ranking kernels
            CHK(hipEventRecord(ev, s1));
            CHK(hipStreamWaitEvent(s2, ev, 0));
        }
        if (b_first) hipLaunchKernelGGL((spin_kernel<128, 256, 1>), dim3(b_wgs), dim3(256), 20480, s2, sb, eb, b_ticks);
        hipLaunchKernelGGL((spin_kernel<AV, AT, AW>), dim3(a_wgs), dim3(a_threads), a_lds, s1, sa, ea, a_ticks);
        if (!b_first) hipLaunchKernelGGL((spin_kernel<128, 256, 1>), dim3(b_wgs), dim3(256), 20480, s2, sb, eb, b_ticks);
        CHK(hipGetLastError());
        CHK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> hsa(a_wgs), hea(a_wgs), hsb(b_wgs), heb(b_wgs);
    CHK(hipMemcpy(hsa.data(), sa, a_wgs * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hea.data(), ea, a_wgs * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hsb.data(), sb, b_wgs * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(heb.data(), eb, b_wgs * 8, hipMemcpyDeviceToHost));
    const unsigned long long a0 = *std::min_element(hsa.begin(), hsa.end()), a_last_start = *std::max_element(hsa.begin(), hsa.end()), a1 = *std::max_element(hea.begin(), hea.end());
    const unsigned long long b0 = *std::min_element(hsb.begin(), hsb.end()), b_last_start = *std::max_element(hsb.begin(), hsb.end()), b1 = *std::max_element(heb.begin(), heb.end());
    auto us = [&](unsigned long long t) { return ((double)t - (double)a0) / 100.0; };
    printf("%-64s A: last start %9.1f us, last end %9.1f us | B: first start %9.1f us, last start %9.1f us, last end %9.1f us\n", name, us(a_last_start), us(a1), us(b0), us(b_last_start), us(b1));
    hipFree(sa); hipFree(ea); hipFree(sb); hipFree(eb); hipStreamDestroy(s1); hipStreamDestroy(s2); hipEventDestroy(ev);
    return 0;
}

int main() {
    // lean family: 2 x 512 threads per CU at <= 96 VGPRs, 45 KB LDS each; B (4 waves, 128 VGPRs, 20 KB) fits beside them
    run_case<96, 512, 4>("lean 512x512 (96 VGPR), B 64 wgs after A, plain stream", 512, 512, 46080, 64, false, false, false);
    run_case<96, 512, 4>("lean 512x512, B 64 wgs after A, high-priority stream", 512, 512, 46080, 64, true, false, false);
    run_case<96, 512, 4>("lean 512x512, B 64 wgs after A, high priority, behind an event", 512, 512, 46080, 64, true, true, false);
    run_case<96, 512, 4>("lean 512x512, B 256 wgs BEFORE A, high priority, behind an event", 512, 512, 46080, 256, true, true, true);
    run_case<96, 512, 4>("lean 512x512, B 512 wgs after A (two per CU), high priority", 512, 512, 46080, 512, true, false, false);
    // the lean kernel as it is today (128 VGPRs): 4 x 128 fills the register file, B cannot co-reside
    run_case<128, 512, 4>("lean 512x512 at 128 VGPR, B 64 wgs after A, high priority", 512, 512, 46080, 64, true, false, false);
    // general family: 3 x 256 threads per CU at 168 VGPRs: no room for B until an A workgroup leaves ...
    run_case<168, 768, 3>("general 768x256 (168 VGPR), B 64 wgs after A, high priority", 768, 256, 1024, 64, true, false, false);
    // ... unless A's grid leaves the room (704 = 768 - 64)
    run_case<168, 768, 3>("general 704x256 (168 VGPR), B 64 wgs after A, high priority", 704, 256, 1024, 64, true, false, false);
    run_case<168, 768, 3>("general 704x256 (168 VGPR), B 64 wgs BEFORE A, behind an event", 704, 256, 1024, 64, true, true, true);
    // final scene: one 768-thread workgroup per CU (102 KB LDS): B at 128 VGPRs next to 3 x 168 does not fit; at 2 x 168 + 128 it does
    run_case<168, 768, 3>("final 256x768 (168 VGPR, 102 KB), B 64 wgs after A", 256, 768, 104448, 64, true, false, false);
    run_case<168, 768, 2>("final 256x512 (168 VGPR, 102 KB), B 64 wgs after A", 256, 512, 104448, 64, true, false, false);
    return 0;
}
