// what a TAKEN scalar branch costs a wavefront that has its SIMD to itself (tools/branch_cost.hip)
//   hipcc --offload-arch=gfx950 -O2 -o tools/branch_cost_bin tools/branch_cost.hip && tools/branch_cost_bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void k(unsigned long long *out, int mode)
{
    unsigned long long t0 = 0, t1 = 0;
    unsigned a = threadIdx.x, s = 0;
    for (int rep = 0 ; rep < 3 ; ++rep)
    {
        t0 = __builtin_amdgcn_s_memtime();
        if (mode == 0)          // 64 not-taken branches (scc = 0)
            asm volatile("s_cmp_eq_u32 0, 1\n" REP64("s_cbranch_scc1 9f\n") "9:\n" ::: "scc");
        else if (mode == 1)     // 64 taken branches, each to the next instruction
            asm volatile(REP64("s_branch 1f\n1:\n") :::);
        else if (mode == 2)     // 64 taken branches, each over 16 instructions (a different fetch line)
            asm volatile(REP64("s_branch 1f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n1:\n") :::);
        else if (mode == 3)     // 64 dependent scalar adds
            asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s) :: "scc");
        else if (mode == 4)     // 64 dependent vector adds
            asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(a));
        else if (mode == 5)     // 64 x (v_readlane -> s_add -> v_readlane ...): the index walk's chain without its test
            asm volatile(REP64("v_readlane_b32 %1, %0, %1\n s_add_u32 %1, %1, 1\n") : "+v"(a), "+s"(s) :: "scc");
        else                    // 64 x chain step with the untaken exit test
            asm volatile("s_mov_b32 s40, 0\n" REP64("v_readlane_b32 s41, %0, s40\n s_add_u32 s40, s40, s41\n s_and_b32 s42, s40, 0x80000000\n s_cbranch_scc1 8f\n") "8:\n" :: "v"(0u) : "s40", "s41", "s42", "scc");
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + s; }
}
int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const char *names[] = { "64 branches not taken", "64 branches taken (next instruction)", "64 branches taken (over 16 instructions)", "64 dependent s_add", "64 dependent v_add",
                            "64 x (v_readlane, s_add) dependent", "64 x chain step (v_readlane, s_add, s_and, s_cbranch not taken)" };
    for (int mode = 0 ; mode < 7 ; ++mode)
    {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-70s %6llu cycles = %.1f each\n", names[mode], h[0], h[0] / 64.0);
    }
    return 0;
}
