// tools/xcd_map.hip -- which XCD runs workgroup i?  Every workgroup of a launch reads the hardware's XCC_ID register; the host
// prints how many workgroups broke the rule "workgroup i runs on XCD i % 8" (the rule the decode kernel's XCD-range chunk mapping
// rests on, dcs_kernels.hip.h), for a launch that fits the chip and for one that does not.
//   hipcc --offload-arch=gfx950 -O2 -o build/xcd_map xcd_map.hip && ./build/xcd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void whoRuns(unsigned *out, int spin)
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    for (int i = 0 ; i < spin ; ++i)
        asm volatile("s_sleep 8");
    if (threadIdx.x == 0)
        out[blockIdx.x] = x & 15u;
}
int main()
{
    for (int blocks : { 64, 1024, 8192, 65536 })
    {
        unsigned *d = nullptr;
        hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned) * blocks);
        hipLaunchKernelGGL(whoRuns, dim3(blocks), dim3(256), 0, 0, d, 200);
        std::vector<unsigned> h(blocks);
        hipMemcpy(h.data(), d, sizeof(unsigned) * blocks, hipMemcpyDeviceToHost);
        int off = 0; unsigned seen = 0;
        for (int i = 0 ; i < blocks ; ++i) { off += h[i] != static_cast<unsigned>(i % 8); seen |= 1u << h[i]; }
        printf("%6d workgroups: %d not on XCD (index %% 8); XCDs seen 0x%x; first sixteen: ", blocks, off, seen);
        for (int i = 0 ; i < 16 ; ++i) printf("%u ", h[i]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
