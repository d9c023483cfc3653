// tools/pcie_bw.hip -- what the link between HBM and pinned host memory does on this box: hipMemcpyAsync device-to-host on one
// and on two streams, and a copy KERNEL that stores straight into the pinned buffer (few workgroups, 16 bytes per lane).
// hipcc --offload-arch=gfx950 -O3 tools/pcie_bw.hip -o tools/build/pcie_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void copyKernel(const uint4 *src, uint4 *dst, size_t n)
{
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x ; i < n ; i += size_t(gridDim.x) * blockDim.x)
        dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = size_t(32) << 20;
    const int reps = 40;
    void *d[2], *h[2];
    hipStream_t s[2];
    for (int i = 0 ; i < 2 ; ++i)
    {
        CK(hipMalloc(&d[i], bytes)); CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault)); CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        CK(hipMemset(d[i], 1, bytes));
    }
    CK(hipDeviceSynchronize());
    for (int nstreams = 1 ; nstreams <= 2 ; ++nstreams)
        for (int dir = 0 ; dir < 2 ; ++dir)
        {
            for (int w = 0 ; w < 2 ; ++w)
            {
                const double t0 = now();
                for (int r = 0 ; r < reps ; ++r)
                    for (int i = 0 ; i < nstreams ; ++i)
                        CK(dir == 0 ? hipMemcpyAsync(h[i], d[i], bytes, hipMemcpyDeviceToHost, s[i]) : hipMemcpyAsync(d[i], h[i], bytes, hipMemcpyHostToDevice, s[i]));
                for (int i = 0 ; i < nstreams ; ++i) CK(hipStreamSynchronize(s[i]));
                if (w == 1)
                    printf("hipMemcpyAsync %s, %d stream(s): %.1f GB/s\n", dir == 0 ? "D2H" : "H2D", nstreams, reps * nstreams * bytes / (now() - t0) / 1e9);
            }
        }
    // both directions at once
    {
        const double t0 = now();
        for (int r = 0 ; r < reps ; ++r)
        {
            CK(hipMemcpyAsync(h[0], d[0], bytes, hipMemcpyDeviceToHost, s[0]));
            CK(hipMemcpyAsync(d[1], h[1], bytes / 4, hipMemcpyHostToDevice, s[1]));
        }
        CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
        printf("D2H with H2D (a quarter as much) next to it: %.1f GB/s D2H\n", reps * bytes / (now() - t0) / 1e9);
    }
    for (int blocks : { 8, 16, 32, 64, 128, 256, 1024 })
    {
        for (int w = 0 ; w < 2 ; ++w)
        {
            const double t0 = now();
            for (int r = 0 ; r < reps ; ++r)
                hipLaunchKernelGGL(copyKernel, dim3(blocks), dim3(256), 0, s[0], static_cast<const uint4 *>(d[0]), static_cast<uint4 *>(h[0]), bytes / 16);
            CK(hipStreamSynchronize(s[0]));
            if (w == 1)
                printf("copy kernel to pinned memory, %4d workgroups: %.1f GB/s\n", blocks, reps * bytes / (now() - t0) / 1e9);
        }
    }
    return 0;
}
