// what a host thread's wait for the GPU costs in CPU time (tools/wait_cost.hip; hipcc --offload-arch=gfx950 -O2 -o /tmp/wait_cost tools/wait_cost.hip)
//   ./wait_cost [kernel ms] : a kernel of about that many milliseconds, waited for in four ways; wall and thread-CPU time of the wait
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <condition_variable>
#include <mutex>

__global__ void spin(unsigned long long ticks, unsigned *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (out) *out = 1;
}
static double now(clockid_t c) { timespec t; clock_gettime(c, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int main(int argc, char **argv)
{
    const double ms = argc > 1 ? atof(argv[1]) : 8.0;
    const unsigned long long ticks = static_cast<unsigned long long>(ms * 1e5);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t evSpin, evBlock;
    hipEventCreateWithFlags(&evSpin, hipEventDisableTiming);
    hipEventCreateWithFlags(&evBlock, hipEventDisableTiming | hipEventBlockingSync);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000ull, nullptr); hipStreamSynchronize(s);
    struct Done { std::mutex m; std::condition_variable cv; bool done = false; } done;
    for (int mode = 0 ; mode < 4 ; ++mode)
    {
        double wall = 0, cpu = 0;
        for (int rep = 0 ; rep < 10 ; ++rep)
        {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, nullptr);
            if (mode == 1) hipEventRecord(evSpin, s);
            if (mode == 2) hipEventRecord(evBlock, s);
            if (mode == 3)
            {
                done.done = false;
                hipLaunchHostFunc(s, [](void *p) { Done *d = static_cast<Done *>(p); { std::lock_guard<std::mutex> l(d->m); d->done = true; } d->cv.notify_one(); }, &done);
            }
            const double w0 = now(CLOCK_MONOTONIC), c0 = now(CLOCK_THREAD_CPUTIME_ID);
            if (mode == 0) hipStreamSynchronize(s);
            if (mode == 1) hipEventSynchronize(evSpin);
            if (mode == 2) hipEventSynchronize(evBlock);
            if (mode == 3) { std::unique_lock<std::mutex> l(done.m); done.cv.wait(l, [&] { return done.done; }); }
            wall += now(CLOCK_MONOTONIC) - w0; cpu += now(CLOCK_THREAD_CPUTIME_ID) - c0;
        }
        static const char *names[] = { "hipStreamSynchronize", "hipEventSynchronize (default event)", "hipEventSynchronize (blocking-sync event)", "hipLaunchHostFunc + condition variable" };
        printf("%-44s wall %.2f ms, this thread's CPU %.2f ms per wait\n", names[mode], wall / 10, cpu / 10);
    }
    timespec pt; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &pt);
    printf("process CPU in total %.1f ms\n", pt.tv_sec * 1e3 + pt.tv_nsec * 1e-6);
    return 0;
}
