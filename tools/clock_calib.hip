// Calibrates clock64() / wall_clock64() against a dependent-FMA chain and host time on gfx950.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(long long* out, int n, float seed)
{
    long long c0 = clock64(), w0 = wall_clock64();
    float x = seed;
    for (int i = 0; i < n; ++i) x = fmaf(x, 1.0000001f, 0.5f);  // dependent chain
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main()
{
    int clk = 0, wclk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("hipDeviceAttributeClockRate %d kHz, WallClockRate %d kHz\n", clk, wclk);
    long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 3; ++rep)
        for (int n : {1000, 100000, 10000000}) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::high_resolution_clock::now();
            hipLaunchKernelGGL(k, dim3(rep == 2 ? 1024 : 1), dim3(64), 0, 0, d, n, 1.0f);
            hipDeviceSynchronize();
            auto t1 = std::chrono::high_resolution_clock::now();
            long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
            printf("grid=%4d n=%8d: clock64 delta %10lld (%.2f per fma)  wall_clock64 delta %10lld  host %.1f us -> clock64 = %.1f MHz, wall = %.1f MHz\n",
                   rep == 2 ? 1024 : 1, n, h[0], (double)h[0] / n, h[1], us, h[0] / us, h[1] / us);
        }
    return 0;
}
