// Microbenchmark: cost of LDS atomics on gfx950 under different lane/address patterns.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, typename T>
__global__ void k(long long* out, int active_lanes, int lanes_per_addr, int iters)
{
    __shared__ T s[2304];
    for (int i = threadIdx.x; i < 2304; i += blockDim.x) s[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool act = lane < active_lanes;
    const int addr = wave * 256 + (lane / lanes_per_addr);  // consecutive addresses = distinct banks
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (act) {
            if (MODE == 0) atomicAdd(&s[addr], (T)1);
            else if (MODE == 1) { T v = atomicAdd(&s[addr], (T)1); if (v == (T)-12345) s[0] = v; }
            else if (MODE == 3) atomicMax(&s[addr], (T)i);
            else { s[addr] = s[addr] + (T)1; }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s[threadIdx.x] == (T)-1) out[0] = 0;
}

template <int MODE, typename T>
void run(const char* name, int threads, int active, int lpa)
{
    long long* d; hipMalloc(&d, 8 * 256);
    const int iters = 1000;
    hipLaunchKernelGGL((k<MODE, T>), dim3(256), dim3(threads), 0, 0, d, active, lpa, iters);
    hipLaunchKernelGGL((k<MODE, T>), dim3(256), dim3(threads), 0, 0, d, active, lpa, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/WG=%d active=%2d lanes/addr=%2d : %.1f clk per wave-instr (per WG wall / iters)\n", name, threads / 64, active, lpa, (double)h / iters);
    hipFree(d);
}

int main()
{
    for (int threads : {64, 512}) {
        run<0, float>("ds_add_f32 (no return)", threads, 64, 1);
        run<0, float>("ds_add_f32 (no return)", threads, 16, 1);
        run<0, float>("ds_add_f32 (no return)", threads, 16, 4);
        run<0, float>("ds_add_f32 (no return)", threads, 64, 4);
        run<0, float>("ds_add_f32 (no return)", threads, 64, 16);
        run<0, float>("ds_add_f32 (no return)", threads, 64, 64);
        run<0, float>("ds_add_f32 (no return)", threads, 4, 1);
        run<0, float>("ds_add_f32 (no return)", threads, 1, 1);
        run<1, float>("ds_add_rtn_f32", threads, 16, 1);
        run<0, unsigned>("ds_add_u32 (no return)", threads, 64, 1);
        run<0, unsigned>("ds_add_u32 (no return)", threads, 16, 4);
        run<0, unsigned>("ds_add_u32 (no return)", threads, 64, 64);
        run<2, float>("plain read+write", threads, 64, 1);
        run<0, unsigned long long>("ds_add_u64 (no return)", threads, 64, 1);
        run<0, unsigned long long>("ds_add_u64 (no return)", threads, 16, 1);
        run<0, unsigned long long>("ds_add_u64 (no return)", threads, 16, 4);
        run<0, unsigned long long>("ds_add_u64 (no return)", threads, 64, 4);
        run<0, unsigned long long>("ds_add_u64 (no return)", threads, 64, 64);
        run<3, unsigned>("ds_max_u32 (no return)", threads, 64, 64);
        run<3, unsigned>("ds_max_u32 (no return)", threads, 1, 1);
    }
    return 0;
}
