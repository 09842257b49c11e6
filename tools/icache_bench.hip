// icache_bench.hip -- does a wave issue straight-line code as fast as a loop?  The same 8 x 512 independent v_add_f32 /
// v_fma_f32 (VOP3, 8 bytes) once as a 64-instruction loop body executed 64 times (hot in the instruction cache) and once
// as 4096 instructions of straight-line code (32 KB: every wave streams it through the instruction cache once).
//   hipcc --offload-arch=gfx950 -O2 tools/icache_bench.hip -o tools/_bin/icache_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define I8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64
#define I4096 I512 I512 I512 I512 I512 I512 I512 I512
#define ASM(body) asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1))
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b0 = 1.0001f, b1 = 0.5f;
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) { for (int r = 0; r < 64; ++r) ASM(I64); }
    else ASM(I4096);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks, int threads, float* out, long long* cyc)
{
    for (int rep = 0; rep < 3; ++rep) {   // rep 0: cold instruction cache
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
        hipDeviceSynchronize();
        const int nw = blocks * threads / 64;
        std::vector<long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        double s = 0, mx = 0; for (auto v : h) { s += (double)v; if (v > mx) mx = (double)v; }
        printf("%-22s blocks=%4d x %3d threads  launch %d: %.2f clk/instr/wave (max %.2f)\n", name, blocks, threads, rep, s / nw / 4096.0, mx / 4096.0);
    }
}
int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 4 * 1024 * 256 * 4); hipMalloc(&cyc, 8 * 8192);
    run<0>("loop of 64 x 64", 1, 64, out, cyc);
    run<1>("straight line 4096", 1, 64, out, cyc);
    run<0>("loop of 64 x 64", 157, 64, out, cyc);
    run<1>("straight line 4096", 157, 64, out, cyc);
    run<0>("loop of 64 x 64", 1024, 256, out, cyc);
    run<1>("straight line 4096", 1024, 256, out, cyc);
    return 0;
}
