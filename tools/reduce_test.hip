// reduce_test.hip -- checks, on the device, the lane mapping of dirt_reduce.h's row_reduce_scatter (bank-masked DPP adds,
// quad permutations) against plain sums.  Prints PASS / FAIL.
//   hipcc --offload-arch=gfx950 -O2 tools/reduce_test.hip -o tools/_bin/reduce_test
#include "../dirt_amd/csrc/dirt_reduce.h"
#include <cstdio>
#include <vector>

template <int N>
__global__ void k(const float* in, float* out)
{
    const int lane = threadIdx.x;
    float v[N];
    for (int i = 0; i < N; ++i) v[i] = in[i * 64 + lane];
    float d0, d1;
    dirt::row_reduce_scatter<N>(v, lane, d0, d1);
    int v0, v1;
    dirt::row_value_of_lane<N>(lane & 15, v0, v1);
    out[lane] = d0; out[64 + lane] = d1; out[128 + lane] = (float)v0; out[192 + lane] = (float)v1;
}

// the same with compile-time constants at the SAME positions of both halves (a padding zero at the end of each half, and an
// equal non-zero constant further in): lo[i] and hi[i] then hold one SSA value, which the register allocator may give one
// register unless the asm blocks' in/out operands are early-clobber
template <int N>
__global__ void kz(const float* in, float* out)
{
    const int lane = threadIdx.x;
    float v[N];
    for (int i = 0; i < N; ++i) v[i] = in[i * 64 + lane];
    v[N / 2 - 1] = 0.f; v[N - 1] = 0.f;
    v[2] = 5.f; v[N / 2 + 2] = 5.f;
    float d0, d1;
    dirt::row_reduce_scatter<N>(v, lane, d0, d1);
    int v0, v1;
    dirt::row_value_of_lane<N>(lane & 15, v0, v1);
    out[lane] = d0; out[64 + lane] = d1; out[128 + lane] = (float)v0; out[192 + lane] = (float)v1;
}

template <int N>
bool run(bool constants = false)
{
    std::vector<float> h(N * 64);
    for (int i = 0; i < N; ++i)
        for (int l = 0; l < 64; ++l) {
            h[i * 64 + l] = (float)((i + 1) * 1000 + ((l * 7 + i * 3) % 61));  // exact in float
            if (constants && (i == N / 2 - 1 || i == N - 1)) h[i * 64 + l] = 0.f;
            if (constants && (i == 2 || i == N / 2 + 2)) h[i * 64 + l] = 5.f;
        }
    float *din, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    if (constants) hipLaunchKernelGGL(kz<N>, dim3(1), dim3(64), 0, 0, din, dout);
    else hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, din, dout);
    float o[256];
    hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    bool ok = true;
    bool seen[4][32] = {};
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4;
        for (int s = 0; s < 2; ++s) {
            const int v = (int)o[128 + 64 * s + l];
            if (v < 0) continue;
            double want = 0;
            for (int m = 16 * row; m < 16 * row + 16; ++m) want += h[v * 64 + m];
            seen[row][v] = true;
            if ((double)o[64 * s + l] != want) { ok = false; printf("N=%d lane %d d%d: got %.1f want %.1f (value %d)\n", N, l, s, o[64 * s + l], want, v); }
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int v = 0; v < N; ++v)
            if (!seen[r][v]) { ok = false; printf("N=%d row %d: value %d lands nowhere\n", N, r, v); }
    hipFree(din); hipFree(dout);
    return ok;
}

int main()
{
    const bool ok = run<24>() & run<16>() & run<32>() & run<24>(true) & run<16>(true) & run<32>(true);
    printf("reduce_test: %s\n", ok ? "PASS" : "FAIL");
    return ok ? 0 : 1;
}
