// permlane_test.hip -- checks, on the device, the lane mapping dirt_grad.hip's wave_reduce_scatter relies on:
// v_permlane32_swap / v_permlane16_swap semantics and the 4-step DPP row sum.  Prints PASS / FAIL.
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_test.hip -o tools/_bin/permlane_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v)
{
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); return dpp_add<0x140>(v);
}
template <int NV4>
__device__ __forceinline__ float wave_reduce_scatter(const float* val, int lane)
{
    constexpr int H1 = NV4 / 2, H2 = NV4 / 4;
    float r1[H1];
#pragma unroll
    for (int i = 0; i < H1; ++i) {
        const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(val[i]), __float_as_uint(val[i + H1]), false, false);
        r1[i] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
    float r2[H2];
#pragma unroll
    for (int i = 0; i < H2; ++i) {
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1[i]), __float_as_uint(r1[i + H2]), false, false);
        r2[i] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
    const int c = lane & 15;
    float out = row_sum16(r2[0]);
#pragma unroll
    for (int i = 1; i < H2; ++i) {
        const float t = row_sum16(r2[i]);
        out = (c == i) ? t : out;
    }
    return out;
}

template <int NV4>
__global__ void k(const float* in, float* out)
{
    const int lane = threadIdx.x;
    float v[NV4];
    for (int i = 0; i < NV4; ++i) v[i] = in[i * 64 + lane];
    out[lane] = wave_reduce_scatter<NV4>(v, lane);
}

template <int NV4>
bool run()
{
    std::vector<float> h(NV4 * 64);
    for (int i = 0; i < NV4; ++i)
        for (int l = 0; l < 64; ++l) h[i * 64 + l] = (float)((i + 1) * 1000 + l);  // exact in float
    float *din, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<NV4>, dim3(1), dim3(64), 0, 0, din, dout);
    float o[64];
    hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < NV4 / 4; ++c) {
            const int v = c + (NV4 / 4) * r;
            double want = 0;
            for (int l = 0; l < 64; ++l) want += h[v * 64 + l];
            if ((double)o[16 * r + c] != want) { ok = false; printf("NV4=%d lane %d: got %.1f want %.1f (value %d)\n", NV4, 16 * r + c, o[16 * r + c], want, v); }
        }
    hipFree(din); hipFree(dout);
    return ok;
}

int main()
{
    const bool ok = run<24>() & run<20>() & run<16>() & run<12>();
    printf("permlane_test: %s\n", ok ? "PASS" : "FAIL");
    return ok ? 0 : 1;
}
