// instr_bench.hip -- cycles per instruction of the cross-lane primitives a wave-level reduction can be built from,
// measured with s_memtime around unrolled runs (one wave, and 4 waves per SIMD to see throughput under contention).
//   hipcc --offload-arch=gfx950 -O2 tools/instr_bench.hip -o tools/_bin/instr_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 1024

template <int KIND>
__device__ __forceinline__ void op(float& a, float& b)
{
    if (KIND == 0) {  // plain v_add
        a = a + b;
    } else if (KIND == 1) {  // DPP quad_perm add
        a = a + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0xB1, 0xF, 0xF, true));
    } else if (KIND == 2) {  // DPP row_mirror add
        a = a + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x140, 0xF, 0xF, true));
    } else if (KIND == 3) {  // DPP row_bcast15 add
        a = a + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x142, 0xF, 0xF, true));
    } else if (KIND == 4) {  // DPP row_bcast31 add
        a = a + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x143, 0xF, 0xF, true));
    } else if (KIND == 5) {  // permlane32_swap + add
        const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    } else if (KIND == 6) {  // permlane16_swap + add
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    } else if (KIND == 7) {  // ds_bpermute (xor 32)
        a = a + __int_as_float(__builtin_amdgcn_ds_bpermute(((int)threadIdx.x ^ 32) << 2, __float_as_int(a)));
    } else if (KIND == 8) {  // ds_swizzle (swap halves of 32: not cross-32) -- quad mode as a cost probe
        a = a + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a), 0x8000 | 0xB1));
    } else if (KIND == 9) {  // v_readlane + v_add with SGPR
        a = a + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    } else if (KIND == 10) {  // wave_shr:1 style DPP (0x138)
        a = a + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x138, 0xF, 0xF, true));
    } else if (KIND == 11) {  // v_cndmask + fma pair (the A-part building block)
        a = fmaf(b > 0.5f ? b : 0.f, 1.0001f, a);
    }
}

template <int KIND>
__global__ void k(float* out, long long* cyc)
{
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = (float)threadIdx.x + i; b[i] = 0.25f * i; }
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    long long t0, t1;
    const long long w0 = wall_clock64();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) op<KIND>(a[i], b[i]);
    }
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
    const long long w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) { cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t1 - t0; cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = w1 - w0; }
}

template <int KIND>
void run(const char* name)
{
    float* out; long long* cyc;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 16 * 16384);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg == 0 ? 1 : 256 * 4, threads = cfg == 0 ? 64 : 256;  // one wave; 4 waves per SIMD on every CU
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
        hipDeviceSynchronize();
        std::vector<long long> h(2 * blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0, w = 0; for (size_t i = 0; i < h.size(); i += 2) { s += (double)h[i]; w += (double)h[i + 1]; }
        const double nw = h.size() / 2;
        printf("%-28s %s: %.1f s_memtime ticks, %.2f ns per op per wave (wall clock)\n", name, cfg == 0 ? "1 wave      " : "4 waves/SIMD",
               s / nw / (REP * 4), w / nw * 10.0 / (REP * 4));
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("v_add_f32");
    run<1>("v_add dpp quad_perm");
    run<2>("v_add dpp row_mirror");
    run<3>("v_add dpp row_bcast15");
    run<4>("v_add dpp row_bcast31");
    run<5>("permlane32_swap + add");
    run<6>("permlane16_swap + add");
    run<7>("ds_bpermute + add");
    run<8>("ds_swizzle + add");
    run<9>("v_readlane + add");
    run<10>("v_add dpp wave_shr");
    run<11>("cmp + cndmask + fma");
    return 0;
}
