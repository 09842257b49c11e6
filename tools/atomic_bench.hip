// atomic_bench.hip -- throughput of global float atomic adds under the gradient kernel's launch shape (1024 workgroups of
// 4 waves), by address pattern, active lanes per instruction and memory scope.  Prints lane-atomics per microsecond.
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics tools/atomic_bench.hip -o tools/_bin/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// PATTERN 0: every active lane its own random row (16-byte rows), component lane & 3
// PATTERN 1: the face pattern: lanes 0 .. 20 of each 32-lane group = 3 vertices x 7 values (x, y, w of a [V,4] row and 4 colours of
//            another [V,4] row); faces local to the tile (vertex ids near tile * 20), neighbouring faces share vertices
// PATTERN 2: as 1, but the 7 values of a vertex go to ONE 32-byte row (x, y, 0, w, c0..c3 interleaved buffer)
// SCOPE 0: agent (what atomicAdd gives), 1: workgroup, 2: wavefront
template <int PATTERN, int SCOPE>
__global__ __launch_bounds__(256) void k(float* gv, float* gvc, int V, int iters, int lanes)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wid = blockIdx.x * 4 + wave;
    for (int it = 0; it < iters; ++it) {
        float* addr = nullptr;
        bool active = false;
        if (PATTERN == 0) {
            active = lane < lanes;
            const uint32_t row = hash32(wid * 977u + it * 131u + lane) % (uint32_t)V;
            addr = gv + (size_t)row * 4 + (lane & 3);
        } else {
            const int grp = lane >> 5, l = lane & 31;
            active = l < 21 && (grp == 0 || lanes > 32);
            const int kk = l / 7, c = l % 7;
            const uint32_t face = hash32(wid * 31u + it * 2u + grp) % 64u;           // a face among the tile's ~64
            const uint32_t vert = ((blockIdx.x * 20u) + (face + kk * 3u) % 40u) % (uint32_t)V;   // shared vertices
            if (PATTERN == 1) addr = c < 3 ? gv + (size_t)vert * 4 + (c == 2 ? 3 : c) : gvc + (size_t)vert * 4 + (c - 3);
            else addr = gv + (size_t)vert * 8 + (c < 3 ? (c == 2 ? 3 : c) : c + 1);
        }
        if (active) {
            if (SCOPE == 0) __hip_atomic_fetch_add(addr, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (SCOPE == 1) __hip_atomic_fetch_add(addr, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(addr, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        // some independent work between the atomics, as in the face loop (~100 VALU)
        float x = (float)lane;
#pragma unroll
        for (int i = 0; i < 100; ++i) x = fmaf(x, 1.0001f, 0.5f);
        if (x == 12345.678f) gv[0] = x;
    }
}

template <int PATTERN, int SCOPE>
void run(const char* name, float* gv, float* gvc, int V, int iters, int lanes)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemsetAsync(gv, 0, (size_t)V * 32, 0);
        (void)hipMemsetAsync(gvc, 0, (size_t)V * 16, 0);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<PATTERN, SCOPE>), dim3(1024), dim3(256), 0, 0, gv, gvc, V, iters, lanes);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int active = PATTERN == 0 ? lanes : (lanes > 32 ? 42 : 21);
    const double n = 4096.0 * iters * active;
    printf("%-34s iters %3d lanes %2d: %8.1f us  %7.1f k lane-atomics  %6.1f per ns\n", name, iters, active, ms * 1e3, n / 1e3, n / (ms * 1e6));
}

int main()
{
    const int V = 30000;
    float *gv, *gvc;
    (void)hipMalloc(&gv, (size_t)V * 32); (void)hipMalloc(&gvc, (size_t)V * 16);
    run<0, 0>("no atomics (0 lanes)", gv, gvc, V, 7, 0);
    for (int iters : {7, 14}) {
        run<0, 0>("random rows, agent", gv, gvc, V, iters, 42);
        run<0, 0>("random rows, agent", gv, gvc, V, iters, 64);
        run<0, 1>("random rows, workgroup", gv, gvc, V, iters, 42);
        run<1, 0>("faces (two buffers), agent", gv, gvc, V, iters, 64);
        run<1, 0>("faces (two buffers), agent", gv, gvc, V, iters, 32);
        run<1, 1>("faces (two buffers), workgroup", gv, gvc, V, iters, 64);
        run<1, 2>("faces (two buffers), wavefront", gv, gvc, V, iters, 64);
        run<2, 0>("faces (one 32-byte row), agent", gv, gvc, V, iters, 64);
        run<2, 1>("faces (one 32-byte row), workgroup", gv, gvc, V, iters, 64);
    }
    return 0;
}
