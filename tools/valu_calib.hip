// Calibration for rocprofv3 SQ counters: a kernel that keeps every SIMD's VALU busy (8 waves/SIMD of
// independent v_fma_f32 chains) and one that is purely f64.  Compare SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES of
// real kernels with these to read VALU utilisation.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ __launch_bounds__(512) void spin(T* out, int iters)
{
    T a = (T)threadIdx.x, b = (T)1.0001, c = (T)0.5, d = a + 1, e = a + 2, f = a + 3;
    for (int i = 0; i < iters; ++i) {
        a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
        a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
    }
    if (a + d + e + f == (T)12345) out[0] = a;
}
int main()
{
    float* o; hipMalloc(&o, 64);
    double* od; hipMalloc(&od, 64);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(spin<float>, dim3(256 * 4), dim3(512), 0, 0, o, 20000);
        hipLaunchKernelGGL(spin<double>, dim3(256 * 4), dim3(512), 0, 0, od, 20000);
    }
    hipDeviceSynchronize();
    return 0;
}
