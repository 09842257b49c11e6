/*
 * Host shim for compiling the reference's own gradient kernel on the CPU (TEST INFRASTRUCTURE).
 *
 * /root/reference/csrc/rasterise_grad_egl.cu includes <tensorflow/core/framework/tensor.h> and is
 * written in CUDA.  Neither TensorFlow nor CUDA exists in this image, so oracle/make_ref.py puts this
 * directory on the include path: this header supplies just enough of both vocabularies for the
 * reference's two CUDA translation units -- csrc/rasterise_grad_egl.cu (Vec3, assemble_grads,
 * launch_grad_assembly, upload_vertices, launch_vertex_upload) and csrc/rasterise_egl.cu
 * (upload_background, download_pixels and their launchers) -- to compile unmodified with g++ and to
 * run as one sequential "thread" on the host.  Nothing here restates the reference's algorithm; it only
 * models the containers and intrinsics the algorithm is written against:
 *
 *   TTypes<T,N>::Tensor / ConstTensor   Eigen::TensorMap, row-major, operator() WITHOUT bounds checks
 *                                       (so pixels(iib,y,x,1) on a 1-channel tensor aliases the next
 *                                       float exactly as Eigen's index arithmetic does: quirk Q1)
 *   tensorflow::Tensor                  shape + borrowed buffer, tensor<T,N>(), dim_size(), NumElements()
 *   surf2Dread<float4> / surf2Dwrite    read / write of a float4 texel of a row-major host array
 *                                       (x is a byte offset, as in CUDA)
 *   atomicAdd(float*, float)            plain fp32 add (one thread => the kernel's own pixel order)
 *   cudaMemsetAsync, surface objects    memset / pass-through handles
 *   blockIdx, blockDim, threadIdx, gridDim   a 1x1x1 grid of 1x1x1 blocks; the reference's
 *                                       CUDA_AXIS_KERNEL_LOOP (csrc/tf_cuda_utils.h:10-12) then walks
 *                                       the whole virtual thread range in one thread
 */
#ifndef DIRT_REF_SHIM_TENSOR_H
#define DIRT_REF_SHIM_TENSOR_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <algorithm>
#include <vector>
#include <iostream>

#define __global__
#define __device__
#define __host__

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct int2 { int x, y; };
struct float4 { float x, y, z, w; };

extern thread_local dim3 blockIdx, blockDim, threadIdx, gridDim;

using std::max;
using std::min;

typedef void *cudaStream_t;
typedef int cudaError_t;

/* A "cudaArray": a row-major float4 image on the host. */
struct RefShimArray { float4 *texels; int width, height; };
typedef RefShimArray const *cudaArray_t;
typedef RefShimArray const *cudaSurfaceObject_t;

enum cudaResourceType { cudaResourceTypeArray };
struct cudaResourceDesc {
    cudaResourceType resType;
    struct { struct { cudaArray_t array; } array; } res;
};

inline cudaError_t cudaCreateSurfaceObject(cudaSurfaceObject_t *surface, cudaResourceDesc const *descriptor) {
    *surface = descriptor->res.array.array;
    return 0;
}
inline cudaError_t cudaDestroySurfaceObject(cudaSurfaceObject_t) { return 0; }
inline cudaError_t cudaMemsetAsync(void *ptr, int value, size_t bytes, cudaStream_t) { std::memset(ptr, value, bytes); return 0; }

template <class T> inline T surf2Dread(cudaSurfaceObject_t surface, int x_bytes, int y);
template <> inline float4 surf2Dread<float4>(cudaSurfaceObject_t surface, int x_bytes, int y) {
    return surface->texels[(size_t) y * surface->width + x_bytes / 16];
}
inline void surf2Dwrite(float4 const &value, cudaSurfaceObject_t surface, int x_bytes, int y) {
    surface->texels[(size_t) y * surface->width + x_bytes / 16] = value;
}
inline char const *cudaGetErrorName(cudaError_t) { return "cudaError (host shim)"; }

inline float atomicAdd(float *address, float value) { float const old = *address; *address = old + value; return old; }

namespace Eigen {
    struct GpuDevice { cudaStream_t stream() const { return nullptr; } };
}

namespace tensorflow {

    template <class T, int N> struct RefShimMap {
        T *ptr;
        long dims[N];
        long dimension(int i) const { return dims[i]; }
        T *data() const { return ptr; }
        template <class... Ix> T &operator ()(Ix... ix) const {
            static_assert(sizeof...(Ix) == N, "index count");
            long const idx[N] = {static_cast<long>(ix)...};
            long flat = 0;
            for (int d = 0; d < N; ++d) flat = flat * dims[d] + idx[d];
            return ptr[flat];  // no bounds check, as Eigen in release builds
        }
        operator RefShimMap<T const, N>() const {
            RefShimMap<T const, N> out;
            out.ptr = ptr;
            for (int d = 0; d < N; ++d) out.dims[d] = dims[d];
            return out;
        }
    };

    template <class T, int N> struct TTypes {
        typedef RefShimMap<T, N> Tensor;
        typedef RefShimMap<T const, N> ConstTensor;
    };

    class Tensor {
    public:
        Tensor(void *buffer, std::vector<long> const &shape) : buffer_(buffer), shape_(shape) {}
        long dim_size(int i) const { return shape_[i]; }
        long NumElements() const { long n = 1; for (long d : shape_) n *= d; return n; }
        template <class T, int N> typename TTypes<T, N>::Tensor tensor() const {
            if ((int) shape_.size() != N) { std::fprintf(stderr, "ref shim: rank mismatch\n"); std::abort(); }
            typename TTypes<T, N>::Tensor out;
            out.ptr = static_cast<T *>(buffer_);
            for (int d = 0; d < N; ++d) out.dims[d] = shape_[d];
            return out;
        }
    private:
        void *buffer_;
        std::vector<long> shape_;
    };

}

struct RefShimFatal {
    ~RefShimFatal() { std::cerr << std::endl; std::abort(); }
    template <class T> RefShimFatal &operator <<(T const &value) { std::cerr << value; return *this; }
};
#define FATAL 0
#define LOG(severity) RefShimFatal()

/* kernel<<<grid, block, shared, stream>>>(args...) is rewritten by oracle/make_ref.py (the only
   edit made to the reference text, g++ cannot parse the chevrons) into REF_SHIM_LAUNCH(kernel, ...)(args...). */
template <class Kernel> struct RefShimLaunch {
    Kernel kernel;
    template <class... Args> void operator ()(Args &&... args) const {
        blockIdx = dim3(0, 0, 0); threadIdx = dim3(0, 0, 0); blockDim = dim3(1, 1, 1); gridDim = dim3(1, 1, 1);
        kernel(std::forward<Args>(args)...);
    }
};
template <class Kernel, class... Config> inline RefShimLaunch<Kernel> ref_shim_launch(Kernel kernel, Config const &...) { return RefShimLaunch<Kernel>{kernel}; }
#define REF_SHIM_LAUNCH(kernel, ...) ref_shim_launch(kernel, __VA_ARGS__)

#endif
