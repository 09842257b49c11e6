/*
 * C entry points around the reference's own launch_grad_assembly / launch_vertex_upload /
 * launch_background_upload / launch_pixels_download
 * (TEST INFRASTRUCTURE; built only into oracle/_ref/, see oracle/make_ref.py).
 *
 * The reference translation units (/root/reference/csrc/rasterise_grad_egl.cu and csrc/rasterise_egl.cu,
 * compiled for the host through oracle/ref_shim/) supply every line of arithmetic and indexing; this file only wraps caller-owned
 * numpy buffers into the shim's tensorflow::Tensor and calls it the way
 * RasteriseGradOpGpu::Compute does (csrc/rasterise_grad_egl.cpp:380-391,464-472).
 */
#include <tensorflow/core/framework/tensor.h>
#include "rasterise_grad_common.h"

thread_local dim3 blockIdx, blockDim, threadIdx, gridDim;

// csrc/rasterise_egl.cu has no header of its own: its two launchers are declared where they are used
// (csrc/rasterise_egl.cpp:24-25)
void launch_background_upload(cudaArray_t &dest_array, tensorflow::Tensor const &src_tensor, int const dest_height, int const dest_width,
                              Eigen::GpuDevice const &device);
void launch_pixels_download(tensorflow::Tensor &dest_tensor, cudaArray_t const &src_array, int const src_height, int const src_width,
                            Eigen::GpuDevice const &device);

extern "C" {

/*
 * One RasteriseGrad op call (C in {1,3}, csrc/hwc.h:27) given the two rendered surfaces.
 *   surfaces   [buffer_height, buffer_width] float4 texels in GL orientation (row 0 = bottom), scenes
 *              tiled as csrc/rasterise_grad_egl.cpp:408-428 lays them out: scene i at
 *              (i % frames_per_row * W, i / frames_per_row * H)
 *   pixels, grad_pixels   [B,H,W,C]; for C == 1 the caller pads them by two floats (quirk Q1 reads
 *              channels 1 and 2 of a 1-channel tensor, csrc/rasterise_grad_egl.cu:119-123,150-151)
 * Outputs are zeroed by the reference itself (csrc/rasterise_grad_egl.cu:244-250).
 */
int dirt_ref_rasterise_grad(const float *vertices, const float *pixels, const float *grad_pixels,
                            const float *barycentrics_and_depth, const float *indices,
                            float *grad_background, float *grad_vertices, float *grad_vertex_colors, float *debug_thingy,
                            int B, int V, int H, int W, int C, int buffer_width, int buffer_height)
{
    using tensorflow::Tensor;
    if (B < 1 || V < 1 || H < 1 || W < 1 || (C != 1 && C != 3)) return -1;
    if (buffer_width % W || buffer_height % H || (buffer_width / W) * (buffer_height / H) < B) return -2;
    Tensor grad_vertices_tensor(grad_vertices, {B, V, 4});
    Tensor grad_vertex_colors_tensor(grad_vertex_colors, {B, V, C});
    Tensor grad_background_tensor(grad_background, {B, H, W, C});
    Tensor debug_thingy_tensor(debug_thingy, {B, H, W, 3});
    Tensor pixels_tensor(const_cast<float *>(pixels), {B, H, W, C});
    Tensor grad_pixels_tensor(const_cast<float *>(grad_pixels), {B, H, W, C});
    Tensor vertices_tensor(const_cast<float *>(vertices), {B, V, 4});
    RefShimArray const barycentrics_and_depth_array{reinterpret_cast<float4 *>(const_cast<float *>(barycentrics_and_depth)), buffer_width, buffer_height};
    RefShimArray const indices_array{reinterpret_cast<float4 *>(const_cast<float *>(indices)), buffer_width, buffer_height};
    Eigen::GpuDevice device;
    launch_grad_assembly(grad_vertices_tensor, grad_vertex_colors_tensor, grad_background_tensor, debug_thingy_tensor,
                         &barycentrics_and_depth_array, &indices_array,
                         pixels_tensor, grad_pixels_tensor, vertices_tensor, buffer_width, buffer_height, device);
    return 0;
}

/*
 * The forward op's two data movers (csrc/rasterise_egl.cpp:348-356,386-392): background [B,H,W,C] -> the RGBA32F
 * framebuffer atlas [buffer_height, buffer_width] float4 (GL orientation, scenes tiled), and back into pixels [B,H,W,C];
 * C in {1, 3}.  What happens in between -- the GL draw -- is not the reference's code (oracle.draw_gl stands in for it).
 */
int dirt_ref_upload_background(const float *background, float *atlas, int B, int H, int W, int C, int buffer_width, int buffer_height)
{
    if (B < 1 || H < 1 || W < 1 || (C != 1 && C != 3) || buffer_width % W || buffer_height % H) return -1;
    tensorflow::Tensor src(const_cast<float *>(background), {B, H, W, C});
    RefShimArray const array{reinterpret_cast<float4 *>(atlas), buffer_width, buffer_height};
    cudaArray_t handle = &array;
    Eigen::GpuDevice device;
    launch_background_upload(handle, src, buffer_height, buffer_width, device);
    return 0;
}

int dirt_ref_download_pixels(const float *atlas, float *pixels, int B, int H, int W, int C, int buffer_width, int buffer_height)
{
    if (B < 1 || H < 1 || W < 1 || (C != 1 && C != 3) || buffer_width % W || buffer_height % H) return -1;
    tensorflow::Tensor dest(pixels, {B, H, W, C});
    RefShimArray const array{reinterpret_cast<float4 *>(const_cast<float *>(atlas)), buffer_width, buffer_height};
    cudaArray_t const handle = &array;
    Eigen::GpuDevice device;
    launch_pixels_download(dest, handle, buffer_height, buffer_width, device);
    return 0;
}

/* The expanded vertex buffer the backward render draws from: `expanded` is [B, 3F] Vertex records
   of 9 words (position[4], barycentric[2], indices[3]), csrc/rasterise_grad_common.h:4-10. */
int dirt_ref_upload_vertices(const float *vertices, const int *faces, void *expanded, int B, int V, int F)
{
    using tensorflow::Tensor;
    if (B < 1 || V < 1 || F < 0) return -1;
    static_assert(sizeof(Vertex) == 36, "Vertex layout");
    Tensor vertices_tensor(const_cast<float *>(vertices), {B, V, 4});
    Tensor faces_tensor(const_cast<int *>(faces), {B, F, 3});
    tensorflow::TTypes<Vertex, 2>::Tensor buffer;
    buffer.ptr = static_cast<Vertex *>(expanded);
    buffer.dims[0] = B;
    buffer.dims[1] = 3L * F;
    Eigen::GpuDevice device;
    launch_vertex_upload(buffer, vertices_tensor, faces_tensor, device);
    return 0;
}

}
