// dirt_launch.h -- kernel parameter blocks and host-side launch prototypes (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dirt_device.h"

namespace dirt {

struct TileRec;   // dirt_raster_common.h

constexpr int MAX_BINS = 256;           // bins of the start / count directory (setup_kernel<NW>: meshes of more than 16 384 faces)
#ifndef DIRT_MASKED_BINS
#define DIRT_MASKED_BINS 1024
#endif
constexpr int MAX_BINS_MASKED = DIRT_MASKED_BINS;   // bins of the masked directory (setup_kernel_v2): 32-pixel bins = raster tiles up to 1024 x 1024

// Coarse binning grid: square bins of (1 << shift) pixels, shift >= 5, bins_x * bins_y <= big (MAX_BINS or MAX_BINS_MASKED);
// `big`: the row of the directory that holds the "big" pseudo-bin (faces touching too many bins), which every tile reads.
struct BinGrid {
    int shift, bins_x, bins_y, big;
    int cell_bin_stride, cell_chunk_stride;   // directory cell (bin, chunk) = cells[bin * cell_bin_stride + chunk * cell_chunk_stride]:
                                              // bin-major for the start / count directory, CHUNK-major for the masked one (a set-up
                                              // wave then stores its chunk's row contiguously; a tile reads one cell per thread either way)
};

// One entry of a bin's face list (also of the per-scene "big" list), 16 bytes.
struct alignas(16) BinEntry {
    FaceBox box;
    int32_t face;
    uint32_t pad;
};
static_assert(sizeof(BinEntry) == 16, "BinEntry must be 16 bytes");

// One cell of the per-scene chunk x bin directory written by setup_kernel: where, inside chunk c's entry
// segment, the entries of bin b start, and how many there are.  Column MAX_BINS is the "big" pseudo-bin
// (faces touching more than 4 bins), which every tile reads.
struct BinCell {
    uint32_t start, count;
};

struct GeomParams {
    const float* vertices;  // [B,V,4]
    const int32_t* faces;   // [B,F,3], or [F,3] when shared_faces
    int shared_faces;       // one topology for every scene of the batch (DIRT_FLAG_SHARED_FACES)
    FaceRec* recs;          // [B*F]
    FaceBox* boxes;         // [B*F]
    BinCell* cells;         // [B][grid.big + 1][nchunk] bin x chunk directory
    BinEntry* entries;      // [B][nchunk][5 * chunk_faces] per-chunk entry segments, sorted by bin
    TileRec* lrecs;  // [B*F] face-local coverage records (make_local_rec), setup_kernel_v2
    float4* crecs;          // [B*F][3] the faces' vertex colours, padded to float4 (setup_kernel_v2, when vertex_colors and C in {1, 3, 4}), or nullptr
    const float* vertex_colors;  // [B,V,C] or nullptr
    int C;
    int B, V, F, H, W;
    int nchunk, chunk_faces;  // faces are processed in nchunk contiguous chunks per scene
    int masked;               // chunk_faces == 64: a directory cell is the 64-bit mask of the chunk's faces that touch the bin (filled by launch_geometry's callers: directory_is_masked)
    int v2_only;              // the launch that follows is raster_kernel_v2: the {box, face} entries and the records' depth-plane tail are not written
    BinGrid grid;
};

struct RasterParams {
    const FaceRec* recs;         // [B*F] set-up records (workspace)
    const BinCell* cells;        // [B][grid.big + 1][nchunk] bin x chunk directory
    const BinEntry* entries;     // [B][nchunk][5 * chunk_faces] per-chunk entry segments, sorted by bin
    const TileRec* lrecs; // [B*F] face-local coverage records (setup_kernel_v2)
    const float4* crecs;         // [B*F][3] vertex colours per face (setup_kernel_v2), or nullptr
    int nchunk, chunk_faces;
    int masked;                  // the directory holds face masks, entries have fixed slots (chunks of 64 faces: setup_kernel_masked)
    const float* background;     // [B,H,W,C]
    const float* vertex_colors;  // [B,V,C]
    float* pixels;               // [B,H,W,C]
    int32_t* vis;                // [B,H,W] front-most face per pixel (dirt_rasterise_visibility's output); or nullptr
    float2* state_a;             // [B,H,W] {clip_w, face} and ...
    float2* state_b;             // [B,H,W] two of the three barycentrics of the front-most fragment (encode_bary): what the backward pass reads; or nullptr (both)
    int V, F, H, W, C;
    BinGrid grid;
    unsigned flags;              // DIRT_FLAG_TILES_*
    int tiles_x, tiles_y;        // filled by launch_raster
    uint32_t tiles_x_magic;      // tile_magic(tiles_x), filled by launch_raster
    void* zero_b;                // optional buffers cleared by the same launch: the backward pass's gradient accumulators
    size_t zero_b_bytes;         //   (the cudaMemsetAsync x4 of csrc/rasterise_grad_egl.cu:244-250)
    void* zero_c;
    size_t zero_c_bytes;
    unsigned zero_b_per, zero_c_per;   // 16-byte units per workgroup, filled by launch_raster
};

struct GradParams {
    const float2* state_a;     // [B,H,W] {clip_w, face} and
    const float2* state_b;     // [B,H,W] two barycentrics (encode_bary): the backward fragment shader's output (csrc/shaders.cpp:64-77) with
                               //         b2 = 1 - b0 - b1 and the face index (bit pattern) in place of the index triple
    const int32_t* faces;      // [B,F,3], or [F,3] when shared_faces
    int shared_faces;
    const float* pixels;       // [B,H,W,C]
    const float* grad_pixels;  // [B,H,W,C]
    float* grad_background;    // [B,H,W,C]
    float* grad_vertices;      // [B,V,4]  (zeroed before launch), rows of gv_stride floats
    float* grad_vertex_colors; // [B,V,C]  (zeroed before launch), rows of gvc_stride floats
    int gv_stride, gvc_stride; // 4 and C for dense tensors; 8 and 8 for the state's interleaved accumulators
    float* debug_thingy;       // [B,H,W,3] or nullptr
    int B, V, F, H, W, C;
    unsigned flags;
    int tiles_x, tiles_y;      // filled by launch_grad
    uint32_t tiles_x_magic;    // tile_magic(tiles_x), filled by launch_grad
    int pixels_aligned16;      // the [B,H,W,C] tensors may be accessed with 16-byte loads / stores, filled by launch_grad
    int c_first, npasses;      // the launch's channel passes (see grad_kernel<CSPEC, STRIDED>), filled by launch_grad
    int last_second;           // the {3,3} shape: what the LAST pass of the launch has as its second group: 0 a triple, 1 a single, 2 nothing
    int gbk_split;             // strided launches: > 0: the launch's passes write grad_background of ALL channels, a share of the tile's
                               // rows each (whole lines); 0: another launch of the call does; < 0: every pass stores its own channels
    float inv_w, inv_h;        // 1 / W, 1 / H (pixel -> NDC: ndc_of), filled by launch_grad
};

BinGrid make_bin_grid(int H, int W, int nchunk, bool masked, int min_shift = 5);
int raster_tile_choice(int H, int W, int B, unsigned flags);   // 32 or 16: the forward / visibility kernels' tile (dirt_forward.hip)
void chunking(int F, int& nchunk, int& chunk_faces);
#ifdef DIRT_NO_MASKED_DIR   // (A/B builds: rounds 1-4's start / count directory for every mesh)
inline bool directory_is_masked(int) { return false; }
#else
inline bool directory_is_masked(int chunk_faces) { return chunk_faces == 64; }   // one face per lane of a one-wave set-up workgroup
#endif
hipError_t launch_zero(void* b, size_t b_bytes, void* c, size_t c_bytes, hipStream_t stream);
hipError_t launch_unpack(const float* acc_gv, const float* acc_gvc, int acc_stride, float* gv, float* gvc, int C, size_t rows, hipStream_t stream);
hipError_t launch_geometry(const GeomParams& g, hipStream_t stream);
hipError_t launch_raster(const RasterParams& p, int B, bool visibility_only, hipStream_t stream);
hipError_t launch_geometry_v2(const GeomParams& g, hipStream_t stream);   // dirt_forward.hip: masked directory, face-local records
bool raster_v2_applies(const RasterParams& p, int B, bool visibility_only);   // ... the two-trip raster kernel (32 x 32 tiles, 1 / 3 / 4 channels or visibility)
hipError_t launch_raster_v2(const RasterParams& p, int B, bool visibility_only, hipStream_t stream);
hipError_t launch_grad(const GradParams& p, hipStream_t stream);
hipError_t launch_grad_small(const GradParams& p, hipStream_t stream);  // dirt_grad_small.hip; p as filled by launch_grad
hipError_t launch_grad_px2(const GradParams& p, hipStream_t stream);    // dirt_grad_px2.hip (two pixels per lane, 32 x 16 tiles); p as filled by launch_grad
bool grad_stream_eligible(const GradParams& p);                         // dirt_grad_stream.hip (4 channels, whole 32 x 32 tiles: loads streamed by LDS-DMA under the compute)
hipError_t launch_grad_stream(const GradParams& p, hipStream_t stream); // ... p as filled by launch_grad

}  // namespace dirt
