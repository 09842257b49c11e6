// dirt_launch.h -- kernel parameter blocks and host-side launch prototypes (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dirt_device.h"

namespace dirt {

struct RasterParams {
    const FaceRec* recs;         // [B*F] set-up records (workspace)
    const FaceBox* boxes;        // [B*F] bounding boxes (workspace)
    const float* background;     // [B,H,W,C]
    const float* vertex_colors;  // [B,V,C]
    float* pixels;               // [B,H,W,C]
    int32_t* vis;                // [B,H,W] visibility export (MODE 1)
    int V, F, H, W, C;
    int tiles_x, tiles_y;
};

struct GradParams {
    const FaceRec* recs;       // [B*F]
    const int32_t* vis;        // [B,H,W] front-most face or -1
    const float* vertices;     // [B,V,4]
    const float* pixels;       // [B,H,W,C]
    const float* grad_pixels;  // [B,H,W,C]
    float* grad_background;    // [B,H,W,C]
    float* grad_vertices;      // [B,V,4]  (zeroed before launch)
    float* grad_vertex_colors; // [B,V,C]  (zeroed before launch)
    float* debug_thingy;       // [B,H,W,3] or nullptr
    int B, V, F, H, W, C;
    unsigned flags;
    int tiles_x, tiles_y;      // filled by launch_grad
    int nslots;                // LDS slot-table capacity, filled by launch_grad
};

hipError_t launch_setup(const float* vertices, const int32_t* faces, FaceRec* recs, FaceBox* boxes, int B, int V,
                        int F, int H, int W, hipStream_t stream);
hipError_t launch_raster(const RasterParams& p, int B, bool visibility_only, hipStream_t stream);
hipError_t launch_grad(const GradParams& p, hipStream_t stream);

}  // namespace dirt
