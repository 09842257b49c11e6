// dirt_grad.hip -- gradient assembly kernel for gfx950.
//
// Replaces assemble_grads / launch_grad_assembly (csrc/rasterise_grad_egl.cu:93-278) and, by
// evaluating every channel group of dirt/rasterise_ops.py:145-165 inside one launch, the N
// per-group RasteriseGrad ops (and N GL re-draws) the reference issues for C not in {1,3}.
//
// Inputs: the visibility buffer (front-most face per pixel, written by raster_kernel<1>) instead of
// the reference's two RGBA32F surfaces; barycentrics and clip-w of a pixel are recomputed from the
// face's set-up record exactly as the forward pass computes them.
//
// One thread per pixel; a wave is 64 consecutive pixels of one row so that grad_pixels reads and
// grad_background writes are fully coalesced.  Variable names follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "../../include/dirt_hip.h"

namespace dirt {

struct Frag {
    float b[3];
    float w;
    int32_t vid[3];
};

// (barycentric, clip_w, indices) of face `f` at pixel (x, r): what the backward fragment shader
// writes (csrc/shaders.cpp:64-77).
__device__ inline Frag frag_eval(const FaceRec* __restrict__ recs, int f, int x, int r, int H)
{
    const FaceRec* __restrict__ rec = recs + f;
    double cf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k] = rec->coef[k];
    double Fk[3];
    edge_eval(cf, (double)x + 0.5, (double)(H - 1 - r) + 0.5, Fk);
    Frag o;
    bary_eval(Fk, rec->flags, rec->inv_det, o.b, o.w);
    o.vid[0] = rec->vid[0]; o.vid[1] = rec->vid[1]; o.vid[2] = rec->vid[2];
    return o;
}

__global__ __launch_bounds__(256) void grad_kernel(GradParams p)
{
    const int x_in_frame = blockIdx.x * 64 + threadIdx.x;
    const int y_in_frame = blockIdx.y * 4 + threadIdx.y;  // tensor row (top row first)
    const int iib = blockIdx.z;
    if (x_in_frame >= p.W || y_in_frame >= p.H) return;
    const int H = p.H, W = p.W, C = p.C;
    const size_t frame = (size_t)H * W;
    const size_t pix = (size_t)iib * frame + (size_t)y_in_frame * W + x_in_frame;
    const size_t total_pix = (size_t)p.B * frame;

    const FaceRec* __restrict__ recs = p.recs + (size_t)iib * p.F;
    const int32_t* __restrict__ vis = p.vis + (size_t)iib * frame;
    const float* __restrict__ vertices = p.vertices + (size_t)iib * p.V * 4;
    const float* __restrict__ g_here = p.grad_pixels + pix * C;
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * 4;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * C;

    const int32_t face_here = vis[(size_t)y_in_frame * W + x_in_frame];
    Frag here;
    if (face_here >= 0) {
        here = frag_eval(recs, face_here, x_in_frame, y_in_frame, H);
    } else {  // clear values, csrc/rasterise_grad_egl.cpp:442-445
        here.b[0] = here.b[1] = here.b[2] = -1.f;
        here.w = INFINITY;
        here.vid[0] = here.vid[1] = here.vid[2] = -1;
    }

    // colour / background gradients, csrc/rasterise_grad_egl.cu:135-148 (group independent)
    if (face_here >= 0) {
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < C; ++c) {
                const float color_grad = g_here[c] * here.b[k];
                atomicAdd(&grad_vertex_colors[(size_t)here.vid[k] * C + c], color_grad);
            }
        for (int c = 0; c < C; ++c) p.grad_background[pix * C + c] = 0.f;
    } else {
        for (int c = 0; c < C; ++c) p.grad_background[pix * C + c] = g_here[c];
    }

    const bool interior = x_in_frame > 0 && y_in_frame > 0 && x_in_frame < W - 1 && y_in_frame < H - 1;
    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;

    for (int c_begin = 0; c_begin < C;) {
        const int G = (c_begin + 3 <= C) ? 3 : 1;  // dirt/rasterise_ops.py:148-152
        const bool alias = (G == 1) && !q1_intended;  // quirk Q1: "channels" 1,2 of a 1-channel tensor

        // 3x3 neighbourhood of `pixels`, edge clamped: at(), csrc/rasterise_grad_egl.cu:113-124
        float sx[3], sy[3];
        {
            float t[3][3][3];
#pragma unroll
            for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
                for (int ox = -1; ox <= 1; ++ox) {
                    const int cx = max(0, min(W - 1, x_in_frame + ox));
                    const int cy = max(0, min(H - 1, y_in_frame - oy));
                    const size_t n = (size_t)iib * frame + (size_t)cy * W + cx;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float v = 0.f;
                        if (G == 3) {
                            v = p.pixels[n * C + c_begin + ch];
                        } else if (ch == 0) {
                            v = p.pixels[n * C + c_begin];
                        } else if (alias) {
                            size_t m = n + ch;
                            if (m > total_pix - 1) m = total_pix - 1;
                            v = p.pixels[m * C + c_begin];
                        }
                        t[oy + 1][ox + 1][ch] = v;
                    }
                }
#define AT(ox, oy, ch) t[(oy) + 1][(ox) + 1][ch]
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {  // :126-127
                float d1 = ((AT(-1, -1, ch) + AT(-1, +1, ch)) - AT(+1, -1, ch)) - AT(+1, +1, ch);
                float d2 = AT(-1, 0, ch) - AT(+1, 0, ch);
                float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
                sx[ch] = m1 + m2;
                d1 = ((AT(-1, -1, ch) + AT(+1, -1, ch)) - AT(-1, +1, ch)) - AT(+1, +1, ch);
                d2 = AT(0, -1, ch) - AT(0, +1, ch);
                m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
                sy[ch] = m1 + m2;
            }
#undef AT
        }

        Frag cur = here;
        bool dilated = false;
        if (interior) {  // :155-194
            float l1x, l1y;
            if (G == 1 && q1_intended) {
                l1x = fabsf(sx[0]); l1y = fabsf(sy[0]);
            } else {
                l1x = (fabsf(sx[0]) + fabsf(sx[1])) + fabsf(sx[2]);
                l1y = (fabsf(sy[0]) + fabsf(sy[1])) + fabsf(sy[2]);
            }
            int off_x = l1x > l1y ? 1 : 0, off_y = l1x > l1y ? 0 : 1;
            if (((x_in_frame + y_in_frame) & 1) == 1) { off_x = -off_x; off_y = -off_y; }
            for (int attempt = 0; attempt < 2 && !dilated; ++attempt) {
                const int ox = attempt == 0 ? off_x : -off_x, oy = attempt == 0 ? off_y : -off_y;
                // the reference offsets in GL buffer orientation (y up): tensor row = y_in_frame - oy
                const int nx = x_in_frame + ox, nr = y_in_frame - oy;
                const int32_t face_off = vis[(size_t)nr * W + nx];
                if (face_off >= 0) {
                    const Frag off = frag_eval(recs, face_off, nx, nr, H);
                    const bool differs =
                        off.vid[0] != cur.vid[0] || off.vid[1] != cur.vid[1] || off.vid[2] != cur.vid[2];
                    if (differs && cur.w > off.w) {  // :165
                        cur = off;
                        dilated = true;
                    }
                }
            }
        }

        if (p.debug_thingy && c_begin == 0) {  // :150-151,172
            float* dbg = p.debug_thingy + pix * 3;
            dbg[0] = dilated ? 1.e-2f : 0.f;
            for (int ch = 1; ch <= 2; ++ch) {
                size_t m = pix * G + ch;
                if (m > total_pix * G - 1) m = total_pix * G - 1;
                // element m of the contiguous [B,H,W,G] slice of grad_pixels
                dbg[ch] = p.grad_pixels[(m / G) * C + c_begin + (m % G)];
            }
        }

        if (cur.b[0] != -1.f) {  // :196-232
            const float width_f = (float)W, height_f = (float)H;
            float dL_dx = 0.f, dL_dy = 0.f;
            for (int channel = 0; channel < G; ++channel) {
                const float dL_dchannel = g_here[c_begin + channel];
                float m = dL_dchannel * sx[channel];
                dL_dx = dL_dx + m;
                m = dL_dchannel * sy[channel];
                dL_dy = dL_dy + m;
            }
            float clip_x = 0.f, clip_y = 0.f;
            for (int k = 0; k < 3; ++k) {
                const float2 vxy = *reinterpret_cast<const float2*>(vertices + (size_t)cur.vid[k] * 4);
                float m = cur.b[k] * vxy.x;
                clip_x = clip_x + m;
                m = cur.b[k] * vxy.y;
                clip_y = clip_y + m;
            }
            const float clip_w = cur.w;
            const float d_xview_by_xclip = (.5f * width_f) / clip_w;
            const float d_yview_by_yclip = (.5f * height_f) / clip_w;
            const float ww = clip_w * clip_w;
            const float d_xview_by_wclip = ((-.5f * width_f) * clip_x) / ww;
            const float d_yview_by_wclip = ((-.5f * height_f) * clip_y) / ww;
            for (int k = 0; k < 3; ++k) {
                const float dLx_b = dL_dx * cur.b[k];
                const float dLy_b = dL_dy * cur.b[k];
                const float gx = dLx_b * d_xview_by_xclip;
                const float gy = dLy_b * d_yview_by_yclip;
                const float gw1 = dLx_b * d_xview_by_wclip, gw2 = dLy_b * d_yview_by_wclip;
                const float gw = gw1 + gw2;
                float* gv = grad_vertices + (size_t)cur.vid[k] * 4;
                atomicAdd(gv + 0, gx);
                atomicAdd(gv + 1, gy);
                atomicAdd(gv + 3, gw);
            }
        }
        c_begin += G;
    }
}

hipError_t launch_grad(const GradParams& p, hipStream_t stream)
{
    if (p.B == 0) return hipSuccess;
    const dim3 block(64, 4, 1);
    const dim3 grid((unsigned)((p.W + 63) / 64), (unsigned)((p.H + 3) / 4), (unsigned)p.B);
    hipLaunchKernelGGL(grad_kernel, grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace dirt
