// dirt_texture.hip -- texture look-up of a deferred shader as one kernel (and its gradient).
//
// Replaces the TensorFlow composition of the reference's samples/textured.py:16-61 -- `uvs_to_pixel_indices` (flip to
// (row, column), repeat / clamp, scale by the texture size) followed by `sample_texture` (floor, fraction, a 4 x gather_nd,
// bilinear blend; or nearest) -- which materialises the index tensor, the four gathered neighbour tensors and five
// products per pixel.  Here one thread reads a (u, v) pair straight out of the G-buffer (any element stride), gathers the
// four texels and writes the blended colour; the backward kernel scatters dL/dtexture with float atomics and writes
// dL/duv.  The arithmetic is the reference's, operation for operation in float32 (so the result equals the composed
// torch / TF expression bit for bit); where the reference's gather would read row Ht or column Wt -- an index inside the
// last texel -- the last texel is used.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dirt_hip.h"

namespace dirt {

struct TexParams {
    const float* texture;   // [Ht, Wt, Ct]
    const float* uvs;       // n pairs (u, v), `uv_stride` floats apart; (0, 0) is the TOP-LEFT of the image
    long long n;
    int Ht, Wt, Ct, uv_stride, guv_stride;
    unsigned flags;
    float* out;             // [n, Ct]
    const float* grad_out;  // [n, Ct]
    float* grad_texture;    // [Ht, Wt, Ct], accumulated into (cleared by the caller)
    float* grad_uvs;        // n pairs, `guv_stride` floats apart; or nullptr
};

// samples/textured.py:16-26: (u, v) -> fractional (row, column) index
__device__ __forceinline__ void uv_to_index(float u, float v, int Ht, int Wt, bool clamp_mode, float& row, float& col, float& drow_dv,
                                            float& dcol_du)
{
    if (clamp_mode) {
        row = fminf(fmaxf(v, 0.f), 1.f) * (float)Ht;
        col = fminf(fmaxf(u, 0.f), 1.f) * (float)Wt;
        drow_dv = (v >= 0.f && v <= 1.f) ? (float)Ht : 0.f;   // the gradient of clip_by_value
        dcol_du = (u >= 0.f && u <= 1.f) ? (float)Wt : 0.f;
    } else {
        row = (v - floorf(v)) * (float)Ht;                    // uvs % 1. (floor-mod)
        col = (u - floorf(u)) * (float)Wt;
        drow_dv = (float)Ht; dcol_du = (float)Wt;
    }
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void texture_kernel(TexParams p)
{
    const bool clamp_mode = (p.flags & DIRT_TEX_CLAMP) != 0, nearest = (p.flags & DIRT_TEX_NEAREST) != 0;
    const int Ct = p.Ct;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x) {
        const float u = p.uvs[i * p.uv_stride], v = p.uvs[i * p.uv_stride + 1];
        float row, col, drow_dv, dcol_du;
        uv_to_index(u, v, p.Ht, p.Wt, clamp_mode, row, col, drow_dv, dcol_du);
        if (nearest) {   // samples/textured.py:31-33: the indices truncated
            const int r = min(max((int)row, 0), p.Ht - 1), c = min(max((int)col, 0), p.Wt - 1);
            const size_t t = ((size_t)r * p.Wt + c) * Ct;
            if (!BACKWARD) {
                for (int ch = 0; ch < Ct; ++ch) p.out[i * Ct + ch] = p.texture[t + ch];
            } else {
                for (int ch = 0; ch < Ct; ++ch) atomicAdd(&p.grad_texture[t + ch], p.grad_out[i * Ct + ch]);
                if (p.grad_uvs) { p.grad_uvs[i * p.guv_stride] = 0.f; p.grad_uvs[i * p.guv_stride + 1] = 0.f; }
            }
            continue;
        }
        const float fr0 = floorf(row), fc0 = floorf(col);
        const float fr = row - fr0, fc = col - fc0;   // frac_indices[..., :1] (row), [..., 1:] (column)
        const int r0 = min(max((int)fr0, 0), p.Ht - 1), r1 = min(r0 + 1, p.Ht - 1);
        const int c0 = min(max((int)fc0, 0), p.Wt - 1), c1 = min(c0 + 1, p.Wt - 1);
        const size_t tl = ((size_t)r0 * p.Wt + c0) * Ct, tr = ((size_t)r0 * p.Wt + c1) * Ct;
        const size_t bl = ((size_t)r1 * p.Wt + c0) * Ct, br = ((size_t)r1 * p.Wt + c1) * Ct;
        const float wc0 = 1.f - fc, wr0 = 1.f - fr;
        if (!BACKWARD) {
            for (int ch = 0; ch < Ct; ++ch) {
                // top_left * (1 - fc) * (1 - fr) + top_right * fc * (1 - fr) + bottom_left * (1 - fc) * fr + bottom_right * fc * fr
                const float a = (p.texture[tl + ch] * wc0) * wr0, b = (p.texture[tr + ch] * fc) * wr0;
                const float c = (p.texture[bl + ch] * wc0) * fr, d = (p.texture[br + ch] * fc) * fr;
                p.out[i * Ct + ch] = ((a + b) + c) + d;
            }
        } else {
            float d_fr = 0.f, d_fc = 0.f;
            for (int ch = 0; ch < Ct; ++ch) {
                const float g = p.grad_out[i * Ct + ch];
                const float t_tl = p.texture[tl + ch], t_tr = p.texture[tr + ch], t_bl = p.texture[bl + ch], t_br = p.texture[br + ch];
                atomicAdd(&p.grad_texture[tl + ch], g * (wc0 * wr0));
                atomicAdd(&p.grad_texture[tr + ch], g * (fc * wr0));
                atomicAdd(&p.grad_texture[bl + ch], g * (wc0 * fr));
                atomicAdd(&p.grad_texture[br + ch], g * (fc * fr));
                d_fr += g * ((t_bl - t_tl) * wc0 + (t_br - t_tr) * fc);
                d_fc += g * ((t_tr - t_tl) * wr0 + (t_br - t_bl) * fr);
            }
            if (p.grad_uvs) {   // floor() has zero gradient: d frac / d index = 1
                p.grad_uvs[i * p.guv_stride] = d_fc * dcol_du;
                p.grad_uvs[i * p.guv_stride + 1] = d_fr * drow_dv;
            }
        }
    }
}

hipError_t launch_texture(const TexParams& p, bool backward, hipStream_t stream)
{
    if (p.n == 0) return hipSuccess;
    long long blocks = (p.n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (backward) hipLaunchKernelGGL(texture_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(texture_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace dirt

extern "C" {

namespace {
thread_local char g_tex_error[256] = "";
}
const char* dirt_texture_last_error(void) { return g_tex_error; }

static int tex_check(const char* who, const void* texture, const void* uvs, long long n, int Ht, int Wt, int Ct, int uv_stride)
{
    if (n < 0 || Ht <= 0 || Wt <= 0 || Ct <= 0 || uv_stride < 2) {
        snprintf(g_tex_error, sizeof(g_tex_error), "%s: bad sizes (n=%lld Ht=%d Wt=%d Ct=%d uv_stride=%d)", who, n, Ht, Wt, Ct, uv_stride);
        return DIRT_E_INVALID_ARGUMENT;
    }
    if (n > 0 && (!texture || !uvs)) {
        snprintf(g_tex_error, sizeof(g_tex_error), "%s: texture / uvs is NULL", who);
        return DIRT_E_INVALID_ARGUMENT;
    }
    return DIRT_OK;
}

int dirt_texture_sample_forward(const float* texture, const float* uvs, float* out, long long n, int Ht, int Wt, int Ct, int uv_stride,
                                unsigned flags, void* stream)
{
    int rc = tex_check("dirt_texture_sample_forward", texture, uvs, n, Ht, Wt, Ct, uv_stride);
    if (rc) return rc;
    if (n > 0 && !out) { snprintf(g_tex_error, sizeof(g_tex_error), "dirt_texture_sample_forward: out is NULL"); return DIRT_E_INVALID_ARGUMENT; }
    dirt::TexParams p{};
    p.texture = texture; p.uvs = uvs; p.n = n; p.Ht = Ht; p.Wt = Wt; p.Ct = Ct; p.uv_stride = uv_stride; p.flags = flags; p.out = out;
    const hipError_t e = dirt::launch_texture(p, false, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) { snprintf(g_tex_error, sizeof(g_tex_error), "dirt_texture_sample_forward: %s", hipGetErrorString(e)); return DIRT_E_HIP; }
    g_tex_error[0] = 0;
    return DIRT_OK;
}

int dirt_texture_sample_backward(const float* texture, const float* uvs, const float* grad_out, float* grad_texture, float* grad_uvs,
                                 long long n, int Ht, int Wt, int Ct, int uv_stride, int grad_uv_stride, unsigned flags, void* stream)
{
    int rc = tex_check("dirt_texture_sample_backward", texture, uvs, n, Ht, Wt, Ct, uv_stride);
    if (rc) return rc;
    if (n > 0 && (!grad_out || !grad_texture)) {
        snprintf(g_tex_error, sizeof(g_tex_error), "dirt_texture_sample_backward: grad_out / grad_texture is NULL");
        return DIRT_E_INVALID_ARGUMENT;
    }
    if (grad_uvs && grad_uv_stride < 2) {
        snprintf(g_tex_error, sizeof(g_tex_error), "dirt_texture_sample_backward: grad_uv_stride < 2");
        return DIRT_E_INVALID_ARGUMENT;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(grad_texture, 0, sizeof(float) * (size_t)Ht * Wt * Ct, s);
    if (e == hipSuccess) {
        dirt::TexParams p{};
        p.texture = texture; p.uvs = uvs; p.n = n; p.Ht = Ht; p.Wt = Wt; p.Ct = Ct; p.uv_stride = uv_stride; p.guv_stride = grad_uv_stride;
        p.flags = flags; p.grad_out = grad_out; p.grad_texture = grad_texture; p.grad_uvs = grad_uvs;
        e = dirt::launch_texture(p, true, s);
    }
    if (e != hipSuccess) { snprintf(g_tex_error, sizeof(g_tex_error), "dirt_texture_sample_backward: %s", hipGetErrorString(e)); return DIRT_E_HIP; }
    g_tex_error[0] = 0;
    return DIRT_OK;
}

}  // extern "C"
