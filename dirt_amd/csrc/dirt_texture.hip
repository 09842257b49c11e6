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

struct Tex3 { float x, y, z; };   // a 3-channel texel / pixel: one 12-byte access

// One texel / output pixel of CT channels as a register array, with the widest access its size and alignment allow
// (CT = 4: 16 bytes; 3: 12; 1: 4; 0: any count `ct`, channel by channel).
template <int CT>
__device__ __forceinline__ void load_ch(const float* __restrict__ p, int ct, float (&v)[CT ? CT : 1], int ch0 = 0)
{
    if constexpr (CT == 4) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else if constexpr (CT == 3) { const Tex3 q = *reinterpret_cast<const Tex3*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; }
    else v[0] = p[ch0];
}
template <int CT>
__device__ __forceinline__ void store_ch(float* __restrict__ p, const float (&v)[CT ? CT : 1], int ch0 = 0)
{
    if constexpr (CT == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else if constexpr (CT == 3) *reinterpret_cast<Tex3*>(p) = Tex3{v[0], v[1], v[2]};
    else p[ch0] = v[0];
}

// The four texels of a bilinear look-up (samples/textured.py:36-60) and their weights
struct Taps {
    int r0, r1, c0, c1;
    float fr, fc, wr0, wc0;
};
__device__ __forceinline__ Taps bilinear_taps(float row, float col, int Ht, int Wt)
{
    Taps t;
    const float fr0 = floorf(row), fc0 = floorf(col);
    t.fr = row - fr0; t.fc = col - fc0;   // frac_indices[..., :1] (row), [..., 1:] (column)
    t.r0 = min(max((int)fr0, 0), Ht - 1); t.r1 = min(t.r0 + 1, Ht - 1);
    t.c0 = min(max((int)fc0, 0), Wt - 1); t.c1 = min(t.c0 + 1, Wt - 1);
    t.wc0 = 1.f - t.fc; t.wr0 = 1.f - t.fr;
    return t;
}

// ---- forward: one thread per pixel, CT channels per access ----
template <int CT>
__global__ __launch_bounds__(256) void texture_forward_kernel(TexParams p)
{
    const bool clamp_mode = (p.flags & DIRT_TEX_CLAMP) != 0, nearest = (p.flags & DIRT_TEX_NEAREST) != 0;
    const int Ct = CT ? CT : p.Ct;
    constexpr int NV = CT ? CT : 1;
    const bool uv_pairs = (p.uv_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(p.uvs) & 7u) == 0;   // (u, v) as one 8-byte load
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x) {
        float u, v;
        if (uv_pairs) { const float2 q = *reinterpret_cast<const float2*>(p.uvs + i * p.uv_stride); u = q.x; v = q.y; }
        else { u = p.uvs[i * p.uv_stride]; v = p.uvs[i * p.uv_stride + 1]; }
        float row, col, drow_dv, dcol_du;
        uv_to_index(u, v, p.Ht, p.Wt, clamp_mode, row, col, drow_dv, dcol_du);
        float* __restrict__ out = p.out + i * Ct;
        if (nearest) {   // samples/textured.py:31-33: the indices truncated
            const int r = min(max((int)row, 0), p.Ht - 1), c = min(max((int)col, 0), p.Wt - 1);
            const float* __restrict__ t = p.texture + ((size_t)r * p.Wt + c) * Ct;
            for (int ch = 0; ch < (CT ? 1 : Ct); ++ch) { float q[NV]; load_ch<CT>(t, Ct, q, ch); store_ch<CT>(out, q, ch); }
            continue;
        }
        const Taps k = bilinear_taps(row, col, p.Ht, p.Wt);
        const float* __restrict__ tl = p.texture + ((size_t)k.r0 * p.Wt + k.c0) * Ct, * __restrict__ tr = p.texture + ((size_t)k.r0 * p.Wt + k.c1) * Ct;
        const float* __restrict__ bl = p.texture + ((size_t)k.r1 * p.Wt + k.c0) * Ct, * __restrict__ br = p.texture + ((size_t)k.r1 * p.Wt + k.c1) * Ct;
        for (int ch = 0; ch < (CT ? 1 : Ct); ++ch) {
            float a[NV], b[NV], c[NV], d[NV], o[NV];
            load_ch<CT>(tl, Ct, a, ch); load_ch<CT>(tr, Ct, b, ch); load_ch<CT>(bl, Ct, c, ch); load_ch<CT>(br, Ct, d, ch);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                // top_left * (1 - fc) * (1 - fr) + top_right * fc * (1 - fr) + bottom_left * (1 - fc) * fr + bottom_right * fc * fr
                const float ta = (a[j] * k.wc0) * k.wr0, tb = (b[j] * k.fc) * k.wr0, tc = (c[j] * k.wc0) * k.fr, td = (d[j] * k.fc) * k.fr;
                o[j] = ((ta + tb) + tc) + td;
            }
            store_ch<CT>(out, o, ch);
        }
    }
}

// ---- backward: a workgroup takes a TW x TH tile of the pixel grid (16 x 16 of an image `cols` wide; 256 x 1 of a flat list).
// The texels a tile's look-ups touch are a compact patch of the texture wherever (u, v) is smooth (a G-buffer: a rendered
// surface): the four products of every pixel are summed in an LDS copy of that patch (ds_add_f32) and each texel of the
// patch goes to memory ONCE, as one float atomic per channel, consecutive lanes on consecutive floats.  The reference's
// gather_nd gradient -- and rounds 2-5 here -- scatter 4 Ct atomics per pixel straight at the texture: at 16 pixels per texel
// that is 64 same-address atomics per texel and channel, serialised by the memory system (2.8 ms for a 2048 x 2048 frame);
// tiles whose patch does not fit (a (u, v) seam, `repeat` wrapping inside the tile) still do.
constexpr int TEX_PATCH = 1600;   // texels of a tile's patch held in LDS (x Ct floats: 19 KB at 3 channels)

template <int CT>
__global__ __launch_bounds__(256) void texture_backward_kernel(TexParams p, int rows, int cols, int tw, int th, int tiles_x)
{
    constexpr int NV = CT ? CT : 1;
    constexpr int LCT = CT ? CT : 4;                 // channels per LDS patch pass (any count: passes of 4)
    __shared__ float s_acc[TEX_PATCH * LCT];
    __shared__ int s_box[4];                         // rmin, rmax, cmin, cmax of the tile's taps
    const bool clamp_mode = (p.flags & DIRT_TEX_CLAMP) != 0, nearest = (p.flags & DIRT_TEX_NEAREST) != 0;
    const int Ct = CT ? CT : p.Ct;
    const int tid = threadIdx.x;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int px = tile_x * tw + tid % tw, py = tile_y * th + tid / tw;
    const bool active = px < cols && py < rows;
    const long long i = active ? (long long)py * cols + px : 0;
    if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = -1; s_box[2] = 0x7fffffff; s_box[3] = -1; }
    float u = 0.f, v = 0.f;
    if (active) { u = p.uvs[i * p.uv_stride]; v = p.uvs[i * p.uv_stride + 1]; }
    float row, col, drow_dv, dcol_du;
    uv_to_index(u, v, p.Ht, p.Wt, clamp_mode, row, col, drow_dv, dcol_du);
    Taps k = bilinear_taps(row, col, p.Ht, p.Wt);
    if (nearest) {   // one tap, weight 1 (samples/textured.py:31-33: the indices truncated)
        k.r0 = k.r1 = min(max((int)row, 0), p.Ht - 1); k.c0 = k.c1 = min(max((int)col, 0), p.Wt - 1);
        k.fr = 0.f; k.fc = 0.f; k.wr0 = 1.f; k.wc0 = 1.f;
    }
    __syncthreads();
    if (active) {
        atomicMin(&s_box[0], k.r0); atomicMax(&s_box[1], k.r1);
        atomicMin(&s_box[2], k.c0); atomicMax(&s_box[3], k.c1);
    }
    __syncthreads();
    const int rmin = s_box[0], cmin = s_box[2];
    const int bh = s_box[1] - rmin + 1, bw = s_box[3] - cmin + 1;
    const bool patch = bh > 0 && bw > 0 && (long long)bh * bw <= TEX_PATCH;   // (workgroup-uniform)
    const float w_tl = k.wc0 * k.wr0, w_tr = k.fc * k.wr0, w_bl = k.wc0 * k.fr, w_br = k.fc * k.fr;
    float d_fr = 0.f, d_fc = 0.f;
    const float* __restrict__ gout = p.grad_out + i * Ct;
    for (int c0 = 0; c0 < Ct; c0 += LCT) {          // (CT = 1, 3, 4: one pass; any other count: passes of four channels)
        const int nc = CT ? CT : min(LCT, Ct - c0);
        if (patch) {
            for (int e = tid; e < bh * bw * LCT; e += 256) s_acc[e] = 0.f;
            __syncthreads();
        }
        if (active) {
            float g[LCT];
            if constexpr (CT != 0) { float q[NV]; load_ch<CT>(gout, Ct, q); for (int j = 0; j < NV; ++j) g[j] = q[j]; }
            else { for (int j = 0; j < LCT; ++j) g[j] = j < nc ? gout[c0 + j] : 0.f; }
            const size_t o_tl = ((size_t)k.r0 * p.Wt + k.c0) * Ct + c0, o_tr = ((size_t)k.r0 * p.Wt + k.c1) * Ct + c0;
            const size_t o_bl = ((size_t)k.r1 * p.Wt + k.c0) * Ct + c0, o_br = ((size_t)k.r1 * p.Wt + k.c1) * Ct + c0;
            const int l_tl = ((k.r0 - rmin) * bw + (k.c0 - cmin)) * LCT, l_tr = ((k.r0 - rmin) * bw + (k.c1 - cmin)) * LCT;
            const int l_bl = ((k.r1 - rmin) * bw + (k.c0 - cmin)) * LCT, l_br = ((k.r1 - rmin) * bw + (k.c1 - cmin)) * LCT;
#pragma unroll
            for (int j = 0; j < LCT; ++j) {
                if (j >= nc) break;
                if (!nearest) {
                    const float t_tl = p.texture[o_tl + j], t_tr = p.texture[o_tr + j], t_bl = p.texture[o_bl + j], t_br = p.texture[o_br + j];
                    d_fr += g[j] * ((t_bl - t_tl) * k.wc0 + (t_br - t_tr) * k.fc);
                    d_fc += g[j] * ((t_tr - t_tl) * k.wr0 + (t_br - t_bl) * k.fr);
                }
                if (patch) {
                    if (nearest) { atomicAdd(&s_acc[l_tl + j], g[j]); continue; }
                    atomicAdd(&s_acc[l_tl + j], g[j] * w_tl); atomicAdd(&s_acc[l_tr + j], g[j] * w_tr);
                    atomicAdd(&s_acc[l_bl + j], g[j] * w_bl); atomicAdd(&s_acc[l_br + j], g[j] * w_br);
                } else {
                    if (nearest) { atomicAdd(&p.grad_texture[o_tl + j], g[j]); continue; }
                    atomicAdd(&p.grad_texture[o_tl + j], g[j] * w_tl); atomicAdd(&p.grad_texture[o_tr + j], g[j] * w_tr);
                    atomicAdd(&p.grad_texture[o_bl + j], g[j] * w_bl); atomicAdd(&p.grad_texture[o_br + j], g[j] * w_br);
                }
            }
        }
        if (patch) {
            __syncthreads();
            // the patch to memory: entry e = (patch row, patch column, channel); consecutive lanes -> consecutive floats of a texture row
            for (int e = tid; e < bh * bw * LCT; e += 256) {
                const float val = s_acc[e];
                const int j = e % LCT, t = e / LCT;
                if (val != 0.f && j < nc) {
                    const int pr = t / bw, pc = t - pr * bw;
                    atomicAdd(&p.grad_texture[((size_t)(rmin + pr) * p.Wt + (cmin + pc)) * Ct + c0 + j], val);
                }
            }
            __syncthreads();
        }
    }
    if (active && p.grad_uvs) {   // floor() has zero gradient: d frac / d index = 1 (nearest: zero)
        p.grad_uvs[i * p.guv_stride] = nearest ? 0.f : d_fc * dcol_du;
        p.grad_uvs[i * p.guv_stride + 1] = nearest ? 0.f : d_fr * drow_dv;
    }
}

hipError_t launch_texture_forward(const TexParams& p, hipStream_t stream)
{
    if (p.n == 0) return hipSuccess;
    long long blocks = (p.n + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const bool a16 = (reinterpret_cast<uintptr_t>(p.texture) & 15u) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0;
    if (p.Ct == 4 && a16) hipLaunchKernelGGL(texture_forward_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else if (p.Ct == 3) hipLaunchKernelGGL(texture_forward_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else if (p.Ct == 1) hipLaunchKernelGGL(texture_forward_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(texture_forward_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_texture_backward(const TexParams& p, long long rows, long long cols, hipStream_t stream)
{
    if (p.n == 0) return hipSuccess;
    // the pixel grid: an image `cols` wide in 16 x 16 tiles, or a flat list (rows == 1) in runs of 256
    const int tw = rows > 1 ? 16 : 256, th = rows > 1 ? 16 : 1;
    const long long tiles_x = (cols + tw - 1) / tw, tiles_y = (rows + th - 1) / th;
    if (tiles_x * tiles_y > 0x7fffffffll || cols > 0x7fffffffll || rows > 0x7fffffffll) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(tiles_x * tiles_y)), block(256);
    const bool a16 = (reinterpret_cast<uintptr_t>(p.grad_out) & 15u) == 0;
    if (p.Ct == 4 && a16) hipLaunchKernelGGL(texture_backward_kernel<4>, grid, block, 0, stream, p, (int)rows, (int)cols, tw, th, (int)tiles_x);
    else if (p.Ct == 3) hipLaunchKernelGGL(texture_backward_kernel<3>, grid, block, 0, stream, p, (int)rows, (int)cols, tw, th, (int)tiles_x);
    else if (p.Ct == 1) hipLaunchKernelGGL(texture_backward_kernel<1>, grid, block, 0, stream, p, (int)rows, (int)cols, tw, th, (int)tiles_x);
    else hipLaunchKernelGGL(texture_backward_kernel<0>, grid, block, 0, stream, p, (int)rows, (int)cols, tw, th, (int)tiles_x);
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
    const hipError_t e = dirt::launch_texture_forward(p, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) { snprintf(g_tex_error, sizeof(g_tex_error), "dirt_texture_sample_forward: %s", hipGetErrorString(e)); return DIRT_E_HIP; }
    g_tex_error[0] = 0;
    return DIRT_OK;
}

int dirt_texture_sample_backward_image(const float* texture, const float* uvs, const float* grad_out, float* grad_texture, float* grad_uvs,
                                       long long rows, long long cols, int Ht, int Wt, int Ct, int uv_stride, int grad_uv_stride, unsigned flags,
                                       void* stream)
{
    const char* who = "dirt_texture_sample_backward";
    if (rows < 0 || cols < 0 || (rows > 0 && cols > 0x7fffffffffffffffll / rows)) {
        snprintf(g_tex_error, sizeof(g_tex_error), "%s: bad pixel grid (rows=%lld cols=%lld)", who, rows, cols);
        return DIRT_E_INVALID_ARGUMENT;
    }
    const long long n = rows * cols;
    int rc = tex_check(who, texture, uvs, n, Ht, Wt, Ct, uv_stride);
    if (rc) return rc;
    if (n > 0 && (!grad_out || !grad_texture)) {
        snprintf(g_tex_error, sizeof(g_tex_error), "%s: grad_out / grad_texture is NULL", who);
        return DIRT_E_INVALID_ARGUMENT;
    }
    if (grad_uvs && grad_uv_stride < 2) {
        snprintf(g_tex_error, sizeof(g_tex_error), "%s: grad_uv_stride < 2", who);
        return DIRT_E_INVALID_ARGUMENT;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(grad_texture, 0, sizeof(float) * (size_t)Ht * Wt * Ct, s);
    if (e == hipSuccess) {
        dirt::TexParams p{};
        p.texture = texture; p.uvs = uvs; p.n = n; p.Ht = Ht; p.Wt = Wt; p.Ct = Ct; p.uv_stride = uv_stride; p.guv_stride = grad_uv_stride;
        p.flags = flags; p.grad_out = grad_out; p.grad_texture = grad_texture; p.grad_uvs = grad_uvs;
        e = dirt::launch_texture_backward(p, rows, cols, s);
    }
    if (e != hipSuccess) { snprintf(g_tex_error, sizeof(g_tex_error), "%s: %s", who, hipGetErrorString(e)); return DIRT_E_HIP; }
    g_tex_error[0] = 0;
    return DIRT_OK;
}

int dirt_texture_sample_backward(const float* texture, const float* uvs, const float* grad_out, float* grad_texture, float* grad_uvs,
                                 long long n, int Ht, int Wt, int Ct, int uv_stride, int grad_uv_stride, unsigned flags, void* stream)
{
    // a flat list of n look-ups: one row of n pixels
    return dirt_texture_sample_backward_image(texture, uvs, grad_out, grad_texture, grad_uvs, 1, n, Ht, Wt, Ct, uv_stride, grad_uv_stride, flags, stream);
}

}  // extern "C"
