// dirt_grad_common.h -- helpers shared by the gradient kernels (dirt_grad.hip: 4 pixels per lane, 32 x 32 tiles;
// dirt_grad_small.hip: 1 pixel per lane, 16 x 16 tiles for small frames).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dirt {

// Quirk Q1 at the right image border: for the pixels of a strip (first column xs, row y) flagged in `which`, the
// aliased "channels" 1, 2 of 1-channel group c -- elements (pixel + 1, + 2) of the flattened [B,H,W,1] slice -- lie in
// the NEXT image row (past the end of the tensor they are clamped to its last element; undefined in the reference).
// Their dilation axis (:185) is decided again from memory -- the 5 x 3 window of elements the three Scharr stencils
// cover, requested together -- and replaces bits shift .. shift+3 of `bits`.  Rare (the last two interior columns of a
// frame): a rolled loop behind a wave-uniform branch.
__device__ __forceinline__ uint32_t alias_wrap_fixup(const float* __restrict__ pixels, int B, int H, int W, int C, int iib, int y,
                                                  int xs, int c, uint32_t which, uint32_t bits, int shift)
{
    const size_t last = (size_t)B * H * W - 1;
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        if (!((which >> j) & 1u)) continue;
        // w[r][i]: element (centre + i - 1) of row y - 1 + r in flat order; at(ox, oy) of "channel" ch = w[1 - oy][ch + 1 + ox]
        float w[3][5];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const size_t base = ((size_t)iib * H + (y - 1 + r)) * W + xs + j - 1;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                size_t m = base + i;
                if (m > last) m = last;
                w[r][i] = pixels[m * C + c];
            }
        }
        float l1x = 0.f, l1y = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float mm = w[2][ch], m0 = w[1][ch], mp = w[0][ch];
            const float zm = w[2][ch + 1], zp = w[0][ch + 1];
            const float pm = w[2][ch + 2], p0 = w[1][ch + 2], pp = w[0][ch + 2];
            float d1 = ((mm + mp) - pm) - pp;
            float d2 = m0 - p0;
            float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
            const float sx = m1 + m2;
            d1 = ((mm + pm) - mp) - pp;
            d2 = zm - zp;
            m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
            const float sy = m1 + m2;
            l1x = ch == 0 ? fabsf(sx) : l1x + fabsf(sx);
            l1y = ch == 0 ? fabsf(sy) : l1y + fabsf(sy);
        }
        bits = (bits & ~(1u << (shift + j))) | ((l1x > l1y) ? (1u << (shift + j)) : 0u);
    }
    return bits;
}

// The reference's diagnostic output (csrc/rasterise_grad_egl.cu:150-151,172) of one pixel, for the first channel group
// (G channels starting at channel 0): [0] = 1e-2 where dilation fired, [1], [2] = elements (pix * G + 1, + 2) of the
// contiguous [B,H,W,G] slice of grad_pixels, clamped to its end.  Optional: only in the DEBUG instantiations.
__device__ __forceinline__ void write_debug(float* __restrict__ debug_thingy, const float* __restrict__ grad_pixels, int B, int H, int W,
                                        int C, int iib, int y, int x, int G, bool dilated)
{
    const size_t total_pix = (size_t)B * H * W;
    const size_t pix = ((size_t)iib * H + y) * W + x;
    float* dbg = debug_thingy + pix * 3;
    dbg[0] = dilated ? 1.e-2f : 0.f;
    for (int ch = 1; ch <= 2; ++ch) {
        size_t mp = G == 3 ? pix : pix + ch;      // pixel of that element
        int mc = G == 3 ? ch : 0;                 // channel inside the group
        if (mp > total_pix - 1) { mp = total_pix - 1; mc = G - 1; }
        dbg[ch] = grad_pixels[mp * C + mc];
    }
}

typedef unsigned long long lanemask;   // one bit per lane of the wave, wave-uniform (a scalar register pair)
typedef float float2v __attribute__((ext_vector_type(2)));   // a register pair for the packed fp32 instructions (v_pk_fma_f32)

// (s, s) * b [+ c] in one packed instruction.  op_sel_hi:[0,1,1] makes both halves take their first factor from the LOW
// register of the first operand's pair, so the scalar needs no copy into a second register (the compiler, given a
// splat, emits a v_mov per use); the pair's high register is never read.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
__device__ __forceinline__ float2v pk_fma_scalar(float s, float2v b, float2v c)
{
    float2v a;
    a.x = s;
    float2v d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float2v pk_mul_scalar(float s, float2v b)
{
    float2v a;
    a.x = s;
    float2v d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
#pragma clang diagnostic pop

// NDC coordinate of the centre of pixel i of n: (2 i + 1 - n) / n.  The numerator is an exact small integer, so the result carries
// two roundings RELATIVE to its own size -- where ((i + 0.5) * (2 / n)) - 1 has an absolute error of an ulp of 1 whatever the
// value, i.e. no correct digit at the centre of the frame: there the reference's clip_x = sum_k b_k * vertex_k.x (:210-217), which
// this stands for, is small and accurate, and the position gradient's w term differed by up to 27 ulps of that sum's scale
// (round 5 fuzz sweep: two elements in 1 727 cases beyond the tolerance's cancellation term; none since).
__device__ __forceinline__ float ndc_of(int i, int n, float inv_n)
{
    return (float)(2 * i + 1 - n) * inv_n;
}

struct Float3 { float x, y, z; };   // three channels of a pixel: one 12-byte load / store (4-byte aligned)

// Loads / stores at a 32-bit byte offset from a wave-uniform base: the address stays "scalar base + vector offset"
// (one VGPR per address instead of two, no 64-bit vector arithmetic).
template <class T>
__device__ __forceinline__ T ld_off(const void* base, uint32_t off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
}
template <class T>
__device__ __forceinline__ void st_off(void* base, uint32_t off, T v)
{
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off) = v;
}

}  // namespace dirt
