// dirt_grad.hip -- gradient assembly kernel for gfx950.
//
// Replaces assemble_grads / launch_grad_assembly (csrc/rasterise_grad_egl.cu:93-278) and, by
// evaluating every channel group of dirt/rasterise_ops.py:145-165 inside one launch, the N
// per-group RasteriseGrad ops (and N GL re-draws) the reference issues for C not in {1,3}.
//
// Input: the per-pixel state the raster kernel leaves behind -- one float4 {b0, b1, clip_w, face} per
// pixel, the counterpart of the reference's two RGBA32F surfaces (csrc/rasterise_grad_egl.cpp:432-456),
// produced by the forward pass itself when it keeps its state -- plus `faces` for the vertex indices.
//
// The reference issues up to 3C+9 global float atomics per covered pixel (:140,228-230) and reads its 3x3
// neighbourhood with 27 scalar loads.  Here:
//   * a 256-thread workgroup stages one 32 x 32 tile (+ halo) in LDS -- the pass's channels of `pixels` as
//     planes, and {clip_w, i0, i1, i2} of every pixel (vertex indices gathered from `faces`) -- then its four
//     waves work independently, with no further barrier: a wave owns 32 x 8 pixels, a lane a 4 x 1 strip, so
//     that Scharr taps, address arithmetic and predicates are shared by four pixels;
//   * nothing is accumulated in memory.  A lane keeps, per pixel, the barycentrics and seven factors
//     (four colour channels, and the x / y / w position factors of the channel groups that were not dilated);
//     the rare dilated (pixel, group) pairs -- which take a NEIGHBOUR's barycentrics and face -- go to a short
//     per-wave list in LDS;
//   * the wave then walks the distinct faces it has seen: for each, every lane forms its masked partial sums
//     (3 vertices x 7 values), the 21 sums are reduced across the wave with a transposing butterfly
//     (v_permlane32_swap / v_permlane16_swap / DPP row operations: 21 totals land in 21 different lanes at
//     ~60 instructions), and ONE global_atomic_add_f32 instruction adds them to the face's three vertices.
//     Per wave and face that is one atomic instruction instead of 21 x 256; no LDS atomics, no hash table,
//     no fixed point.
// Variable names in the per-pixel arithmetic follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "../../include/dirt_hip.h"
#include <type_traits>

namespace dirt {

constexpr int GT = 32;                  // tile side (pixels)
constexpr int GTHREADS = 256;           // 4 waves: wave w owns rows 8w .. 8w+7, lane l the strip x = 4 * (l & 7) .. +3 of row l >> 3
constexpr int PR = GT + 2;              // staged rows: y0-1 .. y0+32
constexpr int PS = 36;                  // plane row stride (floats): column (x - x0) for x0 .. x0+34, column 35 holds x0-1
constexpr int VS = GT + 2;              // state tile row stride (float4)
constexpr int PC = 4;                   // channels per pass: whole channel groups that fit in 4 channels
constexpr int LIST_CAP = 64;            // dilated (pixel, group) pairs listed per wave; the rest use direct atomics
constexpr int ENTRY_FLOATS = 8;         // {i0, i1, i2, target pixel | fx, fy, fw, -}
constexpr int ENTRIES_PER_PLANE = 6 * PS / ENTRY_FLOATS;  // a wave's private rows of one plane: 6 x 36 floats = 27 entries

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the 16 lanes of a DPP row, left in all 16 of them.
__device__ __forceinline__ float row_sum16(float v)
{
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    return dpp_add<0x140>(v);  // row_mirror
}

// Reduce NV4 (a multiple of 4) per-lane values across the 64 lanes of the wave.  Returns, in lane 16 * r + c with
// c < NV4 / 4, the total of value c + (NV4 / 4) * r; other lanes hold nothing of interest.
//   step 1: v_permlane32_swap pairs value i with value i + NV4/2: one add leaves value i's half-sums in lanes 0-31
//           and value (i + NV4/2)'s in lanes 32-63 -- half the registers;
//   step 2: v_permlane16_swap does the same between DPP rows -- a quarter of the registers, each row a different value;
//   step 3: four DPP adds inside the rows.
template <int NV4>
__device__ __forceinline__ float wave_reduce_scatter(const float* val, int lane)
{
    static_assert(NV4 % 4 == 0, "NV4 must be a multiple of 4");
    constexpr int H1 = NV4 / 2, H2 = NV4 / 4;
    float r1[H1];
#pragma unroll
    for (int i = 0; i < H1; ++i) {
        const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(val[i]), __float_as_uint(val[i + H1]), false, false);
        r1[i] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
    float r2[H2];
#pragma unroll
    for (int i = 0; i < H2; ++i) {
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1[i]), __float_as_uint(r1[i + H2]), false, false);
        r2[i] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
    const int c = lane & 15;
    float out = row_sum16(r2[0]);
#pragma unroll
    for (int i = 1; i < H2; ++i) {
        const float t = row_sum16(r2[i]);
        out = (c == i) ? t : out;
    }
    return out;
}

// Scharr responses of one channel from global memory for an aliased (quirk Q1) channel whose taps run past the
// end of the image row: the tap centre is pixel (row y, column x) of scene iib, `shift` elements further on in the
// flattened [B,H,W] slice; reads past the end of the tensor are clamped to its last element (undefined in the
// reference).  Rare (only the last columns of a frame): behind wave-uniform branches, no function call in the kernel.
__device__ __forceinline__ float2 scharr_taps_wrapped(const float* __restrict__ pixels, int B, int H, int W, int C, int iib, int y,
                                                  int x, int shift, int c)
{
    const size_t total_pix = (size_t)B * H * W;
    const size_t centre = ((size_t)iib * H + y) * W + x + shift;
    float sx, sy;
    auto at = [&](int ox, int oy) {
        size_t m = centre + (size_t)ox - (size_t)((long long)oy * W);  // offset_y up = previous row
        if (m > total_pix - 1) m = total_pix - 1;
        return pixels[m * C + c];
    };
    float d1 = ((at(-1, -1) + at(-1, +1)) - at(+1, -1)) - at(+1, +1);
    float d2 = at(-1, 0) - at(+1, 0);
    float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
    sx = m1 + m2;
    d1 = ((at(-1, -1) + at(+1, -1)) - at(-1, +1)) - at(+1, +1);
    d2 = at(0, -1) - at(0, +1);
    m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
    sy = m1 + m2;
    return make_float2(sx, sy);
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

// grad_kernel<CSPEC>: CSPEC = 1, 3, 4: the channel count is that compile-time constant (4: with 16-byte aligned
// pixel tensors) and the kernel is one pass; 0: any channel count, one pass per <= 4 channels of whole groups.
// DEBUG: also write the reference's diagnostic output debug_thingy.
template <int CSPEC, bool DEBUG>
__global__ __launch_bounds__(GTHREADS, CSPEC ? 4 : 3) void grad_kernel(GradParams p)
{
    constexpr int NPLANES = CSPEC ? CSPEC : PC;
    __shared__ __align__(16) float s_pix[NPLANES][PR][PS];  // the pass's channels of `pixels`, edge clamped
    __shared__ float4 s_vw[PR][VS];                         // {clip_w, i0, i1, i2} of every pixel of the halo'd tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int iib = blockIdx.y;
    const int H = p.H, W = p.W, C = CSPEC ? CSPEC : p.C;
    const bool aligned16 = CSPEC == 4 ? true : (p.pixels_aligned16 != 0);
    const size_t frame = (size_t)H * W;
    const int tile = xcd_tile((int)blockIdx.x, p.tiles_x * p.tiles_y);
    const int x0 = (tile % p.tiles_x) * GT, y0 = (tile / p.tiles_x) * GT;

    // Wave-uniform bases at the first staged row of the tile (row0), so that every per-lane address is a small
    // non-negative 32-bit byte offset: (row - row0) * W + column, times the element size.
    const int row0 = max(y0 - 1, 0);
    const size_t origin = (size_t)iib * frame + (size_t)row0 * W;   // pixel index of (row0, column 0)
    const float4* __restrict__ state_t = p.state + origin;
    const float* __restrict__ pixels_t = p.pixels + origin * C;
    const float* __restrict__ gpix_t = p.grad_pixels + origin * C;
    float* __restrict__ gbk_t = p.grad_background + origin * C;
    const int32_t* __restrict__ faces = p.faces + (p.shared_faces ? (size_t)0 : (size_t)iib * p.F * 3);
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * 4;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * C;
    const uint32_t pixel_bytes = 4u * (uint32_t)C;

    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;
    const float width_f = (float)W, height_f = (float)H;

    // ---- this lane's strip ----
    const int sx = lane & 7, ry = lane >> 3;
    const int xs = x0 + 4 * sx;                 // first pixel of the strip
    const int y = y0 + 8 * wave + ry;           // tensor row (top row first)
    const int hr = 8 * wave + ry + 1;           // its row in the halo'd tile
    // pixel index, relative to (row0, 0), of the strip's first pixel; lanes outside the frame address its last pixel
    const uint32_t own_rel = (uint32_t)((min(y, H - 1) - row0) * W + min(xs, W - 1));
    bool in_px[4], interior[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        in_px[j] = (xs + j < W) & (y < H);
        interior[j] = in_px[j] & (xs + j > 0) & (y > 0) & (xs + j < W - 1) & (y < H - 1);
    }

    // channels of a pass: whole channel groups (dirt/rasterise_ops.py:148-152) starting at c0 that fit in PC channels
    auto pass_channels = [&](int c0) {
        if (CSPEC) return (int)CSPEC;
        int nch = 0;
        for (int c = c0; c < C && nch < PC;) {
            const int G = (c + 3 <= C) ? 3 : 1;
            if (nch + G > PC) break;
            nch += G; c += G;
        }
        return nch;
    };

    // ---- staging: loads of the pass's channels of the pixels tile (+halo), edge clamped (at(), :113-124).  Item i is
    //      row i / 36, column x0 - 1 + i % 36; five items per thread, every load issued before any use ----
    constexpr int PITEMS = (PR * PS + GTHREADS - 1) / GTHREADS;
    auto stage_load = [&](int c0, int nch, float (&v)[PITEMS][PC]) {
#pragma unroll
        for (int k = 0; k < PITEMS; ++k) {
            const int i = min(tid + k * GTHREADS, PR * PS - 1);
            const int row = i / PS, ci = i - row * PS;
            const int cy = min(max(y0 - 1 + row, 0), H - 1), cx = min(max(x0 - 1 + ci, 0), W - 1);
            const uint32_t off = (uint32_t)((cy - row0) * W + cx) * pixel_bytes + 4u * (uint32_t)c0;
            if (nch == 4 && (C & 3) == 0 && aligned16) {
                const float4 q = ld_off<float4>(pixels_t, off);  // c0 is a multiple of 4 here
                v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = q.w;
            } else {
#pragma unroll
                for (int ch = 0; ch < PC; ++ch) v[k][ch] = ch < nch ? ld_off<float>(pixels_t, off + 4u * ch) : 0.f;
            }
        }
    };
    auto stage_store = [&](int nch, const float (&v)[PITEMS][PC]) {
#pragma unroll
        for (int k = 0; k < PITEMS; ++k) {
            const int i = tid + k * GTHREADS;
            if (i >= PR * PS) continue;
            const int row = i / PS, ci = i - row * PS;
            const int col = ci == 0 ? PS - 1 : ci - 1;
#pragma unroll
            for (int ch = 0; ch < NPLANES; ++ch)
                if (ch < nch) s_pix[ch][row][col] = v[k][ch];
        }
    };

    // ---- phase A: the visibility "surfaces" of the tile + 1-pixel halo -- what the backward fragment shader writes
    //      (csrc/shaders.cpp:64-77) over the clear values of csrc/rasterise_grad_egl.cpp:442-445 -- as {clip_w,
    //      i0, i1, i2}.  Halo positions outside the frame are clamped; they are only ever consulted for interior
    //      pixels, whose neighbours are inside the frame. ----
    float stage_v[PITEMS][PC];
    stage_load(0, pass_channels(0), stage_v);
    {
        constexpr int VITEMS = (PR * VS + GTHREADS - 1) / GTHREADS;
        float4 rec[VITEMS];
#pragma unroll
        for (int k = 0; k < VITEMS; ++k) {
            const int i = min(tid + k * GTHREADS, PR * VS - 1);
            const int row = i / VS, ci = i - row * VS;
            const int cy = min(max(y0 - 1 + row, 0), H - 1), cx = min(max(x0 - 1 + ci, 0), W - 1);
            rec[k] = ld_off<float4>(state_t, (uint32_t)((cy - row0) * W + cx) * 16u);
        }
        int vid[VITEMS][3];
#pragma unroll
        for (int k = 0; k < VITEMS; ++k) {
            const int f = __float_as_int(rec[k].w);
            const uint32_t fo = (uint32_t)max(f, 0) * 12u;
            vid[k][0] = ld_off<int>(faces, fo); vid[k][1] = ld_off<int>(faces, fo + 4u); vid[k][2] = ld_off<int>(faces, fo + 8u);
            if (f < 0) { vid[k][0] = -1; vid[k][1] = -1; vid[k][2] = -1; }
        }
#pragma unroll
        for (int k = 0; k < VITEMS; ++k) {
            const int i = tid + k * GTHREADS;
            if (i >= PR * VS) continue;
            const int row = i / VS, ci = i - row * VS;
            s_vw[row][ci] = make_float4(rec[k].z, __int_as_float(vid[k][0]), __int_as_float(vid[k][1]), __int_as_float(vid[k][2]));
        }
    }

    // One pass = the channel groups that fit in PC channels, starting at channel c0.  A pass has one of four
    // shapes -- {3}, {3,1}, {1}, {1,1} (dirt/rasterise_ops.py:148-152 packs groups of 3 while >= 3 channels remain,
    // then singles) -- and the body is instantiated for each, so that its loops and branches over channels and
    // groups are static: NCH channels, the first group of size G0, every further group a single channel.
    auto run_pass = [&](auto nch_tag, auto g0_tag, const int c0) {
        constexpr int NCH = decltype(nch_tag)::value;
        constexpr int G0 = decltype(g0_tag)::value;
        constexpr int NG = 1 + (NCH - G0);          // channel groups in the pass
        constexpr int NV = 9 + 3 * NCH;             // values per face: 3 vertices x {x, y, w} + 3 vertices x NCH colours
        constexpr int NV4 = (NV + 3) / 4 * 4;
        constexpr int LCAP = (NPLANES * ENTRIES_PER_PLANE < LIST_CAP) ? NPLANES * ENTRIES_PER_PLANE : LIST_CAP;  // planes beyond NCH are idle

        if (c0 != 0) stage_load(c0, NCH, stage_v);
        // this strip's grad_pixels
        const uint32_t own_off = own_rel * pixel_bytes + 4u * (uint32_t)c0;
        float g[4][NCH];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t off = in_px[j] ? own_off + (uint32_t)j * pixel_bytes : 4u * (uint32_t)c0;  // outside the frame: any valid address
            bool wide = false;
            if constexpr (NCH == 4) {
                if ((C & 3) == 0 && aligned16) {
                    const float4 q = ld_off<float4>(gpix_t, off);
                    g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z; g[j][3] = q.w;
                    wide = true;
                }
            }
            if (!wide) {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) g[j][ch] = ld_off<float>(gpix_t, off + 4u * ch);
            }
        }
        stage_store(NCH, stage_v);
        __syncthreads();

        // ---- Scharr (:126-127, operation for operation: negative-offset minus positive-offset, offset_y is up = the
        //      previous tensor row), streamed per channel into what is needed of it: the direction choice of :185 from the
        //      L1 norms (all three "channels" of the reference's Vec3, in its summation order) and dL/dx, dL/dy of
        //      :203-208 ----
        uint32_t horiz_bits = 0;  // bit 4 * group + j: the pixel's dilation axis is x
        float dLx[NG][4], dLy[NG][4];
        {
            float l1x[4], l1y[4];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int gi = ch < G0 ? 0 : ch - G0 + 1;     // the channel's group
                const bool single = !(ch < G0 && G0 == 3);     // a 1-channel group: quirk Q1 applies
                const bool last_of_group = single || ch == G0 - 1;
                // taps: t[r][0] = column x-1, t[r][1..4] = the strip, t[r][5..7] = columns x+4 .. x+6
                float t[3][8];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float* rowp = &s_pix[ch][hr - 1 + r][0];
                    const float4 q = *reinterpret_cast<const float4*>(rowp + 4 * sx);
                    t[r][0] = rowp[sx == 0 ? PS - 1 : 4 * sx - 1];
                    t[r][1] = q.x; t[r][2] = q.y; t[r][3] = q.z; t[r][4] = q.w;
                    if (single) {
                        const float4 q2 = *reinterpret_cast<const float4*>(rowp + 4 * sx + 4);
                        t[r][5] = q2.x; t[r][6] = q2.y; t[r][7] = q2.z;
                    } else {
                        t[r][5] = rowp[4 * sx + 4]; t[r][6] = 0.f; t[r][7] = 0.f;
                    }
                }
                constexpr int NQ_MAX = 6;
                float Sx[NQ_MAX], Sy[NQ_MAX];
#pragma unroll
                for (int q = 0; q < NQ_MAX; ++q) {
                    if (q >= 4 && !single) { Sx[q] = 0.f; Sy[q] = 0.f; continue; }
                    // at(ox, oy): t[1 - oy][q + 1 + ox]
                    const float mm = t[2][q], m0 = t[1][q], mp = t[0][q];
                    const float zm = t[2][q + 1], zp = t[0][q + 1];
                    const float pm = t[2][q + 2], p0 = t[1][q + 2], pp = t[0][q + 2];
                    float d1 = ((mm + mp) - pm) - pp;
                    float d2 = m0 - p0;
                    float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
                    Sx[q] = m1 + m2;
                    d1 = ((mm + pm) - mp) - pp;
                    d2 = zm - zp;
                    m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
                    Sy[q] = m1 + m2;
                }
                if (!single) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float m = g[j][ch] * Sx[j];
                        dLx[gi][j] = ch == 0 ? m : dLx[gi][j] + m;
                        m = g[j][ch] * Sy[j];
                        dLy[gi][j] = ch == 0 ? m : dLy[gi][j] + m;
                        l1x[j] = ch == 0 ? fabsf(Sx[j]) : l1x[j] + fabsf(Sx[j]);
                        l1y[j] = ch == 0 ? fabsf(Sy[j]) : l1y[j] + fabsf(Sy[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        dLx[gi][j] = g[j][ch] * Sx[j];
                        dLy[gi][j] = g[j][ch] * Sy[j];
                        // quirk Q1: "channels" 1,2 of a 1-channel group = elements (pixel + 1, + 2) of the flattened
                        // [B,H,W,1] slice.  Only the L1 norms of interior pixels use them, and for an interior pixel the
                        // taps are unclamped: column + ch, which is staged unless it runs past the end of the image row
                        // (then it wraps to the next row: read from global memory).
                        float ax1 = Sx[j + 1], ay1 = Sy[j + 1], ax2 = Sx[j + 2], ay2 = Sy[j + 2];
                        if (!q1_intended && x0 + GT + 3 > W) {  // only tiles on the right image border
                            const bool wraps1 = interior[j] && xs + j + 2 > W - 1, wraps2 = interior[j] && xs + j + 3 > W - 1;
                            if (__builtin_amdgcn_ballot_w64(wraps2) != 0ull) {
                                if (wraps1) {
                                    const float2 w2 = scharr_taps_wrapped(p.pixels, p.B, H, W, C, iib, y, xs + j, 1, c0 + ch);
                                    ax1 = w2.x; ay1 = w2.y;
                                }
                                if (wraps2) {
                                    const float2 w2 = scharr_taps_wrapped(p.pixels, p.B, H, W, C, iib, y, xs + j, 2, c0 + ch);
                                    ax2 = w2.x; ay2 = w2.y;
                                }
                            }
                        }
                        const float a0x = fabsf(Sx[j]), a0y = fabsf(Sy[j]);
                        l1x[j] = q1_intended ? a0x : (a0x + fabsf(ax1)) + fabsf(ax2);
                        l1y[j] = q1_intended ? a0y : (a0y + fabsf(ay1)) + fabsf(ay2);
                    }
                }
                if (last_of_group) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) horiz_bits |= (l1x[j] > l1y[j]) ? (1u << (4 * gi + j)) : 0u;  // :185
                    asm volatile("" : "+v"(horiz_bits));  // decided here: the norms' registers are free again
                }
                __builtin_amdgcn_sched_barrier(0);  // one channel's taps at a time
            }
        }

        // ---- own pixels: clip_w and vertex indices (the halo'd tile), barycentrics b0, b1 (the state record) ----
        float w_own[4];
        int key[4][3];
        bool covered[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 q = s_vw[hr][4 * sx + j + 1];
            w_own[j] = q.x;
            covered[j] = in_px[j] & (__float_as_int(q.y) >= 0);
            key[j][0] = covered[j] ? __float_as_int(q.y) : -1;
            key[j][1] = covered[j] ? __float_as_int(q.z) : -1;
            key[j][2] = covered[j] ? __float_as_int(q.w) : -1;
        }
        float bk[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 q = ld_off<float2>(state_t, in_px[j] ? (own_rel + (uint32_t)j) * 16u : 0u);
            bk[j][0] = q.x; bk[j][1] = q.y;
        }

        // ---- background gradient (:143-147): grad_pixels where nothing is covered, zero elsewhere ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!in_px[j]) continue;
            const uint32_t off = own_off + (uint32_t)j * pixel_bytes;
            bool wide = false;
            if constexpr (NCH == 4) {
                if ((C & 3) == 0 && aligned16) {
                    st_off<float4>(gbk_t, off, covered[j] ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(g[j][0], g[j][1], g[j][2], g[j][3]));
                    wide = true;
                }
            }
            if (!wide) {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) st_off<float>(gbk_t, off + 4u * ch, covered[j] ? 0.f : g[j][ch]);
            }
        }

        // ---- per pixel factors: the colour gradients of vertex k are b_k * g[c] (:135-142); the position gradients
        //      b_k * (fx, fy, fw) (:224-230) with the factors summed over the groups that were not dilated ----
        float fpos[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) { fpos[j][0] = 0.f; fpos[j][1] = 0.f; fpos[j][2] = 0.f; }

        // the wave's private rows of the planes hold the list of dilated (pixel, group) pairs: rows 8w+2 .. 8w+7 of the
        // halo'd tile are read by this wave only, and its Scharr reads are complete
        uint32_t nlist = 0;  // wave-uniform
        auto entry_ptr = [&](uint32_t e) -> float* {
            const uint32_t pl = e / ENTRIES_PER_PLANE, k = e - pl * ENTRIES_PER_PLANE;
            return &s_pix[pl][8 * wave + 2][0] + k * ENTRY_FLOATS;
        };

        // ---- dilation (:155-194) and position factors (:196-232) of every (pixel, group) ----
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int c_begin = c0 + (gi == 0 ? 0 : G0 + gi - 1);
            const int G = gi == 0 ? G0 : 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xj = xs + j;
                // direction: x if L1(Sx) > L1(Sy) else y (:185), negated on odd (x + y) (:186-190).  The reference's
                // offsets are in GL buffer orientation (y up): tensor row = y - offset_y.
                const int sgn = ((xj + y) & 1) ? -1 : 1;
                const bool horiz = (horiz_bits >> (4 * gi + j)) & 1u;
                const int dx = horiz ? sgn : 0, dy = horiz ? 0 : -sgn;
                const int e0 = hr * VS + 4 * sx + j + 1;
                const int d = dy * VS + dx;
                const float4 n1 = (&s_vw[0][0])[e0 + d], n2 = (&s_vw[0][0])[e0 - d];
                // index triples differ (:86-89): an uncovered pixel (-1,-1,-1) differs from any face
                const bool c1 = __float_as_int(n1.y) >= 0, c2 = __float_as_int(n2.y) >= 0;
                const bool t1 = ((__float_as_int(n1.y) ^ key[j][0]) | (__float_as_int(n1.z) ^ key[j][1]) | (__float_as_int(n1.w) ^ key[j][2])) != 0;
                const bool t2 = ((__float_as_int(n2.y) ^ key[j][0]) | (__float_as_int(n2.z) ^ key[j][1]) | (__float_as_int(n2.w) ^ key[j][2])) != 0;
                const bool ok1 = interior[j] & c1 & t1 & (w_own[j] > n1.x);          // :165, first attempt (:191)
                const bool ok2 = interior[j] & !ok1 & c2 & t2 & (w_own[j] > n2.x);   // opposite direction if the first failed (:192-193)
                const bool dilated = ok1 | ok2;
                const int ddx = ok1 ? dx : (ok2 ? -dx : 0), ddy = ok1 ? dy : (ok2 ? -dy : 0);
                const float clip_w = ok1 ? n1.x : (ok2 ? n2.x : w_own[j]);
                const bool contributes = dilated | covered[j];

                if constexpr (DEBUG) {
                    if (c_begin == 0 && in_px[j]) write_debug(p.debug_thingy, p.grad_pixels, p.B, H, W, C, iib, y, xj, G, dilated);
                }

                // clip-space x,y of the fragment used (:210-215 sums b_k * vertex_k.xy; perspective-correct
                // barycentrics make that sum the fragment's own clip position = its NDC position times clip_w, so no
                // vertex gather is needed; agrees to float rounding).  :219-222 with one reciprocal (v_rcp_f32, 1 ulp):
                //   gx = dL_dx * b_k * (W/2) / w ; gy likewise ; gw = -(gx * ndc_x + gy * ndc_y)
                const float ndc_x = ((float)(xj + ddx) + 0.5f) * (2.f / width_f) - 1.f;
                const float ndc_y = ((float)(H - 1 - (y + ddy)) + 0.5f) * (2.f / height_f) - 1.f;
                const float rcp_w = __builtin_amdgcn_rcpf(clip_w);
                const float fx = contributes ? (dLx[gi][j] * (.5f * width_f)) * rcp_w : 0.f;
                const float fy = contributes ? (dLy[gi][j] * (.5f * height_f)) * rcp_w : 0.f;
                const float fw = -(fx * ndc_x + fy * ndc_y);
                if (!dilated) { fpos[j][0] += fx; fpos[j][1] += fy; fpos[j][2] += fw; }

                // dilated pairs take the neighbour's face and barycentrics: list them for the face loop
                const unsigned long long dm = __builtin_amdgcn_ballot_w64(dilated);
                if (dm != 0ull) {
                    const uint32_t e = nlist + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
                    const float4 nsel = ok1 ? n1 : n2;
                    const uint32_t trel = (uint32_t)((int)own_rel + j + ddy * W + ddx);  // the neighbour, relative to (row0, 0)
                    if (dilated && e < (uint32_t)LCAP) {
                        float* ep = entry_ptr(e);
                        *reinterpret_cast<float4*>(ep) = make_float4(nsel.y, nsel.z, nsel.w, __uint_as_float(trel));
                        *reinterpret_cast<float4*>(ep + 4) = make_float4(fx, fy, fw, 0.f);
                    }
                    nlist += (uint32_t)__popcll(dm);
                    if (nlist > (uint32_t)LCAP) {  // list full (many dilated pairs in one wave): the reference's direct float atomics
                        if (dilated && e >= (uint32_t)LCAP) {
                            const float2 nb = ld_off<float2>(state_t, trel * 16u);
                            const float bb[3] = {nb.x, nb.y, (1.f - nb.x) - nb.y};
                            const int vi[3] = {__float_as_int(nsel.y), __float_as_int(nsel.z), __float_as_int(nsel.w)};
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                float* gv = grad_vertices + (size_t)vi[k] * 4;
                                atomicAdd(gv + 0, fx * bb[k]);
                                atomicAdd(gv + 1, fy * bb[k]);
                                atomicAdd(gv + 3, fw * bb[k]);
                            }
                        }
                        nlist = (uint32_t)LCAP;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- the list, one entry per lane ----
        int lkey[3] = {-1, -1, -1};
        float lb[3] = {0.f, 0.f, 0.f}, lf[3] = {0.f, 0.f, 0.f};
        if ((uint32_t)lane < nlist) {
            const float* ep = entry_ptr((uint32_t)lane);
            const float4 a = *reinterpret_cast<const float4*>(ep), f = *reinterpret_cast<const float4*>(ep + 4);
            lkey[0] = __float_as_int(a.x); lkey[1] = __float_as_int(a.y); lkey[2] = __float_as_int(a.z);
            const float2 nb = ld_off<float2>(state_t, __float_as_uint(a.w) * 16u);
            lb[0] = nb.x; lb[1] = nb.y; lb[2] = (1.f - nb.x) - nb.y;
            lf[0] = f.x; lf[1] = f.y; lf[2] = f.z;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bk[j][2] = (1.f - bk[j][0]) - bk[j][1];

        // ---- this lane's role in the face loop: lane 16 r + c (c < NV4/4) adds value v = c + (NV4/4) r of the face:
        //      v < 9: component v % 3 (x, y, w) of grad_vertices of vertex v / 3; else colour (v - 9) % NCH of vertex
        //      (v - 9) / NCH ----
        const int role_v = (lane & 15) + (NV4 / 4) * (lane >> 4);
        const bool role_valid = (lane & 15) < NV4 / 4 && role_v < NV;
        const bool role_pos = role_v < 9;
        const int role_k = role_pos ? role_v / 3 : (role_v - 9) / NCH;
        const int role_e = role_pos ? (role_v % 3 == 2 ? 3 : role_v % 3) : c0 + (role_v - 9) % NCH;
        float* const role_base = role_pos ? grad_vertices + role_e : grad_vertex_colors + role_e;
        const uint32_t role_stride = role_pos ? 16u : pixel_bytes;

        // ---- the face loop: pending pixels / list entries as wave-wide masks (scalar registers) ----
        unsigned long long pend[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) pend[j] = __builtin_amdgcn_ballot_w64(covered[j]);
        pend[4] = __builtin_amdgcn_ballot_w64(lkey[0] >= 0);
        for (;;) {
            int K0, K1, K2;
            {
                int src;
                if (pend[0]) { src = __ffsll((long long)pend[0]) - 1; K0 = __builtin_amdgcn_readlane(key[0][0], src); K1 = __builtin_amdgcn_readlane(key[0][1], src); K2 = __builtin_amdgcn_readlane(key[0][2], src); }
                else if (pend[1]) { src = __ffsll((long long)pend[1]) - 1; K0 = __builtin_amdgcn_readlane(key[1][0], src); K1 = __builtin_amdgcn_readlane(key[1][1], src); K2 = __builtin_amdgcn_readlane(key[1][2], src); }
                else if (pend[2]) { src = __ffsll((long long)pend[2]) - 1; K0 = __builtin_amdgcn_readlane(key[2][0], src); K1 = __builtin_amdgcn_readlane(key[2][1], src); K2 = __builtin_amdgcn_readlane(key[2][2], src); }
                else if (pend[3]) { src = __ffsll((long long)pend[3]) - 1; K0 = __builtin_amdgcn_readlane(key[3][0], src); K1 = __builtin_amdgcn_readlane(key[3][1], src); K2 = __builtin_amdgcn_readlane(key[3][2], src); }
                else if (pend[4]) { src = __ffsll((long long)pend[4]) - 1; K0 = __builtin_amdgcn_readlane(lkey[0], src); K1 = __builtin_amdgcn_readlane(lkey[1], src); K2 = __builtin_amdgcn_readlane(lkey[2], src); }
                else break;
            }
            float acc[NV4];
#pragma unroll
            for (int v = NV; v < NV4; ++v) acc[v] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool m = (key[j][0] == K0) & (key[j][1] == K1) & (key[j][2] == K2);
                pend[j] &= ~__builtin_amdgcn_ballot_w64(m);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float bm = m ? bk[j][k] : 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[3 * k + c] = j == 0 ? bm * fpos[j][c] : fmaf(bm, fpos[j][c], acc[3 * k + c]);
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch)
                        acc[9 + NCH * k + ch] = j == 0 ? bm * g[j][ch] : fmaf(bm, g[j][ch], acc[9 + NCH * k + ch]);
                }
            }
            {
                const bool m = (lkey[0] == K0) & (lkey[1] == K1) & (lkey[2] == K2);
                const unsigned long long mm = __builtin_amdgcn_ballot_w64(m);
                if (mm != 0ull) {
                    pend[4] &= ~mm;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float bm = m ? lb[k] : 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[3 * k + c] = fmaf(bm, lf[c], acc[3 * k + c]);
                    }
                }
            }
            const float total = wave_reduce_scatter<NV4>(acc, lane);
            const int vsel = role_k == 0 ? K0 : (role_k == 1 ? K1 : K2);
            if (role_valid && total != 0.f)
                atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(role_base) + (size_t)((uint32_t)vsel * role_stride)), total);
        }
    };

    using std::integral_constant;
    for (int c0 = 0; c0 < C;) {
        const int nch = pass_channels(c0);
        if constexpr (CSPEC != 0) {
            run_pass(integral_constant<int, CSPEC>{}, integral_constant<int, CSPEC == 1 ? 1 : 3>{}, 0);
        } else if (c0 + 3 <= C) {
            if (nch == 4) run_pass(integral_constant<int, 4>{}, integral_constant<int, 3>{}, c0);
            else run_pass(integral_constant<int, 3>{}, integral_constant<int, 3>{}, c0);
        } else {
            if (nch == 2) run_pass(integral_constant<int, 2>{}, integral_constant<int, 1>{}, c0);
            else run_pass(integral_constant<int, 1>{}, integral_constant<int, 1>{}, c0);
        }
        c0 += nch;
        if (CSPEC) break;  // a single pass, statically
        if (c0 < C) __syncthreads();  // every wave is done with the planes (and with its list inside them)
    }
}

hipError_t launch_grad(const GradParams& p_in, hipStream_t stream)
{
    if (p_in.B == 0) return hipSuccess;
    GradParams p = p_in;
    p.tiles_x = (p.W + GT - 1) / GT;
    p.tiles_y = (p.H + GT - 1) / GT;
    p.pixels_aligned16 = ((reinterpret_cast<uintptr_t>(p.pixels) | reinterpret_cast<uintptr_t>(p.grad_pixels) |
                           reinterpret_cast<uintptr_t>(p.grad_background)) & 15u) == 0 ? 1 : 0;
    // the common channel counts get kernels in which the pass / channel-group structure is static
    const int cspec = (p.C == 4 && p.pixels_aligned16) ? 4 : (p.C == 3 ? 3 : (p.C == 1 ? 1 : 0));
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)p.B), block(GTHREADS);
#define DIRT_LAUNCH_GRAD(DBG_)                                                                          \
    do {                                                                                               \
        if (cspec == 4) hipLaunchKernelGGL((grad_kernel<4, DBG_>), grid, block, 0, stream, p);         \
        else if (cspec == 3) hipLaunchKernelGGL((grad_kernel<3, DBG_>), grid, block, 0, stream, p);    \
        else if (cspec == 1) hipLaunchKernelGGL((grad_kernel<1, DBG_>), grid, block, 0, stream, p);    \
        else hipLaunchKernelGGL((grad_kernel<0, DBG_>), grid, block, 0, stream, p);                    \
    } while (0)
    if (p.debug_thingy) DIRT_LAUNCH_GRAD(true);
    else DIRT_LAUNCH_GRAD(false);
#undef DIRT_LAUNCH_GRAD
    return hipGetLastError();
}

}  // namespace dirt
