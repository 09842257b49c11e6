// dirt_grad_px2.hip -- the gradient assembly kernel with TWO pixels per lane (gfx950), round 5.
//
// Same contract as grad_kernel (dirt_grad.hip; replaces assemble_grads, csrc/rasterise_grad_egl.cu:93-236, for all channel
// groups of dirt/rasterise_ops.py:145-165 in one launch), same per-pixel arithmetic and the same face loop -- other
// decomposition.  grad_kernel gives a lane a 4 x 1 strip, a wave 32 x 8 pixels and a workgroup a 32 x 32 tile: at
// 1024 x 1024 that is 4096 waves of ~2 900 instructions at 104-108 VGPRs -- exactly the chip's 4096 wave slots at that
// register count, ONE lockstep round (profiles/EXPERIMENTS.md, round 4).  Here a lane owns a 2 x 1 PAIR (exactly one operand
// pair of the packed fp32 Scharr arithmetic), a DPP row of 16 lanes an 8 x 4 block, a wave 16 x 8 pixels and a workgroup
// (4 waves) a 32 x 16 tile: twice the waves and workgroups, each with about half the instruction chain, 14-23 KB of LDS and
// 60-88 VGPRs (five workgroups per compute unit for 4 channels, eight for 1 and 3).
//   * the two DPP rows that share a face in the loop (a "pair of rows": rows 0, 1 and rows 2, 3) are stacked vertically:
//     an 8 x 8 block walks its distinct faces together -- 3.9 iterations per wave at K3 where the 16 x 8 halves of the
//     4-pixel kernel need 5.6 (before ring cells), each with half the packed multiply-adds;
//   * the price: 1.35 x the (block, face) pairs, i.e. float atomics (52.6 k against 38.8 k row groups x 3 vertices at K3),
//     7.6 % more vector and 43 % more scalar instructions summed over the waves.
// What it was built for -- the 1024 x 1024 x 4-channel headline -- it does NOT speed up (25.3 against 25.7 us; with the op's
// dense output rows +0.7 us on the step): 62 % of its waves still start in one round and wait 3.6 us for their loads, the
// rest run at three waves per SIMD, and the kernel is bound by the instructions it issues (DESIGN.md 4.1a,
// profiles/EXPERIMENTS.md "Round 5": per-wave trace, counters, the all-resident and the persistent two-tile variants).
// It is the library's choice where it measured faster: 3-channel images of every size above 256 tiles (K3-3ch 23.1 -> 22.2,
// K5-3ch 66.5 -> 62.2 us: the image deferred shading differentiates) and frames whose 4-pixel grid leaves compute units
// short of workgroups (257-1023 tiles; K3-768: 24.0 -> 20.8 us).  Channel counts 1, 3, 4 (the image's, a compile-time
// constant); other counts keep grad_kernel's channel passes.  Variable names in the per-pixel arithmetic follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "dirt_reduce.h"
#include "dirt_grad_common.h"
#include "../../include/dirt_hip.h"
#include <type_traits>

namespace dirt {

#ifdef DIRT_TRACE
// Per-wave phase timestamps for tools/trace_grad.py (the layout of dirt_grad.hip's trace); tracing build only.
__device__ long long* g_trace_grad_px2 = nullptr;
extern "C" void dirt_debug_set_trace_grad_px2(void* p)
{
    long long* q = reinterpret_cast<long long*>(p);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_grad_px2), &q, sizeof(q));
}
#define XMARK() do { if (tr_n < 12) { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tr_t[tr_n++] = t_; } } while (0)
#define XCOUNT(i, v) do { tr_c[i] += (v); } while (0)
#else
#define XMARK() do {} while (0)
#define XCOUNT(i, v) do {} while (0)
#endif

#ifndef DIRT_PX2_WAVES
#define DIRT_PX2_WAVES(C_, DEBUG_) ((C_) == 4 ? 5 : ((DEBUG_) ? 7 : 8))   // waves per SIMD the register allocation aims at: 88 VGPRs for 4 channels (two channel
                                                 // groups), 60-64 for 1 and 3 (all 2048 workgroups of a 1024 x 1024 frame resident; the debug_thingy
                                                 // variants would spill at that: scratch memory costs far more than a wave)
#endif

namespace {

constexpr int XT = 32, YT = 16;         // tile: 32 x 16 pixels
constexpr int XTHREADS = 256;           // 4 waves: wave w owns the 16 x 8 region at (16 (w & 1), 8 (w >> 1)) of the tile
constexpr int XR = YT + 2;              // staged rows: y0 - 1 .. y0 + 16
constexpr int XPS = 40;                 // plane row stride (floats): index (x - x0) + 1 for x0 - 1 .. x0 + 34; 40: the 8-byte tap reads of
                                        // the 32 lanes of an 8 x 8 block (rows 40 y + 2 q) fall on 64 distinct banks
constexpr int XVS = 36;                 // state tile row stride (float2): index (x - x0) + 2 for x0 - 1 .. x0 + 32, pairs 16-byte aligned
                                        // (36, not the conflict-free 40: the 3-channel kernel then fits eight workgroups per compute unit)
constexpr int XIS = 20;                 // inbox row stride (float2 cells): cell (ty + 1) * 20 + tx + 2 for ty in -1..8, tx in -1..16
constexpr int XICELLS = 10 * XIS;       // ... of a wave's 16 x 8 region and the one-pixel ring around it
constexpr int XRING = 2 * 18 + 2 * 8;   // ring cells: one per lane (52 of 64)
constexpr int PX = 2;                   // pixels per lane

// alias_wrap_fixup (dirt_grad_common.h) with ROLLED loops: quirk Q1 at the right image border -- for the pixels of a pair
// (first column xs, row y) flagged in `which`, the aliased "channels" 1, 2 of 1-channel group c lie in the NEXT image row (past
// the end of the tensor: clamped to its last element), and their dilation axis (:185) is decided again from memory.  Same
// arithmetic; one Scharr stencil at a time, nine loads each, so that this rare path (the last two interior columns of a
// frame) is not the kernel's register high-water mark (inlined and unrolled it was: 106 VGPRs against ~90).
__device__ __forceinline__ uint32_t alias_wrap_fixup_rolled(const float* __restrict__ pixels, int B, int H, int W, int C, int iib, int y, int xs,
                                                         int c, uint32_t which, uint32_t bits)
{
    const size_t last = (size_t)B * H * W - 1;
#pragma unroll 1
    for (int j = 0; j < PX; ++j) {
        if (!((which >> j) & 1u)) continue;
        const size_t centre = ((size_t)iib * H + y) * W + xs + j;   // flat pixel index of the pixel
        float l1x = 0.f, l1y = 0.f;
#pragma unroll 1
        for (int ch = 0; ch < 3; ++ch) {
            // w[r][i] = element (centre + ch + i - 1) of row y - 1 + r in flat order, clamped to the end of the tensor
            float w[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    size_t m = centre + (size_t)(r * W + ch + i) - (size_t)(W + 1);
                    if (m > last) m = last;
                    w[r][i] = pixels[m * C + c];
                }
            const float mm = w[2][0], m0 = w[1][0], mp = w[0][0];
            const float zm = w[2][1], zp = w[0][1];
            const float pm = w[2][2], p0 = w[1][2], pp = w[0][2];
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
        bits = (bits & ~(1u << j)) | ((l1x > l1y) ? (1u << j) : 0u);
    }
    return bits;
}

}  // namespace

// grad_kernel_px2<CSPEC, DEBUG>: the image has CSPEC = 1, 3 or 4 channels (4 = a 3-channel group and a single; 16-byte
// aligned pixel tensors).  DEBUG: also write the reference's diagnostic output debug_thingy.
template <int CSPEC, bool DEBUG>
__global__ __launch_bounds__(XTHREADS, DIRT_PX2_WAVES(CSPEC, DEBUG)) void grad_kernel_px2(GradParams p)
{
    static_assert(CSPEC == 1 || CSPEC == 3 || CSPEC == 4, "channel counts with a two-pixels-per-lane kernel");
    constexpr int NCH = CSPEC, C = CSPEC;
    constexpr int G0 = CSPEC == 1 ? 1 : 3;          // size of the first channel group
    constexpr int NG = 1 + (NCH - G0);              // channel groups
    __shared__ __align__(16) float s_pix[NCH][XR][XPS];          // the channels of `pixels` as planes, edge clamped (at(), :113-124)
    __shared__ __align__(16) float2 s_vw[XR][XVS];               // {clip_w, face} of every pixel of the halo'd tile
    __shared__ __align__(16) float2 s_inbox[XTHREADS / 64][XICELLS];  // per wave: (fx, fy) sent to each pixel of its region + ring

#ifdef DIRT_TRACE
    long long tr_t[12]; int tr_n = 0; long long tr_c[4] = {0, 0, 0, 0};
    const long long tr_wall0 = wall_clock64();
#endif
    XMARK();  // 0 start
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int iib = blockIdx.y;
    const int H = p.H, W = p.W;
    const size_t frame = (size_t)H * W;
    const int tile = xcd_tile((int)blockIdx.x, p.tiles_x * p.tiles_y);
    int tile_col, tile_row;
    tile_xy(tile, p.tiles_x, p.tiles_x_magic, tile_col, tile_row);
    const int x0 = tile_col * XT, y0 = tile_row * YT;

    // Wave-uniform bases at the first staged row of the tile (row0): every per-lane address is a small non-negative 32-bit
    // byte offset, (row - row0) * W + column, times the element size.
    const int row0 = max(y0 - 1, 0);
    const size_t origin = (size_t)iib * frame + (size_t)row0 * W;   // pixel index of (row0, column 0)
    const float2* __restrict__ state_a = p.state_a + origin;         // {clip_w, face}
    const float2* __restrict__ state_b = p.state_b + origin;         // two barycentrics (encode_bary)
    const float* __restrict__ pixels_t = p.pixels + origin * C;
    const float* __restrict__ gpix_t = p.grad_pixels + origin * C;
    float* __restrict__ gbk_t = p.grad_background + origin * C;
    const int32_t* __restrict__ faces = p.faces + (p.shared_faces ? (size_t)0 : (size_t)iib * p.F * 3);
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * p.gv_stride;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * p.gvc_stride;
    const uint32_t gv_row_bytes = 4u * (uint32_t)p.gv_stride, gvc_row_bytes = 4u * (uint32_t)p.gvc_stride;
    constexpr uint32_t pixel_bytes = 4u * (uint32_t)C;
    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;
    const float width_f = (float)W, height_f = (float)H;

    // ---- this lane's pair: DPP row r = lane >> 4 is the 8 x 4 block (r >> 1, r & 1) of the wave's 16 x 8 region -- rows 0, 1
    //      (and 2, 3), which share a face in the loop, are stacked: an 8 x 8 block; inside a row, lane bits 0-1 choose the
    //      pair of the block's four, bits 2-3 the pixel row ----
    const int blk = lane >> 4;
    const int rx = 8 * (blk >> 1) + 2 * (lane & 3);          // in the wave's region: first pixel of the pair (even)
    const int ry = 4 * (blk & 1) + ((lane >> 2) & 3);
    const int wx0 = 16 * (wave & 1), wy0 = 8 * (wave >> 1);   // the region in the tile
    const int lx = wx0 + rx;                                  // in the tile; plane index of the column LEFT of the pair
    const int xs = x0 + lx;                                   // first pixel of the pair
    const int y = y0 + wy0 + ry;                              // tensor row (top row first)
    const int hr = wy0 + ry + 1;                              // its row in the halo'd tile
    // pixel index, relative to (row0, 0), of the pair's first pixel (lanes outside the frame: a valid one)
    const uint32_t own_rel = (uint32_t)((min(y, H - 1) - row0) * W + min(xs, W - 1));
    bool in_px[PX], interior[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        in_px[j] = (xs + j < W) & (y < H);
        interior[j] = in_px[j] & (xs + j > 0) & (y > 0) & (xs + j < W - 1) & (y < H - 1);
    }
    float2* const inbox = &s_inbox[wave][0];
    const int my_cell = (ry + 1) * XIS + rx + 2;   // the pair's first pixel in the inbox (an even cell: 16-byte aligned)

    // ---- staging: the channels of the pixels tile (+ halo), edge clamped.  A thread keeps one column (x0 - 1 + tid % 36) and
    //      takes rows tid / 36, + 7, + 14 of the 18; every load is issued before any use (threads 252 .. 255 idle). ----
    constexpr int NCOLS = XT + 4;                                // x0 - 1 .. x0 + 34: a single channel's aliased "channels" (quirk Q1) are the next two pixels
    constexpr int PROWS = XTHREADS / NCOLS;                      // rows per sweep: 7
    constexpr int PITEMS = (XR + PROWS - 1) / PROWS;             // 3
    constexpr int LC = CSPEC == 3 ? 3 : CSPEC;
    const int st_row = tid / NCOLS, st_ci = tid - st_row * NCOLS;
    const bool st_on = tid < PROWS * NCOLS;
    const uint32_t st_xoff = (uint32_t)min(max(x0 - 1 + st_ci, 0), W - 1) * pixel_bytes;
    const uint32_t row_bytes = (uint32_t)W * pixel_bytes;
    float stage_v[PITEMS][LC];
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) {
        const int cy = min(max(y0 - 1 + st_row + PROWS * k, 0), H - 1);
        const uint32_t off = (uint32_t)(cy - row0) * row_bytes + st_xoff;
        if constexpr (CSPEC == 4) {
            const float4 q = ld_off<float4>(pixels_t, off);
            stage_v[k][0] = q.x; stage_v[k][1] = q.y; stage_v[k][2] = q.z; stage_v[k][3] = q.w;
        } else if constexpr (CSPEC == 3) {
            const Float3 q = ld_off<Float3>(pixels_t, off);
            stage_v[k][0] = q.x; stage_v[k][1] = q.y; stage_v[k][2] = q.z;
        } else {
            stage_v[k][0] = ld_off<float>(pixels_t, off);
        }
    }
    // ---- the visibility "surface" of the tile + 1-pixel halo -- what the backward fragment shader writes
    //      (csrc/shaders.cpp:64-77) over the clear values of csrc/rasterise_grad_egl.cpp:442-445 -- as {clip_w, face}.  Halo
    //      positions outside the frame are clamped; they are only ever consulted for interior pixels. ----
    {
        constexpr int VCOLS = XT + 2;                            // x0 - 1 .. x0 + 32
        constexpr int VROWS = XTHREADS / VCOLS;                  // 7
        constexpr int VITEMS = (XR + VROWS - 1) / VROWS;         // 3
        const int v_row = tid / VCOLS, v_ci = tid - v_row * VCOLS;
        const bool v_on = tid < VROWS * VCOLS;
        const uint32_t v_xoff = (uint32_t)min(max(x0 - 1 + v_ci, 0), W - 1) * 8u;
        float2 rec[VITEMS];
#pragma unroll
        for (int k = 0; k < VITEMS; ++k) {
            const int cy = min(max(y0 - 1 + v_row + VROWS * k, 0), H - 1);
            rec[k] = ld_off<float2>(state_a, (uint32_t)(cy - row0) * ((uint32_t)W * 8u) + v_xoff);
        }
#pragma unroll
        for (int k = 0; k < VITEMS; ++k) {
            const int row = v_row + VROWS * k;
            if (!v_on || row >= XR) continue;
            s_vw[row][v_ci + 1] = rec[k];
        }
    }
    // own barycentrics (two stored, the largest re-derived: decode_bary) and grad_pixels
    float bk[PX][3];
#pragma unroll
    for (int j = 0; j < PX; ++j) decode_bary(ld_off<float2>(state_b, in_px[j] ? (own_rel + (uint32_t)j) * 8u : 0u), bk[j]);
    const uint32_t own_off = own_rel * pixel_bytes;
    float g[PX][NCH];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const uint32_t off = in_px[j] ? own_off + (uint32_t)j * pixel_bytes : 0u;   // outside the frame: any valid address
        if constexpr (CSPEC == 4) {
            const float4 q = ld_off<float4>(gpix_t, off);
            g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z; g[j][3] = q.w;
        } else if constexpr (CSPEC == 3) {
            const Float3 q = ld_off<Float3>(gpix_t, off);
            g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z;
        } else {
            g[j][0] = ld_off<float>(gpix_t, off);
        }
    }
    // the wave's inbox: 200 cells = 100 16-byte pairs
    {
        float4* z = reinterpret_cast<float4*>(inbox);
        z[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane + 64 < XICELLS / 2) z[lane + 64] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    XMARK();  // 1 loads issued, state tile stored
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) {
        const int row = st_row + PROWS * k;
        if (!st_on || row >= XR) continue;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) s_pix[ch][row][st_ci] = stage_v[k][ch];
    }
    XMARK();  // 2 planes stored
    __syncthreads();
    XMARK();  // 3 barrier passed

    // ---- Scharr (:126-127, operation for operation: negative-offset minus positive-offset, offset_y is up = the previous
    //      tensor row) on the lane's pair with the packed fp32 instructions, streamed per channel into what is needed of it:
    //      the direction choice of :185 from the L1 norms (all three "channels" of the reference's Vec3, in its summation
    //      order) and dL/dx, dL/dy of :203-208 ----
    bool horiz[NG][PX];       // [group][j]: the pixel's dilation axis is x
    float2v dLx[NG], dLy[NG];  // dL/dx, dL/dy of :203-208 per group, of the pair
    {
        float l1x[PX], l1y[PX];
        uint32_t tap_base = (uint32_t)((hr - 1) * XPS + lx);          // the pair's first tap in a plane
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int gi = ch < G0 ? 0 : ch - G0 + 1;                 // the channel's group
            const bool single = !(ch < G0 && G0 == 3);                // a 1-channel group: quirk Q1 applies
            const bool last_of_group = single || ch == G0 - 1;
            const bool first_of_group = ch == 0;                      // (of the 3-channel group)
            // taps of row r as pairs: T[r][i] = columns (xs - 1 + 2i, xs + 2i); a 3-channel group needs columns xs-1 .. xs+2
            // (two pairs), a single xs-1 .. xs+4 (its aliased "channels" are the next two pixels)
            // (the tap address is made to depend on the previous channel's result: LDS reads have no side effects, and the
            // compiler otherwise hoists the taps of ALL channels to the top of the phase -- 40 registers of taps alive at
            // once; the other waves of the SIMD cover the latency of a channel's reads)
            if (ch > 0) asm volatile("" : "+v"(tap_base) : "v"(dLx[ch < G0 ? 0 : ch - G0].x), "v"(dLy[ch < G0 ? 0 : ch - G0].y));
            float2v T[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float* rowp = &s_pix[ch][0][0] + tap_base + r * XPS;
                const float2 qa = *reinterpret_cast<const float2*>(rowp), qb = *reinterpret_cast<const float2*>(rowp + 2);
                T[r][0] = float2v{qa.x, qa.y}; T[r][1] = float2v{qb.x, qb.y};
                if (single) {
                    const float2 qc = *reinterpret_cast<const float2*>(rowp + 4);
                    T[r][2] = float2v{qc.x, qc.y};
                } else {
                    T[r][2] = float2v{0.f, 0.f};
                }
            }
            float2v Sx[2], Sy[2];
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                if (P >= 1 && !single) { Sx[P] = float2v{0.f, 0.f}; Sy[P] = float2v{0.f, 0.f}; continue; }
                // at(ox, oy) of pixel q: row 1 - oy, column q + 1 + ox of the taps; pixels q = 2P, 2P + 1
                const float2v mm = T[2][P], m0 = T[1][P], mp = T[0][P];
                const float2v pm = T[2][P + 1], p0 = T[1][P + 1], pp = T[0][P + 1];
                float2v d1 = ((mm + mp) - pm) - pp;
                float2v d2 = m0 - p0;
                float2v m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
                Sx[P] = m1 + m2;
                d1 = ((mm + pm) - mp) - pp;
                // the middle column of each pixel: the high half of one tap pair and the low half of the next
                d2.x = T[2][P].y - T[0][P].y;
                d2.y = T[2][P + 1].x - T[0][P + 1].x;
                m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
                Sy[P] = m1 + m2;
            }
            auto comp = [](const float2v (&v)[2], int q) { return (q & 1) ? v[q >> 1].y : v[q >> 1].x; };
            const float2v gp = float2v{g[0][ch], g[1][ch]};
            if (!single) {
                float2v m = gp * Sx[0];
                dLx[gi] = first_of_group ? m : dLx[gi] + m;
                m = gp * Sy[0];
                dLy[gi] = first_of_group ? m : dLy[gi] + m;
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    l1x[j] = first_of_group ? fabsf(comp(Sx, j)) : l1x[j] + fabsf(comp(Sx, j));
                    l1y[j] = first_of_group ? fabsf(comp(Sy, j)) : l1y[j] + fabsf(comp(Sy, j));
                }
            } else {
                dLx[gi] = gp * Sx[0];
                dLy[gi] = gp * Sy[0];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    // quirk Q1: "channels" 1, 2 of a 1-channel group = elements (pixel + 1, + 2) of the flattened [B,H,W,1]
                    // slice.  Only the L1 norms of interior pixels use them, and for an interior pixel the taps are
                    // unclamped: column + ch, which is staged unless it runs past the end of the image row (the last two
                    // interior columns of the frame are corrected below: alias_wrap_fixup)
                    const float a0x = fabsf(comp(Sx, j)), a0y = fabsf(comp(Sy, j));
                    l1x[j] = q1_intended ? a0x : (a0x + fabsf(comp(Sx, j + 1))) + fabsf(comp(Sx, j + 2));
                    l1y[j] = q1_intended ? a0y : (a0y + fabsf(comp(Sy, j + 1))) + fabsf(comp(Sy, j + 2));
                }
            }
            if (last_of_group) {
#pragma unroll
                for (int j = 0; j < PX; ++j) horiz[gi][j] = l1x[j] > l1y[j];  // :185
                if (single && !q1_intended && x0 + XT + 3 > W) {  // workgroup-uniform: only tiles on the right image border
                    uint32_t ib = 0;
#pragma unroll
                    for (int j = 0; j < PX; ++j) ib |= (interior[j] && xs + j + 3 > W - 1) ? (1u << j) : 0u;
                    if (__builtin_amdgcn_ballot_w64(ib != 0u) != 0ull) {
                        uint32_t bits = 0;
#pragma unroll
                        for (int j = 0; j < PX; ++j) bits |= horiz[gi][j] ? (1u << j) : 0u;
                        bits = alias_wrap_fixup_rolled(p.pixels, p.B, H, W, C, iib, y, xs, ch, ib, bits);
#pragma unroll
                        for (int j = 0; j < PX; ++j) horiz[gi][j] = ((bits >> j) & 1u) != 0u;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // one channel's taps at a time
        }
    }
    XMARK();  // 4 Scharr done

    // ---- the pair and its six neighbours: clip_w and face ----
    float w_own[PX], w_up[PX], w_dn[PX], w_l, w_r;
    int f_own[PX], f_up[PX], f_dn[PX], f_l, f_r;
    {
        const float2* rowp = &s_vw[hr][lx + 2];
        const float4 a = *reinterpret_cast<const float4*>(rowp);
        w_own[0] = a.x; f_own[0] = __float_as_int(a.y); w_own[1] = a.z; f_own[1] = __float_as_int(a.w);
        const float2 l = rowp[-1], r = rowp[2];
        w_l = l.x; f_l = __float_as_int(l.y); w_r = r.x; f_r = __float_as_int(r.y);
        const float4 c = *reinterpret_cast<const float4*>(rowp - XVS);
        w_up[0] = c.x; f_up[0] = __float_as_int(c.y); w_up[1] = c.z; f_up[1] = __float_as_int(c.w);
        const float4 e = *reinterpret_cast<const float4*>(rowp + XVS);
        w_dn[0] = e.x; f_dn[0] = __float_as_int(e.y); w_dn[1] = e.z; f_dn[1] = __float_as_int(e.w);
    }
    bool covered[PX];
    int key[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        covered[j] = in_px[j] & (f_own[j] >= 0);
        key[j] = covered[j] ? f_own[j] : -1;
    }

    // ---- dilation (:155-194) and position factors (:196-232), as in dirt_grad.hip: a pixel takes the fragment of the
    //      neighbour at +d, else at -d, when that neighbour is another face (:86-89) and closer (:165); d is +-x or +-y by the
    //      L1 norms (:185), the first attempt by the parity dither (:186-191).  The gradients of vertex k are b_k * (fx, fy, fw)
    //      with fx = dL_dx * (W/2) / w, fy = dL_dy * (H/2) / w, fw = -(fx * ndc_x + fy * ndc_y), everything taken at the pixel
    //      whose fragment is used; (fx, fy) are summed per such TARGET pixel -- own pixels in registers, neighbours through
    //      the wave's inbox (ds_add_f32 from the few dilated lanes) -- and fw is formed once per pixel. ----
    float2v fxy[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) fxy[j] = float2v{0.f, 0.f};
    const float NO_NEIGHBOUR = __builtin_nanf("");   // (a value no clip_w that passed `wo > w` can have: a qualifying neighbour with clip_w == +-0 still counts, as in :165)
    const bool pos0 = ((xs + y) & 1) == 0;   // pixel 0 tries +x / up first (:186-191)
    const float2v half_size = float2v{.5f * width_f, .5f * height_f};
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const float wl = j == 0 ? w_l : w_own[0], wr = j == 1 ? w_r : w_own[1];
        const int fl = j == 0 ? f_l : f_own[0], fr = j == 1 ? f_r : f_own[1];
        // (the pixel's own face as the state tile has it: an uncovered pixel, -1, differs from any face)
        const float wo = interior[j] ? w_own[j] : -INFINITY;   // pixels on the frame's border are never dilated (:155)
        const float qL = ((fl != f_own[j]) & (wo > wl)) ? wl : NO_NEIGHBOUR;
        const float qR = ((fr != f_own[j]) & (wo > wr)) ? wr : NO_NEIGHBOUR;
        const float qU = ((f_up[j] != f_own[j]) & (wo > w_up[j])) ? w_up[j] : NO_NEIGHBOUR;
        const float qD = ((f_dn[j] != f_own[j]) & (wo > w_dn[j])) ? w_dn[j] : NO_NEIGHBOUR;
        const bool pos = (j & 1) ? !pos0 : pos0;   // first attempt towards +x / up (:191), else -x / down
        const float qx1 = pos ? qR : qL, qx2 = pos ? qL : qR, qy1 = pos ? qU : qD, qy2 = pos ? qD : qU;
        const float rcp_own = __builtin_amdgcn_rcpf(w_own[j]);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            // direction: x if L1(Sx) > L1(Sy) else y (:185), negated on odd (x + y) (:186-190).  The reference's offsets are
            // in GL buffer orientation (y up): tensor row = y - offset_y.
            const bool hz = horiz[gi][j];
            const float q1 = hz ? qx1 : qy1, q2 = hz ? qx2 : qy2;
            const bool first = q1 == q1;                  // the first attempt found its neighbour (not the NaN sentinel)
            const bool dilated = first | (q2 == q2);      // ... or the opposite one did (:192-193)
            if constexpr (DEBUG) {
                if (gi == 0 && in_px[j]) write_debug(p.debug_thingy, p.grad_pixels, p.B, H, W, C, iib, y, xs + j, G0, dilated);
            }
            const float dLx_j = (j & 1) ? dLx[gi].y : dLx[gi].x, dLy_j = (j & 1) ? dLy[gi].y : dLy[gi].x;
            const float2v t = float2v{dLx_j, dLy_j} * half_size;
            const float2v f = t * float2v{rcp_own, rcp_own};
            const bool own = covered[j] & !dilated;       // contributes to its own pixel
            fxy[j] += float2v{own ? f.x : 0.f, own ? f.y : 0.f};
            if (dilated) {  // few lanes: ds_add_f32 into the neighbour's cell, with the NEIGHBOUR's clip_w
                const float rcp_w = __builtin_amdgcn_rcpf(first ? q1 : q2);
                const float2v fn = t * float2v{rcp_w, rcp_w};
                const int step = hz ? 1 : -XIS;           // +x, or up = the previous row
                const bool fwd = first == pos;            // the neighbour taken lies at +x / up
                float* cell = reinterpret_cast<float*>(inbox + (my_cell + j + (fwd ? step : -step)));
                atomicAdd(cell, fn.x);
                atomicAdd(cell + 1, fn.y);
            }
        }
    }
    XMARK();  // 5 dilation done

    // ---- position totals of the pair's pixels (own sums + what the neighbours sent through the inbox, and fw of the totals)
    //      and the ring: what this wave's pixels sent to pixels of other waves.  Those pixels' faces take it through the face
    //      loop, one ring cell per lane: cells 0-17 the row above, 18-35 the row below, 36-43 / 44-51 the columns left / right. ----
    float2v fpos_xy[PX];
    float fpos_w[PX];
    int lkey = -1;
    float lb[3] = {0.f, 0.f, 0.f}, lf[3] = {0.f, 0.f, 0.f};
    {
        const float4 i01 = *reinterpret_cast<const float4*>(inbox + my_cell);
        fpos_xy[0] = fxy[0] + float2v{i01.x, i01.y}; fpos_xy[1] = fxy[1] + float2v{i01.z, i01.w};
        const float ndc_y_own = ndc_of(H - 1 - y, H, p.inv_h);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const float ndc_x = ndc_of(xs + j, W, p.inv_w);
            fpos_w[j] = -(fpos_xy[j].x * ndc_x + fpos_xy[j].y * ndc_y_own);
        }
        const int r = lane;
        if (r < XRING) {
            const bool top = r < 18, bottom = r >= 18 && r < 36, left = r >= 36 && r < 44;
            const int ty = top ? -1 : (bottom ? 8 : (left ? r - 36 : r - 44));
            const int tx = top ? r - 1 : (bottom ? r - 19 : (left ? -1 : 16));
            const float2 v = inbox[(ty + 1) * XIS + tx + 2];
            if (v.x != 0.f || v.y != 0.f) {  // only pixels inside the frame are ever sent anything
                const int py = y0 + wy0 + ty, px = x0 + wx0 + tx;
                lkey = __float_as_int(s_vw[wy0 + ty + 1][wx0 + tx + 2].y);
                const float2 nb = ld_off<float2>(state_b, (uint32_t)((py - row0) * W + px) * 8u);
                decode_bary(nb, lb);
                const float ndc_x = ndc_of(px, W, p.inv_w);
                const float ndc_y = ndc_of(H - 1 - py, H, p.inv_h);
                lf[0] = v.x; lf[1] = v.y; lf[2] = -(v.x * ndc_x + v.y * ndc_y);
                if (!__builtin_isfinite((v.x + v.y) + ((lb[0] + lb[1]) + lb[2]))) {   // (see the face loop: non-finite factors)
                    const uint32_t fo = (uint32_t)lkey * 12u;
                    const int32_t vk[3] = {ld_off<int32_t>(faces, fo), ld_off<int32_t>(faces, fo + 4u), ld_off<int32_t>(faces, fo + 8u)};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float* row = reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertices) + (size_t)((uint32_t)vk[k] * gv_row_bytes));
                        atomicAdd(row + 0, lb[k] * lf[0]); atomicAdd(row + 1, lb[k] * lf[1]); atomicAdd(row + 3, lb[k] * lf[2]);
                    }
                    lkey = -1;
                    lb[0] = 0.f; lb[1] = 0.f; lb[2] = 0.f; lf[0] = 0.f; lf[1] = 0.f; lf[2] = 0.f;
                }
            }
        }
        XCOUNT(0, __popcll(__builtin_amdgcn_ballot_w64(lkey >= 0)));
    }
    XMARK();  // 6 face loop starts

    // ---- the face loop (dirt_grad.hip): the two rows of an 8 x 8 block walk the distinct faces among their pixels (key[j],
    //      -1 = none) and among the ring cells their lanes hold (lkey), both blocks of the wave at once.  Per face every lane
    //      forms its masked partial sums -- per vertex k the S values b_k * (g_0 .. g_NCH-1, fx, fy, fw) (S = 3 + NCH rounded
    //      up to even; order below), as S / 2 packed pairs: one v_pk_fma_f32 per pair and pixel -- the 3 S sums are reduced
    //      over the lanes of each row (row_reduce_scatter), the two rows' totals joined (v_permlane16_swap) and ONE atomic
    //      instruction adds them to the face's three vertices. ----
    constexpr int S = (3 + NCH + 1) & ~1;       // values per vertex (padded to whole pairs)
    constexpr int HP = S / 2;                   // ... as pairs
    constexpr int NV = 3 * S;                   // values per face
    constexpr int NR = NV <= 16 ? 16 : 24;      // ... padded to what the row reduction takes
    static_assert(NV <= NR, "");
    // Order of a vertex's values: the colours first, then the position factors with (fx, fy) as one aligned pair:
    // NCH even: g.., fx, fy, fw, 0;  odd: g.., fw, fx, fy.
    constexpr int IW = (NCH & 1) ? NCH : NCH + 2, IX = (NCH & 1) ? NCH + 1 : NCH, IY = IX + 1;
    static_assert((IX & 1) == 0 && IY < S && IW < S, "");
    // this lane's role: it adds the pair-of-rows total of value rv of the block's face (row_value_of_lane): vertex rv / S,
    // component c = rv % S: c < NCH: colour c; IX, IY, IW: (x, y, w) of grad_vertices.  The two rows end up with the same
    // totals, so the even row sends d0's value and the odd row d1's: one atomic instruction per iteration.
    int rv0, rv1;
    row_value_of_lane<NR>(lane & 15, rv0, rv1);
    const bool odd_row = (blk & 1) != 0;
    const int rv = odd_row ? rv1 : rv0;
    const int role_c = rv >= 0 ? rv % S : S;
    const int role_k = rv >= 0 && rv < NV ? rv / S : 0;
    const bool role_pos = role_c == IX || role_c == IY || role_c == IW;
    const bool role_valid = rv >= 0 && rv < NV && (role_c < NCH || role_pos);
    float* const role_base = role_pos ? grad_vertices + (role_c == IW ? 3 : role_c - IX) : grad_vertex_colors + (role_c < NCH ? role_c : 0);
    const uint32_t role_stride = role_pos ? gv_row_bytes : gvc_row_bytes;
    // the factors of a pixel, in pairs
    float2v fp[PX][HP];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        float f[S];
#pragma unroll
        for (int c = 0; c < S; ++c) f[c] = c < NCH ? g[j][c < NCH ? c : 0] : 0.f;
        f[IW] = fpos_w[j];
#pragma unroll
        for (int h = 0; h < HP; ++h) { fp[j][h].x = f[2 * h]; fp[j][h].y = f[2 * h + 1]; }
        fp[j][IX / 2] = fpos_xy[j];
    }
    // pending faces: the keys of this lane's pixels / ring cell not yet added (NONE: none or done; "no face" is -1 = NONE)
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t pend[PX + 1];
    // ---- non-finite factors (a NaN / Inf in grad_pixels, in `pixels` through the Scharr filter, a degenerate clip_w).  The
    //      loop multiplies every pixel's factors by a barycentric that is ZEROED where the pixel is not of the block's face:
    //      0 * NaN would carry one pixel's NaN into every face of its 8 x 8 block, where the reference adds a pixel's terms to
    //      the vertices of its own face only (:140,228-230).  Such a pixel adds its 3 (NCH + 3) products itself -- the
    //      reference's own atomics, term for term -- and leaves the loop: factors zeroed, face struck off. ----
    bool gbk_done[PX];   // grad_background of the pixel was written here (a non-finite uncovered pixel: its factors are zeroed for the loop)
    auto store_gbk = [&](int j) {
        const uint32_t off = own_off + (uint32_t)j * pixel_bytes;
        auto gval = [&](int c) { return (c & 1) ? fp[j][c / 2].y : fp[j][c / 2].x; };
        if constexpr (CSPEC == 4) {
            st_off<float4>(gbk_t, off, covered[j] ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(gval(0), gval(1), gval(2), gval(3)));
        } else if constexpr (CSPEC == 3) {
            st_off<Float3>(gbk_t, off, covered[j] ? Float3{0.f, 0.f, 0.f} : Float3{gval(0), gval(1), gval(2)});
        } else {
            st_off<float>(gbk_t, off, covered[j] ? 0.f : gval(0));
        }
    };
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        float2v t = fp[j][0];
#pragma unroll
        for (int h = 1; h < HP; ++h) t += fp[j][h];
        const float u = (t.x + t.y) + ((bk[j][0] + bk[j][1]) + bk[j][2]);   // non-finite iff a factor is, or the sum overflows
        const bool bad = !__builtin_isfinite(u);
        gbk_done[j] = false;
        pend[j] = bad ? NONE : (uint32_t)key[j];
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {   // wave-uniform: not taken on finite data
            if (bad) {
                if (key[j] != -1) {
                    const uint32_t fo = (uint32_t)key[j] * 12u;
                    const int32_t vk[3] = {ld_off<int32_t>(faces, fo), ld_off<int32_t>(faces, fo + 4u), ld_off<int32_t>(faces, fo + 8u)};
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int c = 0; c < S; ++c) {
                            if (!(c < NCH || c == IX || c == IY || c == IW)) continue;
                            const float val = bk[j][k] * ((c & 1) ? fp[j][c / 2].y : fp[j][c / 2].x);
                            float* dstp = c >= NCH
                                ? reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertices) + (size_t)((uint32_t)vk[k] * gv_row_bytes)) + (c == IW ? 3 : c - IX)
                                : reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertex_colors) + (size_t)((uint32_t)vk[k] * gvc_row_bytes)) + c;
                            atomicAdd(dstp, val);
                        }
                } else if (in_px[j]) {   // uncovered: its colour factors are what grad_background gets -- stored now, before they are zeroed
                    store_gbk(j);
                    gbk_done[j] = true;
                }
#pragma unroll
                for (int h = 0; h < HP; ++h) fp[j][h] = float2v{0.f, 0.f};
            }
        }
    }
    pend[PX] = (uint32_t)lkey;
    // the block's next face: the smallest pending key of its 32 lanes (an all-lanes minimum by four DPP rotations and one
    // swap with the other row of the block)
    auto next_face = [&]() {
        uint32_t K = min(min(pend[0], pend[1]), pend[2]);
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x128 /* row_ror:8 */, 0xF, 0xF, true));
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x124 /* row_ror:4 */, 0xF, 0xF, true));
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x122 /* row_ror:2 */, 0xF, 0xF, true));
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x121 /* row_ror:1 */, 0xF, 0xF, true));
        const auto sw = __builtin_amdgcn_permlane16_swap(K, K, false, false);
        return min(sw[0], sw[1]);
    };
    // (the loop is rotated: the next face is chosen as soon as this one's pixels are struck off the pending list, so that its
    // chain of cross-lane minima runs alongside the reduction's chain of cross-lane adds)
    uint32_t K = next_face();
    for (;;) {
        const lanemask live = __builtin_amdgcn_ballot_w64(K != NONE);   // blocks that still have a face
        if (live == 0ull) break;
        // the vertex this lane adds to (requested now, needed after the reduction)
        const uint32_t fbase = (K != NONE ? K : 0u) * 12u;
        const int vsel = ld_off<int32_t>(faces, fbase + 4u * (uint32_t)role_k);
        float2v accp[NR / 2];
#pragma unroll
        for (int i = NV / 2; i < NR / 2; ++i) accp[i] = float2v{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const bool m = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64(pend[j] == K) & live);
            pend[j] = m ? NONE : pend[j];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float bm = m ? bk[j][k] : 0.f;
#pragma unroll
                for (int h = 0; h < HP; ++h)
                    accp[k * HP + h] = j == 0 ? pk_mul_scalar(bm, fp[j][h]) : pk_fma_scalar(bm, fp[j][h], accp[k * HP + h]);
            }
        }
        {
            const lanemask mm = __builtin_amdgcn_ballot_w64(pend[PX] == K) & live;
            if (mm != 0ull) {
                const bool m = __builtin_amdgcn_inverse_ballot_w64(mm);
                pend[PX] = m ? NONE : pend[PX];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float bm = m ? lb[k] : 0.f;
                    accp[k * HP + IX / 2] = pk_fma_scalar(bm, float2v{lf[0], lf[1]}, accp[k * HP + IX / 2]);
                    if (IW & 1) accp[k * HP + IW / 2].y = fmaf(bm, lf[2], accp[k * HP + IW / 2].y);
                    else accp[k * HP + IW / 2].x = fmaf(bm, lf[2], accp[k * HP + IW / 2].x);
                }
            }
        }
        XCOUNT(1, 1);
        const uint32_t K_next = next_face();
        float acc[NR];
#pragma unroll
        for (int i = 0; i < NR / 2; ++i) { acc[2 * i] = accp[i].x; acc[2 * i + 1] = accp[i].y; }
        float d0, d1;
        row_reduce_scatter<NR>(acc, lane, d0, d1);
        // the two rows of a block worked on the same face: their totals, added (both rows get the sum)
        const auto s0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);
        d0 = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);
        if (NR >= 24) {
            const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d1), __float_as_uint(d1), false, false);
            d1 = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
        }
        // (a block without a face this iteration has all-zero totals)
        const float total = odd_row ? d1 : d0;
        // The address is formed BEFORE the branch on purpose: the wait for the vertex index then sits on every path (inside
        // the branch the load stays pending on the path around it and the compiler answers with s_waitcnt vmcnt(0) in the
        // loop header, where it also waits for the previous iteration's atomic: dirt_grad.hip).
        float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(role_base) + (size_t)((uint32_t)vsel * role_stride));
        asm volatile("" : "+v"(dst));
        if (role_valid && total != 0.f)
            asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(dst), "v"(total) : "memory");
        K = K_next;
    }
    XMARK();  // 7 loop done

    // ---- background gradient (:143-147): grad_pixels where nothing is covered, zero elsewhere.  After the face loop (the
    //      stores of a wave then spread over the time in which the waves finish) and from the registers the loop's colour
    //      factors live in: four lanes write the 128 bytes of eight 4-channel pixels. ----
#pragma unroll
    for (int j = 0; j < PX; ++j)
        if (in_px[j] && !gbk_done[j]) store_gbk(j);
    XMARK();  // 8 done
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_grad_px2) {
        long long* o = g_trace_grad_px2 + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16;
        for (int i = 0; i < 12; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
        o[12] = tr_c[0]; o[13] = tr_c[1];
        o[14] = tr_wall0; o[15] = (((long long)wall_clock64() - tr_wall0) << 20) | (long long)(__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4 /* HW_REG_HW_ID */) & 0xFFFFF);
    }
#endif
}

hipError_t launch_grad_px2(const GradParams& p, hipStream_t stream)
{
    GradParams q = p;
    q.tiles_x = (p.W + XT - 1) / XT;
    q.tiles_y = (p.H + YT - 1) / YT;
    q.tiles_x_magic = tile_magic(q.tiles_x);
    const dim3 grid((unsigned)(q.tiles_x * q.tiles_y), (unsigned)p.B), block(XTHREADS);
#define DIRT_LAUNCH_PX2(C_)                                                                          \
    do {                                                                                             \
        if (p.debug_thingy) hipLaunchKernelGGL((grad_kernel_px2<C_, true>), grid, block, 0, stream, q);   \
        else hipLaunchKernelGGL((grad_kernel_px2<C_, false>), grid, block, 0, stream, q);                 \
    } while (0)
    if (p.C == 4) DIRT_LAUNCH_PX2(4);
    else if (p.C == 3) DIRT_LAUNCH_PX2(3);
    else DIRT_LAUNCH_PX2(1);
#undef DIRT_LAUNCH_PX2
    return hipGetLastError();
}

}  // namespace dirt
