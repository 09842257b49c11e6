// dirt_grad.hip -- gradient assembly kernel for gfx950.
//
// Replaces assemble_grads / launch_grad_assembly (csrc/rasterise_grad_egl.cu:93-278) and, by
// evaluating every channel group of dirt/rasterise_ops.py:145-165 inside one launch, the N
// per-group RasteriseGrad ops (and N GL re-draws) the reference issues for C not in {1,3}.
//
// Input: the per-pixel state the raster kernel leaves behind -- two float2 planes, {clip_w, face} and two of the three
// barycentrics (encode_bary, dirt_device.h), the
// counterpart of the reference's two RGBA32F surfaces (csrc/rasterise_grad_egl.cpp:432-456), produced by the forward
// pass itself when it keeps its state -- plus `faces` for the vertex indices.
//
// The reference issues up to 3C+9 global float atomics per covered pixel (:140,228-230) and reads its 3x3
// neighbourhood with 27 scalar loads.  Here:
//   * a 256-thread workgroup stages one 32 x 32 tile (+ halo) in LDS -- the pass's channels of `pixels` as planes and
//     {clip_w, face} of every pixel -- then its four waves work independently, with no further barrier: a wave owns
//     32 x 8 pixels, a lane a 4 x 1 strip, so that Scharr taps, neighbour tests and address arithmetic are shared by four
//     pixels;
//   * the face index stands for the reference's index triple (:86-89): two faces with the same ordered triple have the
//     same set-up record, hence the same coverage and depth at every sample, and the lower index wins every tie -- at
//     most one of them is ever visible, so among visible faces index and triple correspond one to one;
//   * every position gradient is b_k(t) * (fx, fy, fw(t)) for some TARGET pixel t -- the pixel itself, or the neighbour
//     it was dilated from -- with fw linear in (fx, fy).  The (fx, fy) are summed per target first: own pixels in
//     registers, neighbours through a per-wave inbox in LDS (ds_add_f32 from the few dilated lanes), whose one-pixel
//     ring collects what belongs to pixels of other waves; then fw is formed once per pixel;
//   * nothing else is accumulated in memory.  The wave walks the distinct faces it has seen: for each, every lane forms
//     its masked partial sums (3 vertices x {x, y, w, colours}), the <= 21 sums are reduced across the wave with a
//     transposing butterfly (~80 instructions: 21 totals land in 21 different lanes), and ONE global_atomic_add_f32
//     instruction adds them to the face's three vertices.  No LDS accumulators, no hash table, no fixed point.
// Variable names in the per-pixel arithmetic follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "dirt_reduce.h"
#include "dirt_grad_common.h"
#include "../../include/dirt_hip.h"
#include <type_traits>
#include <cstdlib>

namespace dirt {

// Preprocessor switches: none is defined for the product library (dirt_amd/build.py).  They exist so that the A/B and
// knock-out figures of profiles/EXPERIMENTS.md can be reproduced (tools/variants.sh builds a library per flag set,
// tools/ab.sh times it):  DIRT_TRACE (per-wave phase timestamps, tools/trace_grad.py);  DIRT_GRAD_NO_ATOMICS,
// DIRT_KO_LOADS / _PIX / _G / _GBK / _LOOP (knock-outs: wrong results by design);  DIRT_GBK_COALESCED=0, DIRT_NO_WIDE6,
// DIRT_NO_TWO3 (the previous form of a change that was kept: same results, slower).
#ifndef DIRT_GBK_COALESCED
#define DIRT_GBK_COALESCED 1
#endif
#ifndef DIRT_ALIAS_INBOX
#define DIRT_ALIAS_INBOX 1
#endif
#ifndef DIRT_TWO3_WAVES
#define DIRT_TWO3_WAVES (DIRT_ALIAS_INBOX ? 4 : 3)   // waves per SIMD the {3,3} many-channel shape is register-allocated for
#endif
#ifdef DIRT_TRACE
// Per-wave phase timestamps (s_memtime) for tools/trace_grad.py; compiled only into the tracing build of the library.
__device__ long long* g_trace_grad = nullptr;
extern "C" void dirt_debug_set_trace_grad(void* p)
{
    long long* q = reinterpret_cast<long long*>(p);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_grad), &q, sizeof(q));
}
#define GMARK() do { if (tr_n < 12) { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tr_t[tr_n++] = t_; } } while (0)
#define GCOUNT(i, v) do { tr_c[i] += (v); } while (0)
#else
#define GMARK() do {} while (0)
#define GCOUNT(i, v) do {} while (0)
#endif


constexpr int GT = 32;                  // tile side (pixels)
constexpr int GTHREADS = 256;           // 4 waves: wave w owns rows 8w .. 8w+7; a DPP row of 16 lanes owns an 8 x 8 block of them
                                        // (block l >> 4), a lane a 4 x 1 strip of it
constexpr int PR = GT + 2;              // staged rows: y0-1 .. y0+32
constexpr int PS = 36;                  // plane row stride (floats): column (x - x0) + 1 for x0-1 .. x0+34, so that a strip's taps start
                                        // (with the column left of it) at a 16-byte boundary
constexpr int VS = 36;                  // state tile row stride (float2): column (x - x0) + 2, so that strips are 16-byte aligned
constexpr int IS = 34;                  // inbox row stride (float2 cells): cell (ty + 1) * 34 + tx + 2 for ty in -1..8, tx in -1..32
constexpr int ICELLS = 10 * IS + 4;     // ... of a wave's 32 x 8 region and the one-pixel ring around it (a multiple of 2: 16-byte cells pairs)
constexpr int RING = 2 * 34 + 2 * 8;    // ring cells: what the wave's pixels sent to pixels of other waves

// grad_kernel<CSPEC, STRIDED>: a workgroup works on CSPEC = 1, 3, 4 or 6 channels (4 = a 3-channel group and a single;
// 6 = two 3-channel groups, STRIDED only: three workgroups per compute unit, but every 64-byte pixel of a many-channel
// image is fetched by half as many passes) of one tile.  Not STRIDED: that is the image's channel count, a compile-time constant (4: with 16-byte aligned pixel
// tensors).  STRIDED: any channel count, cut into PASSES of whole channel groups (dirt/rasterise_ops.py:148-152): the
// launch covers p.npasses passes of this shape, starting at channel p.c_first, and a workgroup takes one (tile, pass)
// over `pixels` with a runtime channel stride.  The passes of a tile are consecutive work items of one XCD (xcd_tile),
// so they run at about the same time next to the same L2: a pass uses 12 of the 64 bytes of a 16-channel pixel, and
// what one pass brings in from HBM the others find there.  grad_vertices is summed over the groups by the atomics
// (dirt/rasterise_ops.py:167-171).
// DEBUG: also write the reference's diagnostic output debug_thingy.
// ROWS: every DPP row (8 x 8 pixels) of a wave walks its own faces -- four faces per face-loop iteration, ~5 iterations
// where the pairs of rows need ~6.4, but ~1.6 x the float atomics: launched where the memory system has room for them
// (small frames); otherwise the two rows of a pair (16 x 8 pixels) work on one face.
// AI: the per-wave inbox and the ring cells' factors ALIASED onto the pixel planes (always for the {3,3} shape, CSPEC = 6): one more
// workgroup barrier, 11 KB of LDS and twelve registers less -- one workgroup more per compute unit, which pays when the grid
// takes several rounds (launch_grad).
template <int CSPEC, bool STRIDED, bool DEBUG, bool ROWS = false, bool AI = false>
__global__ __launch_bounds__(GTHREADS, CSPEC == 6 ? DIRT_TWO3_WAVES : (AI ? 5 : 4)) void grad_kernel(GradParams p)
{
    static_assert(CSPEC == 1 || CSPEC == 3 || CSPEC == 4 || (CSPEC == 6 && STRIDED), "pass shapes");
    constexpr int NPLANES = CSPEC;
    constexpr int PC = CSPEC == 6 ? 6 : 4;   // channels a staging item holds
    __shared__ __align__(16) float s_pix[NPLANES][PR][PS];  // the pass's channels of `pixels`, edge clamped
    __shared__ __align__(16) float2 s_vw[PR][VS];           // {clip_w, face} of every pixel of the halo'd tile
    // per wave: (fx, fy) sent to each pixel of its region + ring.  The {3,3} shape (six planes) ALIASES it -- and the ring cells'
    // factors, which the face loop reads only in a rare branch -- onto the planes, which are dead once every wave has its Scharr
    // responses (a workgroup barrier there): 50 -> 39 KB of LDS and twelve registers less, so that FOUR workgroups share a
    // compute unit instead of three for the many-channel frames, whose gradient is bound by memory requests in flight
    static_assert(!AI || (CSPEC == 4 && !STRIDED && !DEBUG && !ROWS), "the aliased form exists for the plain 4-channel kernel (and the {3,3} shape)");
    constexpr bool ALIAS_INBOX = DIRT_ALIAS_INBOX && (CSPEC == 6 || AI);
    __shared__ __align__(16) float2 s_inbox_own[ALIAS_INBOX ? 1 : GTHREADS / 64][ALIAS_INBOX ? 1 : ICELLS];
    // (the ring cells' factors are parked in the wave's OWN inbox once gather_positions has read it: cell 0 of the 64 lanes,
    // [component: b0 b1 b2 fx fy fw][lane], then cell 1 of lanes 0 .. 19: 504 floats of the inbox's 688)
    constexpr int RING_E1 = 6 * 64, RING_E1_LANES = RING - 64;
    static_assert(RING_E1 + 6 * RING_E1_LANES <= 2 * ICELLS, "the ring factors fit in the inbox");
    static_assert(!ALIAS_INBOX || sizeof(float2) * (GTHREADS / 64) * ICELLS <= sizeof(float) * NPLANES * PR * PS, "the inboxes fit in the planes");

#ifdef DIRT_TRACE
    long long tr_t[12]; int tr_n = 0; long long tr_c[4] = {0, 0, 0, 0};
    const long long tr_wall0 = wall_clock64();
#endif
    GMARK();  // 0 start
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int iib = blockIdx.y;
    const int H = p.H, W = p.W, C = STRIDED ? p.C : CSPEC;
    const size_t frame = (size_t)H * W;
    // this workgroup's tile and pass: CSPEC channels starting at channel cbase
    // (STRIDED, CSPEC = 4: the launch's passes are 3-channel groups, and only the LAST of them also carries the single that
    // follows the triples -- single_on, wave-uniform.  All passes of an image then go out in one launch and meet in the
    // L2: a pass launched on its own fetches every 64-byte pixel for 4-16 bytes of it, K5: 245 us instead of ~100)
    int tile, cbase = 0, pass_i = 0;
    bool single_on = true;
    int second_mode = 0;   // CSPEC = 6: the pass's second group is a triple (0), a single (1: the last pass of C % 3 != 0) or absent (2)
    if constexpr (!STRIDED) {
        tile = xcd_tile((int)blockIdx.x, p.tiles_x * p.tiles_y);
    } else {
        const int item = xcd_tile((int)blockIdx.x, p.tiles_x * p.tiles_y * p.npasses);
        tile = item / p.npasses;
        pass_i = item - tile * p.npasses;
        cbase = p.c_first + pass_i * (CSPEC == 1 ? 1 : (CSPEC == 6 ? 6 : 3));
        single_on = CSPEC != 4 || pass_i == p.npasses - 1;
        if (CSPEC == 6 && pass_i == p.npasses - 1) second_mode = p.last_second;
    }
    const bool aligned16 = (!STRIDED && CSPEC == 4) ? true : (p.pixels_aligned16 != 0 && (cbase & 3) == 0);
    // ({3,3} passes of frames whose pixels are 16-byte aligned, C % 4 == 0: a pass starts at channel 6 i, i.e. at byte 0 or 8
    // of a 16-byte unit, and a last pass with a single (C % 3 == 1) at a multiple of 4 channels -- a lone last triple does not occur)
#ifdef DIRT_NO_WIDE6
    const bool wide6 = false;
#else
    const bool wide6 = CSPEC == 6 && p.pixels_aligned16 != 0 && (C & 3) == 0;
#endif
    int tile_col, tile_row;
    tile_xy(tile, p.tiles_x, p.tiles_x_magic, tile_col, tile_row);
    const int x0 = tile_col * GT, y0 = tile_row * GT;

    // Wave-uniform bases at the first staged row of the tile (row0), so that every per-lane address is a small
    // non-negative 32-bit byte offset: (row - row0) * W + column, times the element size.
    const int row0 = max(y0 - 1, 0);
    const size_t origin = (size_t)iib * frame + (size_t)row0 * W;   // pixel index of (row0, column 0)
    const float2* __restrict__ state_a = p.state_a + origin;         // {clip_w, face}
    const float2* __restrict__ state_b = p.state_b + origin;         // {b0, b1}
    const float* __restrict__ pixels_t = p.pixels + origin * C + cbase;   // channel 0 of the pass
    const float* __restrict__ gpix_t = p.grad_pixels + origin * C + cbase;
    float* __restrict__ gbk_t = p.grad_background + origin * C + cbase;
    const int32_t* __restrict__ faces = p.faces + (p.shared_faces ? (size_t)0 : (size_t)iib * p.F * 3);
    // (rows of p.gv_stride / p.gvc_stride floats: 4 and C for dense tensors; both 8, the colours 16 bytes behind the
    // positions, for the state's interleaved accumulators, where a vertex's seven values share one 32-byte row)
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * p.gv_stride;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * p.gvc_stride + cbase;
    const uint32_t gv_row_bytes = 4u * (uint32_t)p.gv_stride, gvc_row_bytes = 4u * (uint32_t)p.gvc_stride;
    const uint32_t pixel_bytes = 4u * (uint32_t)C;

    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;
    const float width_f = (float)W, height_f = (float)H;

    // ---- this lane's strip ----
    const int blk = lane >> 4;                  // the lane's DPP row = its 8 x 8 block of the wave's region
    // (which lane of the row takes which strip is free; this assignment -- strip: lane bit 2; row: bits 3, 0, 1 -- is the one
    // whose 16-byte LDS reads of the planes, the state tile and the inbox collide least: 8 / 16 / 16 LDS cycles per wave
    // instruction against 12 / 32 / 24 for the plain order, by the lane groups of the guide's LDS table)
    const int sx = 2 * blk + ((lane >> 2) & 1), ry = ((lane >> 3) & 1) | ((lane & 1) << 1) | (((lane >> 1) & 1) << 2);
    const int xs = x0 + 4 * sx;                 // first pixel of the strip
    const int y = y0 + 8 * wave + ry;           // tensor row (top row first)
    const int hr = 8 * wave + ry + 1;           // its row in the halo'd tile
    // pixel index, relative to (row0, 0), of the strip's first pixel (lanes outside the frame: a valid one)
    const uint32_t own_rel = (uint32_t)((min(y, H - 1) - row0) * W + min(xs, W - 1));
    bool in_px[4], interior[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        in_px[j] = (xs + j < W) & (y < H);
        interior[j] = in_px[j] & (xs + j > 0) & (y > 0) & (xs + j < W - 1) & (y < H - 1);
    }
    float2* const inbox = ALIAS_INBOX ? reinterpret_cast<float2*>(&s_pix[0][0][0]) + wave * ICELLS : &s_inbox_own[ALIAS_INBOX ? 0 : wave][0];
    float* const ringstore = reinterpret_cast<float*>(inbox) + lane;
    auto ring_at = [&](int e, int c) -> float& { return ringstore[e == 0 ? c * 64 : RING_E1 + c * RING_E1_LANES]; };
    const int my_cell = (ry + 1) * IS + 4 * sx + 2;   // the strip's first pixel in the inbox

    // ---- staging: loads of the pass's channels of the pixels tile (+halo), edge clamped (at(), :113-124).  A thread
    //      keeps one column (x0 - 1 + tid % 36, plane column tid % 36) and takes rows tid / 36, + 7, + 14, ... of the 34: the
    //      column clamp and the LDS address are computed once, an item costs a row clamp and one multiply-add.  Five items per thread,
    //      every load issued before any use (threads 252 .. 255 idle). ----
    constexpr int PROWS = GTHREADS / PS;                         // rows per sweep: 7
    constexpr int PITEMS = (PR + PROWS - 1) / PROWS;             // 5
    const int st_row = tid / PS, st_ci = tid - st_row * PS;      // (as unsigned small numbers: a multiply-high, once)
    const bool st_on = tid < PROWS * PS;
    const uint32_t st_xoff = (uint32_t)min(max(x0 - 1 + st_ci, 0), W - 1) * pixel_bytes;
    const uint32_t row_bytes = (uint32_t)W * pixel_bytes;
    auto stage_load = [&](int nch, float (&v)[PITEMS][PC]) {
#pragma unroll
        for (int k = 0; k < PITEMS; ++k) {
            const int cy = min(max(y0 - 1 + st_row + PROWS * k, 0), H - 1);
#if defined(DIRT_KO_LOADS) || defined(DIRT_KO_PIX)
            const uint32_t off = 0u * ((uint32_t)(cy - row0) * row_bytes + st_xoff);
#else
            const uint32_t off = (uint32_t)(cy - row0) * row_bytes + st_xoff;
#endif
            if (nch == 4 && (C & 3) == 0 && aligned16) {
                const float4 q = ld_off<float4>(pixels_t, off);
                v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = q.w;
            } else if (nch == 3) {
                const Float3 q = ld_off<Float3>(pixels_t, off);
                v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = 0.f;
            } else if (CSPEC == 6 && wide6) {   // 16-byte aligned pixels: the pass's 24 (16) bytes as 16 + 8 (8 + 16) byte accesses
                if ((cbase & 3) == 0) {
                    const float4 q = ld_off<float4>(pixels_t, off);
                    v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = q.w; v[k][PC - 2] = 0.f; v[k][PC - 1] = 0.f;
                    if (nch == 6) { const float2 r = ld_off<float2>(pixels_t, off + 16u); v[k][PC - 2] = r.x; v[k][PC - 1] = r.y; }
                } else {
                    const float2 q = ld_off<float2>(pixels_t, off);
                    const float4 r = ld_off<float4>(pixels_t, off + 8u);
                    v[k][0] = q.x; v[k][1] = q.y; v[k][2] = r.x; v[k][3] = r.y; v[k][PC - 2] = r.z; v[k][PC - 1] = r.w;
                }
            } else if (CSPEC == 6) {   // two triples / a triple and a single / a triple
                const Float3 q = ld_off<Float3>(pixels_t, off);
                v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = 0.f; v[k][PC - 2] = 0.f; v[k][PC - 1] = 0.f;
                if (nch == 6) { const Float3 r = ld_off<Float3>(pixels_t, off + 12u); v[k][3] = r.x; v[k][PC - 2] = r.y; v[k][PC - 1] = r.z; }
                else if (nch == 4) v[k][3] = ld_off<float>(pixels_t, off + 12u);
            } else {
#pragma unroll
                for (int ch = 0; ch < PC; ++ch) v[k][ch] = ch < nch ? ld_off<float>(pixels_t, off + 4u * ch) : 0.f;
            }
        }
    };
    auto stage_store = [&](int nch, const float (&v)[PITEMS][PC]) {
        const int col = st_ci;
#pragma unroll
        for (int k = 0; k < PITEMS; ++k) {
            const int row = st_row + PROWS * k;
            if (!st_on || row >= PR) continue;
#pragma unroll
            for (int ch = 0; ch < NPLANES; ++ch)
                if (ch < nch) s_pix[ch][row][col] = v[k][ch];
        }
    };
    auto zero_inbox = [&]() {  // 344 cells = 172 pairs
        float4* z = reinterpret_cast<float4*>(inbox);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (lane + 64 * i < ICELLS / 2) z[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // ---- phase A: the visibility "surface" of the tile + 1-pixel halo -- what the backward fragment shader writes
    //      (csrc/shaders.cpp:64-77) over the clear values of csrc/rasterise_grad_egl.cpp:442-445 -- as {clip_w, face}.
    //      Halo positions outside the frame are clamped; they are only ever consulted for interior pixels, whose
    //      neighbours are inside the frame. ----
    float stage_v[PITEMS][PC];
    const int nch_live = CSPEC == 6 ? (second_mode == 0 ? 6 : (second_mode == 1 ? 4 : 3)) : (single_on ? CSPEC : 3);   // channels the pass reads
    stage_load(nch_live, stage_v);
    {
        // (the same sweep: column x0 - 1 + tid % 34, rows tid / 34, + 7, ...; threads 238 .. 255 idle)
        constexpr int VITEMS = (PR + PROWS - 1) / PROWS;
        const int v_row = tid / PR, v_ci = tid - v_row * PR;
        const bool v_on = tid < PROWS * PR;
        const uint32_t v_xoff = (uint32_t)min(max(x0 - 1 + v_ci, 0), W - 1) * 8u;
        float2 rec[VITEMS];
#pragma unroll
        for (int k = 0; k < VITEMS; ++k) {
            const int cy = min(max(y0 - 1 + v_row + PROWS * k, 0), H - 1);
            rec[k] = ld_off<float2>(state_a, (uint32_t)(cy - row0) * ((uint32_t)W * 8u) + v_xoff);
        }
#pragma unroll
        for (int k = 0; k < VITEMS; ++k) {
            const int row = v_row + PROWS * k;
            if (!v_on || row >= PR) continue;
            s_vw[row][v_ci + 1] = rec[k];
        }
    }
    // own barycentrics (two stored, the largest re-derived: decode_bary)
    float bk[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) decode_bary(ld_off<float2>(state_b, in_px[j] ? (own_rel + (uint32_t)j) * 8u : 0u), bk[j]);
    if (!ALIAS_INBOX) zero_inbox();

    // (fx, fy) sent to this strip's own pixels by themselves (see "position factors" below)
    using std::integral_constant;
    float2v fxy[4];   // (as pairs: what the face loop multiplies is then already in place)
#pragma unroll
    for (int j = 0; j < 4; ++j) fxy[j] = float2v{0.f, 0.f};

    // ---- the face loop.  Every DPP row of the wave (16 lanes = an 8 x 8 pixel block) walks the distinct faces among its
    //      pixels (key[j], -1 = none) and among the ring cells its lanes hold (lkey), all four rows at once: an iteration
    //      takes one face per row.  Per face every lane forms its masked partial sums -- per vertex k the S values
    //      b_k * (g_0 .. g_NCHV-1, fx, fy, fw) (S = 3 + NCHV rounded up to even; order below), as S / 2 packed pairs: one
    //      v_pk_fma_f32 per pair and pixel -- the 3 S sums are reduced over the lanes of the row (row_reduce_scatter,
    //      dirt_reduce.h: the totals of the row's face land in different lanes of the row) and at most two atomic
    //      instructions add the four faces' totals to their vertices.  A block sees ~3 faces where the 16 x 8 half
    //      regions of the two-group version saw ~6. ----
    auto face_loop = [&](auto nchv_tag, const auto& g, const int (&key)[4], const bool (&covered)[4],
                         const float2v (&fpos_xy)[4], const float (&fpos_w)[4], const int (&lkey)[2], const float (&lb)[2][3], const float (&lf)[2][3]) {
        constexpr int NCHV = decltype(nchv_tag)::value;
        // FWS (the {3,3} shape with the aliased inbox): an even channel count leaves the w factor alone in its pair -- g.., fx, fy,
        // fw, 0 -- and the padding costs a register per pixel and per vertex sum.  There fw travels as a SCALAR next to the pairs
        // (one v_fma_f32 instead of one v_pk_fma_f32 per pixel and vertex: the same instruction count) and its three sums
        // follow the vertices' blocks in the reduction's input: 9 registers less in a loop that has to fit 128.
        constexpr bool FWS = ALIAS_INBOX && NCHV == 6;
        constexpr int S = FWS ? NCHV + 2 : (3 + NCHV + 1) & ~1;      // values per vertex held in pairs (padded to whole pairs)
        constexpr int HP = S / 2;                   // ... as pairs
        constexpr int NV = FWS ? 3 * S + 3 : 3 * S; // values per face
        constexpr int NR = NV <= 16 ? 16 : (NV <= 24 ? 24 : 32);   // ... padded to what the row reduction takes
        static_assert(NV <= NR, "");
        // Order of a vertex's values: the colours first (they arrive as whole registers of the grad_pixels loads), then
        // the position factors with (fx, fy) as one aligned pair: NCHV even: g.., fx, fy, fw, 0;  odd: g.., fw, fx, fy
        // (FWS: g.., fx, fy per vertex, then fw of the three vertices: IW = S marks "not in the pairs").
        constexpr int IW = FWS ? S : ((NCHV & 1) ? NCHV : NCHV + 2), IX = (NCHV & 1) ? NCHV + 1 : NCHV, IY = IX + 1;
        static_assert((IX & 1) == 0 && IY < S && (FWS || IW < S), "");
        // this lane's roles: it adds the row totals of values rv[0], rv[1] of the row's face (row_value_of_lane): vertex
        // rv / S, component c = rv % S: c < NCHV: colour c; IX, IY, IW: (x, y, w) of grad_vertices
        // Pairs of rows: the two rows end up with the same totals -- see the end of an iteration -- so the even row sends
        // d0's value and the odd row d1's: one atomic instruction per iteration (role 0 only).  ROWS: every lane sends both.
        int rv[2];
        row_value_of_lane<NR>(lane & 15, rv[0], rv[1]);
        const bool odd_row = (blk & 1) != 0;
        if (!ROWS) { rv[0] = odd_row ? rv[1] : rv[0]; rv[1] = -1; }
        constexpr int NROLES = ROWS && NR >= 24 ? 2 : 1;
        int role_k[NROLES];
        bool role_valid[NROLES];
        float* role_base[NROLES];
        uint32_t role_stride[NROLES];
#pragma unroll
        for (int e = 0; e < NROLES; ++e) {
            const int v = rv[e];
            const bool w_tail = FWS && v >= 3 * S && v < NV;           // (FWS: the three fw sums behind the vertices' blocks)
            const int c = w_tail ? IW : (v >= 0 && v < 3 * S ? v % S : S + 1);
            role_k[e] = w_tail ? v - 3 * S : (v >= 0 && v < 3 * S ? v / S : 0);
            const bool is_pos = c == IX || c == IY || c == IW;
            role_valid[e] = v >= 0 && v < NV && (c < NCHV || is_pos);
            role_base[e] = is_pos ? grad_vertices + (c == IW ? 3 : c - IX) : grad_vertex_colors + (c < NCHV ? c : 0);
            role_stride[e] = is_pos ? gv_row_bytes : gvc_row_bytes;
        }
        // the factors of a pixel, in pairs
        float2v fp[4][HP];
        float fpw[4];   // (FWS: the w factor of a pixel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float f[S + 1];
#pragma unroll
            for (int c = 0; c < S; ++c) f[c] = c < NCHV ? g[j][c < NCHV ? c : 0] : 0.f;
            f[IW] = fpos_w[j];
            fpw[j] = FWS ? fpos_w[j] : 0.f;
#pragma unroll
            for (int h = 0; h < HP; ++h) { fp[j][h].x = f[2 * h]; fp[j][h].y = f[2 * h + 1]; }
            fp[j][IX / 2] = fpos_xy[j];
        }
        // pending faces: the keys of this lane's pixels / ring cells not yet added (NONE: none or done; "no face" is -1 = NONE)
        constexpr uint32_t NONE = 0xFFFFFFFFu;
        uint32_t pend[6];
        // ---- non-finite factors (a NaN / Inf in grad_pixels, in `pixels` through the Scharr filter, a degenerate clip_w).
        //      The loop below multiplies every pixel's factors by a barycentric that is ZEROED where the pixel is not of the
        //      row's face: 0 * NaN would carry one pixel's NaN into every face of its 16 x 8 half region, where the reference
        //      adds a pixel's terms to the vertices of its own face only (:140,228-230).  Such a pixel (rare; a sum of
        //      finite factors that overflows is treated alike) adds its 3 (NCHV + 3) products itself -- the reference's own
        //      atomics, term for term -- and leaves the loop: factors zeroed, face struck off. ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float2v t = fp[j][0];
#pragma unroll
            for (int h = 1; h < HP; ++h) t += fp[j][h];
            if (FWS) t.x += fpw[j];
            const float u = (t.x + t.y) + ((bk[j][0] + bk[j][1]) + bk[j][2]);   // non-finite iff a factor is, or the sum overflows
            const bool bad = !__builtin_isfinite(u);
            pend[j] = bad ? NONE : (uint32_t)key[j];
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {   // wave-uniform: not taken on finite data
                if (bad) {
                    if (key[j] != -1) {
                        const uint32_t fo = (uint32_t)key[j] * 12u;
                        const int32_t vk[3] = {ld_off<int32_t>(faces, fo), ld_off<int32_t>(faces, fo + 4u), ld_off<int32_t>(faces, fo + 8u)};
#pragma unroll
                        for (int k = 0; k < 3; ++k)
#pragma unroll
                            for (int c = 0; c < S + (FWS ? 1 : 0); ++c) {
                                if (!(c < NCHV || c == IX || c == IY || c == IW) || (STRIDED && NCHV == 4 && !single_on && c == 3) || (NCHV == 6 && c < NCHV && c >= nch_live)) continue;
                                const float val = bk[j][k] * ((FWS && c == IW) ? fpw[j] : ((c & 1) ? fp[j][(c < S ? c : 0) / 2].y : fp[j][(c < S ? c : 0) / 2].x));
                                float* dstp = c >= NCHV && (c == IX || c == IY || c == IW)
                                    ? reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertices) + (size_t)((uint32_t)vk[k] * gv_row_bytes)) + (c == IW ? 3 : c - IX)
                                    : reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertex_colors) + (size_t)((uint32_t)vk[k] * gvc_row_bytes)) + c;
                                atomicAdd(dstp, val);
                            }
                    }
#pragma unroll
                    for (int h = 0; h < HP; ++h) fp[j][h] = float2v{0.f, 0.f};
                    fpw[j] = 0.f;
                }
            }
        }
#ifdef DIRT_KO_RING   // (knock-out build, timing only: ring cells' contributions dropped -- what the loop costs without their registers)
        pend[4] = NONE; pend[5] = NONE;
#else
        pend[4] = (uint32_t)lkey[0]; pend[5] = (uint32_t)lkey[1];
#endif
        // the row's next face: the smallest pending key of its 16 lanes (an all-lanes minimum by four DPP rotations)
        auto next_face = [&]() {
            uint32_t K = min(min(min(pend[0], pend[1]), min(pend[2], pend[3])), min(pend[4], pend[5]));
            K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x128 /* row_ror:8 */, 0xF, 0xF, true));
            K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x124 /* row_ror:4 */, 0xF, 0xF, true));
            K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x122 /* row_ror:2 */, 0xF, 0xF, true));
            K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x121 /* row_ror:1 */, 0xF, 0xF, true));
            if (!ROWS) {   // ... and of the other row of its pair: the left (rows 0, 1) and the right (2, 3) 16 x 8 pixels of the region
                const auto sw = __builtin_amdgcn_permlane16_swap(K, K, false, false);
                K = min(sw[0], sw[1]);
            }
            return K;
        };
        // (the loop is rotated: the next face is chosen as soon as this one's pixels are struck off the pending list, so
        // that its chain of cross-lane minima runs alongside the reduction's chain of cross-lane adds)
        uint32_t K = next_face();
        for (;;) {
            const lanemask live = __builtin_amdgcn_ballot_w64(K != NONE);   // rows that still have a face
            if (live == 0ull) break;
            // the vertices this lane adds to (requested now, needed after the reduction)
            // the vertices this lane adds to (requested now, needed after the reduction)
            const uint32_t fbase = (K != NONE ? K : 0u) * 12u;
            int vsel[NROLES];
#pragma unroll
            for (int e = 0; e < NROLES; ++e) vsel[e] = ld_off<int32_t>(faces, fbase + 4u * (uint32_t)role_k[e]);
            float2v accp[NR / 2];
            float accw[3] = {0.f, 0.f, 0.f};   // (FWS: the w sums of the three vertices)
#pragma unroll
            for (int i = (FWS ? 3 * HP : NV / 2); i < NR / 2; ++i) accp[i] = float2v{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool m = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64(pend[j] == K) & live);
                pend[j] = m ? NONE : pend[j];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float bm = m ? bk[j][k] : 0.f;
#pragma unroll
                    for (int h = 0; h < HP; ++h)
                        accp[k * HP + h] = j == 0 ? pk_mul_scalar(bm, fp[j][h]) : pk_fma_scalar(bm, fp[j][h], accp[k * HP + h]);
                    if (FWS) accw[k] = j == 0 ? bm * fpw[j] : fmaf(bm, fpw[j], accw[k]);
                }
            }
#ifndef DIRT_KO_RING
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const lanemask mm = __builtin_amdgcn_ballot_w64(pend[4 + e] == K) & live;
                if (mm != 0ull) {
                    const bool m = __builtin_amdgcn_inverse_ballot_w64(mm);
                    pend[4 + e] = m ? NONE : pend[4 + e];
                    float rb[3], rf[3];
                    if constexpr (ALIAS_INBOX) {   // (the cell's factors were parked in LDS by gather_positions: read here, where a cell's face comes up)
#pragma unroll
                        for (int c = 0; c < 3; ++c) { rb[c] = m ? ring_at(e, c) : 0.f; rf[c] = m ? ring_at(e, 3 + c) : 0.f; }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) { rb[c] = lb[e][c]; rf[c] = lf[e][c]; }
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float bm = m ? rb[k] : 0.f;
                        accp[k * HP + IX / 2] = pk_fma_scalar(bm, float2v{rf[0], rf[1]}, accp[k * HP + IX / 2]);
                        if (FWS) accw[k] = fmaf(bm, rf[2], accw[k]);
                        else if (IW & 1) accp[k * HP + (FWS ? 0 : IW / 2)].y = fmaf(bm, rf[2], accp[k * HP + (FWS ? 0 : IW / 2)].y);
                        else accp[k * HP + (FWS ? 0 : IW / 2)].x = fmaf(bm, rf[2], accp[k * HP + (FWS ? 0 : IW / 2)].x);
                    }
                }
            }
#endif
            GCOUNT(1, 1);
            const uint32_t K_next = next_face();
            float acc[NR];
#pragma unroll
            for (int i = 0; i < NR / 2; ++i) { acc[2 * i] = accp[i].x; acc[2 * i + 1] = accp[i].y; }
            if (FWS) {   // the three w sums behind the vertices' blocks (values 3 S .. 3 S + 2)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[3 * S + k] = accw[k];
            }
            float d0, d1;
            row_reduce_scatter<NR>(acc, lane, d0, d1);
            if (!ROWS) {   // the two rows of a pair worked on the same face: their totals, added (both rows get the sum)
                const auto s0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);
                d0 = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);
                if (NR >= 24) {
                    const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d1), __float_as_uint(d1), false, false);
                    d1 = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
                }
            }
            // (a row / pair without a face this iteration has all-zero totals)
            float total[NROLES];
            total[0] = ROWS ? d0 : (odd_row ? d1 : d0);
            if (NROLES == 2) total[NROLES - 1] = d1;
            // The addresses are formed BEFORE the branches on purpose: the wait for the vertex indices then sits on every
            // path.  Inside a branch it would leave the load pending on the path around it, and the compiler answers that
            // with s_waitcnt vmcnt(0) in the loop header -- where it also waits, every iteration, for the previous
            // iteration's atomic to be acknowledged by the memory system (+3 us at K3, +11 us at K3-256).
            float* dst[NROLES];
#pragma unroll
            for (int e = 0; e < NROLES; ++e) {
                dst[e] = reinterpret_cast<float*>(reinterpret_cast<char*>(role_base[e]) + (size_t)((uint32_t)vsel[e] * role_stride[e]));
                asm volatile("" : "+v"(dst[e]));
            }
#pragma unroll
            for (int e = 0; e < NROLES; ++e) {
#ifdef DIRT_GRAD_NO_ATOMICS   // measurement builds only (what the atomics cost: profiles/README.md); never defined for the product library
                if (role_valid[e] && total[e] == 1.2345e-30f && vsel[e] == -12345)   // (never true; keeps the operands alive)
#else
                if (role_valid[e] && total[e] != 0.f)
#endif
                    // (written as a GLOBAL atomic: behind the asm barrier above the compiler no longer knows the pointer's
                    // address space and emits flat_atomic_add_f32, which is issued to the LDS and the memory pipeline alike
                    // and counts on both wait counters)
                    asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(dst[e]), "v"(total[e]) : "memory");
            }
            K = K_next;
        }
    };

    // ---- position totals of the strip's pixels (own sums + what the neighbours sent through the inbox, and fw of the
    //      totals) and the ring: what this wave's pixels sent to pixels of other waves (the row above / below the region,
    //      the column left / right of the tile).  Those pixels' faces take it through the face loop, at most two ring
    //      cells per lane: cells 0-33 the row above, 34-67 the row below, 68-75 / 76-83 the columns left / right. ----
    auto gather_positions = [&](float2v (&fpos_xy)[4], float (&fpos_w)[4], int (&lkey)[2], float (&lb)[2][3], float (&lf)[2][3]) {
        const float4 i01 = *reinterpret_cast<const float4*>(inbox + my_cell), i23 = *reinterpret_cast<const float4*>(inbox + my_cell + 2);
        fpos_xy[0] = fxy[0] + float2v{i01.x, i01.y}; fpos_xy[1] = fxy[1] + float2v{i01.z, i01.w};
        fpos_xy[2] = fxy[2] + float2v{i23.x, i23.y}; fpos_xy[3] = fxy[3] + float2v{i23.z, i23.w};
        const float ndc_y_own = ndc_of(H - 1 - y, H, p.inv_h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ndc_x = ndc_of(xs + j, W, p.inv_w);
            fpos_w[j] = -(fpos_xy[j].x * ndc_x + fpos_xy[j].y * ndc_y_own);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = lane + 64 * e;
            const bool top = r < 34, bottom = r >= 34 && r < 68, left = r >= 68 && r < 76;
            const int ty = top ? -1 : (bottom ? 8 : (left ? r - 68 : r - 76));
            const int tx = top ? r - 1 : (bottom ? r - 35 : (left ? -1 : 32));
            lkey[e] = -1;
            lb[e][0] = 0.f; lb[e][1] = 0.f; lb[e][2] = 0.f; lf[e][0] = 0.f; lf[e][1] = 0.f; lf[e][2] = 0.f;
            if (r < RING) {
                const float2 v = inbox[(ty + 1) * IS + tx + 2];
                if (v.x != 0.f || v.y != 0.f) {  // only pixels inside the frame are ever sent anything
                    const int py = y0 + 8 * wave + ty, px = x0 + tx;
                    lkey[e] = __float_as_int(s_vw[8 * wave + ty + 1][tx + 2].y);
                    const float2 nb = ld_off<float2>(state_b, (uint32_t)((py - row0) * W + px) * 8u);
                    decode_bary(nb, lb[e]);
                    const float ndc_x = ndc_of(px, W, p.inv_w);
                    const float ndc_y = ndc_of(H - 1 - py, H, p.inv_h);
                    lf[e][0] = v.x; lf[e][1] = v.y; lf[e][2] = -(v.x * ndc_x + v.y * ndc_y);
                    if (!__builtin_isfinite((v.x + v.y) + ((lb[e][0] + lb[e][1]) + lb[e][2]))) {   // (see the face loop: non-finite factors)
                        const uint32_t fo = (uint32_t)lkey[e] * 12u;
                        const int32_t vk[3] = {ld_off<int32_t>(faces, fo), ld_off<int32_t>(faces, fo + 4u), ld_off<int32_t>(faces, fo + 8u)};
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            float* row = reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertices) + (size_t)((uint32_t)vk[k] * gv_row_bytes));
                            atomicAdd(row + 0, lb[e][k] * lf[e][0]); atomicAdd(row + 1, lb[e][k] * lf[e][1]); atomicAdd(row + 3, lb[e][k] * lf[e][2]);
                        }
                        lkey[e] = -1;
                        lb[e][0] = 0.f; lb[e][1] = 0.f; lb[e][2] = 0.f; lf[e][0] = 0.f; lf[e][1] = 0.f; lf[e][2] = 0.f;
                    }
                }
            }
        }
        if constexpr (ALIAS_INBOX) {   // park the cells' factors -- after BOTH cells of every lane have been read: the store is the inbox
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (lkey[e] >= 0) {   // (cell 1: lanes 0 .. RING - 65 only)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ring_at(e, c) = lb[e][c]; ring_at(e, 3 + c) = lf[e][c]; }
                }
        }
        GCOUNT(0, __popcll(__builtin_amdgcn_ballot_w64(lkey[0] >= 0)) + __popcll(__builtin_amdgcn_ballot_w64(lkey[1] >= 0)));
    };

    // One pass = whole channel groups (dirt/rasterise_ops.py:148-152 packs groups of 3 while >= 3 channels remain, then
    // singles) of one of three shapes -- {3}, {3,1}, {1} -- and the body is instantiated for each, so that its loops and
    // branches over channels and groups are static: NCH channels, the first group of size G0, a further one a single.
    auto run_pass = [&](auto nch_tag, auto g0_tag) {
        constexpr int NCH = decltype(nch_tag)::value;
        constexpr int G0 = decltype(g0_tag)::value;
        constexpr bool TWO3 = NCH == 6;             // two 3-channel groups; otherwise a first group of G0 and singles
        constexpr int NG = TWO3 ? 2 : 1 + (NCH - G0);   // channel groups in the pass

        // this strip's grad_pixels
        const uint32_t own_off = own_rel * pixel_bytes;
        float g[4][NCH];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#if defined(DIRT_KO_LOADS) || defined(DIRT_KO_G)
            const uint32_t off = 0u;
#else
            const uint32_t off = in_px[j] ? own_off + (uint32_t)j * pixel_bytes : 0u;  // outside the frame: any valid address
#endif
            bool wide = false;
            if constexpr (NCH == 4) {
                if (single_on && (C & 3) == 0 && aligned16) {
                    const float4 q = ld_off<float4>(gpix_t, off);
                    g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z; g[j][3] = q.w;
                    wide = true;
                } else if (!single_on) {
                    const Float3 q = ld_off<Float3>(gpix_t, off);
                    g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z; g[j][3] = 0.f;   // (a zero fourth channel adds nothing anywhere)
                    wide = true;
                }
            }
            if constexpr (NCH == 3) {
                const Float3 q = ld_off<Float3>(gpix_t, off);
                g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z;
                wide = true;
            }
            if constexpr (NCH == 6) {   // (a channel the pass does not have: zero, adds nothing anywhere)
              if (wide6) {
                if ((cbase & 3) == 0) {
                    const float4 q = ld_off<float4>(gpix_t, off);
                    g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z; g[j][3] = q.w; g[j][4] = 0.f; g[j][5] = 0.f;
                    if (second_mode == 0) { const float2 r = ld_off<float2>(gpix_t, off + 16u); g[j][4] = r.x; g[j][5] = r.y; }
                } else {
                    const float2 q = ld_off<float2>(gpix_t, off);
                    const float4 r = ld_off<float4>(gpix_t, off + 8u);
                    g[j][0] = q.x; g[j][1] = q.y; g[j][2] = r.x; g[j][3] = r.y; g[j][4] = r.z; g[j][5] = r.w;
                }
                wide = true;
              } else {
                const Float3 q = ld_off<Float3>(gpix_t, off);
                g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z; g[j][3] = 0.f; g[j][4] = 0.f; g[j][5] = 0.f;
                if (second_mode == 0) { const Float3 r = ld_off<Float3>(gpix_t, off + 12u); g[j][3] = r.x; g[j][4] = r.y; g[j][5] = r.z; }
                else if (second_mode == 1) g[j][3] = ld_off<float>(gpix_t, off + 12u);
                wide = true;
              }
            }
            if (!wide) {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) g[j][ch] = ld_off<float>(gpix_t, off + 4u * ch);
            }
        }
        GMARK();  // 1 loads issued, state tile stored
        stage_store(nch_live, stage_v);
        GMARK();  // 2 planes stored
        __syncthreads();
        GMARK();  // 3 barrier passed

        // ---- Scharr (:126-127, operation for operation: negative-offset minus positive-offset, offset_y is up = the
        //      previous tensor row), streamed per channel into what is needed of it: the direction choice of :185 from the
        //      L1 norms (all three "channels" of the reference's Vec3, in its summation order) and dL/dx, dL/dy of
        //      :203-208 ----
        // Per-lane predicates are kept as wave-wide lane masks in scalar registers from here on: what combines them is
        // then scalar work (s_and / s_or), and the vector unit -- what bounds this kernel -- only compares and selects.
        lanemask horiz_m[NG][4];  // [group][j]: the pixel's dilation axis is x
        // dL/dx, dL/dy of :203-208 per group, as PAIRS of adjacent pixels (pair P = pixels 2P, 2P + 1 of the strip): the
        // Scharr arithmetic below runs on such pairs with the packed fp32 instructions, two pixels per instruction
        float2v dLx[NG][2], dLy[NG][2];
        {
            float l1x[4], l1y[4];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int gi = TWO3 ? ch / 3 : (ch < G0 ? 0 : ch - G0 + 1);   // the channel's group
                // ({3,3}: the second group of the launch's last pass may be a single or absent -- second_mode, wave-uniform)
                const bool single = TWO3 ? (ch >= 3 && second_mode == 1) : !(ch < G0 && G0 == 3);     // a 1-channel group: quirk Q1 applies
                const bool last_of_group = TWO3 ? (single || ch % 3 == 2) : (single || ch == G0 - 1);
                const bool first_of_group = TWO3 ? ch % 3 == 0 : ch == 0;     // (of a 3-channel group)
                if ((STRIDED && NCH == 4 && ch == 3 && !single_on) || (TWO3 && ch >= 3 && (second_mode == 2 || (second_mode == 1 && ch > 3)))) {   // a pass without the single / the second group: nothing of group 1 is used
                    if (TWO3 && ch > 3 && second_mode == 1) continue;   // (the single was channel 3)
#pragma unroll
                    for (int j = 0; j < 4; ++j) horiz_m[gi][j] = 0ull;
                    dLx[gi][0] = dLx[gi][1] = dLy[gi][0] = dLy[gi][1] = float2v{0.f, 0.f};
                    continue;
                }
                // taps of row r as pairs: T[r][i] = columns (xs - 1 + 2i, xs + 2i); a 3-channel group needs columns
                // xs-1 .. xs+4 (three pairs), a single xs-1 .. xs+6 (its aliased "channels" are the next two pixels)
                constexpr int NT = 4;
                float2v T[3][NT];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float* rowp = &s_pix[ch][hr - 1 + r][4 * sx];
                    const float4 qa = *reinterpret_cast<const float4*>(rowp);
                    T[r][0] = float2v{qa.x, qa.y}; T[r][1] = float2v{qa.z, qa.w};
                    if (single) {
                        const float4 qb = *reinterpret_cast<const float4*>(rowp + 4);
                        T[r][2] = float2v{qb.x, qb.y}; T[r][3] = float2v{qb.z, qb.w};
                    } else {
                        const float2 qb = *reinterpret_cast<const float2*>(rowp + 4);
                        T[r][2] = float2v{qb.x, qb.y}; T[r][3] = float2v{0.f, 0.f};
                    }
                }
                constexpr int NP_MAX = 3;
                float2v Sx[NP_MAX], Sy[NP_MAX];
#pragma unroll
                for (int P = 0; P < NP_MAX; ++P) {
                    if (P >= 2 && !single) { Sx[P] = float2v{0.f, 0.f}; Sy[P] = float2v{0.f, 0.f}; continue; }
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
                auto comp = [](const float2v (&v)[NP_MAX], int q) { return (q & 1) ? v[q >> 1].y : v[q >> 1].x; };
                if (!single) {
#pragma unroll
                    for (int P = 0; P < 2; ++P) {
                        const float2v gp = float2v{g[2 * P][ch], g[2 * P + 1][ch]};
                        float2v m = gp * Sx[P];
                        dLx[gi][P] = first_of_group ? m : dLx[gi][P] + m;
                        m = gp * Sy[P];
                        dLy[gi][P] = first_of_group ? m : dLy[gi][P] + m;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        l1x[j] = first_of_group ? fabsf(comp(Sx, j)) : l1x[j] + fabsf(comp(Sx, j));
                        l1y[j] = first_of_group ? fabsf(comp(Sy, j)) : l1y[j] + fabsf(comp(Sy, j));
                    }
                } else {
#pragma unroll
                    for (int P = 0; P < 2; ++P) {
                        const float2v gp = float2v{g[2 * P][ch], g[2 * P + 1][ch]};
                        dLx[gi][P] = gp * Sx[P];
                        dLy[gi][P] = gp * Sy[P];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // quirk Q1: "channels" 1,2 of a 1-channel group = elements (pixel + 1, + 2) of the flattened
                        // [B,H,W,1] slice.  Only the L1 norms of interior pixels use them, and for an interior pixel the
                        // taps are unclamped: column + ch, which is staged unless it runs past the end of the image row
                        // (the last two interior columns of the frame are corrected below: alias_wrap_fixup)
                        const float a0x = fabsf(comp(Sx, j)), a0y = fabsf(comp(Sy, j));
                        l1x[j] = q1_intended ? a0x : (a0x + fabsf(comp(Sx, j + 1))) + fabsf(comp(Sx, j + 2));
                        l1y[j] = q1_intended ? a0y : (a0y + fabsf(comp(Sy, j + 1))) + fabsf(comp(Sy, j + 2));
                    }
                }
                if (last_of_group) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) horiz_m[gi][j] = __builtin_amdgcn_ballot_w64(l1x[j] > l1y[j]);  // :185
                    if (single && !q1_intended && x0 + GT + 3 > W) {  // wave-uniform: only tiles on the right image border
                        uint32_t ib = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) ib |= (interior[j] && xs + j + 3 > W - 1) ? (1u << j) : 0u;
                        if (__builtin_amdgcn_ballot_w64(ib != 0u) != 0ull) {
                            uint32_t bits = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) bits |= __builtin_amdgcn_inverse_ballot_w64(horiz_m[gi][j]) ? (1u << j) : 0u;
                            bits = alias_wrap_fixup(p.pixels, p.B, H, W, C, iib, y, xs, cbase + ch, ib, bits, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) horiz_m[gi][j] = __builtin_amdgcn_ballot_w64(((bits >> j) & 1u) != 0u);
                        }
                    }
                }
#ifdef DIRT_SCHARR_PAIRS   // (A/B build: two channels' taps in flight)
                if ((ch & 1) || ch == NCH - 1) __builtin_amdgcn_sched_barrier(0);
#elif defined(DIRT_SCHARR_FREE)
#else
                __builtin_amdgcn_sched_barrier(0);  // one channel's taps at a time
#endif
            }
        }
        GMARK();  // 4 Scharr done
        if (ALIAS_INBOX) {   // every wave is done with the planes: they become the inboxes
            __syncthreads();
            zero_inbox();
        }

        // ---- the strip and its eight neighbours: clip_w and face ----
        float w_own[4], w_up[4], w_dn[4], w_l, w_r;
        int f_own[4], f_up[4], f_dn[4], f_l, f_r;
        {
            const float2* rowp = &s_vw[hr][4 * sx + 2];
            const float4 a = *reinterpret_cast<const float4*>(rowp), b = *reinterpret_cast<const float4*>(rowp + 2);
            w_own[0] = a.x; f_own[0] = __float_as_int(a.y); w_own[1] = a.z; f_own[1] = __float_as_int(a.w);
            w_own[2] = b.x; f_own[2] = __float_as_int(b.y); w_own[3] = b.z; f_own[3] = __float_as_int(b.w);
            const float2 l = rowp[-1], r = rowp[4];
            w_l = l.x; f_l = __float_as_int(l.y); w_r = r.x; f_r = __float_as_int(r.y);
            const float4 c = *reinterpret_cast<const float4*>(rowp - VS), d = *reinterpret_cast<const float4*>(rowp - VS + 2);
            w_up[0] = c.x; f_up[0] = __float_as_int(c.y); w_up[1] = c.z; f_up[1] = __float_as_int(c.w);
            w_up[2] = d.x; f_up[2] = __float_as_int(d.y); w_up[3] = d.z; f_up[3] = __float_as_int(d.w);
            const float4 e = *reinterpret_cast<const float4*>(rowp + VS), f = *reinterpret_cast<const float4*>(rowp + VS + 2);
            w_dn[0] = e.x; f_dn[0] = __float_as_int(e.y); w_dn[1] = e.z; f_dn[1] = __float_as_int(e.w);
            w_dn[2] = f.x; f_dn[2] = __float_as_int(f.y); w_dn[3] = f.z; f_dn[3] = __float_as_int(f.w);
        }
        bool covered[4];
        int key[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            covered[j] = in_px[j] & (f_own[j] >= 0);
            key[j] = covered[j] ? f_own[j] : -1;
        }

        // ---- background gradient (:143-147): grad_pixels where nothing is covered, zero elsewhere.  Issued AFTER the face
        //      loop: here, between the Scharr filter and the dilation, every wave of the chip reaches them at about the same
        //      time (one lockstep round) and stalls behind 16.8 MB of stores (the per-wave trace: ~3000 clocks of the
        //      dilation phase); at the end of a wave they spread over the ~10 us in which the waves finish.  What they need
        //      -- grad_pixels and "covered" -- is live in the face loop anyway.  K3-2048: gradient kernel 69 -> 66 us. ----
        auto store_background = [&]() {
        // Many-channel frames with 16-byte aligned pixels: the passes of a launch SHARE the tile -- pass i of n writes rows
        // i, i + n, ... of it, ALL channels of a pixel at once, in a lane <-> float4 linear mapping, so that every store
        // instruction writes 1 KB of whole lines (a pass storing its own 12 bytes of every 64-byte pixel writes each line
        // five times over, partially: K5, ~100 of 550 us).  grad_pixels is read a second time for it (the lines are in
        // the L2: the passes of the tile have just fetched them); "covered" comes from the state tile in LDS.
        if constexpr (STRIDED) {
            if (p.gbk_split >= 0) {   // (uniform over the launch)
                const int n = p.gbk_split;
                if (n == 0) return;    // another launch of this call writes grad_background
                const int per_row = GT * (C >> 2);            // float4 per tile row
                const float inv_q = 4.f / (float)C;
                const float* __restrict__ gp_all = p.grad_pixels + origin * C;
                float* __restrict__ gb_all = p.grad_background + origin * C;
                for (int r = pass_i + n * wave; r < GT; r += 4 * n) {
                    const int yy = y0 + r;
                    if (yy >= H) break;
                    const uint32_t rowoff = ((uint32_t)(yy - row0) * (uint32_t)W + (uint32_t)x0) * pixel_bytes;
                    for (int q0 = lane; q0 < per_row; q0 += 256) {
                        float4 gq[4];
                        bool ok[4], cov[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int q = q0 + 64 * i;
                            const int px = (int)(((float)q + 0.5f) * inv_q);
                            ok[i] = (q < per_row) & (x0 + px < W);
                            cov[i] = __float_as_int(s_vw[r + 1][min(px, GT - 1) + 2].y) >= 0;
                            gq[i] = ld_off<float4>(gp_all, ok[i] ? rowoff + 16u * (uint32_t)q : 0u);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (ok[i]) st_off<float4>(gb_all, rowoff + 16u * (uint32_t)(q0 + 64 * i), cov[i] ? make_float4(0.f, 0.f, 0.f, 0.f) : gq[i]);
                    }
                }
                return;
            }
        }
#if DIRT_GBK_COALESCED
        // 4-channel frames: the wave's 32 x 8 pixels in a lane <-> pixel LINEAR mapping (lane l: column l % 32, rows
        // l / 32 + 2 k), so that every store instruction writes 1 KB of whole lines -- a strip owner's float4 stores are
        // 16 bytes of every 64.  grad_pixels is read a second time for it (the lines are this workgroup's own, a few
        // microseconds old: L2 hits) and "covered" comes from the state tile in LDS.
        if constexpr (NCH == 4 && !STRIDED) {
            const int cx = lane & 31;
            float4 gq[4];
            int fq[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ry_ = 2 * k + (lane >> 5), yy = y0 + 8 * wave + ry_;
                ok[k] = (x0 + cx < W) & (yy < H);
                const uint32_t off = (uint32_t)((min(yy, H - 1) - row0) * W + min(x0 + cx, W - 1)) * 16u;
                gq[k] = ld_off<float4>(gpix_t, off);
                fq[k] = __float_as_int(s_vw[8 * wave + ry_ + 1][cx + 2].y);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ry_ = 2 * k + (lane >> 5), yy = y0 + 8 * wave + ry_;
                if (!ok[k]) continue;
                const uint32_t off = (uint32_t)((yy - row0) * W + x0 + cx) * 16u;
                st_off<float4>(gbk_t, off, fq[k] >= 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : gq[k]);
            }
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!in_px[j]) continue;
            const uint32_t off = own_off + (uint32_t)j * pixel_bytes;
            bool wide = false;
            if constexpr (NCH == 4) {
                if (single_on && (C & 3) == 0 && aligned16) {
                    st_off<float4>(gbk_t, off, covered[j] ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(g[j][0], g[j][1], g[j][2], g[j][3]));
                    wide = true;
                } else if (!single_on) {
                    st_off<Float3>(gbk_t, off, covered[j] ? Float3{0.f, 0.f, 0.f} : Float3{g[j][0], g[j][1], g[j][2]});
                    wide = true;
                }
            }
            if constexpr (NCH == 3) {
                st_off<Float3>(gbk_t, off, covered[j] ? Float3{0.f, 0.f, 0.f} : Float3{g[j][0], g[j][1], g[j][2]});
                wide = true;
            }
            if constexpr (NCH == 6) {
                st_off<Float3>(gbk_t, off, covered[j] ? Float3{0.f, 0.f, 0.f} : Float3{g[j][0], g[j][1], g[j][2]});
                if (second_mode == 0) st_off<Float3>(gbk_t, off + 12u, covered[j] ? Float3{0.f, 0.f, 0.f} : Float3{g[j][3], g[j][4], g[j][5]});
                else if (second_mode == 1) st_off<float>(gbk_t, off + 12u, covered[j] ? 0.f : g[j][3]);
                wide = true;
            }
            if (!wide) {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) st_off<float>(gbk_t, off + 4u * ch, covered[j] ? 0.f : g[j][ch]);
            }
        }
        };

        // ---- dilation (:155-194).  A pixel takes the fragment of the neighbour n at +d, else at -d, when that neighbour
        //      is covered, is another face (:86-89) and is closer (:165); d is +-x or +-y.  An uncovered neighbour has
        //      clip_w = +inf (the clear value), so "closer" already says it is covered.  The four tests of a pixel do not
        //      depend on the channel group, and the horizontal ones are shared by adjacent pixels of the strip.
        // ---- position factors (:196-232): the gradients of vertex k are b_k * (fx, fy, fw) with
        //          fx = dL_dx * (W/2) / w,  fy = dL_dy * (H/2) / w,  fw = -(fx * ndc_x + fy * ndc_y)
        //      (:210-222: clip_x = sum b_k * vertex_k.x is the fragment's own clip position = its NDC position times
        //      clip_w -- perspective-correct barycentrics -- so no vertex gather is needed; one v_rcp_f32, 1 ulp; agrees to
        //      float rounding), everything taken at the pixel whose fragment is used.  (fx, fy) are summed per such
        //      TARGET pixel -- own pixels in registers, neighbours through the inbox -- and fw is formed once per pixel. ----
        // Per pixel: the clip_w of each neighbour that would dilate into it -- another face (:86-89) that is closer (:165) --
        // or a NaN sentinel; the attempt order of :186-193 is then two selects per pixel (by the parity dither) and two per channel
        // group (by its axis), and "which neighbour, if any" two self-compares.  A pixel's own fragment is the common
        // case and runs unconditionally (one rcp of its own clip_w per pixel); the neighbour's clip_w, its reciprocal and
        // the inbox address exist only for the few dilated lanes.  (Rounds 2-3 kept every predicate as a wave-wide lane
        // mask combined on the scalar unit: ~40 masks alive, moved through VGPRs by the compiler, and every pixel a chain
        // v_cmp -> s_and / s_or x 6 -> v_cndmask x 4; this form has 160 fewer scalar instructions per wave.)
        const float NO_NEIGHBOUR = __builtin_nanf("");   // (a value no clip_w that passed `wo > w` can have: a qualifying neighbour with clip_w == +-0 still counts, as in :165)
        const bool pos0 = ((xs + y) & 1) == 0;   // pixel 0 tries +x / up first (:186-191)
        const float2v half_size = float2v{.5f * width_f, .5f * height_f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wl = j == 0 ? w_l : w_own[j - 1], wr = j == 3 ? w_r : w_own[j + 1];
            const int fl = j == 0 ? f_l : f_own[j - 1], fr = j == 3 ? f_r : f_own[j + 1];
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
                if (STRIDED && NCH == 4 && gi == 1 && !single_on) continue;
                if (TWO3 && gi == 1 && second_mode == 2) continue;
                // direction: x if L1(Sx) > L1(Sy) else y (:185), negated on odd (x + y) (:186-190).  The reference's
                // offsets are in GL buffer orientation (y up): tensor row = y - offset_y.
                const bool hz = __builtin_amdgcn_inverse_ballot_w64(horiz_m[gi][j]);
                const float q1 = hz ? qx1 : qy1, q2 = hz ? qx2 : qy2;
                const bool first = q1 == q1;                  // the first attempt found its neighbour (not the NaN sentinel)
                const bool dilated = first | (q2 == q2);      // ... or the opposite one did (:192-193)
                if constexpr (DEBUG) {
                    if (cbase == 0 && gi == 0 && in_px[j]) write_debug(p.debug_thingy, p.grad_pixels, p.B, H, W, C, iib, y, xs + j, G0, dilated);
                }
                const float dLx_j = (j & 1) ? dLx[gi][j >> 1].y : dLx[gi][j >> 1].x, dLy_j = (j & 1) ? dLy[gi][j >> 1].y : dLy[gi][j >> 1].x;
                const float2v t = float2v{dLx_j, dLy_j} * half_size;
                const float2v f = t * float2v{rcp_own, rcp_own};
                const bool own = covered[j] & !dilated;       // contributes to its own pixel
                fxy[j] += float2v{own ? f.x : 0.f, own ? f.y : 0.f};
                if (dilated) {  // few lanes: ds_add_f32 into the neighbour's cell, with the NEIGHBOUR's clip_w
                    const float rcp_w = __builtin_amdgcn_rcpf(first ? q1 : q2);
                    const float2v fn = t * float2v{rcp_w, rcp_w};
                    const int step = hz ? 1 : -IS;            // +x, or up = the previous row
                    const bool fwd = first == pos;            // the neighbour taken lies at +x / up
                    float* cell = reinterpret_cast<float*>(inbox + (my_cell + j + (fwd ? step : -step)));
                    atomicAdd(cell, fn.x);
                    atomicAdd(cell + 1, fn.y);
                }
            }
        }
        GMARK();  // 5 dilation done
        // positions and colours in one face loop
        float2v fpos_xy[4];
        float fpos_w[4];
        int lkey[2];
        float lb[2][3], lf[2][3];
        gather_positions(fpos_xy, fpos_w, lkey, lb, lf);
        GMARK();  // 6 face loop starts
#ifndef DIRT_KO_LOOP
        face_loop(integral_constant<int, NCH>{}, g, key, covered, fpos_xy, fpos_w, lkey, lb, lf);
#endif
#ifndef DIRT_KO_GBK
        store_background();
#endif
    };

    run_pass(integral_constant<int, CSPEC>{}, integral_constant<int, CSPEC == 1 ? 1 : 3>{});
    GMARK();  // 7 done
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_grad) {
        long long* o = g_trace_grad + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16;
        for (int i = 0; i < 12; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
        o[12] = tr_c[0]; o[13] = tr_c[1];
        o[14] = tr_wall0; o[15] = (((long long)wall_clock64() - tr_wall0) << 20) | (long long)(__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4 /* HW_REG_HW_ID */) & 0xFFFFF);
    }
#endif
}

hipError_t launch_grad(const GradParams& p_in, hipStream_t stream)
{
    if (p_in.B == 0) return hipSuccess;
    GradParams p = p_in;
    p.tiles_x = (p.W + GT - 1) / GT;
    p.tiles_y = (p.H + GT - 1) / GT;
    p.tiles_x_magic = tile_magic(p.tiles_x);
    p.inv_w = 1.f / (float)p.W; p.inv_h = 1.f / (float)p.H;   // (IEEE divisions, as the kernel's own would be)
    p.pixels_aligned16 = ((reinterpret_cast<uintptr_t>(p.pixels) | reinterpret_cast<uintptr_t>(p.grad_pixels) |
                           reinterpret_cast<uintptr_t>(p.grad_background)) & 15u) == 0 ? 1 : 0;
#ifdef DIRT_TRACE
    const size_t dyn_lds = getenv("DIRT_TRACE_DYN_LDS") ? (size_t)atoi(getenv("DIRT_TRACE_DYN_LDS")) : 0;  // occupancy experiments
#else
    const size_t dyn_lds = 0;
#endif
    const dim3 block(GTHREADS);
    const unsigned ntiles = (unsigned)(p.tiles_x * p.tiles_y);
#define DIRT_LAUNCH_GRAD(SHAPE_, STRIDED_)                                                                       \
    do {                                                                                                        \
        const dim3 grid(ntiles * (unsigned)p.npasses, (unsigned)p.B);                                           \
        if (p.debug_thingy && p.c_first == 0)                                                                   \
            hipLaunchKernelGGL((grad_kernel<SHAPE_, STRIDED_, true>), grid, block, dyn_lds, stream, p);         \
        else if (!STRIDED_ && rows)                                                                             \
            hipLaunchKernelGGL((grad_kernel<SHAPE_, false, false, true>), grid, block, dyn_lds, stream, p);     \
        else                                                                                                    \
            hipLaunchKernelGGL((grad_kernel<SHAPE_, STRIDED_, false>), grid, block, dyn_lds, stream, p);        \
    } while (0)
    // Which shape (measured on MI355X, gradient kernel in us; faces per 32 x 32 tile = F / tiles):
    //                       tiles  faces/tile   px1    rows   pairs
    //   K3-256  (10k faces)    64     156       10.6   20.7   26.1
    //   K3-384                144      69       18.1   18.1   19.9
    //   K3-512                256      39       22.5   18.7   18.4
    //   K3-768                576      17       39.8   25.8   22.9
    //   12 large faces, 256^2  64     0.2       27.7   14.4   10.6
    // The one-pixel-per-lane kernel (dirt_grad_small.hip: four times the waves, a handful of faces per 4 x 4 block) pays
    // where a small frame is DENSE in faces; with few large faces its per-block atomics all land on the same vertices.
    // Every row its own face (more atomics, fewer iterations) pays for moderately dense small frames; pairs otherwise.
    const long long density = ntiles ? (long long)p.F / (long long)ntiles : 0;   // faces per 32 x 32 tile
    const bool few_tiles = (long long)ntiles * p.B <= 256;                       // at most one workgroup per compute unit
    {
        const bool small_ok = p.C == 1 || p.C == 3 || p.C == 4;   // (many-channel G-buffers keep the strided passes below)
        bool small = small_ok && few_tiles && density >= 96;
        if (p.flags & DIRT_FLAG_GRAD_SMALL) small = small_ok;
        if (p.flags & (DIRT_FLAG_GRAD_ROWS | DIRT_FLAG_GRAD_PAIRS)) small = false;
        if (small) return launch_grad_small(p, stream);
    }
    {
        // The streaming kernel (dirt_grad_stream.hip, round 6): the 4-pixel kernel's decomposition and face loop with wave-private
        // tiles filled by LDS-DMA, slice 1's loads in flight under slice 0's compute.  4 channels, whole 32 x 32 tiles.
        // Measured (round 6, profiles/EXPERIMENTS.md): parity-green, and SLOWER than the 4-pixel kernel at every size tried (K3 30.3
        // against 27.1 us raw, K3-2048 68.8 / 64.7, eight scenes per launch 153 / 152): VMEM issue blocks the wave while the
        // memory pipeline is backed up, so slice 1's requests do not travel under slice 0's compute.  Opt-in only.
        bool streamk = false;
        if (p.flags & DIRT_FLAG_GRAD_STREAM) streamk = grad_stream_eligible(p);
        if (p.flags & (DIRT_FLAG_GRAD_ROWS | DIRT_FLAG_GRAD_PAIRS | DIRT_FLAG_GRAD_PX4 | DIRT_FLAG_GRAD_PX2 | DIRT_FLAG_GRAD_SMALL)) streamk = false;
        if (streamk) return launch_grad_stream(p, stream);
    }
    {
        // Two pixels per lane (dirt_grad_px2.hip: 32 x 16 tiles, twice the waves at half the chain each, 5-7 workgroups per
        // compute unit).  Measured on MI355X (round 5, gradient kernel in us, HIP events, 10 000 faces, dense outputs):
        //                          32x32 tiles    px2     4-pixel kernel
        //   K3-768  (4 ch)             576        20.8        24.0       the 4-pixel grid leaves compute units with 2 or 3 workgroups
        //   K3      (4 ch)            1024        27.1        26.4       one full round of the 4-pixel kernel: nothing to gain
        //   K3-3ch                    1024        22.2        23.1       (64 VGPRs: seven workgroups per compute unit)
        //   K3-1ch                    1024        20.6        18.9
        //   K3-2048 (4 ch)            4096        72.9        68.2       rounds overlap by themselves; px2's extra atomics and instructions cost
        //   K5-3ch  (50 000 faces)    4096        62.2        66.5       3 channels: eight workgroups per compute unit
        // Rule: more than one workgroup per compute unit but less than a full round of the 4-pixel kernel; 3 channels: every such frame.
        const bool px2_ok = p.C == 1 || p.C == 3 || (p.C == 4 && p.pixels_aligned16);
        const long long wgs32 = (long long)ntiles * p.B;
        bool px2 = px2_ok && !few_tiles && (wgs32 < 1024 || p.C == 3);
        if (p.flags & DIRT_FLAG_GRAD_PX2) px2 = px2_ok;
        if (p.flags & (DIRT_FLAG_GRAD_ROWS | DIRT_FLAG_GRAD_PAIRS | DIRT_FLAG_GRAD_PX4)) px2 = false;
        if (px2) return launch_grad_px2(p, stream);
    }
    bool rows = few_tiles && density >= 48;
    if (p.flags & DIRT_FLAG_GRAD_ROWS) rows = true;
    if (p.flags & DIRT_FLAG_GRAD_PAIRS) rows = false;
    p.c_first = 0; p.npasses = 1; p.gbk_split = -1; p.last_second = 0;
    // the common channel counts: kernels in which the channel count is a compile-time constant
    if (p.C == 4 && p.pixels_aligned16) {
        // Grids of several rounds (eight scenes of K3 in one launch: 8192 workgroups): the aliased-inbox form, five workgroups per
        // compute unit instead of four -- K3 x 8 gradient 157.7 -> 147.7 us, K3-2048 66.4 -> 65.9; a single round pays only for its
        // extra barrier (K3: 26.4 -> 27.8), so one scene of up to 2047 tiles keeps the plain form.
        if (DIRT_ALIAS_INBOX && !rows && !p.debug_thingy && (long long)ntiles * p.B >= 2048)
            hipLaunchKernelGGL((grad_kernel<4, false, false, false, true>), dim3(ntiles, (unsigned)p.B), block, dyn_lds, stream, p);
        else
            DIRT_LAUNCH_GRAD(4, false);
    }
    else if (p.C == 3) DIRT_LAUNCH_GRAD(3, false);
    else if (p.C == 1) DIRT_LAUNCH_GRAD(1, false);
    else {
        // any other channel count: passes of whole channel groups (groups of 3 while >= 3 channels remain, then singles,
        // dirt/rasterise_ops.py:148-152).  Every 3-channel pass goes out in ONE launch -- of the {3,1} body when a single
        // follows the triples: its last pass carries that single, the others switch the single's parts off -- so that the
        // passes of a tile meet in the L2 (a second single, C % 3 == 2, is a launch of its own).
        // From 6 channels on the triples go out in PAIRS (the {3,3} shape: one staging of the tile, one walk over its faces
        // and one fetch of every 64-byte pixel for two groups), an odd last triple -- with the single that follows it, if
        // any -- as the last pass of the same launch, so that all passes of a tile meet in the L2 (a launch of its own for
        // the last four channels of K5 read both 268 MB tensors a second time); a second single (C % 3 == 2), or the
        // singles behind an even number of triples, are a launch of the {1} shape.
        const int groups3 = p.C / 3, singles = p.C % 3;
        // grad_background: written by the FIRST launch's passes, all channels, whole lines (store_background), where the
        // pixels are 16-byte aligned; otherwise every pass stores its own channels
        const bool gbk_shared = (p.C & 3) == 0 && p.pixels_aligned16 != 0;
        bool gbk_done = false;
#define DIRT_GBK_PLAN() do { p.gbk_split = !gbk_shared ? -1 : (gbk_done ? 0 : p.npasses); gbk_done = true; } while (0)
        bool two3 = groups3 >= 2;
#ifdef DIRT_NO_TWO3
        two3 = false;
#endif
        if (two3) {
            const bool odd = (groups3 & 1) != 0;
            p.c_first = 0; p.npasses = (groups3 + 1) / 2; p.last_second = !odd ? 0 : (singles >= 1 ? 1 : 2); DIRT_GBK_PLAN();
            if (p.debug_thingy) hipLaunchKernelGGL((grad_kernel<6, true, true>), dim3(ntiles * (unsigned)p.npasses, (unsigned)p.B), block, dyn_lds, stream, p);
            else hipLaunchKernelGGL((grad_kernel<6, true, false>), dim3(ntiles * (unsigned)p.npasses, (unsigned)p.B), block, dyn_lds, stream, p);
            const int rest = singles - (odd && singles >= 1 ? 1 : 0);   // singles not yet done
            if (rest >= 1) { p.c_first = p.C - rest; p.npasses = rest; DIRT_GBK_PLAN(); DIRT_LAUNCH_GRAD(1, true); }
        } else if (groups3 >= 1 && singles >= 1) {
            p.c_first = 0; p.npasses = groups3; DIRT_GBK_PLAN(); DIRT_LAUNCH_GRAD(4, true);
            if (singles == 2) { p.c_first = 3 * groups3 + 1; p.npasses = 1; DIRT_GBK_PLAN(); DIRT_LAUNCH_GRAD(1, true); }
        } else if (groups3 >= 1) {
            p.c_first = 0; p.npasses = groups3; DIRT_GBK_PLAN(); DIRT_LAUNCH_GRAD(3, true);
        } else {
            p.c_first = 0; p.npasses = singles; DIRT_GBK_PLAN(); DIRT_LAUNCH_GRAD(1, true);
        }
#undef DIRT_GBK_PLAN
    }
#undef DIRT_LAUNCH_GRAD
    return hipGetLastError();
}

}  // namespace dirt
