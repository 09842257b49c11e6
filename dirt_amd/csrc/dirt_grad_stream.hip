// dirt_grad_stream.hip -- the gradient assembly kernel for 4-channel frames whose load phase STREAMS under its compute
// (gfx950, round 6).
//
// Same contract as grad_kernel (dirt_grad.hip; replaces assemble_grads, csrc/rasterise_grad_egl.cu:93-236, for the two
// channel groups {0,1,2}, {3} of dirt/rasterise_ops.py:145-165 in one launch), same per-pixel arithmetic, same face loop.
// What is different is how the data gets there.  grad_kernel stages a 32 x 32 tile through registers into LDS planes behind
// ONE workgroup barrier: at 1024 x 1024 every wave of the chip waits 3-9 us for its loads, then computes ~13 us, strictly
// one after the other (profiles/EXPERIMENTS.md, rounds 4-5: HBM idles while the SIMDs work and vice versa).  Here
//   * a WAVE owns its 32 x 8 pixel region alone -- its own halo'd copies of `pixels` (10 x 36 pixels, AoS float4, exactly as
//     they lie in memory: no transposition) and of the state's {clip_w, face} plane (10 x 36 float2) in LDS -- so there is
//     no workgroup barrier anywhere in the kernel;
//   * both arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, no ds_write pass); the only
//     register loads are a lane's own grad_pixels (which it also stores as grad_background) and barycentrics;
//   * the region is worked in two SLICES of 32 x 4 pixels (a lane: one 2 x 1 pair per slice, so its four pixels are a pair in
//     row r and the pair four rows below).  The loads of slice 1 are issued when those of slice 0 have landed and travel
//     while slice 0 is filtered and dilated; the face loop runs once, over all four pixels of every lane, on the 16 x 8 half
//     regions of the 4-pixel kernel (same float-atomic count).  Every wait is a plain `s_waitcnt vmcnt(0)` at a slice boundary:
//     at most one slice's loads are ever in flight, and nothing the compiler counts is first used while a DMA is outstanding;
//   * Scharr runs on CHANNEL pairs of the AoS pixels (v_pk_*_f32 on the two halves of a float4) in the reference's
//     operation order exactly (it decides the dilation axis, a discrete choice), any number of pixels per lane alike;
//   * the per-wave inbox (dilated contributions, ds_add_f32) is ALIASED onto the pixel rows each slice has finished with.
// Eligible: C = 4, 16-byte aligned tensors, W and H multiples of 32, no debug_thingy (launch_grad keeps grad_kernel for the
// rest).  Variable names in the per-pixel arithmetic follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "dirt_reduce.h"
#include "dirt_grad_common.h"
#include "../../include/dirt_hip.h"
#include <type_traits>

namespace dirt {

#ifdef DIRT_TRACE
// Per-wave phase timestamps for tools/trace_grad.py (the layout of dirt_grad.hip's trace); tracing build only.
__device__ long long* g_trace_grad_stream = nullptr;
extern "C" void dirt_debug_set_trace_grad_stream(void* p)
{
    long long* q = reinterpret_cast<long long*>(p);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_grad_stream), &q, sizeof(q));
}
#define SMARK() do { if (tr_n < 12) { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tr_t[tr_n++] = t_; } } while (0)
#define SCOUNT(i, v) do { tr_c[i] += (v); } while (0)
#else
#define SMARK() do {} while (0)
#define SCOUNT(i, v) do {} while (0)
#endif

#ifndef DIRT_STREAM_MODE
#define DIRT_STREAM_MODE 0
#endif
#ifndef DIRT_STREAM_DMA
#define DIRT_STREAM_DMA 1
#endif

namespace {

constexpr int ST = 32;                  // tile side: a workgroup's four waves own rows 8 w .. 8 w + 7 of it
constexpr int STHREADS = 256;
constexpr int SROWS = 10;               // a wave's staged rows: yw0 - 1 .. yw0 + 8
constexpr int SPC = 36;                 // pixel tile: columns x0 - 1 .. x0 + 34 (a single channel's aliased "channels", quirk Q1, are the next two pixels), float4
constexpr int SPSLOTS = SROWS * SPC;    // 360 float4 = 5.6 DMA instructions
constexpr int SAC = 36;                 // state tile: columns x0 - 2 .. x0 + 33 as float2 {clip_w, face}: pairs of pixels are 16-byte units
constexpr int SASLOTS = SROWS * SAC / 2;   // 180 float4 = 2.8 DMA instructions
constexpr int SIS = 34;                 // inbox row stride (float2 cells): cell (ty + 1) * 34 + tx + 2 for ty in -1..8, tx in -1..32
constexpr int SICELLS = 10 * SIS + 4;   // 344 (a multiple of 2: 16-byte cell pairs)
constexpr int SI_FIRST = 288;           // cells [0, 288) = bytes [0, 2304) = pixel rows -1 .. 2: dead once slice 0 is filtered; slice 0 dilates into cells <= 204
constexpr int SRING = 2 * 34 + 2 * 8;   // ring cells: what the wave's pixels sent to pixels of other waves
static_assert(SI_FIRST * 8 == 4 * SPC * 16 && SICELLS * 8 <= SPSLOTS * 16, "the inbox fits in the pixel rows it aliases");

// One LDS-DMA wave-instruction: lane l's 16 bytes at base + voff land at LDS byte address lds_addr + 16 l (lanes switched off
// by the surrounding branch write nothing).  M0 holds the destination; it is saved and restored around the statement (the
// compiler does not know it is written).  Not counted by the compiler: the kernel waits with s_waitcnt vmcnt(0) itself.
__device__ __forceinline__ void glds16(const void* base, uint32_t voff, uint32_t lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ float2v lo2(const float4& q) { return float2v{q.x, q.y}; }
__device__ __forceinline__ float2v hi2(const float4& q) { return float2v{q.z, q.w}; }

// alias_wrap_fixup (dirt_grad_common.h) with ROLLED loops for a pair of pixels: quirk Q1 at the right image border -- for the
// pixels (first column xs, row y) flagged in `which`, the aliased "channels" 1, 2 of 1-channel group c lie in the NEXT image
// row (past the end of the tensor: clamped to its last element), and their dilation axis (:185) is decided again from memory.
__device__ __forceinline__ uint32_t alias_wrap_fixup_pair(const float* __restrict__ pixels, int B, int H, int W, int C, int iib, int y, int xs,
                                                       int c, uint32_t which, uint32_t bits)
{
    const size_t last = (size_t)B * H * W - 1;
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
        if (!((which >> j) & 1u)) continue;
        const size_t centre = ((size_t)iib * H + y) * W + xs + j;   // flat pixel index of the pixel
        float l1x = 0.f, l1y = 0.f;
#pragma unroll 1
        for (int ch = 0; ch < 3; ++ch) {
            float w[3][3];   // w[r][i] = element (centre + ch + i - 1) of row y - 1 + r in flat order, clamped to the end of the tensor
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

__global__ __launch_bounds__(STHREADS, 4) void grad_kernel_stream(GradParams p)
{
    constexpr int C = 4, NCH = 4, NG = 2;
    __shared__ __align__(16) float4 s_pix[STHREADS / 64][SPSLOTS];        // per wave: `pixels` of its region + halo, as in memory; later its inbox
    __shared__ __align__(16) float2 s_a[STHREADS / 64][SROWS][SAC];       // per wave: {clip_w, face} of its region + halo

#ifdef DIRT_TRACE
    long long tr_t[12]; int tr_n = 0; long long tr_c[4] = {0, 0, 0, 0};
    const long long tr_wall0 = wall_clock64();
#endif
    SMARK();  // 0 start
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int iib = blockIdx.y;
    const int H = p.H, W = p.W;
    const size_t frame = (size_t)H * W;
    const int tile = xcd_tile((int)blockIdx.x, p.tiles_x * p.tiles_y);
    int tile_col, tile_row;
    tile_xy(tile, p.tiles_x, p.tiles_x_magic, tile_col, tile_row);
    const int x0 = tile_col * ST, yw0 = tile_row * ST + 8 * wave;   // the wave's region: columns x0 .. x0 + 31, rows yw0 .. yw0 + 7

    // Wave-uniform bases at the wave's first staged row (rowbase): every per-lane address is a small non-negative 32-bit byte
    // offset, (row - rowbase) * W + column, times the element size.
    const int rowbase = max(yw0 - 1, 0);
    const size_t origin = (size_t)iib * frame + (size_t)rowbase * W;   // pixel index of (rowbase, column 0)
    const float2* __restrict__ state_a = p.state_a + origin;            // {clip_w, face}
    const float2* __restrict__ state_b = p.state_b + origin;            // two barycentrics (encode_bary)
    const float* __restrict__ pixels_t = p.pixels + origin * C;
    const float* __restrict__ gpix_t = p.grad_pixels + origin * C;
    float* __restrict__ gbk_t = p.grad_background + origin * C;
    const int32_t* __restrict__ faces = p.faces + (p.shared_faces ? (size_t)0 : (size_t)iib * p.F * 3);
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * p.gv_stride;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * p.gvc_stride;
    const uint32_t gv_row_bytes = 4u * (uint32_t)p.gv_stride, gvc_row_bytes = 4u * (uint32_t)p.gvc_stride;
    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;
    const float width_f = (float)W, height_f = (float)H;

    // ---- this lane's pixels: DPP row blk = lane >> 4 is the 8 x 8 block at columns 8 blk of the region; in it lane bits 0-1
    //      choose the pair (of four across), bits 2-3 the row (of four) of a slice; slice s is rows 4 s .. 4 s + 3.  Pixel
    //      j = 2 s + q is pixel q of the pair in slice s. ----
    const int blk = lane >> 4;
    const int rx = 8 * blk + 2 * (lane & 3);     // region-relative column of the pair (even)
    const int pr = (lane >> 2) & 3;              // row inside a slice
    const int xs = x0 + rx;                      // first pixel of the pair
    const uint32_t lds_pix = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&s_pix[wave][0]);
    const uint32_t lds_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&s_a[wave][0][0]);
    float2* const inbox = reinterpret_cast<float2*>(&s_pix[wave][0]);

    // ---- LDS-DMA pieces.  Pixel tile: slot s = 64 i + lane is row s / 36, column s % 36, edge clamped (at(), :113-124).  State
    //      tile: slot s is row s / 18, pixel pair s % 18 at columns x0 - 2 + 2 (s % 18) (an even column; W is even: a pair never
    //      straddles the end of an image row), clamped as a pair -- halo positions outside the frame are only ever consulted
    //      for interior pixels, whose neighbours are inside the frame. ----
#if !DIRT_STREAM_DMA   // (A/B build: the same tiles staged through registers, plain loads + ds_write_b128 after the wait; DIRT_STREAM_MODE 1 only)
    float4 stage_p[6], stage_a[3];
#endif
    auto dma_pix = [&](int i) {
        const int s = 64 * i + lane;
        if (s < SPSLOTS) {
            const int row = s / SPC, col = s - row * SPC;
            const int cy = min(max(yw0 - 1 + row, 0), H - 1), cx = min(max(x0 - 1 + col, 0), W - 1);
#if DIRT_STREAM_DMA
            glds16(pixels_t, (uint32_t)((cy - rowbase) * W + cx) * 16u, lds_pix + 1024u * (uint32_t)i);
#else
            stage_p[i] = ld_off<float4>(pixels_t, (uint32_t)((cy - rowbase) * W + cx) * 16u);
#endif
        }
    };
    auto dma_a = [&](int i) {
        const int s = 64 * i + lane;
        if (s < SASLOTS) {
            const int row = s / (SAC / 2), pc = s - row * (SAC / 2);
            const int cy = min(max(yw0 - 1 + row, 0), H - 1), cx = min(max(x0 - 2 + 2 * pc, 0), W - 2);
#if DIRT_STREAM_DMA
            glds16(state_a, (uint32_t)((cy - rowbase) * W + cx) * 8u, lds_a + 1024u * (uint32_t)i);
#else
            stage_a[i] = ld_off<float4>(state_a, (uint32_t)((cy - rowbase) * W + cx) * 8u);
#endif
        }
    };
    // own pixels of slice s: byte offsets of the pair's first pixel
    auto own_rel = [&](int s) { return (uint32_t)((yw0 + 4 * s + pr - rowbase) * W + xs); };

    // ---- slice 0's loads: pixel rows -1 .. 4 are slots 0 .. 215 (pieces 0-3 bring rows -1 .. 6.1), state rows -1 .. 4 are
    //      slots 0 .. 107 (pieces 0-1), this lane's grad_pixels of the slice ----
    float4 gq[4];    // grad_pixels of the lane's pixels
    float4 bq[2];    // encoded barycentrics of the pair in slice s: {p, q} of pixel 0, {p, q} of pixel 1
    dma_pix(0); dma_pix(1); dma_pix(2); dma_pix(3);
    dma_a(0); dma_a(1);
    gq[0] = ld_off<float4>(gpix_t, own_rel(0) * 16u);
    gq[1] = ld_off<float4>(gpix_t, own_rel(0) * 16u + 16u);
    // ---- slice 1's loads: the rest of both tiles, grad_pixels and the barycentrics of all four pixels (two stored, the
    //      largest re-derived: decode_bary; needed in the face loop only) ----
    auto issue_slice1 = [&]() {
        dma_pix(4); dma_pix(5);
        dma_a(2);
        gq[2] = ld_off<float4>(gpix_t, own_rel(1) * 16u);
        gq[3] = ld_off<float4>(gpix_t, own_rel(1) * 16u + 16u);
        bq[0] = ld_off<float4>(state_b, own_rel(0) * 8u);
        bq[1] = ld_off<float4>(state_b, own_rel(1) * 8u);
    };
#if DIRT_STREAM_MODE == 1   // everything requested up front, one wait (no streaming: the 4-pixel kernel's order with this kernel's data path)
    issue_slice1();
#endif
    SMARK();  // 1 slice 0 requested
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(gq[0].x), "+v"(gq[0].y), "+v"(gq[0].z), "+v"(gq[0].w), "+v"(gq[1].x), "+v"(gq[1].y), "+v"(gq[1].z), "+v"(gq[1].w));
#if !DIRT_STREAM_DMA
    static_assert(DIRT_STREAM_MODE == 1, "register staging: everything up front");
#pragma unroll
    for (int i = 0; i < 6; ++i) if (64 * i + lane < SPSLOTS) s_pix[wave][64 * i + lane] = stage_p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) if (64 * i + lane < SASLOTS) reinterpret_cast<float4*>(&s_a[wave][0][0])[64 * i + lane] = stage_a[i];
#endif
    SMARK();  // 2 slice 0 landed
#if DIRT_STREAM_MODE == 0   // slice 1's loads travel while slice 0 is worked on
    issue_slice1();
#endif

    // per pixel, filled slice by slice
    float dLx[NG][4], dLy[NG][4];   // dL/dx, dL/dy of :203-208 per group
    float2v fxy[4];                 // (fx, fy) sent to the pixel by itself (see "position factors" below)
    float w_own[4];
    int key[4];
    bool covered[4];
    const float NO_NEIGHBOUR = __builtin_nanf("");   // (a value no clip_w that passed `wo > w` can have: a qualifying neighbour with clip_w == +-0 still counts, as in :165)
    const float2v half_size = float2v{.5f * width_f, .5f * height_f};

    auto slice = [&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        const int ry = 4 * s + pr;               // region-relative row
        const int y = yw0 + ry;                  // tensor row (top row first)
        bool interior[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) interior[q] = (xs + q > 0) & (y > 0) & (xs + q < W - 1) & (y < H - 1);

        // ---- Scharr (:126-127, operation for operation: negative-offset minus positive-offset, offset_y is up = the previous
        //      tensor row) on CHANNEL pairs of the lane's two pixels, streamed into what is needed of it: the direction choice of
        //      :185 from the L1 norms (all three "channels" of the reference's Vec3, in its summation order) and dL/dx, dL/dy
        //      of :203-208.  Taps: rows y - 1 .. y + 1 (tile rows ry .. ry + 2), columns xs - 1 .. xs + 2 (tile columns
        //      rx .. rx + 3) of all channels; of channel 3 also columns xs + 3, xs + 4 (quirk Q1, below). ----
        bool horiz[NG][2];
        {
            const float4* tap = &s_pix[wave][0] + ry * SPC + rx;
            float4 T[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) T[r][c] = tap[r * SPC + c];
            float2v Sx[2][2], Sy[2][2];   // [pixel][channel pair]
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    auto half = [&](const float4& v) { return h ? hi2(v) : lo2(v); };
                    // at(ox, oy) of the pixel: row 1 - oy, column q + 1 + ox of the taps
                    const float2v mm = half(T[2][q]), m0 = half(T[1][q]), mp = half(T[0][q]);
                    const float2v zm = half(T[2][q + 1]), zp = half(T[0][q + 1]);
                    const float2v pm = half(T[2][q + 2]), p0 = half(T[1][q + 2]), pp = half(T[0][q + 2]);
                    float2v d1 = ((mm + mp) - pm) - pp;
                    float2v d2 = m0 - p0;
                    float2v m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
                    Sx[q][h] = m1 + m2;
                    d1 = ((mm + pm) - mp) - pp;
                    d2 = zm - zp;
                    m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
                    Sy[q][h] = m1 + m2;
                }
            // quirk Q1: "channels" 1, 2 of the 1-channel group = elements (pixel + 1, + 2) of the flattened [B,H,W,1] slice, i.e.
            // channel 3 of the next two pixels.  Only the L1 norms of interior pixels use them, and for an interior pixel the taps
            // are unclamped: column + ch, which is staged unless it runs past the end of the image row (the last two interior
            // columns of the frame are corrected below: alias_wrap_fixup_pair).  Scharr of channel 3 at pixels xs + 2, xs + 3:
            float2v SxE = float2v{0.f, 0.f}, SyE = float2v{0.f, 0.f};
            if (!q1_intended) {   // (wave-uniform)
                float E[3][2];
#pragma unroll
                for (int r = 0; r < 3; ++r) { E[r][0] = tap[r * SPC + 4].w; E[r][1] = tap[r * SPC + 5].w; }
                const float2v mm = float2v{T[2][2].w, T[2][3].w}, m0 = float2v{T[1][2].w, T[1][3].w}, mp = float2v{T[0][2].w, T[0][3].w};
                const float2v zm = float2v{T[2][3].w, E[2][0]}, zp = float2v{T[0][3].w, E[0][0]};
                const float2v pm = float2v{E[2][0], E[2][1]}, p0 = float2v{E[1][0], E[1][1]}, pp = float2v{E[0][0], E[0][1]};
                float2v d1 = ((mm + mp) - pm) - pp;
                float2v d2 = m0 - p0;
                float2v m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
                SxE = m1 + m2;
                d1 = ((mm + pm) - mp) - pp;
                d2 = zm - zp;
                m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
                SyE = m1 + m2;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = 2 * s + q;
                // :203-208 per group, products and sums in the reference's order
                const float2v gl = lo2(gq[j]), gh = hi2(gq[j]);
                const float2v xl = gl * Sx[q][0], xh = gh * Sx[q][1], yl = gl * Sy[q][0], yh = gh * Sy[q][1];
                dLx[0][j] = (xl.x + xl.y) + xh.x; dLy[0][j] = (yl.x + yl.y) + yh.x;
                dLx[1][j] = xh.y; dLy[1][j] = yh.y;
                // :185: x if L1(Sx) > L1(Sy)
                const float l1x0 = (fabsf(Sx[q][0].x) + fabsf(Sx[q][0].y)) + fabsf(Sx[q][1].x);
                const float l1y0 = (fabsf(Sy[q][0].x) + fabsf(Sy[q][0].y)) + fabsf(Sy[q][1].x);
                horiz[0][q] = l1x0 > l1y0;
                const float a0x = fabsf(Sx[q][1].y), a0y = fabsf(Sy[q][1].y);
                const float a1x = q == 0 ? fabsf(Sx[1][1].y) : fabsf(SxE.x), a1y = q == 0 ? fabsf(Sy[1][1].y) : fabsf(SyE.x);
                const float a2x = q == 0 ? fabsf(SxE.x) : fabsf(SxE.y), a2y = q == 0 ? fabsf(SyE.x) : fabsf(SyE.y);
                const float l1x1 = q1_intended ? a0x : (a0x + a1x) + a2x;
                const float l1y1 = q1_intended ? a0y : (a0y + a1y) + a2y;
                horiz[1][q] = l1x1 > l1y1;
            }
            if (!q1_intended && x0 + ST + 3 > W) {  // workgroup-uniform: only tiles on the right image border
                uint32_t ib = 0;
#pragma unroll
                for (int q = 0; q < 2; ++q) ib |= (interior[q] && xs + q + 3 > W - 1) ? (1u << q) : 0u;
                if (__builtin_amdgcn_ballot_w64(ib != 0u) != 0ull) {
                    uint32_t bits = (horiz[1][0] ? 1u : 0u) | (horiz[1][1] ? 2u : 0u);
                    bits = alias_wrap_fixup_pair(p.pixels, p.B, H, W, C, iib, y, xs, 3, ib, bits);
                    horiz[1][0] = (bits & 1u) != 0u; horiz[1][1] = (bits & 2u) != 0u;
                }
            }
        }
        // ---- the pixel rows this slice was the last to read become (part of) the inbox: cleared now, written by the
        //      dilation below.  (LDS serves a wave's instructions in order: the clears cannot overtake the tap reads.) ----
        {
            float4* z = reinterpret_cast<float4*>(inbox);
            if (s == 0) {
                z[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                z[lane + 64] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < SI_FIRST / 2 - 128) z[lane + 128] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                if (lane < (SICELLS - SI_FIRST) / 2) z[SI_FIRST / 2 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }

        // ---- the pair and its six neighbours: clip_w and face (state tile row ry + 1, float2 column rx + 2) ----
        float w_up[2], w_dn[2], w_l, w_r;
        int f_own[2], f_up[2], f_dn[2], f_l, f_r;
        {
            const float2* rowp = &s_a[wave][ry + 1][rx + 2];
            const float4 a = *reinterpret_cast<const float4*>(rowp);
            w_own[2 * s] = a.x; f_own[0] = __float_as_int(a.y); w_own[2 * s + 1] = a.z; f_own[1] = __float_as_int(a.w);
            const float2 l = rowp[-1], r = rowp[2];
            w_l = l.x; f_l = __float_as_int(l.y); w_r = r.x; f_r = __float_as_int(r.y);
            const float4 c = *reinterpret_cast<const float4*>(rowp - SAC);
            w_up[0] = c.x; f_up[0] = __float_as_int(c.y); w_up[1] = c.z; f_up[1] = __float_as_int(c.w);
            const float4 e = *reinterpret_cast<const float4*>(rowp + SAC);
            w_dn[0] = e.x; f_dn[0] = __float_as_int(e.y); w_dn[1] = e.z; f_dn[1] = __float_as_int(e.w);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            covered[2 * s + q] = f_own[q] >= 0;
            key[2 * s + q] = f_own[q];   // (-1 = none)
        }

        // ---- dilation (:155-194) and position factors (:196-232), as in dirt_grad.hip: a pixel takes the fragment of the
        //      neighbour at +d, else at -d, when that neighbour is another face (:86-89) and closer (:165); d is +-x or +-y by
        //      the L1 norms (:185), the first attempt by the parity dither (:186-191).  The gradients of vertex k are
        //      b_k * (fx, fy, fw) with fx = dL_dx * (W/2) / w, fy = dL_dy * (H/2) / w, fw = -(fx * ndc_x + fy * ndc_y), everything
        //      taken at the pixel whose fragment is used; (fx, fy) are summed per such TARGET pixel -- own pixels in registers,
        //      neighbours through the wave's inbox (ds_add_f32 from the few dilated lanes) -- and fw is formed once per pixel. ----
        const int my_cell = (ry + 1) * SIS + rx + 2;   // the pair's first pixel in the inbox (an even cell: 16-byte aligned)
        const bool pos0 = ((xs + y) & 1) == 0;         // pixel 0 tries +x / up first (:186-191)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = 2 * s + q;
            fxy[j] = float2v{0.f, 0.f};
            const float wl = q == 0 ? w_l : w_own[2 * s], wr = q == 1 ? w_r : w_own[2 * s + 1];
            const int fl = q == 0 ? f_l : f_own[0], fr = q == 1 ? f_r : f_own[1];
            // (the pixel's own face as the state tile has it: an uncovered pixel, -1, differs from any face)
            const float wo = interior[q] ? w_own[j] : -INFINITY;   // pixels on the frame's border are never dilated (:155)
            const float qL = ((fl != f_own[q]) & (wo > wl)) ? wl : NO_NEIGHBOUR;
            const float qR = ((fr != f_own[q]) & (wo > wr)) ? wr : NO_NEIGHBOUR;
            const float qU = ((f_up[q] != f_own[q]) & (wo > w_up[q])) ? w_up[q] : NO_NEIGHBOUR;
            const float qD = ((f_dn[q] != f_own[q]) & (wo > w_dn[q])) ? w_dn[q] : NO_NEIGHBOUR;
            const bool pos = (q & 1) ? !pos0 : pos0;   // first attempt towards +x / up (:191), else -x / down
            const float qx1 = pos ? qR : qL, qx2 = pos ? qL : qR, qy1 = pos ? qU : qD, qy2 = pos ? qD : qU;
            const float rcp_own = __builtin_amdgcn_rcpf(w_own[j]);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                // direction: x if L1(Sx) > L1(Sy) else y (:185), negated on odd (x + y) (:186-190).  The reference's offsets are
                // in GL buffer orientation (y up): tensor row = y - offset_y.
                const bool hz = horiz[gi][q];
                const float q1 = hz ? qx1 : qy1, q2 = hz ? qx2 : qy2;
                const bool first = q1 == q1;                  // the first attempt found its neighbour (not the NaN sentinel)
                const bool dilated = first | (q2 == q2);      // ... or the opposite one did (:192-193)
                const float2v t = float2v{dLx[gi][j], dLy[gi][j]} * half_size;
                const float2v f = t * float2v{rcp_own, rcp_own};
                const bool own = covered[j] & !dilated;       // contributes to its own pixel
                fxy[j] += float2v{own ? f.x : 0.f, own ? f.y : 0.f};
                if (dilated) {  // few lanes: ds_add_f32 into the neighbour's cell, with the NEIGHBOUR's clip_w
                    const float rcp_w = __builtin_amdgcn_rcpf(first ? q1 : q2);
                    const float2v fn = t * float2v{rcp_w, rcp_w};
                    const int step = hz ? 1 : -SIS;           // +x, or up = the previous row
                    const bool fwd = first == pos;            // the neighbour taken lies at +x / up
                    float* cell = reinterpret_cast<float*>(inbox + (my_cell + q + (fwd ? step : -step)));
                    atomicAdd(cell, fn.x);
                    atomicAdd(cell + 1, fn.y);
                }
            }
        }
    };

    slice(std::integral_constant<int, 0>{});
#if DIRT_STREAM_MODE == 2   // no prefetch: slice 1 is requested when slice 0 is done (overlap only between waves)
    issue_slice1();
#endif
    SMARK();  // 3 slice 0 done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(gq[2].x), "+v"(gq[2].y), "+v"(gq[2].z), "+v"(gq[2].w), "+v"(gq[3].x), "+v"(gq[3].y), "+v"(gq[3].z), "+v"(gq[3].w));
    asm volatile("" : "+v"(bq[0].x), "+v"(bq[0].y), "+v"(bq[0].z), "+v"(bq[0].w), "+v"(bq[1].x), "+v"(bq[1].y), "+v"(bq[1].z), "+v"(bq[1].w));
    SMARK();  // 4 slice 1 landed
    slice(std::integral_constant<int, 1>{});
    SMARK();  // 5 slice 1 done

    // own barycentrics
    float bk[4][3];
    decode_bary(make_float2(bq[0].x, bq[0].y), bk[0]); decode_bary(make_float2(bq[0].z, bq[0].w), bk[1]);
    decode_bary(make_float2(bq[1].x, bq[1].y), bk[2]); decode_bary(make_float2(bq[1].z, bq[1].w), bk[3]);

    // ---- position totals of the lane's pixels (own sums + what the neighbours sent through the inbox, and fw of the totals)
    //      and the ring: what this wave's pixels sent to pixels of other waves (the row above / below the region, the column
    //      left / right of the tile).  Those pixels' faces take it through the face loop, at most two ring cells per lane:
    //      cells 0-33 the row above, 34-67 the row below, 68-75 / 76-83 the columns left / right. ----
    float2v fpos_xy[4];
    float fpos_w[4];
    int lkey[2];
    float lb[2][3], lf[2][3];
    {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ry = 4 * s + pr;
            const float4 i01 = *reinterpret_cast<const float4*>(inbox + ((ry + 1) * SIS + rx + 2));
            fpos_xy[2 * s] = fxy[2 * s] + float2v{i01.x, i01.y}; fpos_xy[2 * s + 1] = fxy[2 * s + 1] + float2v{i01.z, i01.w};
            const float ndc_y_own = ndc_of(H - 1 - (yw0 + ry), H, p.inv_h);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float ndc_x = ndc_of(xs + q, W, p.inv_w);
                fpos_w[2 * s + q] = -(fpos_xy[2 * s + q].x * ndc_x + fpos_xy[2 * s + q].y * ndc_y_own);
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = lane + 64 * e;
            const bool top = r < 34, bottom = r >= 34 && r < 68, left = r >= 68 && r < 76;
            const int ty = top ? -1 : (bottom ? 8 : (left ? r - 68 : r - 76));
            const int tx = top ? r - 1 : (bottom ? r - 35 : (left ? -1 : 32));
            lkey[e] = -1;
            lb[e][0] = 0.f; lb[e][1] = 0.f; lb[e][2] = 0.f; lf[e][0] = 0.f; lf[e][1] = 0.f; lf[e][2] = 0.f;
            if (r < SRING) {
                const float2 v = inbox[(ty + 1) * SIS + tx + 2];
                if (v.x != 0.f || v.y != 0.f) {  // only pixels inside the frame are ever sent anything
                    const int py = yw0 + ty, px = x0 + tx;
                    lkey[e] = __float_as_int(s_a[wave][ty + 1][tx + 2].y);
                    const float2 nb = ld_off<float2>(state_b, (uint32_t)((py - rowbase) * W + px) * 8u);
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
        SCOUNT(0, __popcll(__builtin_amdgcn_ballot_w64(lkey[0] >= 0)) + __popcll(__builtin_amdgcn_ballot_w64(lkey[1] >= 0)));
    }
    SMARK();  // 6 face loop starts

    // ---- the face loop (dirt_grad.hip): the two DPP rows of a PAIR (a 16 x 8 half of the region) walk the distinct faces among
    //      their pixels (key[j], -1 = none) and among the ring cells their lanes hold (lkey), both pairs of the wave at once.
    //      Per face every lane forms its masked partial sums -- per vertex k the 8 values b_k * (g_0 .. g_3, fx, fy, fw, 0) as
    //      four packed pairs: one v_pk_fma_f32 per pair and pixel -- the 24 sums are reduced over the lanes of each row
    //      (row_reduce_scatter), the two rows' totals joined (v_permlane16_swap) and ONE atomic instruction adds them to the
    //      face's three vertices. ----
    constexpr int S = 8, HP = 4, NV = 24, NR = 24;
    constexpr int IX = 4, IY = 5, IW = 6;
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
    // the factors of a pixel, in pairs: (g0, g1), (g2, g3), (fx, fy), (fw, 0)
    float2v fp[4][HP];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        fp[j][0] = lo2(gq[j]); fp[j][1] = hi2(gq[j]);
        fp[j][2] = fpos_xy[j];
        fp[j][3] = float2v{fpos_w[j], 0.f};
    }
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t pend[6];
    // ---- non-finite factors (a NaN / Inf in grad_pixels, in `pixels` through the Scharr filter, a degenerate clip_w).  The
    //      loop multiplies every pixel's factors by a barycentric that is ZEROED where the pixel is not of the pair's face:
    //      0 * NaN would carry one pixel's NaN into every face of its 16 x 8 half region, where the reference adds a pixel's
    //      terms to the vertices of its own face only (:140,228-230).  Such a pixel adds its 3 x 7 products itself -- the
    //      reference's own atomics, term for term -- and leaves the loop: factors zeroed, face struck off. ----
    bool gbk_done[4];   // grad_background of the pixel was written here (a non-finite uncovered pixel: its factors are zeroed for the loop)
    auto store_gbk = [&](int j) {
        const uint32_t off = own_rel(j >> 1) * 16u + (uint32_t)(j & 1) * 16u;
        st_off<float4>(gbk_t, off, covered[j] ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(fp[j][0].x, fp[j][0].y, fp[j][1].x, fp[j][1].y));
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
                        for (int c = 0; c < S - 1; ++c) {
                            const float val = bk[j][k] * ((c & 1) ? fp[j][c / 2].y : fp[j][c / 2].x);
                            float* dstp = c >= NCH
                                ? reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertices) + (size_t)((uint32_t)vk[k] * gv_row_bytes)) + (c == IW ? 3 : c - IX)
                                : reinterpret_cast<float*>(reinterpret_cast<char*>(grad_vertex_colors) + (size_t)((uint32_t)vk[k] * gvc_row_bytes)) + c;
                            atomicAdd(dstp, val);
                        }
                } else {   // uncovered: its colour factors are what grad_background gets -- stored now, before they are zeroed
                    store_gbk(j);
                    gbk_done[j] = true;
                }
#pragma unroll
                for (int h = 0; h < HP; ++h) fp[j][h] = float2v{0.f, 0.f};
            }
        }
    }
    pend[4] = (uint32_t)lkey[0]; pend[5] = (uint32_t)lkey[1];
    // the pair's next face: the smallest pending key of its 32 lanes (an all-lanes minimum by four DPP rotations and one swap
    // with the other row of the pair)
    auto next_face = [&]() {
        uint32_t K = min(min(min(pend[0], pend[1]), min(pend[2], pend[3])), min(pend[4], pend[5]));
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
        const lanemask live = __builtin_amdgcn_ballot_w64(K != NONE);   // pairs that still have a face
        if (live == 0ull) break;
        // the vertex this lane adds to (requested now, needed after the reduction)
        const uint32_t fbase = (K != NONE ? K : 0u) * 12u;
        const int vsel = ld_off<int32_t>(faces, fbase + 4u * (uint32_t)role_k);
        float2v accp[NR / 2];
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
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const lanemask mm = __builtin_amdgcn_ballot_w64(pend[4 + e] == K) & live;
            if (mm != 0ull) {
                const bool m = __builtin_amdgcn_inverse_ballot_w64(mm);
                pend[4 + e] = m ? NONE : pend[4 + e];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float bm = m ? lb[e][k] : 0.f;
                    accp[k * HP + IX / 2] = pk_fma_scalar(bm, float2v{lf[e][0], lf[e][1]}, accp[k * HP + IX / 2]);
                    accp[k * HP + IW / 2].x = fmaf(bm, lf[e][2], accp[k * HP + IW / 2].x);
                }
            }
        }
        SCOUNT(1, 1);
        const uint32_t K_next = next_face();
        float acc[NR];
#pragma unroll
        for (int i = 0; i < NR / 2; ++i) { acc[2 * i] = accp[i].x; acc[2 * i + 1] = accp[i].y; }
        float d0, d1;
        row_reduce_scatter<NR>(acc, lane, d0, d1);
        // the two rows of a pair worked on the same face: their totals, added (both rows get the sum)
        const auto s0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);
        d0 = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);
        const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d1), __float_as_uint(d1), false, false);
        d1 = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
        // (a pair without a face this iteration has all-zero totals)
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
    SMARK();  // 7 loop done

    // ---- background gradient (:143-147): grad_pixels where nothing is covered, zero elsewhere.  After the face loop (the
    //      stores of a wave then spread over the time in which the waves finish) and from the registers the loop's colour
    //      factors live in: the four lanes of a block row write the 128 bytes of eight pixels. ----
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!gbk_done[j]) store_gbk(j);
    SMARK();  // 8 done
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_grad_stream) {
        long long* o = g_trace_grad_stream + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16;
        for (int i = 0; i < 12; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
        o[12] = tr_c[0]; o[13] = tr_c[1];
        o[14] = tr_wall0; o[15] = (((long long)wall_clock64() - tr_wall0) << 20) | (long long)(__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4 /* HW_REG_HW_ID */) & 0xFFFFF);
    }
#endif
}

// Which launches take the streaming kernel: 4 channels, 16-byte aligned image tensors, whole 32 x 32 tiles (the DMA pieces
// carry no partial-tile masks; a pair of state pixels is one 16-byte unit: W even), no diagnostic output.
bool grad_stream_eligible(const GradParams& p)
{
    return p.C == 4 && p.pixels_aligned16 != 0 && (p.W % ST) == 0 && (p.H % ST) == 0 && p.debug_thingy == nullptr &&
           (reinterpret_cast<uintptr_t>(p.state_a) & 15u) == 0 && (reinterpret_cast<uintptr_t>(p.state_b) & 15u) == 0;
}

hipError_t launch_grad_stream(const GradParams& p, hipStream_t stream)
{
    // (p as filled by launch_grad: tiles_x / tiles_y are those of 32 x 32 tiles)
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)p.B), block(STHREADS);
    hipLaunchKernelGGL(grad_kernel_stream, grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace dirt
