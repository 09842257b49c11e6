// dirt_grad_small.hip -- the gradient assembly kernel for SMALL frames (gfx950): one pixel per lane, 16 x 16 tiles.
//
// Same contract as grad_kernel (dirt_grad.hip; replaces assemble_grads, csrc/rasterise_grad_egl.cu:93-236, for all
// channel groups of dirt/rasterise_ops.py:145-165 in one launch), other shape.  grad_kernel gives a lane a 4 x 1 strip
// and a wave 32 x 8 pixels: on a 256 x 256 frame that is 64 workgroups for 256 compute units, every wave alone on its
// SIMD (a lone wave issues a dependent instruction every ~8 cycles), walking the ~18 faces of its region one (pair
// of) face(s) per iteration: 20 of that configuration's 42 microseconds.  Here a lane owns ONE pixel, a DPP row of 16
// lanes a 4 x 4 block, a wave 8 x 8 pixels, a workgroup a 16 x 16 tile: four times the waves (every SIMD of the chip
// busy on a 256 x 256 frame), each with ~3 faces per block instead of ~18 per region.  Every row walks its own faces
// (row_reduce_scatter: the totals of a row's face land in its 16 lanes, one or two atomic instructions add them to the
// face's vertices).  The price is more float atomics per pixel (one group per (4 x 4 block, face)), which is why the
// library launches this shape only where the frame is small (launch_grad).
//
// Per-pixel arithmetic (Scharr, dilation, position factors) is that of dirt_grad.hip -- same operation order where the
// reference's result depends on it (the L1 norms that choose the dilation axis) -- written per pixel instead of per
// strip; variable names follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "dirt_reduce.h"
#include "dirt_grad_common.h"
#include "../../include/dirt_hip.h"

namespace dirt {

#ifdef DIRT_TRACE
// Per-wave phase timestamps for tools/trace_grad.py (the layout of dirt_grad.hip's trace); tracing build only.
__device__ long long* g_trace_grad_small = nullptr;
extern "C" void dirt_debug_set_trace_grad_small(void* p)
{
    long long* q = reinterpret_cast<long long*>(p);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_grad_small), &q, sizeof(q));
}
#define SMARK() do { if (tr_n < 12) { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tr_t[tr_n++] = t_; } } while (0)
#define SCOUNT(i, v) do { tr_c[i] += (v); } while (0)
#else
#define SMARK() do {} while (0)
#define SCOUNT(i, v) do {} while (0)
#endif

namespace {

constexpr int ST = 16;            // tile side
constexpr int SROWS = ST + 2;     // staged rows y0-1 .. y0+16
constexpr int SCOLS = ST + 4;     // staged columns x0-1 .. x0+18: a single channel's aliased "channels" (quirk Q1) are the next two pixels
constexpr int SIB = 10;           // inbox row stride: cell (ty + 1) * 10 + tx + 1 for ty, tx in -1..8 (the wave's 8 x 8 region + ring)
constexpr int SRING = 36;         // ring cells of a region

// Scharr response of one "channel" from its 3 x 3 taps w[r][i] = (row y - 1 + r, column x - 1 + i), operation for operation
// as csrc/rasterise_grad_egl.cu:126-127 (negative-offset minus positive-offset; offset_y is up = the previous tensor row).
__device__ __forceinline__ void scharr(const float (&w)[3][3], float& sx, float& sy)
{
    const float mm = w[2][0], m0 = w[1][0], mp = w[0][0];
    const float zm = w[2][1], zp = w[0][1];
    const float pm = w[2][2], p0 = w[1][2], pp = w[0][2];
    float d1 = ((mm + mp) - pm) - pp;
    float d2 = m0 - p0;
    float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
    sx = m1 + m2;
    d1 = ((mm + pm) - mp) - pp;
    d2 = zm - zp;
    m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
    sy = m1 + m2;
}

}  // namespace

// grad_kernel_px1<CSPEC, DEBUG>: the image has CSPEC = 1, 3 or 4 channels (4 = a 3-channel group and a single).
template <int CSPEC, bool DEBUG>
__global__ __launch_bounds__(256) void grad_kernel_px1(GradParams p)
{
    static_assert(CSPEC == 1 || CSPEC == 3 || CSPEC == 4, "channel counts with a specialised small-frame kernel");
    constexpr int NCH = CSPEC;
    constexpr int G0 = CSPEC == 1 ? 1 : 3;           // size of the first channel group
    constexpr int NG = 1 + (NCH - G0);               // channel groups
    __shared__ __align__(16) float s_pix[SROWS][SCOLS][CSPEC == 3 ? 4 : CSPEC];  // pixels of the halo'd tile, edge clamped (at(), :113-124)
    __shared__ __align__(16) float2 s_vw[SROWS][SROWS];                          // {clip_w, face}
    __shared__ __align__(16) float2 s_inbox[4][SIB * SIB + 2];                   // per wave: (fx, fy) sent to each pixel of its region + ring
    constexpr int LC = CSPEC == 3 ? 4 : CSPEC;       // floats per staged pixel

#ifdef DIRT_TRACE
    long long tr_t[12]; int tr_n = 0; long long tr_c[4] = {0, 0, 0, 0};
    const long long tr_wall0 = wall_clock64();
#endif
    SMARK();  // 0 start
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int iib = blockIdx.y;
    const int H = p.H, W = p.W;
    constexpr int C = CSPEC;
    const int tile = xcd_tile((int)blockIdx.x, p.tiles_x * p.tiles_y);
    int tile_col, tile_row;
    tile_xy(tile, p.tiles_x, p.tiles_x_magic, tile_col, tile_row);
    const int x0 = tile_col * ST, y0 = tile_row * ST;
    const size_t frame = (size_t)H * W;
    const float2* __restrict__ state_a = p.state_a + (size_t)iib * frame;
    const float2* __restrict__ state_b = p.state_b + (size_t)iib * frame;
    const float* __restrict__ pixels = p.pixels + (size_t)iib * frame * C;
    const float* __restrict__ gpix = p.grad_pixels + (size_t)iib * frame * C;
    float* __restrict__ gbk = p.grad_background + (size_t)iib * frame * C;
    const int32_t* __restrict__ faces = p.faces + (p.shared_faces ? (size_t)0 : (size_t)iib * p.F * 3);
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * p.gv_stride;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * p.gvc_stride;
    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;
    const bool wide = p.pixels_aligned16 != 0;   // 16-byte accesses to the [B,H,W,4] tensors

    // ---- staging: the halo'd tile of `pixels` and of the visibility surface, every load issued before the barrier ----
    {
        constexpr int NPOS = SROWS * SCOLS, ITEMS = (NPOS + 255) / 256;
        float v[ITEMS][LC];
        float2 rec[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int n = min(tid + 256 * k, NPOS - 1);
            const int row = n / SCOLS, col = n - row * SCOLS;
            const int cy = min(max(y0 - 1 + row, 0), H - 1), cx = min(max(x0 - 1 + col, 0), W - 1);
            const size_t pix = (size_t)cy * W + cx;
            bool done = false;
            if constexpr (CSPEC == 4) {
                if (wide) {
                    const float4 q = *reinterpret_cast<const float4*>(pixels + pix * 4);
                    v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = q.w;
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int ch = 0; ch < LC; ++ch) v[k][ch] = ch < NCH ? pixels[pix * C + ch] : 0.f;
            }
            rec[k] = state_a[pix];
        }
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int n = tid + 256 * k;
            if (n >= NPOS) continue;
            const int row = n / SCOLS, col = n - row * SCOLS;
#pragma unroll
            for (int ch = 0; ch < LC; ++ch) s_pix[row][col][ch] = v[k][ch];
            if (col < SROWS) s_vw[row][col] = rec[k];
        }
    }
    float2* const inbox = &s_inbox[wave][0];
    if (lane < (SIB * SIB + 2) / 2) reinterpret_cast<float4*>(inbox)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- this lane's pixel: DPP row = 4 x 4 block (lane >> 4), lane & 15 = position inside it ----
    const int blk = lane >> 4;
    const int lx = 8 * (wave & 1) + 4 * (blk & 1) + (lane & 3);      // in the tile
    const int ly = 8 * (wave >> 1) + 4 * (blk >> 1) + ((lane >> 2) & 3);
    const int x = x0 + lx, y = y0 + ly;
    const bool in_px = (x < W) & (y < H);
    const bool interior = in_px & (x > 0) & (y > 0) & (x < W - 1) & (y < H - 1);
    const size_t own = (size_t)min(y, H - 1) * W + min(x, W - 1);
    const int rx = 4 * (blk & 1) + (lane & 3), ry = 4 * (blk >> 1) + ((lane >> 2) & 3);   // in the wave's region
    const int my_cell = (ry + 1) * SIB + rx + 1;

    float g[NCH];
    {
        bool done = false;
        if constexpr (CSPEC == 4) {
            if (wide) {
                const float4 q = *reinterpret_cast<const float4*>(gpix + own * 4);
                g[0] = q.x; g[1] = q.y; g[2] = q.z; g[3] = q.w;
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) g[ch] = gpix[own * C + ch];
        }
    }
    float bk[3];
    decode_bary(state_b[own], bk);
    SMARK();  // 1 loads issued, tile stored
    SMARK();  // 2
    __syncthreads();
    SMARK();  // 3 barrier passed

    // ---- Scharr, the L1 norms that choose the dilation axis (:185, all three "channels" of the reference's Vec3 in its
    //      summation order) and dL/dx, dL/dy of :203-208, per channel group ----
    bool horiz[NG];
    float dLx[NG], dLy[NG];
    {
        float l1x = 0.f, l1y = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int gi = ch < G0 ? 0 : ch - G0 + 1;
            const bool single = !(ch < G0 && G0 == 3);
            float w[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 3; ++i) w[r][i] = s_pix[ly + r][lx + i][ch];
            float sx, sy;
            scharr(w, sx, sy);
            if (!single) {
                float m = g[ch] * sx;
                dLx[gi] = ch == 0 ? m : dLx[gi] + m;
                m = g[ch] * sy;
                dLy[gi] = ch == 0 ? m : dLy[gi] + m;
                l1x = ch == 0 ? fabsf(sx) : l1x + fabsf(sx);
                l1y = ch == 0 ? fabsf(sy) : l1y + fabsf(sy);
                if (ch == G0 - 1) horiz[gi] = l1x > l1y;
            } else {
                dLx[gi] = g[ch] * sx;
                dLy[gi] = g[ch] * sy;
                // quirk Q1 (:119-123): "channels" 1, 2 of a 1-channel group are elements (pixel + 1, + 2) of the flattened
                // [B,H,W,1] slice: for an interior pixel the same channel of the next two pixels of the row (staged; past the
                // end of the image row they lie in the next one: corrected from memory below)
                float a1x = 0.f, a1y = 0.f, a2x = 0.f, a2y = 0.f;
                if (!q1_intended) {
                    float w1[3][3], w2[3][3];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int i = 0; i < 3; ++i) { w1[r][i] = s_pix[ly + r][lx + 1 + i][ch]; w2[r][i] = s_pix[ly + r][lx + 2 + i][ch]; }
                    scharr(w1, a1x, a1y);
                    scharr(w2, a2x, a2y);
                }
                const float a0x = fabsf(sx), a0y = fabsf(sy);
                const float sl1x = q1_intended ? a0x : (a0x + fabsf(a1x)) + fabsf(a2x);
                const float sl1y = q1_intended ? a0y : (a0y + fabsf(a1y)) + fabsf(a2y);
                bool hz = sl1x > sl1y;
                if (!q1_intended && x0 + ST + 3 > W) {   // workgroup-uniform: only tiles on the right image border
                    const uint32_t ib = (interior && x + 3 > W - 1) ? 1u : 0u;
                    if (__builtin_amdgcn_ballot_w64(ib != 0u) != 0ull)
                        hz = (alias_wrap_fixup(p.pixels, p.B, H, W, C, iib, y, x, ch, ib, hz ? 1u : 0u, 0) & 1u) != 0u;
                }
                horiz[gi] = hz;
            }
        }
    }

    SMARK();  // 4 Scharr done
    // ---- the pixel and its four neighbours: clip_w and face ----
    const float2 s_own = s_vw[ly + 1][lx + 1], s_l = s_vw[ly + 1][lx], s_r = s_vw[ly + 1][lx + 2];
    const float2 s_u = s_vw[ly][lx + 1], s_d = s_vw[ly + 2][lx + 1];
    const float w_own = s_own.x;
    const int f_own = __float_as_int(s_own.y);
    const bool covered = in_px & (f_own >= 0);
    // The vertex indices of this pixel's face, requested now: the face loop needs the indices of the face a row works on,
    // and every face a row meets is the face of one of its lanes (or ring cells), so they are passed on by cross-lane
    // maxima instead of a load per iteration -- a wave is about alone on its SIMD here, nothing would cover that round trip
    // (per-wave trace at K3-256: 1727 clocks per iteration with the load, 7.3 iterations per wave).
    struct Int3 { int32_t x, y, z; };
    Int3 vid_own = {0, 0, 0};
    if (p.F > 0) vid_own = *reinterpret_cast<const Int3*>(faces + (size_t)(covered ? f_own : 0) * 3);   // (F == 0: `faces` may be null)

    // ---- background gradient (:143-147): grad_pixels where nothing is covered, zero elsewhere ----
    if (in_px) {
        bool done = false;
        if constexpr (CSPEC == 4) {
            if (wide) {
                *reinterpret_cast<float4*>(gbk + own * 4) = covered ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(g[0], g[1], g[2], g[3]);
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) gbk[own * C + ch] = covered ? 0.f : g[ch];
        }
    }

    // ---- dilation (:155-194) and position factors (:196-232), as in dirt_grad.hip: the gradients of vertex k are
    //      b_k * (fx, fy, fw) taken at the pixel whose fragment is used (this one, or the closer neighbour of another face it
    //      is dilated from); (fx, fy) are summed per such target pixel -- own in registers, neighbours through the wave's
    //      inbox (whose ring collects what belongs to pixels of other waves) -- and fw is formed once per pixel ----
    const bool ok_l = interior & (__float_as_int(s_l.y) != f_own) & (w_own > s_l.x);
    const bool ok_r = interior & (__float_as_int(s_r.y) != f_own) & (w_own > s_r.x);
    const bool ok_u = interior & (__float_as_int(s_u.y) != f_own) & (w_own > s_u.x);
    const bool ok_d = interior & (__float_as_int(s_d.y) != f_own) & (w_own > s_d.x);
    const bool pos = ((x + y) & 1) == 0;   // first attempt towards +x / up (:186-191)
    const float half_w = .5f * (float)W, half_h = .5f * (float)H;
    float fx = 0.f, fy = 0.f;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const bool hz = horiz[gi];
        const bool ok_a = hz ? (pos ? ok_r : ok_l) : (pos ? ok_u : ok_d);
        const bool ok_b = hz ? (pos ? ok_l : ok_r) : (pos ? ok_d : ok_u);
        const bool dil = ok_a | ok_b;            // the opposite direction if the first failed (:192-193)
        const bool fwd = pos == ok_a;            // the neighbour taken lies at +x / up
        const float clip_w = dil ? (hz ? (fwd ? s_r.x : s_l.x) : (fwd ? s_u.x : s_d.x)) : w_own;
        if constexpr (DEBUG) {
            if (gi == 0 && in_px) write_debug(p.debug_thingy, p.grad_pixels, p.B, H, W, C, iib, y, x, G0, dil);
        }
        const float rcp_w = __builtin_amdgcn_rcpf(clip_w);
        const float f0 = (dLx[gi] * half_w) * rcp_w, f1 = (dLy[gi] * half_h) * rcp_w;
        if (covered & !dil) { fx += f0; fy += f1; }
        if (dil) {
            const int step = hz ? 1 : -SIB;   // +x, or up = the previous row
            float* cell = reinterpret_cast<float*>(inbox + (my_cell + (fwd ? step : -step)));
            atomicAdd(cell, f0);
            atomicAdd(cell + 1, f1);
        }
    }

    SMARK();  // 5 dilation done
    // ---- totals per target pixel: own + what the neighbours sent; the ring cells (targets outside this wave's region) ----
    const float2 in_own = inbox[my_cell];
    const float px_x = fx + in_own.x, px_y = fy + in_own.y;
    const float ndc_x = ndc_of(x, W, p.inv_w), ndc_y = ndc_of(H - 1 - y, H, p.inv_h);
    const float px_w = -(px_x * ndc_x + px_y * ndc_y);
    int lkey = -1;
    Int3 lvid = {0, 0, 0};
    float lb[3] = {0.f, 0.f, 0.f}, lf[3] = {0.f, 0.f, 0.f};
    if (lane < SRING) {
        const int r = lane;
        const int ty = r < 10 ? -1 : (r < 20 ? 8 : (r < 28 ? r - 20 : r - 28));
        const int tx = r < 10 ? r - 1 : (r < 20 ? r - 11 : (r < 28 ? -1 : 8));
        const float2 v = inbox[(ty + 1) * SIB + tx + 1];
        if (v.x != 0.f || v.y != 0.f) {   // only pixels inside the frame are ever sent anything
            const int tly = 8 * (wave >> 1) + ty, tlx = 8 * (wave & 1) + tx;   // in the tile
            const int py = y0 + tly, pxx = x0 + tlx;
            lkey = __float_as_int(s_vw[tly + 1][tlx + 1].y);
            lvid = *reinterpret_cast<const Int3*>(faces + (size_t)lkey * 3);
            decode_bary(state_b[(size_t)py * W + pxx], lb);
            const float nx = ndc_of(pxx, W, p.inv_w), ny = ndc_of(H - 1 - py, H, p.inv_h);
            lf[0] = v.x; lf[1] = v.y; lf[2] = -(v.x * nx + v.y * ny);
        }
    }

    // ---- the face loop: every DPP row walks the distinct faces of its 4 x 4 block (and of the ring cells its lanes hold),
    //      smallest key first; per face the S = NCH + 3 values of each vertex -- b_k * (g_0 .. g_NCH-1, fx, fy, fw) -- are
    //      reduced over the row and added to the face's vertices by the lanes the totals land in ----
    constexpr int S = NCH + 3, NV = 3 * S, NR = NV <= 16 ? 16 : 24, NROLES = NR == 24 ? 2 : 1;
    float fval[S];
#pragma unroll
    for (int c = 0; c < NCH; ++c) fval[c] = g[c];
    fval[NCH] = px_x; fval[NCH + 1] = px_y; fval[NCH + 2] = px_w;
    int rv[2];
    row_value_of_lane<NR>(lane & 15, rv[0], rv[1]);
    int role_k[NROLES];
    bool role_valid[NROLES];
    float* role_base[NROLES];
    uint32_t role_stride[NROLES];
#pragma unroll
    for (int e = 0; e < NROLES; ++e) {
        const int v = rv[e];
        const bool ok = v >= 0 && v < NV;
        const int c = ok ? v % S : 0;
        role_k[e] = ok ? v / S : 0;
        role_valid[e] = ok;
        const bool is_pos = c >= NCH;
        role_base[e] = is_pos ? grad_vertices + (c - NCH == 2 ? 3 : c - NCH) : grad_vertex_colors + c;   // (.z is never written, :228-230)
        role_stride[e] = 4u * (uint32_t)(is_pos ? p.gv_stride : p.gvc_stride);
    }
    // the lane's products b_k * (g.., fx, fy, fw) of its pixel and b_k * (fx, fy, fw) of its ring cell: loop invariant
    float prod_own[3][S], prod_ring[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int c = 0; c < S; ++c) prod_own[k][c] = bk[k] * fval[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) prod_ring[k][c] = lb[k] * lf[c];
    }
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t pend0 = covered ? (uint32_t)f_own : NONE, pend1 = (uint32_t)lkey;
    auto next_face = [&]() {
        uint32_t K = min(pend0, pend1);
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x128 /* row_ror:8 */, 0xF, 0xF, true));
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x124 /* row_ror:4 */, 0xF, 0xF, true));
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x122 /* row_ror:2 */, 0xF, 0xF, true));
        K = min(K, (uint32_t)__builtin_amdgcn_mov_dpp((int)K, 0x121 /* row_ror:1 */, 0xF, 0xF, true));
        return K;
    };
    SCOUNT(0, __popcll(__builtin_amdgcn_ballot_w64(lkey >= 0)));
    SMARK();  // 6 face loop starts
    uint32_t K = next_face();
    for (;;) {
        if (__builtin_amdgcn_ballot_w64(K != NONE) == 0ull) break;
        SCOUNT(1, 1);
        const bool live = K != NONE;
        const bool m0 = live & (pend0 == K), m1 = live & (pend1 == K);
        // the face's three vertex indices, from whichever lanes of the row hold it (+1: 0 = "not mine"); all holders agree
        uint32_t rvid[3];
        {
            const uint32_t h[3] = {m0 ? (uint32_t)vid_own.x + 1u : (m1 ? (uint32_t)lvid.x + 1u : 0u),
                                   m0 ? (uint32_t)vid_own.y + 1u : (m1 ? (uint32_t)lvid.y + 1u : 0u),
                                   m0 ? (uint32_t)vid_own.z + 1u : (m1 ? (uint32_t)lvid.z + 1u : 0u)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint32_t r = h[k];
                r = max(r, (uint32_t)__builtin_amdgcn_mov_dpp((int)r, 0x128 /* row_ror:8 */, 0xF, 0xF, true));
                r = max(r, (uint32_t)__builtin_amdgcn_mov_dpp((int)r, 0x124 /* row_ror:4 */, 0xF, 0xF, true));
                r = max(r, (uint32_t)__builtin_amdgcn_mov_dpp((int)r, 0x122 /* row_ror:2 */, 0xF, 0xF, true));
                r = max(r, (uint32_t)__builtin_amdgcn_mov_dpp((int)r, 0x121 /* row_ror:1 */, 0xF, 0xF, true));
                rvid[k] = r != 0u ? r - 1u : 0u;   // (a row without a face this iteration: vertex 0, and `live` keeps it from adding anything)
            }
        }
        int vsel[NROLES];
#pragma unroll
        for (int e = 0; e < NROLES; ++e) vsel[e] = (int)(role_k[e] == 0 ? rvid[0] : (role_k[e] == 1 ? rvid[1] : rvid[2]));
        pend0 = m0 ? NONE : pend0;
        pend1 = m1 ? NONE : pend1;
        // (selects of the lane's own products, not products with a zeroed factor: a non-finite grad_pixels / position
        // factor of a pixel that is NOT of this face must not reach the face's totals -- 0 * NaN -- where the reference adds
        // a pixel's terms to its own face's vertices only, csrc/rasterise_grad_egl.cu:140,228-230)
        float acc[NR];
#pragma unroll
        for (int i = NV; i < NR; ++i) acc[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int c = 0; c < S; ++c) acc[k * S + c] = m0 ? prod_own[k][c] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[k * S + NCH + c] += m1 ? prod_ring[k][c] : 0.f;
        }
        const uint32_t K_next = next_face();
        float d0, d1;
        row_reduce_scatter<NR>(acc, lane, d0, d1);
        float total[NROLES];
        total[0] = d0;
        if (NROLES == 2) total[NROLES - 1] = d1;
        // (addresses before the branches: see the face loop of dirt_grad.hip)
        float* dst[NROLES];
#pragma unroll
        for (int e = 0; e < NROLES; ++e) {
            dst[e] = reinterpret_cast<float*>(reinterpret_cast<char*>(role_base[e]) + (size_t)((uint32_t)vsel[e] * role_stride[e]));
            asm volatile("" : "+v"(dst[e]));
        }
#pragma unroll
        for (int e = 0; e < NROLES; ++e)
            if (role_valid[e] && live && total[e] != 0.f)
                asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(dst[e]), "v"(total[e]) : "memory");
        K = K_next;
    }
    SMARK();  // 7 done
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_grad_small) {
        long long* o = g_trace_grad_small + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16;
        for (int i = 0; i < 12; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
        o[12] = tr_c[0]; o[13] = tr_c[1];
        o[14] = tr_wall0; o[15] = (((long long)wall_clock64() - tr_wall0) << 20) | (long long)(__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4 /* HW_REG_HW_ID */) & 0xFFFFF);
    }
#endif
}

hipError_t launch_grad_small(const GradParams& p, hipStream_t stream)
{
    GradParams q = p;
    q.tiles_x = (p.W + ST - 1) / ST;
    q.tiles_y = (p.H + ST - 1) / ST;
    q.tiles_x_magic = tile_magic(q.tiles_x);
    const dim3 grid((unsigned)(q.tiles_x * q.tiles_y), (unsigned)p.B), block(256);
#define DIRT_LAUNCH_SMALL(C_)                                                                        \
    do {                                                                                             \
        if (p.debug_thingy) hipLaunchKernelGGL((grad_kernel_px1<C_, true>), grid, block, 0, stream, q);   \
        else hipLaunchKernelGGL((grad_kernel_px1<C_, false>), grid, block, 0, stream, q);                 \
    } while (0)
    if (p.C == 4) DIRT_LAUNCH_SMALL(4);
    else if (p.C == 3) DIRT_LAUNCH_SMALL(3);
    else DIRT_LAUNCH_SMALL(1);
#undef DIRT_LAUNCH_SMALL
    return hipGetLastError();
}

}  // namespace dirt
