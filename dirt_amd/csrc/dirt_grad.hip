// dirt_grad.hip -- gradient assembly kernel for gfx950.
//
// Replaces assemble_grads / launch_grad_assembly (csrc/rasterise_grad_egl.cu:93-278) and, by
// evaluating every channel group of dirt/rasterise_ops.py:145-165 inside one launch, the N
// per-group RasteriseGrad ops (and N GL re-draws) the reference issues for C not in {1,3}.
//
// Inputs: the visibility buffer (front-most face per pixel) and the fragment buffer ((b0,b1,b2,clip_w) per
// pixel), both written by the raster kernel -- the counterpart of the reference's two RGBA32F surfaces
// (csrc/rasterise_grad_egl.cpp:432-456), produced by the forward pass itself when it keeps its state.
//
// The reference issues up to 3C+9 global float atomics per covered pixel (:140,228-230) and reads
// its 3x3 neighbourhood with 27 scalar loads.  Here one workgroup owns a 32 x GH tile (each wave one
// 8x8 block, one pixel per lane); channel groups are processed in turn, the group's channels of the
// `pixels` tile (+halo) staged in LDS, and per-face partial sums are accumulated in LDS:
//   * the faces that receive gradient in the tile get a slot in a small LDS hash table (LDS CAS);
//   * each value is first summed over the 4 lanes of a 4x1 pixel quad with two DPP quad_perm adds
//     when the quad targets one face (the common case);
//   * the sums are accumulated in FIXED POINT with 64-bit integer LDS atomics.  Measured on MI355X
//     (tools/lds_atomic_bench.hip): ds_add_f32 is serialised per active lane (~3 clk/lane for the
//     whole CU, whatever the addresses), ds_add_u64 runs at full rate for distinct addresses and
//     2 clk/lane for equal ones.  The power-of-two scale comes from tile-wide bounds that are known
//     before any contribution is computed (max |grad_pixels|, max |pixels|, max 1/clip_w over the
//     tile: three LDS atomicMax per wave), so no barrier separates computing a contribution from
//     adding it; contributions keep >= 2^-20 relative precision against the tile's largest one in
//     practice and the tile sum is exact and order independent;
//   * per pass (<= 4 channels = the channel groups that fit) the replicas are summed and ONE global
//     float atomic per (face, vertex, component) is issued for the whole tile.  Faces that do not
//     fit the slot table, and tiles that see an inf / NaN, fall back to the reference's direct
//     float atomics.
// Variable names in the per-pixel arithmetic follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "../../include/dirt_hip.h"
#include <type_traits>
#include <cstdlib>

namespace dirt {

#ifdef DIRT_TRACE
__device__ long long* g_trace_grad = nullptr;
extern "C" void dirt_debug_set_trace_grad(void* p)
{
    long long* q = reinterpret_cast<long long*>(p);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_grad), &q, sizeof(q));
}
#define GMARK() do { if (tr_n < 16) tr_t[tr_n++] = clock64(); } while (0)
#else
#define GMARK() do {} while (0)
#endif

// grad_kernel<GH, COPIES>: a tile is 32 x GH pixels, one pixel per lane, GH / 8 rows of four 8 x 8 blocks.
//   <16, 4>  the normal shape: 8 waves, 71 KB of LDS, two workgroups per CU;
//   <8, 2>   for small frames and dense meshes: twice the workgroups, and half the faces per tile, so the 64-slot
//            table does not overflow where triangles are only a few pixels large; 38 KB of LDS, four workgroups per CU.
constexpr int GW = 32;
constexpr int PWU = GW + 4;            // staged `pixels` columns: x0-1 .. x0+34 (halo + 2 for the Q1 alias taps)
constexpr int PW = 40;                 // ... padded: row stride = 8 (mod 32) dwords keeps an 8x8 block's reads conflict free
constexpr int VWU = GW + 2;            // visibility tile with a 1-pixel halo
constexpr int VW = 40;                 // ... padded likewise (32 (mod 64) dwords for the float4 rows)
constexpr int MAX_SLOTS = 64;          // slot table capacity (LDS)
constexpr int PC = 4;                  // channels per pass: whole channel groups that fit in 4 channels
constexpr int NVAL = 9 + 3 * PC;       // 9 position values (3 vertices x {x,y,w}) + 3 vertices x PC colour values
#ifndef GRAD_WAVES_PER_SIMD
#define GRAD_WAVES_PER_SIMD 4
#endif
static_assert(MAX_SLOTS == 64, "the slot bookkeeping uses one wave for the table");
constexpr int FIX_BITS = 32;           // fixed-point contributions: |q| <= 2^(FIX_BITS-2) (the bounds carry 2x slack), an int32

__device__ __forceinline__ float quad_sum(float v)
{
    // v_add_f32 with DPP quad_perm: lanes 4q..4q+3 all end up with the sum of the quad
    float t = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // [1,0,3,2]
    v = v + t;
    t = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));        // [2,3,0,1]
    return v + t;
}

// Maximum of a non-negative 32-bit pattern over the wave (DPP within rows of 16, then across rows).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
    x = max(x, (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x = max(x, (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x = max(x, (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true));  // row_half_mirror
    x = max(x, (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true));  // row_mirror
    const uint32_t a = __builtin_amdgcn_readlane((int)x, 0), b = __builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = __builtin_amdgcn_readlane((int)x, 32), d = __builtin_amdgcn_readlane((int)x, 48);
    return max(max(a, b), max(c, d));
}

// Open-addressing insert of `face` into the tile's slot table; returns the slot or -1 when full.  `claimed` is
// set for the one thread whose CAS created the slot.
__device__ inline int slot_insert(int32_t* keys, int nslots, int face, bool& claimed)
{
    claimed = false;
    uint32_t h = ((uint32_t)face * 2654435761u) % (uint32_t)nslots;
    for (int probe = 0; probe < nslots; ++probe) {
        const int32_t cur = __hip_atomic_load(keys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == face) return (int)h;
        if (cur == -1) {
            const int32_t prev = atomicCAS(&keys[h], -1, face);
            if (prev == -1) { claimed = true; return (int)h; }
            if (prev == face) return (int)h;
        }
        h = (h + 1 == (uint32_t)nslots) ? 0u : h + 1;
    }
    return -1;
}

// What one lane adds for one target face: after the 4x1 quad pre-reduction either the quad leader
// adds the quad's sum (when the quad targets one slot) or every lane adds its own value.
struct Target {
    int slot;     // LDS slot, -1 = none (no contribution), -2 = table full: direct global atomics
    bool uniform; // the lane's 4x1 quad targets one slot
    bool active;  // this lane issues the LDS adds
    int copy;     // accumulator replica
};

template <int COPIES>
__device__ __forceinline__ Target make_target(int slot, int lane)
{
    Target t;
    t.slot = slot;
    const int s0 = __builtin_amdgcn_mov_dpp(slot, 0x00, 0xF, 0xF, true);  // quad_perm [0,0,0,0]
    const unsigned long long m = __builtin_amdgcn_ballot_w64(slot == s0);
    t.uniform = (((m >> (lane & ~3)) & 0xFull) == 0xFull) & (slot != -2);  // -2 lanes may belong to different faces
    t.active = (slot >= 0) & (!t.uniform | ((lane & 3) == 0));
    t.copy = (lane >> 2) & (COPIES - 1);
    return t;
}

// Quad pre-reduction of one contribution (0 for lanes that contribute nothing).  Must be called by
// all 64 lanes (DPP).  Returns what this lane will add (meaningful where t.active).
__device__ __forceinline__ float quad_reduce(const Target& t, float v)
{
    const float q = quad_sum(v);
    return t.uniform ? q : v;
}

// Power-of-two scale for fixed-point accumulation.  `bound` (> 0, finite) is TWICE a rigorous bound on the quad-summed
// contributions, so |v * to_fix| < 2^(FIX_BITS - 2) = 2^30: an int32 with a bit to spare, at the finest resolution the
// one-instruction float -> int32 conversion allows.
struct FixScale {
    float to_fix;    // 2^(FIX_BITS - 1 - E), E = exponent of the bound
    float from_fix;  // its inverse
    bool finite;     // false: inf / NaN in the tile -> direct float atomics keep IEEE semantics
};

__device__ __forceinline__ FixScale fix_scale(float bound)
{
    const uint32_t bits = __float_as_uint(bound) & 0x7FFFFFFFu;
    uint32_t e = bits >> 23;  // biased exponent: |v| <= bound < 2^(e - 126)
    FixScale f;
    f.finite = e < 255u;
    e = min(max(e, 32u), 220u);
    f.to_fix = __uint_as_float((uint32_t)(127 + FIX_BITS - 1 + 126 - (int)e) << 23);
    f.from_fix = __uint_as_float((uint32_t)(127 - (FIX_BITS - 1) - 126 + (int)e) << 23);
    return f;
}

// N fixed-point adds (values idx0 .. idx0+N-1 of the lane's slot) under one exec mask.  The values are already
// scaled to the fixed-point unit (the scale is folded into the per-pixel factors they are products of); they are
// rounded to the nearest integer with one v_cvt_rpi_i32_f32 (floor(x + 0.5)).
template <int N, int COPIES>
__device__ __forceinline__ void fix_add(unsigned long long* acc, const Target& t, int idx0, const float* v)
{
    if (t.active) {
        unsigned long long* a = &acc[(t.slot * NVAL + idx0) * COPIES + t.copy];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int q;
            asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(v[i]));
            atomicAdd(a + i * COPIES, (unsigned long long)(long long)q);
        }
    }
}

// Index-triple comparison of two faces through their records (only when the slot table is full).
__device__ __forceinline__ bool triple_differs_global(const FaceRec* __restrict__ recs, int fa, int fb)
{
    return recs[fa].vid[0] != recs[fb].vid[0] || recs[fa].vid[1] != recs[fb].vid[1] || recs[fa].vid[2] != recs[fb].vid[2];
}

// Scharr responses of one channel at the pixel `c` points to (row stride PW), from the staged tile:
// csrc/rasterise_grad_egl.cu:126-127, operation for operation (negative-offset minus positive-offset,
// offset_y is up = the previous tensor row).
__device__ __forceinline__ void scharr_taps(const float* c, float& sx, float& sy)
{
    const float mm = c[PW - 1], m0 = c[-1], mp = c[-PW - 1];     // at(-1,-1) at(-1,0) at(-1,+1)
    const float zm = c[PW], zp = c[-PW];                         // at(0,-1)           at(0,+1)
    const float pm = c[PW + 1], p0 = c[1], pp = c[-PW + 1];      // at(+1,-1) at(+1,0) at(+1,+1)
    float d1 = ((mm + mp) - pm) - pp;
    float d2 = m0 - p0;
    float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
    sx = m1 + m2;
    d1 = ((mm + pm) - mp) - pp;
    d2 = zm - zp;
    m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
    sy = m1 + m2;
}

// The same from global memory for an aliased (quirk Q1) channel whose taps run past the end of the
// image row: `centre` is the flat pixel index of the tap centre in the [B,H,W] slice; reads past the end
// of the tensor are clamped to its last element (undefined in the reference).  Rare: kept out of line.
__device__ __forceinline__ float2 scharr_taps_wrapped(const float* __restrict__ pixels, size_t total_pix, size_t centre, int W,
                                                   int C, int c)
{
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

template <int GH, int COPIES, int CSPEC, int SLOTS = 64>
__global__ __launch_bounds__(GW * GH, SLOTS == 32 ? 6 : GRAD_WAVES_PER_SIMD) void grad_kernel(GradParams p)
{
    constexpr int MAX_SLOTS = SLOTS;  // shadows the namespace constant
    constexpr int GTHREADS = GW * GH;  // GH / 8 x 4 waves, one 8 x 8 block each
    constexpr int PH = GH + 2;         // staged rows: y0-1 .. y0+GH
    __shared__ float s_pix[PC][PH][PW];                              // the pass's channels of `pixels`, edge clamped
    __shared__ __align__(16) unsigned long long s_acc[MAX_SLOTS * NVAL * COPIES];  // fixed-point partial sums
    __shared__ float4 s_frag[PH][VW];                                // (b0,b1,b2,clip_w) of every pixel of the halo'd tile
    __shared__ int32_t s_vis[PH][VW];                                // its front-most face
    __shared__ int16_t s_slot[PH][VW];                               // and that face's slot (-1 none, -2 table full)
    __shared__ int32_t s_key[MAX_SLOTS];                             // slot -> face
    __shared__ int32_t s_vid[MAX_SLOTS][3];                          // slot -> the face's vertex indices
    __shared__ uint8_t s_used[MAX_SLOTS];                             // compacted list of the occupied slots
    __shared__ int s_nused;
    __shared__ uint32_t s_bound[3];                                  // tile maxima (float bits): |grad_pixels|, |pixels|, 1/clip_w

#ifdef DIRT_TRACE
    long long tr_t[16]; int tr_n = 0;
#endif
    GMARK();  // 0 start
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int iib = blockIdx.y;
    const int ntiles = p.tiles_x * p.tiles_y;
    // CSPEC = 1, 3, 4: the channel count is that compile-time constant (4: with 16-byte aligned pixel tensors), which
    // makes the pass / channel-group structure static; 0: any channel count
    const int H = p.H, W = p.W, C = CSPEC ? CSPEC : p.C;
    const bool aligned16 = CSPEC == 4 ? true : (p.pixels_aligned16 != 0);
    const size_t frame = (size_t)H * W;
    const size_t total_pix = (size_t)p.B * frame;

    const FaceRec* __restrict__ recs = p.recs + (size_t)iib * p.F;
    const int32_t* __restrict__ vis = p.vis + (size_t)iib * frame;
    const float4* __restrict__ frag = p.frag + (size_t)iib * frame;
    const float* __restrict__ pixels = p.pixels + (size_t)iib * frame * C;
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * 4;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * C;

    // ---- this lane's pixel inside the halo'd tile ----
    const int px_l = (wave & 3) * 8 + (lane & 7) + 1, py_l = (wave >> 2) * 8 + (lane >> 3) + 1;
    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;
    const float width_f = (float)W, height_f = (float)H;

    // channels of a pass: whole channel groups (dirt/rasterise_ops.py:148-152) starting at c0 that fit in PC channels
    auto pass_channels = [&](int c0) {
        if (CSPEC) return (int)CSPEC;  // 1, 3 or 4 channels are one pass
        int nch = 0;
        for (int c = c0; c < C && nch < PC;) {
            const int G = (c + 3 <= C) ? 3 : 1;
            if (nch + G > PC) break;
            nch += G; c += G;
        }
        return nch;
    };
    // loads of the pass's channels of the pixels tile (+halo), edge clamped (at(), :113-124): two positions per
    // thread (PH * PWU <= 2 * GTHREADS), every load issued before any use so the tile costs one memory latency
    auto stage_load = [&](int tx0, int tr0, int c0, int nch, float (&v)[2][PC]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ii = min(tid + j * GTHREADS, PH * PWU - 1);
            const int yy = ii / PWU, xx = ii - yy * PWU;
            const int cy = min(max(tr0 + yy - 1, 0), H - 1), cx = min(max(tx0 + xx - 1, 0), W - 1);
            const float* src = pixels + ((size_t)cy * W + cx) * C + c0;
            if (nch == 4 && (C & 3) == 0 && aligned16) {
                const float4 q = *reinterpret_cast<const float4*>(src);  // c0 is a multiple of 4 here
                v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z; v[j][3] = q.w;
            } else {
#pragma unroll
                for (int ch = 0; ch < PC; ++ch) v[j][ch] = ch < nch ? src[ch] : 0.f;
            }
        }
    };
    // What a tile needs from global memory before anything can happen: the visibility and fragments of the halo'd
    // tile (PH * VWU positions, APOS per thread), the first pass's channels of the `pixels` tile and this pixel's
    // grad_pixels.  All loads are issued back to back; a workgroup that processes several tiles requests the next
    // tile's while it computes on the current one.
    constexpr int APOS = (PH * VWU + GTHREADS - 1) / GTHREADS;
    struct TileIn {
        int32_t a_face[APOS];
        float4 a_frag[APOS];
        float stage0_v[2][PC];
        float g0v[PC];
    };
    auto load_inputs = [&](int tile_linear, TileIn& in) {
        const int tile = xcd_tile(tile_linear, ntiles);
        const int tx0 = (tile % p.tiles_x) * GW, tr0 = (tile / p.tiles_x) * GH;
        stage_load(tx0, tr0, 0, pass_channels(0), in.stage0_v);
        {
            const int xs = min(tx0 + px_l - 1, W - 1), ys = min(tr0 + py_l - 1, H - 1);  // safe addresses for idle lanes
            const float* __restrict__ g = p.grad_pixels + ((size_t)iib * frame + (size_t)ys * W + xs) * C;
            const int nch0 = pass_channels(0);
#pragma unroll
            for (int c = 0; c < PC; ++c) in.g0v[c] = c < nch0 ? g[c] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < APOS; ++j) {
            const int i = tid + j * GTHREADS;
            in.a_face[j] = -1;
            in.a_frag[j] = make_float4(-1.f, -1.f, -1.f, INFINITY);
            if (i < PH * VWU) {
                const int vy = i / VWU, vx = i - vy * VWU;
                const int rr = min(max(tr0 + vy - 1, 0), H - 1), xx = min(max(tx0 + vx - 1, 0), W - 1);
                in.a_face[j] = vis[(size_t)rr * W + xx];
                in.a_frag[j] = frag[(size_t)rr * W + xx];
            }
        }
    };

    // One tile.  `in`: its inputs (already requested); next_linear >= 0: the tile this workgroup processes next, whose
    // inputs are requested into `nxt` as soon as this tile's are consumed.  Entered with an empty slot table, zero
    // bounds and zero accumulators.
    auto process_tile = [&](const int tile_linear, const TileIn& in, const int next_linear, TileIn& nxt) {
    const int tile = xcd_tile(tile_linear, ntiles);
    const int tx0 = (tile % p.tiles_x) * GW;
    const int tr0 = (tile / p.tiles_x) * GH;
    const int x_in_frame = tx0 + px_l - 1;
    const int y_in_frame = tr0 + py_l - 1;  // tensor row (top row first)
    const bool inside = x_in_frame < W && y_in_frame < H;
    const int xs = min(x_in_frame, W - 1), ys = min(y_in_frame, H - 1);  // safe addresses for idle lanes
    const size_t pix = (size_t)iib * frame + (size_t)ys * W + xs;
    const float* __restrict__ g_here = p.grad_pixels + pix * C;
    const bool interior = inside && x_in_frame > 0 && y_in_frame > 0 && x_in_frame < W - 1 && y_in_frame < H - 1;

    // ---- phase A: the visibility "surfaces" of the tile + 1-pixel halo, what the backward fragment
    //      shader writes (csrc/shaders.cpp:64-77) over the clear values of
    //      csrc/rasterise_grad_egl.cpp:442-445.  Halo positions outside the frame are clamped; they
    //      are only ever consulted for interior pixels, whose neighbours are inside the frame. ----
    float w_min = INFINITY;
#pragma unroll
    for (int j = 0; j < APOS; ++j) {
        const int i = tid + j * GTHREADS;
        if (i >= PH * VWU) continue;
        const int vy = i / VWU, vx = i - vy * VWU;
        const int32_t face = in.a_face[j];
        int slot = -1;
        if (face >= 0) {
            w_min = fminf(w_min, fabsf(in.a_frag[j].w));
            bool claimed;
            slot = slot_insert(s_key, MAX_SLOTS, face, claimed);
            if (claimed) {  // the thread that created the slot fetches the face's vertex indices for everybody
                const FaceRec* __restrict__ rec = recs + face;
                s_vid[slot][0] = rec->vid[0]; s_vid[slot][1] = rec->vid[1]; s_vid[slot][2] = rec->vid[2];
            }
            if (slot < 0) slot = -2;
        }
        s_vis[vy][vx] = face;
        s_frag[vy][vx] = in.a_frag[j];
        s_slot[vy][vx] = (int16_t)slot;
    }
    {
        const float rcpw_max = __builtin_amdgcn_rcpf(w_min);  // a bound (used with 2x slack): 1 ulp is plenty
        const uint32_t m = wave_max_u32(__float_as_uint(rcpw_max));
        if (lane == 0 && m) atomicMax(&s_bound[2], m);
    }
    GMARK();  // 2 phase A body

    // One pass = the channel groups that fit in PC channels, starting at channel c0.  A pass has one of four
    // shapes -- {3}, {3,1}, {1}, {1,1} (dirt/rasterise_ops.py:148-152 packs groups of 3 while >= 3 channels remain,
    // then singles) -- and the body is instantiated for each, so that its loops and branches over channels and
    // groups are static: NCH channels, the first group of size G0, every further group a single channel.
    auto run_pass = [&](auto nch_tag, auto g0_tag, const int c0) {
        constexpr int NCH = decltype(nch_tag)::value;
        constexpr int G0 = decltype(g0_tag)::value;
        constexpr int nch = NCH;
        // ---- stage the pass's channels of the pixels tile (+halo), edge clamped: at(), :113-124 ----
        float pmax = 0.f, gmax = 0.f;
        float stage_v[2][PC];
        if (c0 == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ch = 0; ch < PC; ++ch) stage_v[j][ch] = in.stage0_v[j][ch];  // requested before phase A
        } else {
            stage_load(tx0, tr0, c0, nch, stage_v);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pos = tid + j * GTHREADS;
            if (pos >= PH * PWU) continue;
            const int yy = pos / PWU, xx = pos - yy * PWU;
#pragma unroll
            for (int ch = 0; ch < PC; ++ch) {
                if (ch < nch) {
                    s_pix[ch][yy][xx] = stage_v[j][ch];
                    pmax = fmaxf(pmax, fabsf(stage_v[j][ch]));
                    if (!(stage_v[j][ch] == stage_v[j][ch])) pmax = INFINITY;
                }
            }
        }
        float gch[PC];
#pragma unroll
        for (int c = 0; c < PC; ++c) {
            gch[c] = c0 == 0 ? in.g0v[c] : ((c < nch) ? g_here[c0 + c] : 0.f);
            gmax = fmaxf(gmax, fabsf(gch[c]));
        }
        // (NaNs do not survive fmaxf: they are folded in explicitly so the inf/NaN fallback sees them)
        {
            bool gnan = false;
#pragma unroll
            for (int c = 0; c < PC; ++c) gnan |= !(gch[c] == gch[c]);
            if (gnan) gmax = INFINITY;
            const uint32_t mg = wave_max_u32(__float_as_uint(gmax)), mp = wave_max_u32(__float_as_uint(pmax));
            if (lane == 0) {
                if (mg) atomicMax(&s_bound[0], mg);
                if (mp) atomicMax(&s_bound[1], mp);
            }
        }
        __syncthreads();
        GMARK();  // 3 staged
        if (CSPEC && next_linear >= 0) load_inputs(next_linear, nxt);  // in flight while this tile is computed
        if (c0 == 0 && wave == 0) {
            // the slot table is complete: list the occupied slots for the flush (typically ~20 of 64); read after the
            // barrier that separates accumulation from flush
            const bool used = lane < MAX_SLOTS && s_key[lane & (MAX_SLOTS - 1)] >= 0;
            const unsigned long long um = __builtin_amdgcn_ballot_w64(used);
            if (used) s_used[__popcll(um & ((1ull << lane) - 1ull))] = (uint8_t)lane;
            if (lane == 0) s_nused = __popcll(um);
        }
        GMARK();  // 4 vids

        // ---- fixed-point scales from tile-wide bounds (no data-dependent barrier needed):
        //      |colour contribution|   = |g * b|, b <= 1                      <= gmax
        //      |position contribution| <= |dL_dx| * b * max(W,H) * |1/w|,  |dL_dx| <= 3 * gmax * |Scharr| <= 3 * gmax * pmax
        //      both times 4 for the quad pre-reduction and 2 for rounding slack ----
        const float gb = __uint_as_float(s_bound[0]), pb = __uint_as_float(s_bound[1]), wb = __uint_as_float(s_bound[2]);
        const FixScale fc = fix_scale(8.f * gb);
        const FixScale fp = fix_scale(24.f * gb * pb * fmaxf(width_f, height_f) * wb);
        const bool finite = fc.finite && fp.finite;

        const int32_t face_here = inside ? s_vis[py_l][px_l] : -1;
        const int slot_here = inside ? (int)s_slot[py_l][px_l] : -1;
        const Target t_here = make_target<COPIES>(slot_here, lane);
        const int hv0 = s_vid[max(slot_here, 0)][0], hv1 = s_vid[max(slot_here, 0)][1], hv2 = s_vid[max(slot_here, 0)][2];
        const float4 fh4 = s_frag[py_l][px_l];

        // ---- background gradient (:143-147) and colour gradients (:135-142) of the pass's channels ----
        if (inside && CSPEC) {
            float* gbk = p.grad_background + pix * C + c0;
            if (nch == 4 && (C & 3) == 0 && aligned16) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(gbk) = face_here >= 0 ? z : make_float4(gch[0], gch[1], gch[2], gch[3]);
            } else {
#pragma unroll
                for (int c = 0; c < PC; ++c)
                    if (c < nch) gbk[c] = face_here >= 0 ? 0.f : gch[c];
            }
        } else if (inside && c0 == 0) {
            // any channel count: the whole pixel at once in the first pass (zero where covered, grad_pixels where
            // not), so that a pixel's 4 * C bytes are written once instead of 12 bytes of them in every pass
            float* gbk = p.grad_background + pix * C;
            if ((C & 3) == 0 && aligned16) {
                for (int c = 0; c < C; c += 4) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (face_here < 0) v = *reinterpret_cast<const float4*>(g_here + c);
                    *reinterpret_cast<float4*>(gbk + c) = v;
                }
            } else {
                for (int c = 0; c < C; ++c) gbk[c] = face_here >= 0 ? 0.f : g_here[c];
            }
        }
        {
            const bool col_lds = finite && slot_here != -2;
            const bool col_direct = !col_lds && face_here >= 0 && (t_here.active || slot_here == -2);
            // g * b_k in fixed-point units: the scale is folded into g once (a power of two: exact)
            float gs[PC];
#pragma unroll
            for (int c = 0; c < PC; ++c) gs[c] = (face_here >= 0 && c < nch) ? gch[c] * fc.to_fix : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float hbk = k == 0 ? fh4.x : (k == 1 ? fh4.y : fh4.z);
                float cv[PC];
#pragma unroll
                for (int c = 0; c < PC; ++c) cv[c] = quad_reduce(t_here, gs[c] * hbk);
                if (col_lds) {
                    fix_add<PC, COPIES>(s_acc, t_here, 9 + k * PC, cv);
                } else if (col_direct) {
                    // table full, or inf / NaN in the tile: the reference's direct float atomics
#pragma unroll
                    for (int c = 0; c < PC; ++c)
                        if (c < nch) atomicAdd(&grad_vertex_colors[(size_t)recs[face_here].vid[k] * C + c0 + c], cv[c] * fc.from_fix);
                }
            }
        }

        GMARK();  // colour done
        // ---- channel groups of the pass: Scharr, dilation, position gradients ----
        int cg = 0;
#pragma unroll
        for (int gi = 0; gi < PC; ++gi) {  // at most PC groups in a pass
            if (cg >= nch) break;
            const int c_begin = c0 + cg;
            const int G = (gi == 0) ? G0 : 1;
            const bool alias = (G == 1) && !q1_intended;  // quirk Q1: "channels" 1,2 of a 1-channel tensor

            // Scharr (:126-127), streamed per channel into what is needed of it: the L1 norms of :185 (all three
            // "channels" of the reference's Vec3, in its summation order) and dL/dx, dL/dy of :203-208 (the
            // group's real channels, in channel order)
            float l1x = 0.f, l1y = 0.f, dL_dx = 0.f, dL_dy = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float sxc = 0.f, syc = 0.f;
                if (ch < G) {
                    scharr_taps(&s_pix[cg + ch][py_l][px_l], sxc, syc);
                    const float gcv = (cg + ch == 0) ? gch[0] : (cg + ch == 1) ? gch[1] : (cg + ch == 2) ? gch[2] : gch[3];
                    float m = gcv * sxc;
                    dL_dx = dL_dx + m;
                    m = gcv * syc;
                    dL_dy = dL_dy + m;
                } else if (alias) {
                    // quirk Q1: "channels" 1,2 of a 1-channel group = elements (pixel + ch) of the flattened
                    // [B,H,W,1] slice.  Only the L1 norms of interior pixels use them, and for an interior pixel
                    // the taps are unclamped: column + ch, which is staged unless it runs past the end of the
                    // image row (then it wraps to the next row: read from global memory).
                    scharr_taps(&s_pix[cg][py_l][min(px_l + ch, PW - 2)], sxc, syc);
                    const bool wraps = interior && x_in_frame + 1 + ch > W - 1;
                    if (__builtin_amdgcn_ballot_w64(wraps) != 0ull) {  // only tiles on the right image border
                        if (wraps) {
                            const float2 w2 = scharr_taps_wrapped(p.pixels, total_pix,
                                                                  (size_t)iib * frame + (size_t)y_in_frame * W + x_in_frame + ch, W, C, c_begin);
                            sxc = w2.x; syc = w2.y;
                        }
                    }
                }
                if (!(G == 1 && q1_intended && ch > 0)) { l1x = l1x + fabsf(sxc); l1y = l1y + fabsf(syc); }
            }
            GMARK();  // scharr
            // dilation, :155-194: which pixel's (barycentric, indices, clip_w) this pixel uses.  Both candidate
            // neighbours are read unconditionally (LDS) and the choice is predicated -- no divergent branches.
            int cy_l = py_l, cx_l = px_l;  // position (in the halo'd tile) of the fragment used
            bool dilated = false;
            {
                // direction: x if L1(Sx) > L1(Sy) else y (:185), negated on odd (x + y) (:186-190).  The reference's
                // offsets are in GL buffer orientation (y up): tensor row = y - offset_y.  The three LDS tiles share
                // the row stride VW, so a neighbour is one signed element offset `d` away in each of them.
                const bool horiz = l1x > l1y;
                const int sgn = ((x_in_frame + y_in_frame) & 1) ? -1 : 1;
                const int d = horiz ? sgn : -sgn * VW;
                const int e0 = py_l * VW + px_l, e1 = e0 + d, e2 = e0 - d;
                const int32_t f1 = (&s_vis[0][0])[e1], f2 = (&s_vis[0][0])[e2];
                const int s1 = (&s_slot[0][0])[e1], s2 = (&s_slot[0][0])[e2];
                const float w1 = (&s_frag[0][0])[e1].w, w2 = (&s_frag[0][0])[e2].w;
                const float w_here = fh4.w;
                // index triples differ (:86-89): an uncovered pixel (-1,-1,-1) differs from any face; two faces
                // with slots compare by canonical slot; a face without a slot (table full) through its record
                // (bitwise operators throughout: straight-line predicated code, no divergent branches around the LDS reads)
                bool d1 = (f1 >= 0) & (f1 != face_here), d2 = (f2 >= 0) & (f2 != face_here);
                {
                    // distinct faces over the same three vertices count as equal (:86-89): compare the triples
                    const bool here = face_here >= 0;
                    const int c1 = max(s1, 0), c2 = max(s2, 0);
                    const bool t1 = (s_vid[c1][0] != hv0) | (s_vid[c1][1] != hv1) | (s_vid[c1][2] != hv2);
                    const bool t2 = (s_vid[c2][0] != hv0) | (s_vid[c2][1] != hv1) | (s_vid[c2][2] != hv2);
                    const bool g1 = here & d1 & ((slot_here < 0) | (s1 < 0)), g2 = here & d2 & ((slot_here < 0) | (s2 < 0));
                    d1 = d1 & (t1 | !here); d2 = d2 & (t2 | !here);
                    if (__builtin_amdgcn_ballot_w64(g1 | g2) != 0ull) {  // some face has no slot (table full): compare through the records
                        if (g1) d1 = triple_differs_global(recs, face_here, f1);
                        if (g2) d2 = triple_differs_global(recs, face_here, f2);
                    }
                }
                const bool ok1 = interior & d1 & (w_here > w1);          // :165, first attempt (:191)
                const bool ok2 = interior & !ok1 & d2 & (w_here > w2);   // opposite direction if the first failed (:192-193)
                dilated = ok1 | ok2;
                const int dsel = ok1 ? d : (ok2 ? -d : 0);
                cy_l = py_l + (horiz ? 0 : (dsel > 0 ? 1 : (dsel < 0 ? -1 : 0)));
                cx_l = px_l + (horiz ? dsel : 0);
            }

            if (p.debug_thingy && c_begin == 0 && inside) {  // :150-151,172
                float* dbg = p.debug_thingy + pix * 3;
                dbg[0] = dilated ? 1.e-2f : 0.f;
                for (int ch = 1; ch <= 2; ++ch) {
                    // element (pix*G + ch) of the contiguous [B,H,W,G] slice of grad_pixels, clamped to its end
                    size_t mp = G == 3 ? pix : pix + ch;      // pixel of that element
                    int mc = G == 3 ? ch : 0;                 // channel inside the group
                    if (mp > total_pix - 1) { mp = total_pix - 1; mc = G - 1; }
                    dbg[ch] = p.grad_pixels[mp * C + c_begin + mc];
                }
            }

            GMARK();  // dilation
            // position gradients, :196-232
            const int32_t face_cur = inside ? s_vis[cy_l][cx_l] : -1;
            const bool covered = face_cur >= 0;
            const int slot_cur = covered ? (int)s_slot[cy_l][cx_l] : -1;
            const Target t_cur = make_target<COPIES>(slot_cur, lane);
            const float4 fc4 = s_frag[cy_l][cx_l];
            // clip-space x,y of the fragment used (:210-215 sums b_k * vertex_k.xy; perspective-correct
            // barycentrics make that sum the fragment's own clip position = its NDC position times clip_w,
            // so no vertex gather is needed; agrees to float rounding)
            const float clip_w = fc4.w;
            const float ndc_x = ((float)(tx0 + cx_l - 1) + 0.5f) * (2.f / width_f) - 1.f;
            const float ndc_y = ((float)(H - 1 - (tr0 + cy_l - 1)) + 0.5f) * (2.f / height_f) - 1.f;
            const float clip_x = ndc_x * clip_w, clip_y = ndc_y * clip_w;
            // :219-222 with one reciprocal (v_rcp_f32, 1 ulp) instead of four divisions
            const float rcp_w = __builtin_amdgcn_rcpf(clip_w);
            const float d_xview_by_xclip = (.5f * width_f) * rcp_w;
            const float d_yview_by_yclip = (.5f * height_f) * rcp_w;
            const float rcp_ww = rcp_w * rcp_w;
            const float d_xview_by_wclip = ((-.5f * width_f) * clip_x) * rcp_ww;
            const float d_yview_by_wclip = ((-.5f * height_f) * clip_y) * rcp_ww;
            const bool pos_lds = finite && slot_cur != -2;
            const bool pos_direct = !pos_lds && covered && (t_cur.active || slot_cur == -2);
            // :224-230: the three components of vertex k are b_k times per-pixel factors; those carry the
            // fixed-point scale (a power of two: exact) and are zero where nothing is covered
            const float fx = covered ? (dL_dx * d_xview_by_xclip) * fp.to_fix : 0.f;
            const float fy = covered ? (dL_dy * d_yview_by_yclip) * fp.to_fix : 0.f;
            const float fw = covered ? (dL_dx * d_xview_by_wclip + dL_dy * d_yview_by_wclip) * fp.to_fix : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float cbk = k == 0 ? fc4.x : (k == 1 ? fc4.y : fc4.z);
                float pv[3];
                pv[0] = quad_reduce(t_cur, fx * cbk);
                pv[1] = quad_reduce(t_cur, fy * cbk);
                pv[2] = quad_reduce(t_cur, fw * cbk);
                if (pos_lds) {
                    fix_add<3, COPIES>(s_acc, t_cur, 3 * k, pv);
                } else if (pos_direct) {
                    float* gv = grad_vertices + (size_t)recs[face_cur].vid[k] * 4;
                    atomicAdd(gv + 0, pv[0] * fp.from_fix);
                    atomicAdd(gv + 1, pv[1] * fp.from_fix);
                    atomicAdd(gv + 3, pv[2] * fp.from_fix);
                }
            }
            GMARK();  // fix_add
            cg += G;
        }
        GMARK();  // 5 accumulated
        __syncthreads();
        if (CSPEC && next_linear >= 0) {  // the next tile's empty slot table and bounds (nothing reads them any more)
            for (int i = tid; i < MAX_SLOTS; i += GTHREADS) s_key[i] = -1;
            if (tid < 3) s_bound[tid] = 0u;
        }

        // ---- flush: one global atomic per (face, vertex, component) for the whole tile and pass ----
        const int nused_f = s_nused;
        for (int e = tid; e < nused_f * NVAL; e += GTHREADS) {
            const int u = e / NVAL, v = e - u * NVAL;
            const int slot = s_used[u];
            unsigned long long* a = &s_acc[(slot * NVAL + v) * COPIES];
            long long sum = 0;
#pragma unroll
            for (int cp = 0; cp < COPIES; ++cp) { sum += (long long)a[cp]; a[cp] = 0ull; }
            if (sum == 0) continue;
            // |sum| < 2^45: two exact conversions and one fma instead of the generic int64 -> double sequence
            const double dsum = fma((double)(int32_t)(sum >> 32), 4294967296.0, (double)(uint32_t)sum);
            const float f = (float)(dsum * (double)(v < 9 ? fp.from_fix : fc.from_fix));
            if (v < 9) {
                const int k = v / 3, comp = v - k * 3;
                atomicAdd(&grad_vertices[(size_t)s_vid[slot][k] * 4 + (comp == 2 ? 3 : comp)], f);
            } else {
                const int k = (v - 9) / PC, c = (v - 9) - k * PC;
                if (c < nch) atomicAdd(&grad_vertex_colors[(size_t)s_vid[slot][k] * C + c0 + c], f);
            }
        }
        GMARK();  // 6 flushed
        if (CSPEC && next_linear >= 0) __syncthreads();  // the flush has read s_vid / s_used; the next tile may claim slots
    };
    using std::integral_constant;
    for (int c0 = 0; c0 < C;) {
        const int nch = pass_channels(c0);
        if (CSPEC) {
            run_pass(integral_constant<int, CSPEC ? CSPEC : 1>{}, integral_constant<int, CSPEC == 1 ? 1 : 3>{}, 0);
        } else if (c0 + 3 <= C) {
            if (nch == 4) run_pass(integral_constant<int, 4>{}, integral_constant<int, 3>{}, c0);
            else run_pass(integral_constant<int, 3>{}, integral_constant<int, 3>{}, c0);
        } else {
            if (nch == 2) run_pass(integral_constant<int, 2>{}, integral_constant<int, 1>{}, c0);
            else run_pass(integral_constant<int, 1>{}, integral_constant<int, 1>{}, c0);
        }
        c0 += nch;
        if (CSPEC) break;  // a single pass, statically
        if (c0 < C) {
            __syncthreads();
            if (tid < 2) s_bound[tid] = 0u;  // |grad_pixels| and |pixels| bounds are per pass; 1/w is per tile
            __syncthreads();
        }
    }
    };  // process_tile

    // ---- the workgroup's tiles: blockIdx.x, blockIdx.x + gridDim.x, ...  (one tile when the grid covers them all) ----
    TileIn cur, nxt;
    load_inputs((int)blockIdx.x, cur);
    // empty slot table, cleared accumulators (under the latency of the loads above); a flush leaves them cleared
    for (int i = tid; i < MAX_SLOTS * NVAL * COPIES / 2; i += GTHREADS) reinterpret_cast<uint4*>(s_acc)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < MAX_SLOTS; i += GTHREADS) s_key[i] = -1;
    if (tid < 3) s_bound[tid] = 0u;
    __syncthreads();
    GMARK();  // 1 init
    for (int t = (int)blockIdx.x;;) {
        const int tn = (CSPEC && SLOTS == 64 && t + (int)gridDim.x < ntiles) ? t + (int)gridDim.x : -1;
        process_tile(t, cur, tn, nxt);
        if (tn < 0) break;
        cur = nxt;
        t = tn;
    }
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_grad) {
        long long* o = g_trace_grad + ((size_t)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 16; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
    }
#endif
}

hipError_t launch_grad(const GradParams& p_in, hipStream_t stream)
{
    if (p_in.B == 0) return hipSuccess;
    GradParams p = p_in;
    p.tiles_x = (p.W + GW - 1) / GW;
    // 32 x 16 tiles unless the frame is small (fewer than two workgroups per CU) or the mesh is dense (the expected
    // number of faces in a halo'd tile approaches the 64 slots of the table): then 32 x 8
    const double faces_per_tile = (double)p.F * (34.0 * 18.0) / ((double)p.H * (double)p.W);
    const long long tiles16 = (long long)p.tiles_x * ((p.H + 15) / 16) * p.B;
    int gh = (tiles16 < 512 || faces_per_tile > 40.0) ? 8 : 16;
    if (p.flags & DIRT_FLAG_TILES_LARGE) gh = 16;
    if (p.flags & DIRT_FLAG_TILES_SMALL) gh = 8;
    p.tiles_y = (p.H + gh - 1) / gh;
    p.pixels_aligned16 = ((reinterpret_cast<uintptr_t>(p.pixels) | reinterpret_cast<uintptr_t>(p.grad_background)) & 15u) == 0 ? 1 : 0;
    // the common channel counts get kernels in which the pass / channel-group structure is static
    const int cspec = (p.C == 4 && p.pixels_aligned16) ? 4 : (p.C == 3 ? 3 : (p.C == 1 ? 1 : 0));
    // The channel-specialised kernels process several tiles per workgroup (the next tile's inputs are requested while
    // the current one is computed) when there are enough tiles to keep every CU busy regardless.
    const long long ntiles = (long long)p.tiles_x * p.tiles_y;
    int tiles_per_wg = 1;
    if (cspec) {
        for (tiles_per_wg = 4; tiles_per_wg > 1 && ntiles * p.B / tiles_per_wg < 512; tiles_per_wg >>= 1) {}
        if (const char* env = getenv("DIRT_GRAD_TILES_PER_WG")) {  // tests pin it (any value >= 1)
            const int v = atoi(env);
            if (v >= 1) tiles_per_wg = v;
        }
    }
    dim3 grid((unsigned)((ntiles + tiles_per_wg - 1) / tiles_per_wg), (unsigned)p.B);
    // The channel-specialised 32 x 16 kernels exist with a 64-slot table (two workgroups per CU, next-tile prefetch) and
    // with a 32-slot table (50 KB of LDS and no prefetch registers: three workgroups per CU, 6 waves per SIMD) for
    // meshes whose tiles see few faces; a tile that overflows its table is still correct, only slower.
    int slots = (cspec && gh == 16 && faces_per_tile <= 10.0) ? 32 : 64;
    if (const char* env = getenv("DIRT_GRAD_SLOTS")) {  // tests pin it
        const int v = atoi(env);
        if ((v == 32 && cspec && gh == 16) || v == 64) slots = v;
    }
#define DIRT_LAUNCH_GRAD(GH_, CP_, SL_)                                                                          \
    do {                                                                                                         \
        const dim3 block(GW * GH_);                                                                              \
        if (cspec == 4) hipLaunchKernelGGL((grad_kernel<GH_, CP_, 4, SL_>), grid, block, 0, stream, p);          \
        else if (cspec == 3) hipLaunchKernelGGL((grad_kernel<GH_, CP_, 3, SL_>), grid, block, 0, stream, p);     \
        else if (cspec == 1) hipLaunchKernelGGL((grad_kernel<GH_, CP_, 1, SL_>), grid, block, 0, stream, p);     \
        else hipLaunchKernelGGL((grad_kernel<GH_, CP_, 0, 64>), grid, block, 0, stream, p);                      \
    } while (0)
    if (gh == 16 && slots == 32) {
        grid = dim3((unsigned)ntiles, (unsigned)p.B);  // one tile per workgroup
        DIRT_LAUNCH_GRAD(16, 4, 32);
    } else if (gh == 16) {
        DIRT_LAUNCH_GRAD(16, 4, 64);
    } else {
        DIRT_LAUNCH_GRAD(8, 2, 64);
    }
#undef DIRT_LAUNCH_GRAD
    return hipGetLastError();
}

}  // namespace dirt
