// dirt_raster.hip -- triangle set-up and the tiled visibility / shading kernels for gfx950.
//
// Replaces, for the whole batch in one launch each:
//   * the GL vertex pipeline + upload_vertices      (csrc/rasterise_grad_egl.cu:12-34)
//   * B x { glViewport, glScissor, glClear(DEPTH), glDrawElementsBaseVertex }
//                                                   (csrc/rasterise_egl.cpp:362-380)
//   * upload_background / download_pixels           (csrc/rasterise_egl.cu:10-38,65-91): there is no
//     RGBA32F atlas; tiles read `background` and write `pixels` in place, top row first.
//
// Structure of raster_kernel<MODE, NB = 2> (one 256-thread workgroup = one 32x32 pixel tile of one scene = 4x4
// blocks of 8x8 pixels; each of its 4 waves owns a 16x16 region = 2x2 blocks, 4 pixels per lane, so
// that one record fetch serves 256 pixels):
//   scan   : the threads walk column `bin` of the chunk x bin directory (binary search over the run
//            prefix held in LDS) and append the faces whose box touches the tile to an LDS list
//            (wave-aggregated LDS atomic), together with a 16-bit mask of the blocks the box touches;
//   raster : 64 candidates at a time are staged in LDS as TileRecs (one lane per candidate: float32
//            tile-local edge coefficients + certified error bounds + the f64 depth plane); every wave
//            then visits the candidates whose mask touches one of its four blocks: float32 edge
//            functions classify each sample as certainly inside / certainly outside, the few in
//            between take the specification's f64 test (covered_exact); then depth from the f64
//            plane and a (z24, face) lexicographic min held in registers -- no LDS or global
//            atomics, and the result is independent of list order;
//   shade  : the winner's record is re-read per pixel, barycentrics and all C channels are
//            interpolated once, and the HWC pixel is written (background copied where uncovered).
#include "dirt_device.h"
#include "dirt_launch.h"
#include "../../include/dirt_hip.h"

namespace dirt {

#ifdef DIRT_TRACE
__device__ long long* g_trace_buf = nullptr;
#endif

// ---- binning ---------------------------------------------------------------------------------
// The frame is cut into at most MAX_BINS square bins of 2^shift pixels (>= 32, so a raster tile never
// straddles two bins).  ONE kernel builds, per scene, exact-size lists of the faces touching each bin -- the
// replacement for the GL driver's own binning hardware -- without a single global atomic, without any
// buffer that needs clearing and without any cross-workgroup dependency:
//   setup_kernel : the scene's faces are cut into `nchunk` contiguous chunks, one 256-thread workgroup
//                  each.  Pass 1, per face: set-up record, bounding box, and an LDS count in every bin the
//                  box touches (faces touching more than 4 bins count for the "big" pseudo-bin, which every
//                  tile reads, so a chunk produces at most 5 * chunk_faces entries).  An LDS prefix over the
//                  bins turns the counts into the chunk's own bin-sorted segment layout, stored as one row
//                  of the chunk x bin directory.  Pass 2: the faces claim their slots with LDS cursors.
// A raster tile then reads column `bin` of the directory (one cell per chunk) and walks those segments.
// Entry order inside a (chunk, bin) run is whatever the LDS atomics produce; visibility does not depend on it.

__device__ __forceinline__ bool bin_range(const FaceBox& box, const BinGrid& grid, int& bx0, int& bx1, int& by0, int& by1)
{
    bx0 = box.i_min >> grid.shift; bx1 = box.i_max >> grid.shift;
    by0 = box.r_min >> grid.shift; by1 = box.r_max >> grid.shift;
    return (bx1 - bx0 + 1) * (by1 - by0 + 1) <= 4;  // false: the face goes on the big list
}

__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ b, size_t nb, uint32_t* __restrict__ c, size_t nc)
{
    // clears caller buffers of any alignment in 4-byte units (the gradients of the backward pass)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += stride) c[i] = 0u;
}

__global__ __launch_bounds__(256) void setup_kernel(GeomParams g)
{
    __shared__ uint32_t s_cnt[MAX_BINS + 1];    // [MAX_BINS] = big faces
    __shared__ uint32_t s_start[MAX_BINS + 1];  // exclusive prefix of s_cnt = the chunk's segment layout
    __shared__ uint32_t s_wave[4];
    const int ib = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // side job: clear the gradient accumulators of the backward pass (the cudaMemsetAsync x4 of
    // csrc/rasterise_grad_egl.cu:244-250) so that no separate launch is needed for it
    {
        // 16 bytes per store (the buffers are 16-byte aligned and their sizes multiples of 16: [B,V,4] floats and the
        // 256-byte aligned workspace regions; caller tensors of other sizes get a dword tail)
        const size_t nthreads = (size_t)gridDim.x * gridDim.y * 256;
        const size_t gtid = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid;
        uint4* zb = reinterpret_cast<uint4*>(g.zero_b);
        uint4* zc = reinterpret_cast<uint4*>(g.zero_c);
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
        for (size_t i = gtid; i < g.zero_b_bytes / 16; i += nthreads) zb[i] = z4;
        for (size_t i = gtid; i < g.zero_c_bytes / 16; i += nthreads) zc[i] = z4;
        uint32_t* tb = reinterpret_cast<uint32_t*>(g.zero_b) + (g.zero_b_bytes / 16) * 4;
        uint32_t* tc = reinterpret_cast<uint32_t*>(g.zero_c) + (g.zero_c_bytes / 16) * 4;
        if (gtid < (g.zero_b_bytes % 16) / 4) tb[gtid] = 0u;
        if (gtid < (g.zero_c_bytes % 16) / 4) tc[gtid] = 0u;
    }
    s_cnt[tid] = 0;
    if (tid == 0) s_cnt[MAX_BINS] = 0;
    __syncthreads();
    const int f0 = chunk * g.chunk_faces, f1 = min(g.F, f0 + g.chunk_faces);
    const float* __restrict__ verts = g.vertices + (size_t)ib * g.V * 4;

    // ---- pass 1: set-up + histogram ----
    FaceBox first_box;  // the box of this thread's first face stays in registers for pass 2
    first_box.i_min = 32767; first_box.i_max = -32768; first_box.r_min = 32767; first_box.r_max = -32768;
    for (int f = f0 + tid; f < f1; f += 256) {
        const size_t n = (size_t)ib * g.F + f;
        FaceRec rec;
        FaceBox box;
        if (setup_face(verts, g.V, g.faces + (g.shared_faces ? (size_t)f : n) * 3, g.H, g.W, rec, box)) {
            g.recs[n] = rec;
            int bx0, bx1, by0, by1;
            if (bin_range(box, g.grid, bx0, bx1, by0, by1)) {
                for (int by = by0; by <= by1; ++by)
                    for (int bx = bx0; bx <= bx1; ++bx) atomicAdd(&s_cnt[by * g.grid.bins_x + bx], 1u);
            } else {
                atomicAdd(&s_cnt[MAX_BINS], 1u);
            }
        } else {
            g.recs[n].flags = 0;
            box.i_min = 32767; box.i_max = -32768; box.r_min = 32767; box.r_max = -32768;
        }
        if (f == f0 + tid) first_box = box;
        if (g.chunk_faces > 256) g.boxes[n] = box;  // re-read in pass 2 when a thread owns several faces
    }
    __syncthreads();

    // ---- the chunk's segment layout: exclusive prefix over the 257 (pseudo-)bins ----
    const uint32_t cnt = s_cnt[tid];
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    const uint32_t start = off + incl - cnt;
    BinCell* __restrict__ row = g.cells + ((size_t)ib * g.nchunk + chunk) * (MAX_BINS + 1);
    s_start[tid] = start;
    row[tid] = BinCell{start, cnt};
    if (tid == 255) {
        s_start[MAX_BINS] = start + cnt;
        row[MAX_BINS] = BinCell{start + cnt, s_cnt[MAX_BINS]};
    }
    __syncthreads();
    s_cnt[tid] = 0;  // reused as the fill cursors
    if (tid == 0) s_cnt[MAX_BINS] = 0;
    __syncthreads();

    // ---- pass 2: faces claim their slots in the chunk's segment ----
    BinEntry* __restrict__ out = g.entries + ((size_t)ib * g.nchunk + chunk) * (5 * (size_t)g.chunk_faces);
    for (int f = f0 + tid; f < f1; f += 256) {
        BinEntry e;
        e.box = (f == f0 + tid) ? first_box : g.boxes[(size_t)ib * g.F + f];
        if (e.box.i_min > e.box.i_max) continue;  // culled at set-up
        e.face = f; e.pad = 0;
        int bx0, bx1, by0, by1;
        if (bin_range(e.box, g.grid, bx0, bx1, by0, by1)) {
            for (int by = by0; by <= by1; ++by)
                for (int bx = bx0; bx <= bx1; ++bx) {
                    const int b = by * g.grid.bins_x + bx;
                    out[s_start[b] + atomicAdd(&s_cnt[b], 1u)] = e;
                }
        } else {
            out[s_start[MAX_BINS] + atomicAdd(&s_cnt[MAX_BINS], 1u)] = e;
        }
    }
}

// A tile is 2 x 2 wave regions; a wave region is NB x NB blocks of 8 x 8 pixels (NB * NB pixels per lane).
// NB = 2 (32 x 32 tiles, one record fetch serves 256 pixels) is the normal shape; NB = 1 (16 x 16 tiles) gives four
// times as many workgroups for small frames, where 32 x 32 tiles would leave most of the 1024 SIMDs idle.
constexpr int RTHREADS = 256;     // 4 waves: one per region (NB = 2); NB = 1 tiles use two waves per region (512 threads)
constexpr int LIST_CAP = 2048;    // bin entries scanned (and at most listed) per round

// What the coverage / depth loop reads per candidate: built once per (tile, candidate) by one lane when the
// candidate is staged in LDS, 80 bytes = five 16-byte LDS reads.
//
// The specification decides coverage by the SIGN of E_k = fma(a_k, px, fma(b_k, py, c_k)) in f64 (and a tie
// rule on exact zeros).  Here E_k is first evaluated in float32 in tile-local coordinates (dx = px - ox in
// 0..31, dy = py - oy in -31..0, c' = E_k(ox, oy) rounded from f64) together with a certified bound on
// |E32_k - E_k| over the tile:
//   |E32 - E| <= u32*(2*31|a| + 3*31|b| + 3|c'|) + 2^-51*Mg,  u32 = 2^-24,  Mg = |a|W + |b|H + |c|
//            <= 2^-22*(32|a32| + 32|b32| + |c32|) + 2^-49*Mg32 = bnd        (Mg32: Mg from the rounded values)
// E32_k > bnd_k for every k: certainly inside; some E32_k < -bnd_k: certainly outside; anything else (a
// ~1e-5 pixel strip along an edge, or exactly on it) takes the specification's f64 path, covered_exact().
// Results are bit-identical to the specification at a fraction of the f64 work.
struct alignas(16) TileRec {
    float a[3], b[3], c[3], bnd[3];  //  0: tile-local float32 edge functions (true sign: inside = positive), bounds
    double zp[3];                    // 48: depth plane scaled to the 24-bit range, global coordinates
    uint32_t flags;                  // 72
    int32_t face;                    // 76
};
static_assert(sizeof(TileRec) == 80, "TileRec is 80 bytes");

// One lane builds the TileRec of one candidate.  ox, oy: sample position of the tile's top-left pixel.
__device__ __forceinline__ void make_tile_rec(const FaceRec* __restrict__ rec, int face, double ox, double oy, float wf, float hf,
                                              TileRec* out)
{
    const uint32_t flags = rec->flags;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = rec->coef[3 * k], b = rec->coef[3 * k + 1], c = rec->coef[3 * k + 2];
        const double cl = fma(a, ox, fma(b, oy, c));
        const float sg = (flags & (1u << k)) ? -1.f : 1.f;  // undo the sign folding: E_k = sg * F_k
        const float a32 = (float)a, b32 = (float)b, c32 = (float)cl, cg = (float)c;
        const float mg = fabsf(a32) * wf + fabsf(b32) * hf + fabsf(cg);
        out->bnd[k] = (0x1p-22f * (32.f * fabsf(a32) + 32.f * fabsf(b32) + fabsf(c32)) + 0x1p-49f * mg) * 1.0001f;
        out->a[k] = sg * a32; out->b[k] = sg * b32; out->c[k] = sg * c32;
    }
    out->zp[0] = rec->zp[0]; out->zp[1] = rec->zp[1]; out->zp[2] = rec->zp[2];
    out->flags = flags; out->face = face;
}

// The specification's f64 coverage test for the samples the float filter cannot decide.  Rare; kept out of
// line (and reading the FaceRec from global memory) so that nothing of it is speculated into the main loop.
__device__ __noinline__ bool covered_exact(const FaceRec* __restrict__ rec, double px, double py)
{
    const uint32_t flags = rec->flags;
    const double F0 = fma(rec->coef[0], px, fma(rec->coef[1], py, rec->coef[2]));
    const double F1 = fma(rec->coef[3], px, fma(rec->coef[4], py, rec->coef[5]));
    const double F2 = fma(rec->coef[6], px, fma(rec->coef[7], py, rec->coef[8]));
    return ((F0 >= 0.0) != ((flags & 1u) != 0)) && ((F1 >= 0.0) != ((flags & 2u) != 0)) && ((F2 >= 0.0) != ((flags & 4u) != 0));
}

// Coverage + depth + visibility update of one block (one pixel per lane) for one candidate.
// dx: this lane's tile-local column; trow[k] = fmaf(b_k, dy, c_k) of the block row; px, py the sample
// position; qrow = fma(zp[1], py, zp[2]) of the block row.
__device__ __forceinline__ void raster_block(const TileRec& t, const FaceRec* __restrict__ recs, float dx, const float trow[3],
                                             double px, double py, double qrow, uint32_t& zbest, int32_t& fbest)
{
    const float E0 = fmaf(t.a[0], dx, trow[0]);
    const float E1 = fmaf(t.a[1], dx, trow[1]);
    const float E2 = fmaf(t.a[2], dx, trow[2]);
    const float lo = fminf(fminf(E0 - t.bnd[0], E1 - t.bnd[1]), E2 - t.bnd[2]);
    const float hi = fminf(fminf(E0 + t.bnd[0], E1 + t.bnd[1]), E2 + t.bnd[2]);
    bool cov = lo > 0.f;                             // certainly inside
    const bool unsure = !(lo > 0.f) && !(hi < 0.f);  // neither certainly inside nor outside (NaN lands here)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(unsure) != 0ull, 0)) {
        if (unsure) cov = covered_exact(recs + t.face, px, py);
    }
    if (__builtin_amdgcn_ballot_w64(cov) == 0ull) return;
    const double q = fma(t.zp[0], px, qrow);  // depth scaled to [0, 2^24-1]; kept iff inside (the depth clip)
    const uint32_t z24 = (uint32_t)rint(q);
    // GL_LESS against the stored depth; equal depth keeps the lower face index, which is what drawing the
    // faces in index order does (csrc/rasterise_egl.cpp:373-379)
    // (bitwise, not short-circuit, operators: one predicated update instead of nested divergent branches)
    const bool wins = cov & (q >= 0.0) & (q <= 16777215.0) & ((z24 < zbest) | ((z24 == zbest) & (t.face < fbest)));
    zbest = wins ? z24 : zbest;
    fbest = wins ? t.face : fbest;
}

// One candidate against the wave's NB x NB blocks; `m4` (wave-uniform) says which blocks its box touches.
template <int NB>
__device__ __forceinline__ void raster_candidate(const TileRec& t, const FaceRec* __restrict__ recs, uint32_t m4, const float* dx,
                                                 const float* dy, const double* px, const double* py, uint32_t* zbest,
                                                 int32_t* fbest)
{
#pragma unroll
    for (int by = 0; by < NB; ++by) {
        if ((m4 >> (NB * by)) & ((1u << NB) - 1u)) {
            float trow[3];
            trow[0] = fmaf(t.b[0], dy[by], t.c[0]);
            trow[1] = fmaf(t.b[1], dy[by], t.c[1]);
            trow[2] = fmaf(t.b[2], dy[by], t.c[2]);
            const double qrow = fma(t.zp[1], py[by], t.zp[2]);
#pragma unroll
            for (int bx = 0; bx < NB; ++bx)
                if ((m4 >> (NB * by + bx)) & 1u)
                    raster_block(t, recs, dx[bx], trow, px[bx], py[by], qrow, zbest[NB * by + bx], fbest[NB * by + bx]);
        }
    }
}

// The backward pass's state of one pixel -- csrc/shaders.cpp:64-77: {clip_w, face} and {b0, b1} (b2 = 1 - b0 - b1; the
// face index stands for the index triple) -- or the clear values of csrc/rasterise_grad_egl.cpp:442-445.
__device__ __forceinline__ void store_state(const RasterParams& p, size_t pix, float b0, float b1, float clip_w, int32_t face)
{
    p.state_a[pix] = make_float2(clip_w, __int_as_float(face));
    p.state_b[pix] = make_float2(b0, b1);
}

__device__ __forceinline__ void export_state(const RasterParams& p, const FaceRec* __restrict__ recs, int ib, int x, int r,
                                             double px, double py, int32_t f)
{
    const size_t pix = ((size_t)ib * p.H + r) * p.W + x;
    if (f < 0) { store_state(p, pix, -1.f, -1.f, INFINITY, -1); return; }
    const FaceRec* __restrict__ rec = recs + f;
    double cf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k] = rec->coef[k];
    double Fk[3];
    edge_eval(cf, px, py, Fk);
    float b[3], cw;
    bary_eval(Fk, rec->flags, rec->inv_det, b, cw);
    store_state(p, pix, b[0], b[1], cw, f);
}

// Shade one pixel: the winner's record is re-read, barycentrics and all C channels interpolated once.
template <int CSPEC>
__device__ __forceinline__ void shade_pixel(const RasterParams& p, const FaceRec* __restrict__ recs, int ib, int x, int r,
                                            double px, double py, int32_t f)
{
    const size_t pix = ((size_t)ib * p.H + r) * p.W + x;
    const int C = CSPEC ? CSPEC : p.C;  // CSPEC = 1, 3, 4: compile-time channel count; 0: any
    float* __restrict__ out = p.pixels + pix * C;
    if (f < 0) {  // pixels start as the background: csrc/rasterise_egl.cpp:348-356
        if (p.state_a) store_state(p, pix, -1.f, -1.f, INFINITY, -1);
        const float* __restrict__ bg = p.background + pix * C;
        if ((C & 3) == 0) {
            for (int c = 0; c < C; c += 4)
                *reinterpret_cast<float4*>(out + c) = *reinterpret_cast<const float4*>(bg + c);
        } else {
            for (int c = 0; c < C; ++c) out[c] = bg[c];
        }
        return;
    }
    const FaceRec* __restrict__ rec = recs + f;
    double cf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k] = rec->coef[k];
    double Fk[3];
    edge_eval(cf, px, py, Fk);
    float b[3], cw;
    bary_eval(Fk, rec->flags, rec->inv_det, b, cw);
    if (p.state_a) store_state(p, pix, b[0], b[1], cw, f);
    const float* __restrict__ cols = p.vertex_colors + (size_t)ib * p.V * C;
    const float* __restrict__ c0 = cols + (size_t)rec->vid[0] * C;
    const float* __restrict__ c1 = cols + (size_t)rec->vid[1] * C;
    const float* __restrict__ c2 = cols + (size_t)rec->vid[2] * C;
    if ((C & 3) == 0) {
        for (int c = 0; c < C; c += 4) {
            const float4 u0 = *reinterpret_cast<const float4*>(c0 + c);
            const float4 u1 = *reinterpret_cast<const float4*>(c1 + c);
            const float4 u2 = *reinterpret_cast<const float4*>(c2 + c);
            float4 o;
            o.x = fmaf(b[2], u2.x, fmaf(b[1], u1.x, b[0] * u0.x));
            o.y = fmaf(b[2], u2.y, fmaf(b[1], u1.y, b[0] * u0.y));
            o.z = fmaf(b[2], u2.z, fmaf(b[1], u1.z, b[0] * u0.z));
            o.w = fmaf(b[2], u2.w, fmaf(b[1], u1.w, b[0] * u0.w));
            *reinterpret_cast<float4*>(out + c) = o;
        }
    } else {
        for (int c = 0; c < C; ++c) out[c] = fmaf(b[2], c2[c], fmaf(b[1], c1[c], b[0] * c0[c]));
    }
}

template <int MODE, int NB, int CSPEC>
__global__ __launch_bounds__(NB == 1 ? 2 * RTHREADS : RTHREADS) void raster_kernel(RasterParams p)
{
    // Small tiles exist for frames with few tiles: there the candidates of a region are shared by SPLIT waves (each
    // takes every SPLIT-th one) whose (z24, face) minima are merged through LDS, which doubles the waves in flight.
    constexpr int SPLIT = NB == 1 ? 2 : 1;
    constexpr int RT = RTHREADS * SPLIT;
    constexpr int TILE_W = 16 * NB, TILE_H = 16 * NB;  // pixels
    constexpr int BT = 2 * NB;                         // blocks per tile side: block (bx, by) = mask bit BT * by + bx
    __shared__ int32_t s_face[LIST_CAP];
    __shared__ uint16_t s_mask[LIST_CAP];  // bit (4*by + bx): the face's box touches block (bx, by) of the tile
    __shared__ uint32_t s_count;
    __shared__ uint32_t s_pre[2 * MAX_BINS + 1];       // exclusive prefix of the run lengths (nchunk <= MAX_BINS... 256 chunks)
    __shared__ uint32_t s_run_base[2 * MAX_BINS];      // first entry of each run, relative to the scene's entries
    __shared__ TileRec s_rec[64];          // tile-local records of the 64 list entries being rasterised
    __shared__ int32_t s_vis[TILE_W * TILE_H];  // the tile's visibility, for the row-major resolve
    __shared__ int32_t s_mf[(SPLIT > 1 ? SPLIT - 1 : 1) * (SPLIT > 1 ? TILE_W * TILE_H : 1)];   // minima of the waves with part > 0
    __shared__ uint32_t s_mz[(SPLIT > 1 ? SPLIT - 1 : 1) * (SPLIT > 1 ? TILE_W * TILE_H : 1)];

#ifdef DIRT_TRACE
    long long tr_t[8]; int tr_n = 0;
#define TRACE_MARK() do { if (tr_n < 8) tr_t[tr_n++] = clock64(); } while (0)
    long long tr_acc[4] = {0, 0, 0, 0}, tr_last = 0; int tr_cnt = 0;
#define TRACE_ACC(i) do { long long now_ = clock64(); if ((i) > 0) tr_acc[i] += now_ - tr_last; tr_last = now_; } while (0)
#define TRACE_CNT() do { ++tr_cnt; } while (0)
#else
#define TRACE_MARK() do {} while (0)
#define TRACE_ACC(i) do {} while (0)
#define TRACE_CNT() do {} while (0)
#endif
    TRACE_MARK();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ib = blockIdx.y;
    const int tile = xcd_tile(blockIdx.x, p.tiles_x * p.tiles_y);
    const int tx0 = (tile % p.tiles_x) * TILE_W;
    const int tr0 = (tile / p.tiles_x) * TILE_H;
    const int tx1 = tx0 + TILE_W - 1, tr1 = tr0 + TILE_H - 1;

    const FaceRec* __restrict__ recs = p.recs + (size_t)ib * p.F;
    const int bin = (tr0 >> p.grid.shift) * p.grid.bins_x + (tx0 >> p.grid.shift);
    // column `bin` (and the big pseudo-bin) of the chunk x bin directory: entry e of the tile's input is
    // entry (e - s_pre[j]) of run j, where the 2 * nchunk runs are (chunk, bin) then (chunk, big)
    const BinCell* __restrict__ cells = p.cells + (size_t)ib * p.nchunk * (MAX_BINS + 1);
    const BinEntry* __restrict__ scene_entries = p.entries + (size_t)ib * p.nchunk * (5 * (size_t)p.chunk_faces);
    const int nruns = 2 * p.nchunk;
    for (int j = tid; j < nruns; j += RT) {
        const int c = j < p.nchunk ? j : j - p.nchunk;
        const BinCell cell = cells[(size_t)c * (MAX_BINS + 1) + (j < p.nchunk ? bin : MAX_BINS)];
        s_run_base[j] = (uint32_t)c * (5u * (uint32_t)p.chunk_faces) + cell.start;
        s_pre[j] = cell.count;
    }
    __syncthreads();
    if (wave == 0) {  // exclusive prefix of the run lengths (<= 512 runs: 8 per lane)
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = lane * 8 + i;
            v[i] = j < nruns ? s_pre[j] : 0u;
            sum += v[i];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = lane * 8 + i;
            if (j < nruns) s_pre[j] = run;
            run += v[i];
        }
        if (lane == 63) s_pre[nruns] = incl;
    }
    __syncthreads();
    const int n_all = (int)s_pre[nruns];

    // this wave's region (blocks NB*wx .. NB*wx + NB-1, NB*wy .. of the tile) and this lane's NB x NB pixels
    const int region = wave & 3, part = wave >> 2;  // waves r, r + 4, ... take alternate candidates of region r
    const int wx = region & 1, wy = region >> 1;
    const int x0 = tx0 + wx * (8 * NB) + (lane & 7);
    const int r0 = tr0 + wy * (8 * NB) + (lane >> 3);
    double px[NB], py[NB];
    // tile-local sample coordinates (exact small integers): dx = px - (tx0 + 0.5), dy = py - py(tile's top row)
    float dxl[NB], dyl[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        px[k] = (double)(x0 + 8 * k) + 0.5;
        py[k] = (double)(p.H - 1 - (r0 + 8 * k)) + 0.5;
        dxl[k] = (float)(x0 + 8 * k - tx0);
        dyl[k] = (float)(tr0 - (r0 + 8 * k));
    }
    const float wf = (float)p.W, hf = (float)p.H;

    uint32_t zbest[NB * NB];
    int32_t fbest[NB * NB];
#pragma unroll
    for (int k = 0; k < NB * NB; ++k) { zbest[k] = Z24_CLEAR; fbest[k] = -1; }  // -1: a tie with the cleared depth never wins

    TRACE_MARK();  // 1: directory loaded
    for (int round = 0; round < n_all; round += LIST_CAP) {
        if (tid == 0) s_count = 0;
        __syncthreads();
        TRACE_MARK();  // 2: first barrier
        const int round_end = min(n_all, round + LIST_CAP);
        for (int base = round; base < round_end; base += RT) {
            const int e = base + tid;
            bool hit = false;
            BinEntry en;
            if (e < round_end) {
                // run j with s_pre[j] <= e < s_pre[j + 1] (binary search; empty runs are skipped by the order)
                int lo = 0, hi = nruns - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_pre[mid] <= (uint32_t)e) lo = mid; else hi = mid - 1;
                }
                en = scene_entries[s_run_base[lo] + ((uint32_t)e - s_pre[lo])];
                hit = en.box.i_min <= tx1 && en.box.i_max >= tx0 && en.box.r_min <= tr1 && en.box.r_max >= tr0;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            TRACE_MARK();  // 3: entries loaded
            if (m) {
                uint32_t off = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) off = atomicAdd(&s_count, (uint32_t)__popcll(m));
                off = __shfl(off, leader);
                if (hit) {
                    const uint32_t slot = off + __popcll(m & ((1ull << lane) - 1ull));
                    const int bx0 = max(en.box.i_min - tx0, 0) >> 3, bx1 = min(en.box.i_max - tx0, TILE_W - 1) >> 3;
                    const int by0 = max(en.box.r_min - tr0, 0) >> 3, by1 = min(en.box.r_max - tr0, TILE_H - 1) >> 3;
                    const uint32_t rowbits = ((2u << bx1) - (1u << bx0)) & ((1u << BT) - 1u);
                    uint32_t mask = 0;
                    for (int by = by0; by <= by1; ++by) mask |= rowbits << (BT * by);
                    s_face[slot] = en.face;
                    s_mask[slot] = (uint16_t)mask;
                }
            }
        }
        TRACE_MARK();  // 4: appended
        __syncthreads();
        const int n = (int)s_count;
        TRACE_MARK();  // 5: list built

        // ---- candidates, 64 at a time: one lane per candidate builds its tile-local record in LDS (one
        //      memory latency per chunk instead of one per candidate), then every wave walks the ones that
        //      touch its blocks; with ~64 VGPRs there are enough waves in flight to hide the LDS reads ----
        uint32_t seen = 0;  // candidates of this region so far (wave-uniform)
        for (int cb = 0; cb < n; cb += 64) {
            const int m_chunk = min(64, n - cb);
            TRACE_ACC(0);
            if (tid < m_chunk) {
                const int face = s_face[cb + tid];
                make_tile_rec(recs + face, face, (double)tx0 + 0.5, (double)(p.H - 1 - tr0) + 0.5, wf, hf, &s_rec[tid]);
            }
            __syncthreads();
            TRACE_ACC(1);
            const int idx = cb + lane;
            uint32_t mym4 = 0;
            if (idx < n) {
                const uint32_t mk = s_mask[idx];
                // the wave's block bits inside the tile mask, gathered into NB * NB bits (NB * by + bx)
#pragma unroll
                for (int by = 0; by < NB; ++by)
                    mym4 |= ((mk >> ((NB * wy + by) * BT + NB * wx)) & ((1u << NB) - 1u)) << (NB * by);
            }
            unsigned long long m = __builtin_amdgcn_ballot_w64(mym4 != 0);
            if (SPLIT > 1) {  // this wave's share: the candidates of the region whose running rank is `part` modulo SPLIT
                const uint32_t rank = seen + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                seen += (uint32_t)__popcll(m);
                m = __builtin_amdgcn_ballot_w64(mym4 != 0 && (int)(rank % (uint32_t)SPLIT) == part);
            }
            while (m) {
                const int k = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t m4 = (uint32_t)__builtin_amdgcn_readlane((int)mym4, k);
                const TileRec t = s_rec[k];
                raster_candidate<NB>(t, recs, m4, dxl, dyl, px, py, zbest, fbest);
                TRACE_CNT();
            }
            TRACE_ACC(2);
            __syncthreads();
            TRACE_ACC(3);
        }
    }

    TRACE_MARK();  // 6: candidates done

    // ---- resolve through an LDS visibility tile: the per-lane results are scattered to it, then the 256
    //      threads walk the tile row-major (32 consecutive pixels of a row per half-wave: coalesced HWC
    //      stores) to export visibility and / or shade ----
    if (SPLIT > 1) {
        // merge: waves with part > 0 publish their minima, the region's first wave folds them in -- the same
        // (z24, face) lexicographic minimum; -1 = that wave saw no fragment
        if (part > 0) {
#pragma unroll
            for (int k = 0; k < NB * NB; ++k) {
                const int i = (wy * (8 * NB) + (k / NB) * 8 + (lane >> 3)) * TILE_W + wx * (8 * NB) + (k % NB) * 8 + (lane & 7);
                s_mf[(part - 1) * TILE_W * TILE_H + i] = fbest[k];
                s_mz[(part - 1) * TILE_W * TILE_H + i] = zbest[k];
            }
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int k = 0; k < NB * NB; ++k) {
                const int i = (wy * (8 * NB) + (k / NB) * 8 + (lane >> 3)) * TILE_W + wx * (8 * NB) + (k % NB) * 8 + (lane & 7);
#pragma unroll
                for (int q = 0; q < SPLIT - 1; ++q) {
                    const int32_t f2 = s_mf[q * TILE_W * TILE_H + i];
                    const uint32_t z2 = s_mz[q * TILE_W * TILE_H + i];
                    const bool take = (f2 >= 0) & ((z2 < zbest[k]) | ((z2 == zbest[k]) & (f2 < fbest[k])));
                    zbest[k] = take ? z2 : zbest[k];
                    fbest[k] = take ? f2 : fbest[k];
                }
            }
        }
    }
    if (part == 0) {
#pragma unroll
        for (int by = 0; by < NB; ++by)
#pragma unroll
            for (int bx = 0; bx < NB; ++bx)
                s_vis[(wy * (8 * NB) + by * 8 + (lane >> 3)) * TILE_W + wx * (8 * NB) + bx * 8 + (lane & 7)] = fbest[NB * by + bx];
    }
    __syncthreads();
#pragma unroll 1
    for (int i = tid; i < TILE_W * TILE_H; i += RT) {
        const int x = tx0 + (i & (TILE_W - 1)), r = tr0 + i / TILE_W;
        if (r >= p.H || x >= p.W) continue;
        const int32_t f = s_vis[i];
        if (p.vis) p.vis[((size_t)ib * p.H + r) * p.W + x] = f;
        if (MODE == 0) shade_pixel<CSPEC>(p, recs, ib, x, r, (double)x + 0.5, (double)(p.H - 1 - r) + 0.5, f);
        else if (p.state_a) export_state(p, recs, ib, x, r, (double)x + 0.5, (double)(p.H - 1 - r) + 0.5, f);
    }
    TRACE_MARK();  // 7: stored
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_buf) {
        long long* o = g_trace_buf + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = tr_t[i];
        o[1] = tr_acc[1]; o[2] = tr_acc[2]; o[3] = tr_acc[3]; o[4] = tr_cnt;
    }
#endif
}

// ------------------------------------------------------------------------------------------------

#ifdef DIRT_TRACE
extern "C" void dirt_debug_set_trace(void* p)
{
    long long* q = reinterpret_cast<long long*>(p);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &q, sizeof(q));
}
#endif

hipError_t launch_zero(void* b, size_t b_bytes, void* c, size_t c_bytes, hipStream_t stream)
{
    const size_t nb = b_bytes / 4, nc = c_bytes / 4;
    const size_t most = nb > nc ? nb : nc;
    if (most == 0) return hipSuccess;
    unsigned grid = (unsigned)((most + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(b), nb,
                       reinterpret_cast<uint32_t*>(c), nc);
    return hipGetLastError();
}

hipError_t launch_geometry(const GeomParams& g, hipStream_t stream)
{
    if (g.B == 0) return hipSuccess;
    // also with F == 0: the (all-zero) directory row is what the raster kernel reads
    const dim3 grid((unsigned)g.nchunk, (unsigned)g.B);
    hipLaunchKernelGGL(setup_kernel, grid, dim3(256), 0, stream, g);
    return hipGetLastError();
}

void chunking(int F, int& nchunk, int& chunk_faces)
{
    // <= 256 chunks of >= 256 faces: a raster tile reads one directory cell per chunk (2 * nchunk <= 512 runs)
    chunk_faces = 256;
    if ((long long)chunk_faces * 256 < F) chunk_faces = (F + 255) / 256;
    nchunk = F > 0 ? (F + chunk_faces - 1) / chunk_faces : 1;
}

BinGrid make_bin_grid(int H, int W)
{
    BinGrid g;
    g.shift = 5;  // bins of >= 32 pixels: a raster tile (32 or 16 pixels) never straddles two bins
    while (((W + (1 << g.shift) - 1) >> g.shift) * ((H + (1 << g.shift) - 1) >> g.shift) > MAX_BINS) ++g.shift;
    g.bins_x = (W + (1 << g.shift) - 1) >> g.shift;
    g.bins_y = (H + (1 << g.shift) - 1) >> g.shift;
    return g;
}

hipError_t launch_raster(const RasterParams& p_in, int B, bool visibility_only, hipStream_t stream)
{
    if (B == 0) return hipSuccess;
    RasterParams p = p_in;
    // 32 x 32 tiles unless that leaves the chip mostly idle (fewer than two workgroups per CU): then 16 x 16
    const long long tiles32 = (long long)((p.W + 31) / 32) * ((p.H + 31) / 32) * B;
    int tile = tiles32 >= 512 ? 32 : 16;
    if (p.flags & DIRT_FLAG_TILES_LARGE) tile = 32;
    if (p.flags & DIRT_FLAG_TILES_SMALL) tile = 16;
    p.tiles_x = (p.W + tile - 1) / tile;
    p.tiles_y = (p.H + tile - 1) / tile;
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)B);
    const int cspec = visibility_only ? 0 : (p.C == 4 ? 4 : (p.C == 3 ? 3 : (p.C == 1 ? 1 : 0)));
#define DIRT_LAUNCH_RASTER(NB_)                                                                               \
    do {                                                                                                      \
        if (visibility_only) hipLaunchKernelGGL((raster_kernel<1, NB_, 0>), grid, dim3(NB_ == 1 ? 2 * RTHREADS : RTHREADS), 0, stream, p);   \
        else if (cspec == 4) hipLaunchKernelGGL((raster_kernel<0, NB_, 4>), grid, dim3(NB_ == 1 ? 2 * RTHREADS : RTHREADS), 0, stream, p);   \
        else if (cspec == 3) hipLaunchKernelGGL((raster_kernel<0, NB_, 3>), grid, dim3(NB_ == 1 ? 2 * RTHREADS : RTHREADS), 0, stream, p);   \
        else if (cspec == 1) hipLaunchKernelGGL((raster_kernel<0, NB_, 1>), grid, dim3(NB_ == 1 ? 2 * RTHREADS : RTHREADS), 0, stream, p);   \
        else hipLaunchKernelGGL((raster_kernel<0, NB_, 0>), grid, dim3(NB_ == 1 ? 2 * RTHREADS : RTHREADS), 0, stream, p);                   \
    } while (0)
    if (tile == 32) DIRT_LAUNCH_RASTER(2);
    else DIRT_LAUNCH_RASTER(1);
#undef DIRT_LAUNCH_RASTER
    return hipGetLastError();
}

}  // namespace dirt
