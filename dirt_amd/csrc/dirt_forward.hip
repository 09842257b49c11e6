// dirt_forward.hip -- the forward pass in TWO dependent trips per tile (gfx950, round 6): set-up that leaves tile-ready,
// face-local coverage records behind, and a raster kernel that copies them into LDS by LDS-DMA.
//
// Replaces (as dirt_raster.hip does, whose kernels remain for meshes of more than 16 384 faces, 16 x 16 tiles and channel
// counts other than 1, 3, 4): the GL vertex pipeline + upload_vertices (csrc/rasterise_grad_egl.cu:12-34), the per-scene
// { glViewport, glScissor, glClear(DEPTH), glDrawElementsBaseVertex } (csrc/rasterise_egl.cpp:362-380) and
// upload_background / download_pixels (csrc/rasterise_egl.cu:10-38,65-91).
//
// Rounds 1-5: a raster tile walked directory -> entries {box, face} -> set-up records -> vertex colours: FOUR dependent memory
// round trips and three workgroup barriers before the first sample was tested -- 7.3 of the kernel's 18.7 us at K3, the same
// whether 576 or 1024 tiles are rendered (profiles/EXPERIMENTS.md, round 4 knock-outs).  Here:
//   * setup_kernel_v2 (a workgroup of four waves per chunk of 64 faces, one face per lane in each, every wave its own part of
//     the face's set-up) writes per face, besides the 128-byte set-up
//     record, its FACE-LOCAL coverage record (make_local_rec: float32 edge functions about the top-left pixel of the face's own
//     box + certified bound + f64 depth plane + box: the 80 bytes the coverage loop reads, valid for every tile) and its three
//     vertex colours as float4s; bins are 32-pixel squares -- the raster tiles themselves up to 1024 x 1024 -- so a bin's
//     faces ARE a tile's candidates and no box list is consulted; a face may touch up to 16 bins before it goes to the
//     "big" list.  For meshes whose faces are (3f, 3f+1, 3f+2) -- split vertices, what dirt/lighting.py's
//     split_vertices_by_face produces -- the vertices and colours are requested together with the indices (one round trip).
//   * raster_kernel_v2: trip 1 reads the tile's row of the bin x chunk mask directory; the set bits are the candidates; trip 2
//     is ONE round of LDS-DMA: the candidates' 80-byte coverage records straight into LDS (lane <-> 16-byte piece, linear),
//     then -- issued behind them, landing while the coverage loop runs -- the 96-byte heads of their set-up records and their
//     colours for the shading pass.  Coverage, depth and shading arithmetic are dirt_raster.hip's (dirt_raster_common.h).
//     Two shapes: four waves per 32 x 32 tile (16 x 16 pixels each), and eight half-size waves for launches of at most 2048
//     tiles, which end with their heaviest wave.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "dirt_raster_common.h"
#include "../../include/dirt_hip.h"

namespace dirt {

#ifdef DIRT_TRACE
// Per-wave phase timestamps (s_memtime) for tools/trace_forward.py; compiled only into the tracing build of the library.
__device__ long long* g_trace_fwd_setup = nullptr;
__device__ long long* g_trace_fwd_raster = nullptr;
extern "C" void dirt_debug_set_trace_forward(void* ps, void* pr)
{
    long long* q = reinterpret_cast<long long*>(ps);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_fwd_setup), &q, sizeof(q));
    q = reinterpret_cast<long long*>(pr);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_fwd_raster), &q, sizeof(q));
}
#define FMARK() do { if (tr_n < 12) { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tr_t[tr_n++] = t_; } } while (0)
#define FTRACE_DECL() long long tr_t[12]; int tr_n = 0; const long long tr_wall0 = wall_clock64()
#else
#define FMARK() do {} while (0)
#define FTRACE_DECL() do {} while (0)
#endif

namespace {

constexpr int MAX_FACE_BINS = 16;   // bins a face may touch before it goes to the "big" pseudo-bin (read by every tile)

// One LDS-DMA wave-instruction: lane l's 16 bytes at base + voff land at LDS byte address lds_addr + 16 l (lanes switched off
// by the surrounding branch write nothing).  M0 holds the destination; saved and restored around the statement.  Not counted
// by the compiler: the kernel waits with s_waitcnt vmcnt(0) itself.
__device__ __forceinline__ void glds16(const void* base, uint32_t voff, uint32_t lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

struct Float3v { float x, y, z; };   // three channels of a vertex colour: one 12-byte load

template <class T>
__device__ __forceinline__ uint32_t lds_address(T* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p;
}

}  // namespace

// ---- set-up ----------------------------------------------------------------------------------------------------------
// CK: the channel count whose vertex colours ride along (1, 3, 4), or 0: none (the visibility / stateless backward passes).
// One workgroup of FOUR waves per chunk of 64 faces, one face per lane in EVERY wave, each wave with its own part of the face's
// set-up.  The chip is almost empty while this kernel runs (157 chunks at K3) and a wave's time is its instruction count
// (profiles/EXPERIMENTS.md), so the ~700-instruction chain of a face is cut across waves instead of lanes:
//   wave 0  the edge functions, det, depth plane, sign folding -> the set-up record (setup_face_edges), left in LDS;
//   wave 1  the conservative pixel box (setup_face_box); after the barrier, once wave 0's verdict is known: the face's bit in
//           the mask of every bin the box touches;
//   wave 2  clears the chunk's 1 026 masks while the others wait for their vertices; after the barrier: the face-local float32
//           coverage record from wave 0's record and wave 1's box (make_local_rec);
//   wave 3  fetches the three vertex colours (its own request of the indices) -> the colour record.
// Waves 0, 1, 3 each request the indices and -- speculatively, for the identity triple -- their vertices / colours themselves
// (a few hundred bytes more through the L2 instead of a dependent LDS hand-over).  All records go through LDS and leave as
// linear 16-byte runs: the chunk's records are contiguous in memory (8 + 5 + 3 KB).
// (Rounds 5-6a: one wave did everything, 8 195 clocks; four waves with the arithmetic still on one: 7 621.)
constexpr int STHREADS = 256;
// (Leading scalar arguments = the fields the first trip is addressed with, PRELOADED into SGPRs at wave launch: see raster_kernel_v2.)
template <int CK>
__global__ __launch_bounds__(STHREADS) void setup_kernel_v2(const float* __restrict__ a_vertices, const int32_t* __restrict__ a_faces,
                                                            const float* __restrict__ a_vertex_colors, int a_shared_faces, int a_V, int a_F,
                                                            GeomParams g_in)
{
    GeomParams g = g_in;
    g.vertices = a_vertices; g.faces = a_faces; g.vertex_colors = a_vertex_colors; g.shared_faces = a_shared_faces; g.V = a_V; g.F = a_F;
    __shared__ __align__(16) unsigned long long s_mask[MAX_BINS_MASKED + 2];    // [grid.big] = big faces
    __shared__ __align__(16) FaceRec s_recs[64];
    __shared__ __align__(16) TileRec s_lrecs[64];
    __shared__ __align__(16) float4 s_crecs[CK ? 3 * 64 : 1];
    __shared__ FaceBox s_box[64];
    __shared__ uint8_t s_ok_edges[64], s_ok_box[64];
    const int ib = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0 edges, 1 box + masks, 2 mask clearing + local records, 3 colours
    FTRACE_DECL();
    FMARK();  // 0 start
    const int nbins = g.grid.bins_x * g.grid.bins_y, big = g.grid.big;
    BinCell* __restrict__ row = g.cells + ((size_t)ib * g.nchunk + chunk) * (size_t)g.grid.cell_chunk_stride;   // chunk-major: this chunk's cells are contiguous
    if (g.F == 0 || g.V <= 0) {   // (uniform) nothing to set up: an all-zero row is what the raster kernel reads
        for (int i = tid; i <= big; i += STHREADS) if (i < nbins || i == big) row[i] = BinCell{0u, 0u};
        return;
    }
    const int f0 = chunk * 64;
    const int nf = min(64, g.F - f0);            // faces of this chunk (the last one may be short)
    const size_t n0 = (size_t)ib * g.F + f0;     // ... and their first record
    const int f = f0 + lane;
    const bool have = lane < nf;
    const int fs = have ? f : 0;                 // (lanes past the last face: face 0, nothing stored)
    const float* __restrict__ verts = g.vertices + (size_t)ib * g.V * 4;
    const float* __restrict__ cols = CK ? g.vertex_colors + (size_t)ib * g.V * CK : nullptr;
    // what a role reads of a vertex: its clip-space position, or (role 3) its colour
    auto fetch = [&](int vid) {
        if (role != 3) return *reinterpret_cast<const float4*>(verts + (size_t)vid * 4);
        if constexpr (CK == 0) return make_float4(0.f, 0.f, 0.f, 0.f);
        else {
            const float* __restrict__ cp = cols + (size_t)vid * CK;
            if constexpr (CK == 4) return *reinterpret_cast<const float4*>(cp);
            else if constexpr (CK == 3) { const Float3v q = *reinterpret_cast<const Float3v*>(cp); return make_float4(q.x, q.y, q.z, 0.f); }
            else return make_float4(cp[0], 0.f, 0.f, 0.f);
        }
    };
    int32_t idx[3] = {0, 0, 0};
    float4 vv[3];
    const bool fetches = role != 2 && (role != 3 || CK != 0);
    if (fetches) {
        // The face's indices and -- speculatively, for the identity triple (3f, 3f+1, 3f+2) of split-vertex meshes -- its vertices
        // (colours) are requested TOGETHER, branch-free (clamped addresses: a load inside a divergent branch is waited for inside
        // it): one memory round trip instead of two for such meshes; any other mesh pays the unused requests and takes the second
        // trip below.
        const int32_t* __restrict__ fp = g.faces + (g.shared_faces ? (size_t)fs : (size_t)ib * g.F + fs) * 3;
        idx[0] = fp[0]; idx[1] = fp[1]; idx[2] = fp[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) vv[k] = fetch(min(3 * fs + k, g.V - 1));
    } else if (role == 2) {
        for (int i = lane; i < nbins; i += 64) s_mask[i] = 0ull;
        if (lane == 0) s_mask[big] = 0ull;
    }
    FMARK();  // 1 requests issued / masks cleared
    if (fetches) {
        const bool identity = 3 * fs + 2 < g.V && idx[0] == 3 * fs && idx[1] == 3 * fs + 1 && idx[2] == 3 * fs + 2;
        if (__builtin_amdgcn_ballot_w64(have && !identity) != 0ull) {   // (wave-uniform) some face of the chunk is not the identity triple
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ci = identity ? 3 * fs + k : ((uint32_t)idx[k] < (uint32_t)g.V ? idx[k] : 0);   // a bad index reads vertex 0 (the face is dropped by setup_face_edges)
                vv[k] = fetch(ci);
            }
        }
    }
    FMARK();  // 2 indices and vertices there
    bool ok_mine = false;
    FaceBox box;
    box.i_min = 0; box.i_max = -1; box.r_min = 0; box.r_max = -1;
    if (role == 0) {
        FaceRec rec;
        ok_mine = have && setup_face_edges(vv, idx, g.V, g.H, g.W, rec);
        if (ok_mine) s_recs[lane] = rec;
        else s_recs[lane].flags = 0;   // (no bit anywhere: the rest of its records is never read)
        s_ok_edges[lane] = ok_mine ? 1 : 0;
    } else if (role == 1) {
        ok_mine = have && setup_face_box(vv, g.H, g.W, box);
        s_box[lane] = box;
        s_ok_box[lane] = ok_mine ? 1 : 0;
    } else if (role == 3) {
        if constexpr (CK != 0) { s_crecs[3 * lane] = vv[0]; s_crecs[3 * lane + 1] = vv[1]; s_crecs[3 * lane + 2] = vv[2]; }
    }
    FMARK();  // 3 this wave's part of the set-up, in LDS
    __syncthreads();
    if (role == 1) {
        if (ok_mine && s_ok_edges[lane]) {
            const unsigned long long bit = 1ull << lane;
            const int bx0 = box.i_min >> g.grid.shift, bx1 = box.i_max >> g.grid.shift;
            const int by0 = box.r_min >> g.grid.shift, by1 = box.r_max >> g.grid.shift;
            if ((bx1 - bx0 + 1) * (by1 - by0 + 1) <= MAX_FACE_BINS) {
                for (int by = by0; by <= by1; ++by)
                    for (int bx = bx0; bx <= bx1; ++bx) atomicOr(&s_mask[by * g.grid.bins_x + bx], bit);
            } else {
                atomicOr(&s_mask[big], bit);
            }
            if (!g.v2_only) {
                // (the {box, face} entry at its fixed slot: what dirt_raster.hip's kernels -- 16 x 16 tiles, other channel counts --
                // read behind the same masks)
                BinEntry e;
                e.box = box; e.face = f; e.pad = 0;
                g.entries[((size_t)ib * g.nchunk + chunk) * (5 * (size_t)g.chunk_faces) + lane] = e;
            }
        }
    } else if (role == 2) {
        if (have && s_ok_edges[lane] && s_ok_box[lane]) {
            TileRec lr;
            make_local_rec(s_recs[lane], f, s_box[lane], g.H, (float)g.W, (float)g.H, &lr);
            s_lrecs[lane] = lr;
        }
    } else if (role == 0) {
        // (meanwhile: the set-up records leave, 8 pieces of 16 bytes per face, linear)
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(&s_recs[0]);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(g.recs + n0);
        for (int i = lane; i < 8 * nf; i += 64) dst[i] = src[i];
    } else {
        if constexpr (CK != 0) {
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(&s_crecs[0]);
            uint4* __restrict__ dst = reinterpret_cast<uint4*>(g.crecs + 3 * n0);
            for (int i = lane; i < 3 * nf; i += 64) dst[i] = src[i];
        }
    }
    __syncthreads();
    FMARK();  // 4 masks and local records complete
    // ---- the chunk's coverage records (5 pieces per face, linear) and its row of the directory (two cells per store: 16-byte
    //      accesses on both sides; rows start at 16-byte boundaries) ----
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(&s_lrecs[0]);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(g.lrecs + n0);
        for (int i = tid; i < 5 * nf; i += STHREADS) dst[i] = src[i];
    }
    for (int i = 2 * tid; i < nbins; i += 2 * STHREADS) {
        if (i + 1 < nbins) {
            const ulonglong2 m = *reinterpret_cast<const ulonglong2*>(&s_mask[i]);
            *reinterpret_cast<ulonglong2*>(&row[i]) = m;
        } else {
            const unsigned long long m = s_mask[i];
            row[i] = BinCell{(uint32_t)m, (uint32_t)(m >> 32)};
        }
    }
    if (tid == STHREADS - 1) {
        const unsigned long long m = s_mask[big];
        row[big] = BinCell{(uint32_t)m, (uint32_t)(m >> 32)};
    }
    FMARK();  // 5 record and directory stores issued
#ifdef DIRT_TRACE
    if (tid == 0 && g_trace_fwd_setup) {
        long long* o = g_trace_fwd_setup + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
        for (int i = 0; i < 12; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
        o[12] = tr_wall0; o[13] = (long long)wall_clock64() - tr_wall0;
    }
#endif
}

hipError_t launch_geometry_v2(const GeomParams& g, hipStream_t stream)
{
    const dim3 grid((unsigned)g.nchunk, (unsigned)g.B);
    const int ck = g.crecs ? g.C : 0;
    if (ck == 4) hipLaunchKernelGGL(setup_kernel_v2<4>, grid, dim3(STHREADS), 0, stream, g.vertices, g.faces, g.vertex_colors, g.shared_faces, g.V, g.F, g);
    else if (ck == 3) hipLaunchKernelGGL(setup_kernel_v2<3>, grid, dim3(STHREADS), 0, stream, g.vertices, g.faces, g.vertex_colors, g.shared_faces, g.V, g.F, g);
    else if (ck == 1) hipLaunchKernelGGL(setup_kernel_v2<1>, grid, dim3(STHREADS), 0, stream, g.vertices, g.faces, g.vertex_colors, g.shared_faces, g.V, g.F, g);
    else hipLaunchKernelGGL(setup_kernel_v2<0>, grid, dim3(STHREADS), 0, stream, g.vertices, g.faces, g.vertex_colors, g.shared_faces, g.V, g.F, g);
    return hipGetLastError();
}

// ---- raster ----------------------------------------------------------------------------------------------------------
namespace {

#ifndef DIRT_V2_CAP
#define DIRT_V2_CAP 96
#endif
constexpr int V2_CAP = DIRT_V2_CAP;       // candidates per round whose records live in LDS

// What the shading pass reads of a candidate, as LDS-DMA leaves it: the head of its FaceRec, then (MODE 0) its colours.
template <bool COLOURS>
struct alignas(16) ShadeSlot {
    double coef[9];
    double inv_det;
    uint32_t flags;
    int32_t vid[3];
    float4 col[COLOURS ? 3 : 0];
};
static_assert(sizeof(ShadeSlot<true>) == 144 && sizeof(ShadeSlot<false>) == 96, "16-byte pieces: 6 of the record, 3 colours");

}  // namespace

// raster_kernel_v2<MODE, CSPEC, WAVES>: MODE 0 renders CSPEC = 1, 3, 4 channels (pixels, and the backward pass's state when
// p.state_a); MODE 1 is the visibility pass (p.vis and / or the state; no colours).  32 x 32-pixel tiles.  WAVES = 4: each
// wave owns a 16 x 16 region = 2 x 2 blocks of 8 x 8 pixels (one pixel of every block per lane), from the coverage loop to the
// stores: dirt_raster.hip's decomposition.  WAVES = 8 (512 threads, <= 64 registers, eight waves per SIMD): a wave owns a
// 16 x 8 region = 2 x 1 blocks -- for launches whose workgroups are all resident at once (one scene of 1024 x 1024): the
// kernel then ends with its heaviest wave, and half-size waves halve that wave's serial candidate loop and its shading.
#ifndef DIRT_V2_WAVES
#define DIRT_V2_WAVES 6   // waves per SIMD the four-wave shape is compiled for: 80 VGPRs, six 22 KB workgroups per CU (round 6: 4 -> 6: raster -2.4...-5 % in multi-round launches)
#endif
// (The leading scalar arguments repeat the fields of `p_in` that the first memory trip is addressed with: gfx950 PRELOADS the
// first kernel arguments into SGPRs while the wave is launched (-amdgpu-kernarg-preload-count, dirt_amd/build.py), so the
// directory reads are issued without waiting for a first-touch scalar load of the kernel argument segment.)
template <int MODE, int CSPEC, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 8 : DIRT_V2_WAVES) void raster_kernel_v2(
    const BinCell* __restrict__ a_cells, int a_nchunk, int a_tiles_x, int a_tiles_y, uint32_t a_tiles_x_magic, int a_shift, int a_bins_x,
    int a_big, int a_chunk_stride, RasterParams p_in)
{
    RasterParams p = p_in;
    p.cells = a_cells; p.nchunk = a_nchunk; p.tiles_x = a_tiles_x; p.tiles_y = a_tiles_y; p.tiles_x_magic = a_tiles_x_magic;
    p.grid.shift = a_shift; p.grid.bins_x = a_bins_x; p.grid.big = a_big; p.grid.cell_chunk_stride = a_chunk_stride;
    constexpr int NB = 2, NBY = WAVES == 8 ? 1 : 2, TILE = 32, PPL = NB * NBY, THREADS = 64 * WAVES, RH = 8 * NBY;
    constexpr bool COLOURS = MODE == 0;
    constexpr int SPARTS = COLOURS ? 9 : 6;
    using Slot = ShadeSlot<COLOURS>;
    __shared__ __align__(16) TileRec s_rec[V2_CAP];
    __shared__ __align__(16) Slot s_shade[V2_CAP];
    __shared__ int32_t s_face[V2_CAP];
    __shared__ uint32_t s_count;
    FTRACE_DECL();
    FMARK();  // 0 start
#ifdef DIRT_TRACE
    int tr_cand = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ib = blockIdx.y;
    const int tile = xcd_tile(blockIdx.x, p.tiles_x * p.tiles_y);
    int tile_col, tile_row;
    tile_xy(tile, p.tiles_x, p.tiles_x_magic, tile_col, tile_row);
    const int tx0 = tile_col * TILE, tr0 = tile_row * TILE;
    constexpr int C = CSPEC;

    const FaceRec* __restrict__ recs = p.recs + (size_t)ib * p.F;
    const TileRec* __restrict__ lrecs = p.lrecs + (size_t)ib * p.F;
    const float4* __restrict__ crecs = COLOURS ? p.crecs + (size_t)ib * p.F * 3 : nullptr;
    // ---- trip 1: this tile's cell of every chunk -- the mask of the chunk's faces that touch the tile's bin -- and the
    //      chunk's mask of "big" faces.  One chunk per thread (<= 256 chunks of 64 faces). ----
    const int bin = (tr0 >> p.grid.shift) * p.grid.bins_x + (tx0 >> p.grid.shift);
    const BinCell* __restrict__ cells = p.cells + (size_t)ib * p.nchunk * (size_t)p.grid.cell_chunk_stride;
    unsigned long long m_bin = 0ull, m_big = 0ull;
    if (tid < p.nchunk) {
        const BinCell* __restrict__ rowc = cells + (size_t)tid * p.grid.cell_chunk_stride;   // (chunk-major: cell_bin_stride is 1)
        const BinCell a = rowc[bin], b = rowc[p.grid.big];
        m_bin = ((unsigned long long)a.count << 32) | a.start;
        m_big = ((unsigned long long)b.count << 32) | b.start;
    }

    // this wave's region (16 pixels wide, RH tall) and this lane's NB x NBY pixels, one of every 8 x 8 block
    const int wx = wave & 1, wy = wave >> 1;
    const int x0 = tx0 + wx * 16 + (lane & 7);
    const int r0 = tr0 + wy * RH + (lane >> 3);
    double px[NB], py[NBY];
#pragma unroll
    for (int k = 0; k < NB; ++k) px[k] = (double)(x0 + 8 * k) + 0.5;
#pragma unroll
    for (int k = 0; k < NBY; ++k) py[k] = (double)(p.H - 1 - (r0 + 8 * k)) + 0.5;
    unsigned long long best[PPL];   // (z24 << 32 | face) of the front-most fragment so far
    int cbest[PPL];                 // the winner's slot (its shading data is in LDS when < V2_CAP and lds_records)
#pragma unroll
    for (int k = 0; k < PPL; ++k) { best[k] = (unsigned long long)Z24_CLEAR << 32; cbest[k] = V2_CAP; }  // a tie with the cleared depth never wins
    bool lds_records = true;   // false once a second round has reused the slots (dense meshes): shading data comes from memory then

    // side job: this workgroup's share of the buffers the launch clears (the backward pass's gradient accumulators)
    if ((p.zero_b_bytes | p.zero_c_bytes) != 0) {
        const unsigned gwg = blockIdx.y * gridDim.x + blockIdx.x;
        if (p.zero_b_bytes) zero_share<THREADS>(p.zero_b, p.zero_b_bytes, p.zero_b_per, gwg, tid);
        if (p.zero_c_bytes) zero_share<THREADS>(p.zero_c, p.zero_c_bytes, p.zero_c_per, gwg, tid);
    }
    const uint32_t lds_rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(&s_rec[0]));
    const uint32_t lds_shade = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(&s_shade[0]));

    FMARK();  // 1 cells requested, side job issued
    for (int round = 0;; ++round) {
        if (tid == 0) s_count = 0;
        __syncthreads();
        FMARK();  // 2 (round 0)
        // ---- the candidates: every set bit claims a slot (a wave-wide prefix of the threads' bit counts, ONE LDS atomic per
        //      wave); bits beyond the round's capacity stay for the next round ----
        {
            const uint32_t cnt = (uint32_t)__popcll(m_bin) + (uint32_t)__popcll(m_big);
            uint32_t incl = cnt;   // inclusive prefix over the 64 lanes: DPP row shifts, then the rows' totals carried across
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112 /* row_shr:2 */, 0xF, 0xF, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t base = 0;
            if (total != 0u) {   // (wave-uniform)
                if (lane == 0) base = atomicAdd(&s_count, total);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            }
            uint32_t slot = base + incl - cnt;
            const int face0 = tid * 64;
            while (m_bin != 0ull && slot < (uint32_t)V2_CAP) {
                s_face[slot++] = face0 + (__ffsll((long long)m_bin) - 1);
                m_bin &= m_bin - 1ull;
            }
            while (m_big != 0ull && slot < (uint32_t)V2_CAP) {
                s_face[slot++] = face0 + (__ffsll((long long)m_big) - 1);
                m_big &= m_big - 1ull;
            }
        }
        FMARK();  // 3 cells there, slots claimed
        __syncthreads();
        FMARK();  // 4
        const int n = (int)min(s_count, (uint32_t)V2_CAP);
        if (round != 0) lds_records = false;

        // ---- trip 2: the candidates' coverage records, 16-byte piece idx = 5 slot + part, lane-linear into s_rec ----
        // (the pieces' slot / part are functions of the thread index alone; an opaque copy of it keeps the compiler from hoisting
        // a dozen of them out of the round loop, where they would be live -- in the eight-wave shape: spilled -- across everything)
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < (5 * V2_CAP + THREADS - 1) / THREADS; ++k) {
            const int idx = tid_o + THREADS * k;
            if (idx < 5 * n) {
                const int slot = idx / 5, part = idx - 5 * slot;
                glds16(lrecs, (uint32_t)s_face[slot] * (uint32_t)sizeof(TileRec) + 16u * (uint32_t)part, lds_rec + 1024u * (uint32_t)(wave + WAVES * k));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FMARK();  // 5 coverage records landed
        __syncthreads();
        FMARK();  // 6
        // ---- ... and, landing while the coverage loop runs, what the shading pass reads of them: piece idx = SPARTS slot +
        //      part: the six pieces of the set-up record's head, then the three colours ----
        if (round == 0) {
#pragma unroll
            for (int k = 0; k < (SPARTS * V2_CAP + THREADS - 1) / THREADS; ++k) {
                const int idx = tid_o + THREADS * k;
                if (idx < SPARTS * n) {
                    const int slot = idx / SPARTS, part = idx - SPARTS * slot;
                    const uint32_t face = (uint32_t)s_face[slot];
                    const uint32_t dst = lds_shade + 1024u * (uint32_t)(wave + WAVES * k);
                    if (part < 6) glds16(recs, face * (uint32_t)sizeof(FaceRec) + 16u * (uint32_t)part, dst);
                    if (COLOURS && part >= 6) glds16(crecs, face * 48u + 16u * (uint32_t)(part - 6), dst);
                }
            }
        }

        // ---- coverage + depth: every wave visits the candidates whose box touches one of its four blocks ----
        for (int cb = 0; cb < n; cb += 64) {
            const int idx = cb + lane;
            uint32_t mym4 = 0;
            if (idx < n) {
                const FaceBox box = s_rec[idx].box;
                // the wave's blocks (bx, by) the box touches, as bits 2 by + bx
                const int rx0 = tx0 + 16 * wx, ry0 = tr0 + RH * wy;
                const int bx0 = max(box.i_min - rx0, 0) >> 3, bx1 = min(box.i_max - rx0, 15) >> 3;
                const int by0 = max(box.r_min - ry0, 0) >> 3, by1 = min(box.r_max - ry0, RH - 1) >> 3;
                if (box.i_max >= rx0 && box.i_min <= rx0 + 15 && box.r_max >= ry0 && box.r_min <= ry0 + RH - 1) {
                    const uint32_t rowbits = (bx0 == 0 ? 1u : 0u) | (bx1 == 1 ? 2u : 0u);
                    mym4 = (by0 == 0 ? rowbits : 0u) | (NBY == 2 && by1 == 1 ? rowbits << 2 : 0u);
#ifndef DIRT_NO_BLOCK_CULL
                    // ... minus the blocks the TRIANGLE misses although its box touches them (cull_blocks, dirt_raster_common.h;
                    // face-local record: offsets from the top-left sample of the face's box)
                    mym4 = cull_blocks<NB, NBY>(s_rec[idx], mym4, (float)(rx0 - (int)box.i_min), (float)((int)box.r_min - ry0));
#endif
                }
            }
            unsigned long long m = __builtin_amdgcn_ballot_w64(mym4 != 0);
            auto visit = [&](const TileRec& t, int k) {
                const uint32_t m4 = (uint32_t)__builtin_amdgcn_readlane((int)mym4, k);
                // sample offsets from the face's origin (the top-left pixel of its box): exact small integers
                float dxl[NB], dyl[NBY];
#pragma unroll
                for (int q = 0; q < NB; ++q) dxl[q] = (float)(x0 + 8 * q - (int)t.box.i_min);
#pragma unroll
                for (int q = 0; q < NBY; ++q) dyl[q] = (float)((int)t.box.r_min - (r0 + 8 * q));
                raster_candidate<NB, NBY>(t, recs, round == 0 ? cb + k : V2_CAP, m4, dxl, dyl, px, py, best, cbest);
#ifdef DIRT_TRACE
                ++tr_cand;
#endif
            };
            // (no software prefetch of the next candidate's record -- two candidates per trip, the second record requested before
            // the first is worked on, measured the same within noise at K3, K3-2048 and eight scenes per launch for 19 more
            // registers: profiles/EXPERIMENTS.md round 6)
            while (m) {
                const int k = __ffsll((long long)m) - 1;
                m &= m - 1;
                const TileRec t = s_rec[cb + k];
                visit(t, k);
            }
        }
        FMARK();  // 7 coverage loop done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the shading data has landed
        const bool more = (m_bin | m_big) != 0ull;
        if (!__syncthreads_or(more)) break;
    }
    FMARK();  // 8 shading data landed, barrier

    int32_t fbest[PPL];   // the front-most face per pixel, -1: none (nothing was less than the cleared depth)
#pragma unroll
    for (int k = 0; k < PPL; ++k) fbest[k] = (uint32_t)(best[k] >> 32) != Z24_CLEAR ? (int32_t)(uint32_t)best[k] : -1;

    // ---- shade ----
    // Per pixel: barycentrics of the winner (csrc/shaders.cpp:52-57,74) from its record in LDS -- or, for candidates of later
    // rounds, in memory --, the backward pass's state, the interpolated colours, the HWC pixel; background where nothing is
    // visible.  (WAVES = 4 requests the background of all its pixels first and uses it last; the eight-wave shape has 64
    // registers and twice the waves to hide a load behind: it fetches per pixel.  Bringing it in by LDS-DMA before the
    // candidates instead was measured: K3 raster 20.7 against 18.3 us -- 16 MB of requests in front of the directory reads.)
    constexpr bool BG_FIRST = WAVES != 8;
    const float* __restrict__ cols = COLOURS ? p.vertex_colors + (size_t)ib * p.V * C : nullptr;
    // (opaque copies of the lane's first pixel: the other pixels' coordinates are re-derived here, one add each, instead of
    // being kept -- at a register bound: spilled -- from the prologue through the coverage loop)
    int xs0 = x0, rs0 = r0;
    asm volatile("" : "+v"(xs0), "+v"(rs0));
    auto pixel_index = [&](int k, bool& inside) {
        const int x = xs0 + 8 * (k % NB), r = rs0 + 8 * (k / NB);
        inside = x < p.W && r < p.H;
        return ((size_t)ib * p.H + min(r, p.H - 1)) * p.W + min(x, p.W - 1);
    };
    auto background_at = [&](size_t pix) {
        const float* __restrict__ bg = p.background + pix * C;
        if (CSPEC == 4) return *reinterpret_cast<const float4*>(bg);
        else if (CSPEC == 3) return make_float4(bg[0], bg[1], bg[2], 0.f);
        else return make_float4(bg[0], 0.f, 0.f, 0.f);
    };
    float4 bgv[BG_FIRST ? PPL : 1];
    if constexpr (BG_FIRST) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            bool inside;
            const size_t pix = pixel_index(k, inside);
            bgv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 0 && inside && fbest[k] < 0) bgv[k] = background_at(pix);
        }
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int32_t f = fbest[k];
        const bool has = f >= 0;
        const bool from_lds = has && lds_records && cbest[k] < V2_CAP;
        const int ci = from_lds ? cbest[k] : 0;   // (lanes without a winner read slot 0; what they compute is not used)
        double cf[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) cf[i] = s_shade[ci].coef[i];
        uint32_t flags = s_shade[ci].flags;
        double inv_det = s_shade[ci].inv_det;
        const bool from_mem = has && !from_lds;
        const bool any_from_mem = __builtin_amdgcn_ballot_w64(from_mem) != 0ull;   // (wave-uniform; rare: tiles of more than V2_CAP candidates)
        if (any_from_mem && from_mem) {
            const FaceRec* __restrict__ rec = recs + f;
#pragma unroll
            for (int i = 0; i < 9; ++i) cf[i] = rec->coef[i];
            flags = rec->flags; inv_det = rec->inv_det;
        }
        double Fk[3];
        edge_eval(cf, (double)(xs0 + 8 * (k % NB)) + 0.5, (double)(p.H - 1 - (rs0 + 8 * (k / NB))) + 0.5, Fk);
        float b[3], cw;
        bary_eval(Fk, flags, inv_det, b, cw);
        const float b0 = b[0], b1 = b[1], b2 = b[2];
        bool inside;
        const size_t pix = pixel_index(k, inside);
        if (!inside) continue;
        // the backward pass's state and the visibility export
        if (p.vis) p.vis[pix] = f;
        if (p.state_a) store_state(p, pix, has, b0, b1, b2, cw, f);
        if (MODE != 0) continue;
        float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0, u2 = u0;
        if constexpr (COLOURS) { u0 = s_shade[ci].col[0]; u1 = s_shade[ci].col[1]; u2 = s_shade[ci].col[2]; }
        if (any_from_mem && from_mem) {
            if constexpr (COLOURS) {
                const FaceRec* __restrict__ rec = recs + f;
                const float* __restrict__ c0 = cols + (size_t)rec->vid[0] * C;
                const float* __restrict__ c1 = cols + (size_t)rec->vid[1] * C;
                const float* __restrict__ c2 = cols + (size_t)rec->vid[2] * C;
                if (CSPEC == 4) { u0 = *reinterpret_cast<const float4*>(c0); u1 = *reinterpret_cast<const float4*>(c1); u2 = *reinterpret_cast<const float4*>(c2); }
                else if (CSPEC == 3) { u0 = make_float4(c0[0], c0[1], c0[2], 0.f); u1 = make_float4(c1[0], c1[1], c1[2], 0.f); u2 = make_float4(c2[0], c2[1], c2[2], 0.f); }
                else { u0 = make_float4(c0[0], 0.f, 0.f, 0.f); u1 = make_float4(c1[0], 0.f, 0.f, 0.f); u2 = make_float4(c2[0], 0.f, 0.f, 0.f); }
            }
        }
        float* __restrict__ out = p.pixels + pix * C;
        float4 o;   // pixels start as the background: csrc/rasterise_egl.cpp:348-356
        if constexpr (BG_FIRST) o = bgv[k];
        else o = has ? make_float4(0.f, 0.f, 0.f, 0.f) : background_at(pix);
        if (has) {
            o.x = fmaf(b2, u2.x, fmaf(b1, u1.x, b0 * u0.x));
            if (CSPEC >= 3) { o.y = fmaf(b2, u2.y, fmaf(b1, u1.y, b0 * u0.y)); o.z = fmaf(b2, u2.z, fmaf(b1, u1.z, b0 * u0.z)); }
            if (CSPEC == 4) o.w = fmaf(b2, u2.w, fmaf(b1, u1.w, b0 * u0.w));
        }
        if (CSPEC == 4) *reinterpret_cast<float4*>(out) = o;
        else if (CSPEC == 3) { out[0] = o.x; out[1] = o.y; out[2] = o.z; }
        else out[0] = o.x;
    }
    FMARK();  // 9 shaded, stores issued
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_fwd_raster) {
        long long* o = g_trace_fwd_raster + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 16;
        for (int i = 0; i < 12; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
        o[12] = tr_wall0; o[13] = (long long)wall_clock64() - tr_wall0; o[14] = blockIdx.x; o[15] = tr_cand;
    }
#endif
}

// The forward / visibility kernels' tile: 32 x 32 pixels unless that leaves the chip mostly idle (fewer than two workgroups
// per CU), then 16 x 16 -- or pinned.  The bin grid is sized for this very choice (dirt_capi.hip::geom_params: 16-pixel bins
// under 16 x 16 tiles, so that a bin's faces are a tile's candidates there too: K3-256 step -0.5 us).
int raster_tile_choice(int H, int W, int B, unsigned flags)
{
    const long long tiles32 = (long long)((W + 31) / 32) * ((H + 31) / 32) * B;
    int tile = tiles32 >= 512 ? 32 : 16;
    if (flags & DIRT_FLAG_TILES_SMALL) tile = 16;
    if (flags & DIRT_FLAG_TILES_LARGE) tile = 32;   // (both bits: 32 x 32 tiles, eight waves each: launch_raster_v2)
    return tile;
}

// Which launches take the two-trip kernel: the masked directory with its face-local records (meshes of up to 16 384 faces),
// 32 x 32-pixel tiles, 1 / 3 / 4 channels or the visibility pass.  (A 16 x 16-tile shape of it -- a wave per 8 x 8 block --
// was built and measured at K3-256: 29.2-29.3 us against 28.7-29.2 for dirt_raster.hip's 16 x 16 kernel on the same
// 16-pixel bins, whose two-candidates-per-trip loop it lacks: not kept; profiles/EXPERIMENTS.md round 6.)
bool raster_v2_applies(const RasterParams& p, int B, bool visibility_only)
{
    if (!p.masked || p.lrecs == nullptr) return false;
    if (raster_tile_choice(p.H, p.W, B, p.flags) != 32) return false;
    if (visibility_only) return true;
    return (p.C == 1 || p.C == 3 || p.C == 4) && p.crecs != nullptr;
}

hipError_t launch_raster_v2(const RasterParams& p_in, int B, bool visibility_only, hipStream_t stream)
{
    RasterParams p = p_in;
    p.tiles_x = (p.W + 31) / 32;
    p.tiles_y = (p.H + 31) / 32;
    p.tiles_x_magic = tile_magic(p.tiles_x);
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)B);
    {
        const size_t nwg = (size_t)grid.x * grid.y;
        p.zero_b_per = (unsigned)((p.zero_b_bytes / 16 + nwg - 1) / nwg);
        p.zero_c_per = (unsigned)((p.zero_c_bytes / 16 + nwg - 1) / nwg);
    }
    // half-size waves (eight per workgroup) where (nearly) every workgroup of the launch is resident at once -- four 512-thread
    // workgroups per compute unit --: such a launch ends with its heaviest wave (profiles/EXPERIMENTS.md round 6)
#ifdef DIRT_V2_NO_W8
    const bool w8 = false;
#elif defined(DIRT_V2_W8_ALWAYS)   // (A/B build)
    const bool w8 = !visibility_only;
#else
    const bool pinned8 = (p.flags & (DIRT_FLAG_TILES_LARGE | DIRT_FLAG_TILES_SMALL)) == (DIRT_FLAG_TILES_LARGE | DIRT_FLAG_TILES_SMALL);
    const bool w8 = !visibility_only && (pinned8 || ((size_t)grid.x * grid.y <= 2048 && !(p.flags & DIRT_FLAG_TILES_LARGE)));   // (measured: 1024 tiles -2 us, 2048 -1.5, 4096 and more +5...10)
#endif
#define V2_ARGS p.cells, p.nchunk, p.tiles_x, p.tiles_y, p.tiles_x_magic, p.grid.shift, p.grid.bins_x, p.grid.big, p.grid.cell_chunk_stride, p
    if (visibility_only) hipLaunchKernelGGL((raster_kernel_v2<1, 4>), grid, dim3(RTHREADS), 0, stream, V2_ARGS);
    else if (w8 && p.C == 4) hipLaunchKernelGGL((raster_kernel_v2<0, 4, 8>), grid, dim3(512), 0, stream, V2_ARGS);
    else if (w8 && p.C == 3) hipLaunchKernelGGL((raster_kernel_v2<0, 3, 8>), grid, dim3(512), 0, stream, V2_ARGS);
    else if (w8) hipLaunchKernelGGL((raster_kernel_v2<0, 1, 8>), grid, dim3(512), 0, stream, V2_ARGS);
    else if (p.C == 4) hipLaunchKernelGGL((raster_kernel_v2<0, 4>), grid, dim3(RTHREADS), 0, stream, V2_ARGS);
    else if (p.C == 3) hipLaunchKernelGGL((raster_kernel_v2<0, 3>), grid, dim3(RTHREADS), 0, stream, V2_ARGS);
    else hipLaunchKernelGGL((raster_kernel_v2<0, 1>), grid, dim3(RTHREADS), 0, stream, V2_ARGS);
#undef V2_ARGS
    return hipGetLastError();
}

}  // namespace dirt
