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
#include "dirt_raster_common.h"
#include "../../include/dirt_hip.h"

namespace dirt {

#ifdef DIRT_TRACE
__device__ long long* g_trace_buf = nullptr;
__device__ long long* g_trace_setup = nullptr;
extern "C" void dirt_debug_set_trace_setup(void* p)
{
    long long* q = reinterpret_cast<long long*>(p);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_setup), &q, sizeof(q));
}
#define SETUP_MARK() do { if (st_n < 8) { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_t[st_n++] = t_; } } while (0)
#else
#define SETUP_MARK() do {} while (0)
#endif

// ---- binning ---------------------------------------------------------------------------------
// The frame is cut into at most MAX_BINS square bins of 2^shift pixels (>= 32, so a raster tile never
// straddles two bins).  ONE kernel builds, per scene, exact-size lists of the faces touching each bin -- the
// replacement for the GL driver's own binning hardware -- without a single global atomic, without any
// buffer that needs clearing and without any cross-workgroup dependency:
//   setup_kernel : the scene's faces are cut into `nchunk` contiguous chunks, one single-wave workgroup
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

constexpr int STHREADS = 64;   // one wave per chunk of <= 64 faces: ~160 workgroups for 10 000 faces, and no barrier between its phases
                               // but LDS order; NW = 4 waves per chunk for larger chunks (meshes of more than 16 384 faces: a
                               // thread still owns one face, where a single wave would walk its chunk in several trips)

template <int NW>
__global__ __launch_bounds__(STHREADS * NW) void setup_kernel(GeomParams g)
{
    constexpr int NT = STHREADS * NW;
    __shared__ uint32_t s_cnt[MAX_BINS + 1];    // [MAX_BINS] = big faces
    __shared__ uint32_t s_start[MAX_BINS + 1];  // exclusive prefix of s_cnt = the chunk's segment layout
    const int ib = blockIdx.y, chunk = blockIdx.x, lane = threadIdx.x & 63, tid = threadIdx.x;
#ifdef DIRT_TRACE
    long long st_t[8]; int st_n = 0;
    const long long st_wall0 = wall_clock64();
#endif
    SETUP_MARK();  // 0 start
    static_assert(MAX_BINS == 4 * STHREADS, "four bins per lane");
    const int f0 = chunk * g.chunk_faces, f1 = min(g.F, f0 + g.chunk_faces);
    const float* __restrict__ verts = g.vertices + (size_t)ib * g.V * 4;
    // This thread's first face: its indices, then its vertices, are requested before anything else.  (Rounds 1-3 cleared the
    // backward pass's gradient accumulators here as a side job: 38 400 16-byte stores over 157 single-wave workgroups,
    // 2 600 clocks of every wave's 9 200; the raster kernel's 1024 workgroups do it now, one store per 7 threads.)
    const bool have_first = f0 + tid < f1;
    int32_t idx_first[3] = {0, 0, 0};
    float4 vv_first[3];
    if (have_first) face_fetch_indices(g.faces + (g.shared_faces ? (size_t)(f0 + tid) : (size_t)ib * g.F + f0 + tid) * 3, idx_first);
    for (int i = tid; i <= MAX_BINS; i += NT) s_cnt[i] = 0;
    if (have_first) face_fetch_vertices(verts, g.V, idx_first, vv_first);
    __syncthreads();
    SETUP_MARK();  // 1 cleared, first face requested

    // ---- pass 1: set-up + histogram ----
    // the box of this thread's first face stays in registers for pass 2 (as four scalars: kept as a FaceBox the compiler
    // left a dead 8-byte store to a stack slot behind, and with it a scratch allocation at every dispatch of this kernel)
    int fb_i_min = 32767, fb_i_max = -32768, fb_r_min = 32767, fb_r_max = -32768;
    for (int f = f0 + tid; f < f1; f += NT) {
        const size_t n = (size_t)ib * g.F + f;
        FaceRec rec;
        FaceBox box;
        const bool ok = (f == f0 + tid) ? setup_face_from(vv_first, idx_first, g.V, g.H, g.W, rec, box)
                                        : setup_face(verts, g.V, g.faces + (g.shared_faces ? (size_t)f : n) * 3, g.H, g.W, rec, box);
        if (ok) {
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
        if (f == f0 + tid) { fb_i_min = box.i_min; fb_i_max = box.i_max; fb_r_min = box.r_min; fb_r_max = box.r_max; }
        if (g.chunk_faces > NT) g.boxes[n] = box;  // re-read in pass 2 when a thread owns several faces
    }
    SETUP_MARK();  // 2 pass 1 done (set-up + histogram)
    __syncthreads();
    SETUP_MARK();  // 3

    // ---- the chunk's segment layout: exclusive prefix over the 257 (pseudo-)bins, four bins per lane of the first wave ----
    if (tid < STHREADS) {
        uint32_t cnt[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { cnt[i] = s_cnt[4 * lane + i]; sum += cnt[i]; }
        // inclusive prefix over the 64 lanes: four DPP row shifts inside each row of 16, then the rows' totals carried
        // across with row_bcast:15 / row_bcast:31 (six ds_bpermute shuffles before: an LDS round trip each)
        uint32_t incl = sum;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112 /* row_shr:2 */, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);   // rows 1, 3 += last lane of rows 0, 2
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);   // rows 2, 3 += last lane of row 1
        uint32_t start = incl - sum;
        // the directory is stored bin-major, [bin][chunk]: what a raster tile reads -- its bin's cell of every chunk -- is
        // then contiguous (nchunk x 8 bytes = a few lines, instead of one line per chunk)
        BinCell* __restrict__ col = g.cells + (size_t)ib * (MAX_BINS + 1) * g.nchunk + chunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_start[4 * lane + i] = start;
            col[(size_t)(4 * lane + i) * g.nchunk] = BinCell{start, cnt[i]};
            start += cnt[i];
        }
        if (lane == STHREADS - 1) {
            s_start[MAX_BINS] = start;
            col[(size_t)MAX_BINS * g.nchunk] = BinCell{start, s_cnt[MAX_BINS]};
        }
    }
    __syncthreads();
    for (int i = tid; i <= MAX_BINS; i += NT) s_cnt[i] = 0;  // reused as the fill cursors
    __syncthreads();
    SETUP_MARK();  // 4 prefix + directory done

    // ---- pass 2: faces claim their slots in the chunk's segment ----
    BinEntry* __restrict__ out = g.entries + ((size_t)ib * g.nchunk + chunk) * (5 * (size_t)g.chunk_faces);
    for (int f = f0 + tid; f < f1; f += NT) {
        BinEntry e;
        if (f == f0 + tid) { e.box.i_min = (int16_t)fb_i_min; e.box.i_max = (int16_t)fb_i_max; e.box.r_min = (int16_t)fb_r_min; e.box.r_max = (int16_t)fb_r_max; }
        else e.box = g.boxes[(size_t)ib * g.F + f];
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
    SETUP_MARK();  // 5 pass 2 done
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_setup) {
        long long* o = g_trace_setup + ((size_t)blockIdx.x * NW + (tid >> 6)) * 16;
        for (int i = 0; i < 8; ++i) o[i] = i < st_n ? st_t[i] : 0;
        o[8] = st_wall0; o[9] = (long long)wall_clock64() - st_wall0;
    }
#endif
}

// raster_kernel<MODE, NB, CSPEC>: MODE 0 renders (pixels, and the backward pass's state when p.state_a), MODE 1 is the
// visibility pass (p.vis and / or the state; no colours).  CSPEC = 1, 3, 4: that channel count, vertex colours of the
// listed candidates staged in LDS; 0: any channel count.
//
// One 256-thread workgroup = one tile of one scene; each of its 4 waves owns a region of NB x NB blocks of 8 x 8 pixels
// (one pixel of every block per lane), from the candidate loop to the stores -- no resolve step:
//   scan   : every thread takes runs of the bin's column of the chunk x bin directory (one cell per chunk, and the
//            chunk's run of the "big" pseudo-bin) and appends the faces whose box touches the tile to an LDS list, with a
//            16-bit mask of the blocks the box touches: two dependent memory round trips;
//   stage  : 64 candidates at a time, one lane per candidate: the set-up record is read once; its tile-local float32
//            form (TileRec) goes to LDS, and so does the record itself for the first SHADE_CAP candidates (the shading
//            pass reads the winners' records from LDS, not from memory);
//   raster : every wave visits the candidates whose mask touches one of its blocks: float32 edge functions classify each sample as certainly
//            inside / outside, the few in between take the specification's f64 test; then depth from the f64 plane and
//            a (z24, face) lexicographic min held in registers -- no atomics, independent of list order;
//   shade  : every lane evaluates the barycentrics of its pixels' winners in f64 (csrc/shaders.cpp:52-57,74) from the
//            records in LDS, exports the state, interpolates the colours (requested per candidate at staging, left in
//            LDS after the candidate loop) and writes the HWC pixels (background copied where uncovered).
template <int MODE, int NB, int CSPEC>
__global__ __launch_bounds__(RTHREADS, 4) void raster_kernel(RasterParams p)
{
    constexpr int TILE = 16 * NB;                      // pixels
    constexpr int BT = 2 * NB;                         // blocks per tile side: block (bx, by) = mask bit BT * by + bx
    constexpr int PPL = NB * NB;                       // pixels per lane
    constexpr bool LDS_COLORS = MODE == 0 && CSPEC != 0;
    __shared__ int32_t s_face[LIST_CAP];
    __shared__ uint16_t s_mask[LIST_CAP];  // bit (BT*by + bx): the face's box touches block (bx, by) of the tile
    __shared__ uint32_t s_count;
    __shared__ TileRec s_rec[64];          // tile-local records of the 64 list entries being rasterised
    // what the shading pass needs of the set-up records of the first listed candidates: 96 of a record's 128 bytes, at a
    // stride of 104 (26 dwords) -- lanes of a wave read the SAME field of DIFFERENT candidates' records, which at a stride of
    // 128 bytes is the same two banks for every candidate (round 5: SHADE_STRIDE; the records took 12.3 KB, now 9.75)
    __shared__ __align__(8) ShadeRec s_shade[SHADE_CAP];
    __shared__ float4 s_col[LDS_COLORS ? SHADE_CAP : 1][3];  // ... and their vertex colours (channel-specialised kernels)
    // any channel count that is a multiple of 4, up to 16: the colours of the first QCAP listed candidates, [candidate][vertex][quad]
    constexpr bool LDS_QUADS = MODE == 0 && CSPEC == 0;
    constexpr int QCAP = 48;
    __shared__ float4 s_colq[LDS_QUADS ? QCAP * 3 * 4 : 1];

#ifdef DIRT_TRACE
    long long tr_t[8]; int tr_n = 0;
    const long long tr_wall0 = wall_clock64();
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
    int tile_col, tile_row;
    tile_xy(tile, p.tiles_x, p.tiles_x_magic, tile_col, tile_row);
    const int tx0 = tile_col * TILE, tr0 = tile_row * TILE;
    const int tx1 = tx0 + TILE - 1, tr1 = tr0 + TILE - 1;
    const int C = CSPEC ? CSPEC : p.C;

    const FaceRec* __restrict__ recs = p.recs + (size_t)ib * p.F;
    const int bin = (tr0 >> p.grid.shift) * p.grid.bins_x + (tx0 >> p.grid.shift);
    // rows `bin` and "big" of the bin x chunk directory: the 2 * nchunk runs (chunk, bin) then (chunk, big)
    const BinCell* __restrict__ cells = p.cells + (size_t)ib * p.nchunk * (size_t)(p.masked ? p.grid.cell_chunk_stride : p.grid.big + 1);
    const BinEntry* __restrict__ scene_entries = p.entries + (size_t)ib * p.nchunk * (5 * (size_t)p.chunk_faces);
    const int nruns = 2 * p.nchunk;  // <= 512: two per thread
    // (masked directory -- chunks of 64 faces, setup_kernel_masked --: a cell is the mask of the chunk's faces in the bin, and
    // face l of the chunk has entry l of the chunk's segment; otherwise start / count of a run inside the segment)
    const bool masked = p.masked != 0;
    uint32_t run_base[2], run_count[2], run_pos[2];
    unsigned long long run_mask[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = tid + k * RTHREADS;
        run_base[k] = 0; run_count[k] = 0; run_pos[k] = 0; run_mask[k] = 0ull;
        if (j < nruns) {
            const int c = j < p.nchunk ? j : j - p.nchunk;
            const BinCell cell = cells[(size_t)(j < p.nchunk ? bin : p.grid.big) * p.grid.cell_bin_stride + (size_t)c * p.grid.cell_chunk_stride];
            run_base[k] = (uint32_t)c * (5u * (uint32_t)p.chunk_faces) + (masked ? 0u : cell.start);
            run_count[k] = masked ? 0u : cell.count;
            run_mask[k] = masked ? ((unsigned long long)cell.count << 32) | (unsigned long long)cell.start : 0ull;
        }
    }

    // this wave's region (blocks NB*wx .. NB*wx + NB-1, NB*wy .. of the tile) and this lane's NB x NB pixels
    const int wx = wave & 1, wy = wave >> 1;
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

    unsigned long long best[PPL];   // (z24 << 32 | face) of the front-most fragment so far
    int cbest[PPL];   // the winner's position in the list (its record is in LDS when < SHADE_CAP and lds_records)
#pragma unroll
    for (int k = 0; k < PPL; ++k) { best[k] = (unsigned long long)Z24_CLEAR << 32; cbest[k] = SHADE_CAP; }  // a tie with the cleared depth never wins
    bool lds_records = true;   // false once a second round has reused the list (dense meshes): records come from memory then
    int n_first = 0;           // candidates listed in the first round

    // side job: this workgroup's share of the buffers the launch clears (the backward pass's gradient accumulators)
    if ((p.zero_b_bytes | p.zero_c_bytes) != 0) {
        const unsigned gwg = blockIdx.y * gridDim.x + blockIdx.x;
        if (p.zero_b_bytes) zero_share(p.zero_b, p.zero_b_bytes, p.zero_b_per, gwg, tid);
        if (p.zero_c_bytes) zero_share(p.zero_c, p.zero_c_bytes, p.zero_c_per, gwg, tid);
    }
    TRACE_MARK();  // 1: directory requested
    for (int round = 0;; ++round) {
        if (tid == 0) s_count = 0;
        __syncthreads();
        TRACE_MARK();  // 2: cleared, barrier
        // ---- scan: this thread's runs, four entries per trip.  The four box tests of a trip are evaluated together and the
        //      trip's hits of the whole wave claim their list slots with ONE wave-aggregated LDS atomic (rounds 1-3: one per
        //      entry, i.e. four dependent LDS round trips per trip behind nested divergent branches) ----
        bool full = false;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            while ((masked ? run_mask[k] != 0ull : run_pos[k] < run_count[k]) && !full) {
                BinEntry en[4];
                uint32_t left, at[4];   // entries left in the run; where the next four are
                if (masked) {
                    unsigned long long t = run_mask[k];
                    left = (uint32_t)__popcll(t);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        at[i] = t != 0ull ? (uint32_t)(__ffsll((long long)t) - 1) : (i ? at[i - 1] : 0u);   // (past the last: the last again; not a hit)
                        t &= t - 1ull;
                    }
                } else {
                    left = run_count[k] - run_pos[k];
#pragma unroll
                    for (int i = 0; i < 4; ++i) at[i] = run_pos[k] + min((uint32_t)i, left - 1u);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) en[i] = scene_entries[run_base[k] + at[i]];
                bool hit[4];
                unsigned long long hm[4];
                uint32_t before = 0, off[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const FaceBox box = en[i].box;
                    hit[i] = ((uint32_t)i < left) & (box.i_min <= tx1) & (box.i_max >= tx0) & (box.r_min <= tr1) & (box.r_max >= tr0);
                    hm[i] = __builtin_amdgcn_ballot_w64(hit[i]);
                    off[i] = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm[i], 0u));
                    before += (uint32_t)__popcll(hm[i]);
                }
                uint32_t base = 0;
                if (before != 0u) {   // (wave-uniform among the lanes of this trip)
                    const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
                    if ((int)__builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0u)) == 0)
                        base = atomicAdd(&s_count, before);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                }
                uint32_t consumed = min(4u, left);
#pragma unroll
                for (int i = 3; i >= 0; --i) {
                    const uint32_t slot = base + off[i];
                    if (hit[i] && slot >= (uint32_t)LIST_CAP) { consumed = (uint32_t)i; full = true; }   // not consumed: next round
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t slot = base + off[i];
                    if (hit[i] && slot < (uint32_t)LIST_CAP) {
                        const FaceBox box = en[i].box;
                        const int bx0 = max(box.i_min - tx0, 0) >> 3, bx1 = min(box.i_max - tx0, TILE - 1) >> 3;
                        const int by0 = max(box.r_min - tr0, 0) >> 3, by1 = min(box.r_max - tr0, TILE - 1) >> 3;
                        const uint32_t rowbits = ((2u << bx1) - (1u << bx0)) & ((1u << BT) - 1u);
                        // rows by0 .. by1 of the mask get `rowbits`: one multiplication by the rows' unit bits (no loop, no
                        // carries: rowbits < 2^BT)
                        constexpr uint32_t UNIT = BT == 4 ? 0x1111u : 0x5u;   // bit BT * by of every block row
                        const uint32_t rows = UNIT & ((2u << (BT * by1 + BT - 1)) - (1u << (BT * by0)));
                        s_face[slot] = en[i].face;
                        s_mask[slot] = (uint16_t)(rowbits * rows);
                    }
                }
                if (masked) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if ((uint32_t)i < consumed) run_mask[k] &= run_mask[k] - 1ull;   // strike the consumed faces off
                } else {
                    run_pos[k] += consumed;
                }
            }
        }
        TRACE_MARK();  // 3: appended
        __syncthreads();
        const int n = (int)min(s_count, (uint32_t)LIST_CAP);
        if (round != 0) lds_records = false;
        if (round == 0) n_first = n;
        TRACE_MARK();  // 4: list built

        // ---- candidates, 64 at a time: one lane per candidate reads its set-up record once and leaves the tile-local
        //      form (and, for the shading pass, the record itself) in LDS; then every wave walks the ones that touch its
        //      blocks ----
        for (int cb = 0; cb < n; cb += 64) {
            const int m_chunk = min(64, n - cb);
            TRACE_ACC(0);
            // vertex colours of this lane's candidate: requested now, stored after the candidate loop
            float4 colv0 = make_float4(0.f, 0.f, 0.f, 0.f), colv1 = colv0, colv2 = colv0;
            bool stage_colors = false;
            if (tid < m_chunk) {
                const int face = s_face[cb + tid];
                const FaceRec rec = recs[face];
                make_tile_rec(rec, face, (double)tx0 + 0.5, (double)(p.H - 1 - tr0) + 0.5, wf, hf, &s_rec[tid]);
                if (round == 0 && cb + tid < SHADE_CAP) {
                    {
                        ShadeRec sr;
#pragma unroll
                        for (int i = 0; i < 9; ++i) sr.coef[i] = rec.coef[i];
                        sr.inv_det = rec.inv_det; sr.flags = rec.flags; sr.vid[0] = rec.vid[0]; sr.vid[1] = rec.vid[1]; sr.vid[2] = rec.vid[2]; sr.pad = 0.0;
                        s_shade[cb + tid] = sr;
                    }
                    if (LDS_COLORS) {
                        stage_colors = true;
                        const float* __restrict__ cols = p.vertex_colors + (size_t)ib * p.V * C;
                        auto fetch = [&](int vid) {
                            const float* __restrict__ cp = cols + (size_t)vid * C;
                            if (CSPEC == 4) return *reinterpret_cast<const float4*>(cp);
                            if (CSPEC == 3) return make_float4(cp[0], cp[1], cp[2], 0.f);
                            return make_float4(cp[0], 0.f, 0.f, 0.f);
                        };
                        colv0 = fetch(rec.vid[0]); colv1 = fetch(rec.vid[1]); colv2 = fetch(rec.vid[2]);
                    }
                }
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
#ifndef DIRT_NO_BLOCK_CULL
                // ... minus the blocks the triangle itself misses (tile-local record: offsets from the tile's top-left sample).
                // (Not in the 32 x 32 shapes specialised for 1 / 3 / 4 channels -- meshes of more than 16 384 faces; smaller ones
                // take raster_kernel_v2 --: they sit at 128 registers and the cull's temporaries would spill 16-32 bytes.)
                if constexpr (!(MODE == 0 && NB == 2 && CSPEC != 0))
                    if (mym4) mym4 = cull_blocks<NB>(s_rec[lane], mym4, (float)(8 * NB * wx), -(float)(8 * NB * wy));
#endif
            }
            unsigned long long m = __builtin_amdgcn_ballot_w64(mym4 != 0);
            if (m) {
                // (no software prefetch of the next candidate's record: holding two records costs 20 registers in a kernel
                // that sits at the limit for four workgroups per CU, and the other waves of the SIMD cover the LDS latency:
                // K3 raster 24.3 -> 21.7 us without it)
#ifndef DIRT_RASTER_NO_PAIRS
                if constexpr (NB == 1) {   // small frames: two candidates per trip (raster_candidate_pair)
                    while (m & (m - 1)) {
                        const int k0 = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const int k1 = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const TileRec t0 = s_rec[k0], t1 = s_rec[k1];
                        raster_candidate_pair(t0, t1, recs, round == 0 ? cb + k0 : SHADE_CAP, round == 0 ? cb + k1 : SHADE_CAP, dxl[0], dyl[0], px[0], py[0], best[0], cbest[0]);
                        TRACE_CNT();
                    }
                }
#endif
                while (m) {
                    const int k = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const TileRec t = s_rec[k];
                    const uint32_t m4 = (uint32_t)__builtin_amdgcn_readlane((int)mym4, k);
                    raster_candidate<NB>(t, recs, round == 0 ? cb + k : SHADE_CAP, m4, dxl, dyl, px, py, best, cbest);
                    TRACE_CNT();
                }
            }
            TRACE_ACC(2);
            if (LDS_COLORS && stage_colors) { s_col[cb + tid][0] = colv0; s_col[cb + tid][1] = colv1; s_col[cb + tid][2] = colv2; }
            __syncthreads();
            TRACE_ACC(3);
        }
        // another round? (a thread whose append found the list full still holds entries)
        const bool more = masked ? ((run_mask[0] | run_mask[1]) != 0ull) : ((run_pos[0] < run_count[0]) | (run_pos[1] < run_count[1]));
        if (!__syncthreads_or(more)) break;
    }

    TRACE_MARK();  // 5: candidates done (one round)
    int32_t fbest[PPL];   // the front-most face per pixel, -1: none (nothing was less than the cleared depth)
#pragma unroll
    for (int k = 0; k < PPL; ++k) fbest[k] = (uint32_t)(best[k] >> 32) != Z24_CLEAR ? (int32_t)(uint32_t)best[k] : -1;

    // Many-channel images (C = 8, 12, 16): the shading pass would gather 3 x C floats per PIXEL from memory (K5: 0.8 GB per
    // frame through the L2); instead the workgroup copies the colours of its first QCAP candidates' vertices into LDS
    // once, every thread a few 16-byte pieces, and the pixels read them there.
    bool quads_in_lds = false;
    if (LDS_QUADS) {
        const int Cq = CSPEC ? CSPEC : p.C;
        quads_in_lds = lds_records && (Cq & 3) == 0 && Cq <= 16 && Cq > 4;
        if (quads_in_lds) {
            const int nq = Cq >> 2, ncand = min(min(n_first, QCAP), SHADE_CAP);
            const float* __restrict__ colsq = p.vertex_colors + (size_t)ib * p.V * Cq;
            for (int it = tid; it < ncand * 3 * nq; it += RTHREADS) {
                const int cand = it / (3 * nq), rem = it - cand * 3 * nq, kv = rem / nq, q = rem - kv * nq;
                const int vid = s_shade[cand].vid[kv];
                s_colq[(cand * 3 + kv) * 4 + q] = *reinterpret_cast<const float4*>(colsq + (size_t)vid * Cq + 4 * q);
            }
            __syncthreads();
        }
    }

    // ---- shade ----
    // this lane's pixels: background where nothing is visible (requested now, used last)
    bool inside[PPL];
    size_t pix[PPL];
    float4 bgv[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int x = x0 + 8 * (k % NB), r = r0 + 8 * (k / NB);
        inside[k] = x < p.W && r < p.H;
        pix[k] = ((size_t)ib * p.H + min(r, p.H - 1)) * p.W + min(x, p.W - 1);
        bgv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0 && CSPEC != 0 && inside[k] && fbest[k] < 0) {
            const float* __restrict__ bg = p.background + pix[k] * C;
            if (CSPEC == 4) bgv[k] = *reinterpret_cast<const float4*>(bg);
            else if (CSPEC == 3) bgv[k] = make_float4(bg[0], bg[1], bg[2], 0.f);
            else bgv[k] = make_float4(bg[0], 0.f, 0.f, 0.f);
        }
    }
    // Per pixel: barycentrics of the winner (csrc/shaders.cpp:52-57,74) from its record in LDS -- or, for candidates beyond
    // SHADE_CAP / later rounds, in memory --, the backward pass's state, the interpolated colours, the HWC pixel.  The LDS
    // reads are written as such (indexing s_shade / s_col, not through a pointer that may also point to memory: that
    // would be flat loads, each waiting for LDS AND memory, i.e. for the stores of the pixel before); the rare records in
    // memory are fetched behind a wave-uniform branch.
    const float* __restrict__ cols = p.vertex_colors + (size_t)ib * p.V * C;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int32_t f = fbest[k];
        const bool has = f >= 0;
        const bool from_lds = has && lds_records && cbest[k] < SHADE_CAP;
        const int ci = from_lds ? cbest[k] : 0;   // (lanes without a winner read slot 0; what they compute is not used)
        double cf[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) cf[i] = s_shade[ci].coef[i];
        uint32_t flags = s_shade[ci].flags;
        double inv_det = s_shade[ci].inv_det;
        int32_t vid0 = 0, vid1 = 0, vid2 = 0;
        if (MODE == 0 && !LDS_COLORS) { vid0 = s_shade[ci].vid[0]; vid1 = s_shade[ci].vid[1]; vid2 = s_shade[ci].vid[2]; }
        float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0, u2 = u0;
        if (LDS_COLORS) { u0 = s_col[ci][0]; u1 = s_col[ci][1]; u2 = s_col[ci][2]; }
        if (__builtin_amdgcn_ballot_w64(has && !from_lds) != 0ull) {
            if (has && !from_lds) {
                const FaceRec* __restrict__ rec = recs + f;
#pragma unroll
                for (int i = 0; i < 9; ++i) cf[i] = rec->coef[i];
                flags = rec->flags; inv_det = rec->inv_det;
                vid0 = rec->vid[0]; vid1 = rec->vid[1]; vid2 = rec->vid[2];
                if (LDS_COLORS) {
                    const float* __restrict__ c0 = cols + (size_t)vid0 * C;
                    const float* __restrict__ c1 = cols + (size_t)vid1 * C;
                    const float* __restrict__ c2 = cols + (size_t)vid2 * C;
                    if (CSPEC == 4) { u0 = *reinterpret_cast<const float4*>(c0); u1 = *reinterpret_cast<const float4*>(c1); u2 = *reinterpret_cast<const float4*>(c2); }
                    else if (CSPEC == 3) { u0 = make_float4(c0[0], c0[1], c0[2], 0.f); u1 = make_float4(c1[0], c1[1], c1[2], 0.f); u2 = make_float4(c2[0], c2[1], c2[2], 0.f); }
                    else { u0 = make_float4(c0[0], 0.f, 0.f, 0.f); u1 = make_float4(c1[0], 0.f, 0.f, 0.f); u2 = make_float4(c2[0], 0.f, 0.f, 0.f); }
                }
            }
        }
        double Fk[3];
        edge_eval(cf, (double)(x0 + 8 * (k % NB)) + 0.5, (double)(p.H - 1 - (r0 + 8 * (k / NB))) + 0.5, Fk);
        float b[3], cw;
        bary_eval(Fk, flags, inv_det, b, cw);
        const float b0 = b[0], b1 = b[1], b2 = b[2];
        if (!inside[k]) continue;
        // the backward pass's state and the visibility export
        if (p.vis) p.vis[pix[k]] = f;
        if (p.state_a) store_state(p, pix[k], has, b0, b1, b2, cw, f);
        if (MODE != 0) continue;
        float* __restrict__ out = p.pixels + pix[k] * C;
        if (CSPEC != 0) {
            float4 o = bgv[k];   // pixels start as the background: csrc/rasterise_egl.cpp:348-356
            if (has) {
                o.x = fmaf(b2, u2.x, fmaf(b1, u1.x, b0 * u0.x));
                if (CSPEC >= 3) { o.y = fmaf(b2, u2.y, fmaf(b1, u1.y, b0 * u0.y)); o.z = fmaf(b2, u2.z, fmaf(b1, u1.z, b0 * u0.z)); }
                if (CSPEC == 4) o.w = fmaf(b2, u2.w, fmaf(b1, u1.w, b0 * u0.w));
            }
            if (CSPEC == 4) *reinterpret_cast<float4*>(out) = o;
            else if (CSPEC == 3) { out[0] = o.x; out[1] = o.y; out[2] = o.z; }
            else out[0] = o.x;
        } else if (!has) {
            const float* __restrict__ bg = p.background + pix[k] * C;
            if ((C & 3) == 0) {
                for (int c = 0; c < C; c += 4) *reinterpret_cast<float4*>(out + c) = *reinterpret_cast<const float4*>(bg + c);
            } else {
                for (int c = 0; c < C; ++c) out[c] = bg[c];
            }
        } else if (LDS_QUADS && quads_in_lds && from_lds && cbest[k] < QCAP) {
            const float4* __restrict__ w = &s_colq[cbest[k] * 12];
            for (int q = 0; q < (C >> 2); ++q) {
                const float4 w0 = w[q], w1 = w[4 + q], w2 = w[8 + q];
                float4 o;
                o.x = fmaf(b2, w2.x, fmaf(b1, w1.x, b0 * w0.x));
                o.y = fmaf(b2, w2.y, fmaf(b1, w1.y, b0 * w0.y));
                o.z = fmaf(b2, w2.z, fmaf(b1, w1.z, b0 * w0.z));
                o.w = fmaf(b2, w2.w, fmaf(b1, w1.w, b0 * w0.w));
                *reinterpret_cast<float4*>(out + 4 * q) = o;
            }
        } else {
            const float* __restrict__ c0 = cols + (size_t)vid0 * C;
            const float* __restrict__ c1 = cols + (size_t)vid1 * C;
            const float* __restrict__ c2 = cols + (size_t)vid2 * C;
            if ((C & 3) == 0) {
                for (int c = 0; c < C; c += 4) {
                    const float4 w0 = *reinterpret_cast<const float4*>(c0 + c);
                    const float4 w1 = *reinterpret_cast<const float4*>(c1 + c);
                    const float4 w2 = *reinterpret_cast<const float4*>(c2 + c);
                    float4 o;
                    o.x = fmaf(b2, w2.x, fmaf(b1, w1.x, b0 * w0.x));
                    o.y = fmaf(b2, w2.y, fmaf(b1, w1.y, b0 * w0.y));
                    o.z = fmaf(b2, w2.z, fmaf(b1, w1.z, b0 * w0.z));
                    o.w = fmaf(b2, w2.w, fmaf(b1, w1.w, b0 * w0.w));
                    *reinterpret_cast<float4*>(out + c) = o;
                }
            } else {
                for (int c = 0; c < C; ++c) out[c] = fmaf(b2, c2[c], fmaf(b1, c1[c], b0 * c0[c]));
            }
        }
    }
    TRACE_MARK();  // 6: stored
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_buf) {
        long long* o = g_trace_buf + ((size_t)blockIdx.x * 4 + wave) * 16;
        for (int i = 0; i < 8; ++i) o[i] = i < tr_n ? tr_t[i] : 0;
        o[8] = tr_acc[1]; o[9] = tr_acc[2]; o[10] = tr_acc[3]; o[11] = tr_cnt;
        o[12] = tr_wall0; o[13] = (long long)wall_clock64() - tr_wall0; o[14] = blockIdx.x;
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

// The state's interleaved gradient accumulators (one row {x, y, z, w, c_0 ..} of acc_stride floats per vertex) copied out
// into dense [rows, 4] / [rows, C] tensors: one thread per vertex (DIRT_FLAG_DENSE_FROM_STATE).
__global__ __launch_bounds__(256) void unpack_kernel(const float* __restrict__ acc_gv, const float* __restrict__ acc_gvc, int acc_stride,
                                                     float* __restrict__ gv, float* __restrict__ gvc, int C, size_t rows)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    *reinterpret_cast<float4*>(gv + 4 * r) = *reinterpret_cast<const float4*>(acc_gv + r * (size_t)acc_stride);
    const float* __restrict__ a = acc_gvc + r * (size_t)acc_stride;
    float* __restrict__ o = gvc + r * (size_t)C;
    for (int c = 0; c < C; ++c) o[c] = a[c];
}

hipError_t launch_unpack(const float* acc_gv, const float* acc_gvc, int acc_stride, float* gv, float* gvc, int C, size_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, acc_gv, acc_gvc, acc_stride, gv, gvc, C, rows);
    return hipGetLastError();
}

hipError_t launch_geometry(const GeomParams& g, hipStream_t stream)
{
    if (g.B == 0) return hipSuccess;
    // also with F == 0: the (all-zero) directory row is what the raster kernel reads
    const dim3 grid((unsigned)g.nchunk, (unsigned)g.B);
    if (g.masked) return launch_geometry_v2(g, stream);   // dirt_forward.hip
    if (g.chunk_faces > STHREADS) hipLaunchKernelGGL(setup_kernel<4>, grid, dim3(4 * STHREADS), 0, stream, g);
    else hipLaunchKernelGGL(setup_kernel<1>, grid, dim3(STHREADS), 0, stream, g);
    return hipGetLastError();
}

void chunking(int F, int& nchunk, int& chunk_faces)
{
    // <= 256 chunks of >= 64 faces (one wave each): a raster tile reads one directory cell per chunk (2 * nchunk <= 512 runs)
    chunk_faces = 64;
    if ((long long)chunk_faces * 256 < F) chunk_faces = (F + 255) / 256;
    nchunk = F > 0 ? (F + chunk_faces - 1) / chunk_faces : 1;
}

BinGrid make_bin_grid(int H, int W, int nchunk, bool masked, int min_shift)
{
    BinGrid g;
    const int max_bins = masked ? MAX_BINS_MASKED : MAX_BINS;
    g.big = max_bins;
    g.cell_bin_stride = masked ? 1 : nchunk;
    g.cell_chunk_stride = masked ? ((max_bins + 2) & ~1) : 1;   // (even: a chunk's row starts at a 16-byte boundary)
    g.shift = masked ? min_shift : 5;  // bins of >= 32 pixels (16 for launches on 16 x 16 tiles: a bin's faces are then a tile's candidates there too): a raster tile never straddles two bins
    while (((W + (1 << g.shift) - 1) >> g.shift) * ((H + (1 << g.shift) - 1) >> g.shift) > max_bins) ++g.shift;
    g.bins_x = (W + (1 << g.shift) - 1) >> g.shift;
    g.bins_y = (H + (1 << g.shift) - 1) >> g.shift;
    return g;
}

hipError_t launch_raster(const RasterParams& p_in, int B, bool visibility_only, hipStream_t stream)
{
    if (B == 0) return hipSuccess;
    if (raster_v2_applies(p_in, B, visibility_only)) return launch_raster_v2(p_in, B, visibility_only, stream);
    RasterParams p = p_in;
    const int tile = raster_tile_choice(p.H, p.W, B, p.flags);   // (the bin grid was sized for this very choice: dirt_capi.hip::geom_params)
    p.tiles_x = (p.W + tile - 1) / tile;
    p.tiles_y = (p.H + tile - 1) / tile;
    p.tiles_x_magic = tile_magic(p.tiles_x);
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)B);
    {
        const size_t nwg = (size_t)grid.x * grid.y;
        p.zero_b_per = (unsigned)((p.zero_b_bytes / 16 + nwg - 1) / nwg);
        p.zero_c_per = (unsigned)((p.zero_c_bytes / 16 + nwg - 1) / nwg);
    }
    const int cspec = visibility_only ? 0 : (p.C == 4 ? 4 : (p.C == 3 ? 3 : (p.C == 1 ? 1 : 0)));
#define DIRT_LAUNCH_RASTER(NB_)                                                                               \
    do {                                                                                                      \
        if (visibility_only) hipLaunchKernelGGL((raster_kernel<1, NB_, 0>), grid, dim3(RTHREADS), 0, stream, p);   \
        else if (cspec == 4) hipLaunchKernelGGL((raster_kernel<0, NB_, 4>), grid, dim3(RTHREADS), 0, stream, p);   \
        else if (cspec == 3) hipLaunchKernelGGL((raster_kernel<0, NB_, 3>), grid, dim3(RTHREADS), 0, stream, p);   \
        else if (cspec == 1) hipLaunchKernelGGL((raster_kernel<0, NB_, 1>), grid, dim3(RTHREADS), 0, stream, p);   \
        else hipLaunchKernelGGL((raster_kernel<0, NB_, 0>), grid, dim3(RTHREADS), 0, stream, p);                   \
    } while (0)
    if (tile == 32) DIRT_LAUNCH_RASTER(2);
    else DIRT_LAUNCH_RASTER(1);
#undef DIRT_LAUNCH_RASTER
    return hipGetLastError();
}

}  // namespace dirt
