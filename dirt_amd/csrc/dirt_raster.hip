// dirt_raster.hip -- triangle set-up and the tiled visibility / shading kernels for gfx950.
//
// Replaces, for the whole batch in one launch each:
//   * the GL vertex pipeline + upload_vertices      (csrc/rasterise_grad_egl.cu:12-34)
//   * B x { glViewport, glScissor, glClear(DEPTH), glDrawElementsBaseVertex }
//                                                   (csrc/rasterise_egl.cpp:362-380)
//   * upload_background / download_pixels           (csrc/rasterise_egl.cu:10-38,65-91): there is no
//     RGBA32F atlas; tiles read `background` and write `pixels` in place, top row first.
//
// Structure of raster_kernel (one 256-thread workgroup = one 32x32 pixel tile of one scene = 4x4
// blocks of 8x8 pixels; each of its 4 waves owns a 16x16 region = 2x2 blocks, 4 pixels per lane, so
// that one record fetch serves 256 pixels):
//   scan   : the threads stride over the face list of the tile's bin (plus the scene's big list)
//            and append the faces whose box touches the tile to an LDS list (wave-aggregated LDS
//            atomic), together with a 16-bit mask of the blocks the box touches;
//   raster : each wave tests 64 list entries at a time against its four block bits (ballot -> 64-bit
//            survivor mask); survivors are visited with a scalar bit-scan, their record fetched with
//            wave-uniform (scalar) loads so the nine f64 edge coefficients sit in SGPRs, the next
//            survivor's loads in flight while the current one is evaluated; each lane evaluates the
//            three edge functions at its pixel centre exactly as the specification writes them,
//            then depth, then a (z24, face) lexicographic min held in registers -- no LDS or global
//            atomics, and the result is independent of list order;
//   shade  : the winner's record is re-read per pixel, barycentrics and all C channels are
//            interpolated once, and the HWC pixel is written (background copied where uncovered).
#include "dirt_device.h"
#include "dirt_launch.h"

namespace dirt {

#ifdef DIRT_TRACE
__device__ long long* g_trace_buf = nullptr;
#endif

// ---- binning ---------------------------------------------------------------------------------
// The frame is cut into at most MAX_BINS square bins of 2^shift pixels (>= 128, so a raster tile
// never straddles two bins).  Two kernels build, per scene, an exact-size list of the faces touching
// each bin -- the replacement for the GL driver's own binning hardware -- without a single global
// atomic and without any buffer that needs clearing:
//   setup_kernel : the scene's faces are cut into `nchunk` contiguous chunks, one 256-thread
//                  workgroup each.  Per face: set-up record, bounding box, and an LDS count in every
//                  bin the box touches (faces touching more than 4 bins are counted for the scene's
//                  "big" list, which every tile reads, so the bin lists hold <= 4F entries in total).
//                  The chunk's histogram row is stored to chunk_count[chunk][bin].
//   fill_kernel  : same chunks.  Column sums of the count matrix give each bin's size, an exclusive
//                  prefix over bins gives its segment, the partial column sum over earlier chunks
//                  gives this chunk's offset inside the segment; faces then claim slots with LDS
//                  cursors.  Chunk 0 publishes count / start / big_count for the raster kernel.
// List order inside a chunk is whatever the LDS atomics produce; visibility does not depend on it.

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
    __shared__ uint32_t s_cnt[MAX_BINS + 1];  // [MAX_BINS] = big faces
    const int ib = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    // side job: clear the gradient accumulators of the backward pass (the cudaMemsetAsync x4 of
    // csrc/rasterise_grad_egl.cu:244-250) so that no separate launch is needed for it
    {
        const size_t nthreads = (size_t)gridDim.x * gridDim.y * 256;
        const size_t gtid = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid;
        uint32_t* zb = reinterpret_cast<uint32_t*>(g.zero_b);
        uint32_t* zc = reinterpret_cast<uint32_t*>(g.zero_c);
        for (size_t i = gtid; i < g.zero_b_bytes / 4; i += nthreads) zb[i] = 0u;
        for (size_t i = gtid; i < g.zero_c_bytes / 4; i += nthreads) zc[i] = 0u;
    }
    s_cnt[tid] = 0;
    if (tid == 0) s_cnt[MAX_BINS] = 0;
    __syncthreads();
    const int f0 = chunk * g.chunk_faces, f1 = min(g.F, f0 + g.chunk_faces);
    const float* __restrict__ verts = g.vertices + (size_t)ib * g.V * 4;
    for (int f = f0 + tid; f < f1; f += 256) {
        const size_t n = (size_t)ib * g.F + f;
        FaceRec rec;
        FaceBox box;
        if (setup_face(verts, g.V, g.faces + n * 3, g.H, g.W, rec, box)) {
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
        g.boxes[n] = box;
    }
    __syncthreads();
    uint32_t* __restrict__ row = g.chunk_count + ((size_t)ib * g.nchunk + chunk) * (MAX_BINS + 1);
    row[tid] = s_cnt[tid];
    if (tid == 0) row[MAX_BINS] = s_cnt[MAX_BINS];
}

__global__ __launch_bounds__(256) void fill_kernel(GeomParams g)
{
    __shared__ uint32_t s_base[MAX_BINS + 1];  // first slot of this chunk in each bin's segment ([MAX_BINS]: big list)
    __shared__ uint32_t s_cur[MAX_BINS + 1];
    __shared__ uint32_t s_wave[4];
    const int ib = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t* __restrict__ mat = g.chunk_count + (size_t)ib * g.nchunk * (MAX_BINS + 1);

    // column sums over all chunks (bin sizes) and over the chunks before this one
    uint32_t total = 0, before = 0;
#pragma unroll 8
    for (int c = 0; c < g.nchunk; ++c) {
        const uint32_t v = mat[(size_t)c * (MAX_BINS + 1) + tid];
        total += v;
        before += (c < chunk) ? v : 0u;
    }
    uint32_t incl = total;  // exclusive prefix of the bin sizes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    // the big-list column, summed by wave 3 (its lanes stride over the chunks)
    uint32_t btotal = 0, bbefore = 0;
    if (wave == 3) {
        for (int c = lane; c < g.nchunk; c += 64) {
            const uint32_t v = mat[(size_t)c * (MAX_BINS + 1) + MAX_BINS];
            btotal += v;
            bbefore += (c < chunk) ? v : 0u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            btotal += __shfl_xor(btotal, d);
            bbefore += __shfl_xor(bbefore, d);
        }
    }
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    const uint32_t start = off + incl - total;
    s_base[tid] = start + before;
    s_cur[tid] = 0;
    BinCounters* __restrict__ ctr = g.ctrs + ib;
    if (chunk == 0) { ctr->count[tid] = total; ctr->start[tid] = start; }
    if (tid == 192) {  // lane 0 of wave 3
        s_base[MAX_BINS] = bbefore;
        s_cur[MAX_BINS] = 0;
        if (chunk == 0) ctr->big_count = btotal;
    }
    __syncthreads();

    const int f0 = chunk * g.chunk_faces, f1 = min(g.F, f0 + g.chunk_faces);
    BinEntry* __restrict__ out = g.entries + (size_t)ib * 4 * g.F;
    BinEntry* __restrict__ big = g.big + (size_t)ib * g.F;
    for (int f = f0 + tid; f < f1; f += 256) {
        BinEntry e;
        e.box = g.boxes[(size_t)ib * g.F + f];
        if (e.box.i_min > e.box.i_max) continue;  // culled at set-up
        e.face = f; e.pad = 0;
        int bx0, bx1, by0, by1;
        if (bin_range(e.box, g.grid, bx0, bx1, by0, by1)) {
            for (int by = by0; by <= by1; ++by)
                for (int bx = bx0; bx <= bx1; ++bx) {
                    const int b = by * g.grid.bins_x + bx;
                    out[s_base[b] + atomicAdd(&s_cur[b], 1u)] = e;
                }
        } else {
            big[s_base[MAX_BINS] + atomicAdd(&s_cur[MAX_BINS], 1u)] = e;
        }
    }
}

constexpr int TILE_W = 32;        // tile = 32 x 32 pixels = 4 x 4 blocks
constexpr int TILE_H = 32;
constexpr int RTHREADS = 256;     // 4 waves; each owns a 16 x 16 region = 2 x 2 blocks, 4 pixels per lane
constexpr int LIST_CAP = 2048;    // bin entries scanned (and at most listed) per round

// The part of a FaceRec the coverage / depth loop needs (its first 104 bytes).  Loaded through a
// wave-uniform address, so it is fetched with scalar loads and lives in SGPRs.
struct RecCore {
    double coef[9];
    double zp[3];  // depth plane, scaled to the 24-bit range
    uint32_t flags;
    uint32_t pad;
};
static_assert(sizeof(RecCore) == 104, "RecCore is the head of FaceRec");

// Coverage + depth + visibility update of one block (one pixel per lane) for one candidate.
// F_k = fma(a_k, px, fma(b_k, py, c_k)) exactly as the specification writes it; the inner fma is
// shared by the two blocks of a block row (`trow`), as is that of the depth plane (`zrow`).
__device__ __forceinline__ void raster_block(const RecCore& rec, int face, double px, const double trow[3], double zrow,
                                             uint32_t& zbest, int32_t& fbest)
{
    const double F0 = fma(rec.coef[0], px, trow[0]);
    const double F1 = fma(rec.coef[3], px, trow[1]);
    const double F2 = fma(rec.coef[6], px, trow[2]);
    const bool c0 = (F0 >= 0.0) != ((rec.flags & 1u) != 0);
    const bool c1 = (F1 >= 0.0) != ((rec.flags & 2u) != 0);
    const bool c2 = (F2 >= 0.0) != ((rec.flags & 4u) != 0);
    if (c0 && c1 && c2) {
        const double q = fma(rec.zp[0], px, zrow);  // depth scaled to [0, 2^24-1]; kept iff inside (the depth clip)
        if (q >= 0.0 && q <= 16777215.0) {
            const uint32_t z24 = (uint32_t)rint(q);
            // GL_LESS against the stored depth; equal depth keeps the lower face index, which is
            // what drawing the faces in index order does (csrc/rasterise_egl.cpp:373-379).
            if (z24 < zbest || (z24 == zbest && face < fbest)) { zbest = z24; fbest = face; }
        }
    }
}

// One candidate against the wave's 2 x 2 blocks; `m4` (wave-uniform) says which blocks its box touches.
__device__ __forceinline__ void raster_candidate(const RecCore& rec, int face, uint32_t m4, const double px[2],
                                                 const double py[2], uint32_t zbest[4], int32_t fbest[4])
{
#pragma unroll
    for (int by = 0; by < 2; ++by) {
        if ((m4 >> (2 * by)) & 3u) {
            double trow[3];
            trow[0] = fma(rec.coef[1], py[by], rec.coef[2]);
            trow[1] = fma(rec.coef[4], py[by], rec.coef[5]);
            trow[2] = fma(rec.coef[7], py[by], rec.coef[8]);
            const double zrow = fma(rec.zp[1], py[by], rec.zp[2]);
#pragma unroll
            for (int bx = 0; bx < 2; ++bx)
                if ((m4 >> (2 * by + bx)) & 1u)
                    raster_block(rec, face, px[bx], trow, zrow, zbest[2 * by + bx], fbest[2 * by + bx]);
        }
    }
}

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8); give every XCD a
// contiguous run of tiles (a band of tile rows) so neighbouring tiles, which share faces, hit the
// same L2.  Bijective for any tile count.  Speed only; nothing depends on placement.
__device__ __forceinline__ int xcd_tile(int b, int ntiles)
{
    const int x = b & 7, j = b >> 3;
    const int q = ntiles >> 3, rem = ntiles & 7;
    return x * q + min(x, rem) + j;
}

// Shade one pixel: the winner's record is re-read, barycentrics and all C channels interpolated once.
__device__ __forceinline__ void shade_pixel(const RasterParams& p, const FaceRec* __restrict__ recs, int ib, int x, int r,
                                            double px, double py, int32_t f)
{
    const size_t pix = ((size_t)ib * p.H + r) * p.W + x;
    const int C = p.C;
    float* __restrict__ out = p.pixels + pix * C;
    if (f < 0) {  // pixels start as the background: csrc/rasterise_egl.cpp:348-356
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

template <int MODE>
__global__ __launch_bounds__(RTHREADS) void raster_kernel(RasterParams p)
{
    __shared__ int32_t s_face[LIST_CAP];
    __shared__ uint16_t s_mask[LIST_CAP];  // bit (4*by + bx): the face's box touches block (bx, by) of the tile
    __shared__ uint32_t s_count;
    __shared__ uint4 s_rec[64 * 8];       // the FaceRecs of the 64 list entries being rasterised

#ifdef DIRT_TRACE
    long long tr_t[8]; int tr_n = 0;
#define TRACE_MARK() do { if (tr_n < 8) tr_t[tr_n++] = clock64(); } while (0)
#else
#define TRACE_MARK() do {} while (0)
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
    const BinCounters* __restrict__ ctr = p.ctrs + ib;
    const int bin = (tr0 >> p.grid.shift) * p.grid.bins_x + (tx0 >> p.grid.shift);
    const int n_bin = (int)ctr->count[bin];
    const int n_all = n_bin + (int)ctr->big_count;
    const BinEntry* __restrict__ bin_entries = p.entries + (size_t)ib * 4 * p.F + ctr->start[bin];
    const BinEntry* __restrict__ big_entries = p.big + (size_t)ib * p.F;

    // this wave's 16 x 16 region (blocks 2wx..2wx+1, 2wy..2wy+1 of the tile) and this lane's 4 pixels
    const int wx = wave & 1, wy = wave >> 1;
    const int x0 = tx0 + wx * 16 + (lane & 7);
    const int r0 = tr0 + wy * 16 + (lane >> 3);
    const double px[2] = {(double)x0 + 0.5, (double)(x0 + 8) + 0.5};
    const double py[2] = {(double)(p.H - 1 - r0) + 0.5, (double)(p.H - 1 - (r0 + 8)) + 0.5};
    // the wave's four block bits inside a 16-bit tile mask, gathered into 4 bits (2*by + bx)
    const int sh0 = (2 * wy) * 4 + 2 * wx, sh1 = (2 * wy + 1) * 4 + 2 * wx;

    uint32_t zbest[4];
    int32_t fbest[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { zbest[k] = Z24_CLEAR; fbest[k] = -1; }  // -1: a tie with the cleared depth never wins

    TRACE_MARK();  // 1: directory loaded
    for (int round = 0; round < n_all; round += LIST_CAP) {
        if (tid == 0) s_count = 0;
        __syncthreads();
        TRACE_MARK();  // 2: first barrier
        const int round_end = min(n_all, round + LIST_CAP);
        for (int base = round; base < round_end; base += RTHREADS) {
            const int e = base + tid;
            bool hit = false;
            BinEntry en;
            if (e < round_end) {
                en = e < n_bin ? bin_entries[e] : big_entries[e - n_bin];
                hit = en.box.i_min <= tx1 && en.box.i_max >= tx0 && en.box.r_min <= tr1 && en.box.r_max >= tr0;
            }
            const unsigned long long m = __ballot(hit);
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
                    const uint32_t rowbits = ((2u << bx1) - (1u << bx0)) & 0xFu;
                    uint32_t mask = 0;
                    for (int by = by0; by <= by1; ++by) mask |= rowbits << (4 * by);
                    s_face[slot] = en.face;
                    s_mask[slot] = (uint16_t)mask;
                }
            }
        }
        TRACE_MARK();  // 4: appended
        __syncthreads();
        const int n = (int)s_count;
        TRACE_MARK();  // 5: list built

        // ---- candidates, 64 at a time: their records are staged in LDS by one parallel batch of
        //      coalesced 16-byte loads (a single memory latency for the whole tile instead of one
        //      per candidate), then every wave walks the ones that touch its blocks ----
        for (int cb = 0; cb < n; cb += 64) {
            const int m_chunk = min(64, n - cb);
            for (int i = tid; i < m_chunk * 8; i += RTHREADS) {
                const int rec = i >> 3, piece = i & 7;
                s_rec[rec * 8 + piece] = reinterpret_cast<const uint4*>(recs + s_face[cb + rec])[piece];
            }
            __syncthreads();
            const int idx = cb + lane;
            uint32_t mym4 = 0;
            if (idx < n) {
                const uint32_t mk = s_mask[idx];
                mym4 = ((mk >> sh0) & 3u) | (((mk >> sh1) & 3u) << 2);
            }
            unsigned long long m = __ballot(mym4 != 0);
            if (m) {
                // software pipeline over the survivors: the next record's LDS reads are in flight
                // while the current one is evaluated
                int k = __ffsll((long long)m) - 1;
                m &= m - 1;
                RecCore cur = *reinterpret_cast<const RecCore*>(&s_rec[k * 8]);
                while (true) {
                    const bool more = m != 0;
                    const int kc = k;
                    if (more) {
                        k = __ffsll((long long)m) - 1;
                        m &= m - 1;
                    }
                    const RecCore nxt = *reinterpret_cast<const RecCore*>(&s_rec[k * 8]);
                    const int face = s_face[cb + kc];
                    const uint32_t m4 = (uint32_t)__builtin_amdgcn_readlane((int)mym4, kc);
                    raster_candidate(cur, face, m4, px, py, zbest, fbest);
                    if (!more) break;
                    cur = nxt;
                }
            }
            __syncthreads();
        }
    }

    TRACE_MARK();  // 6: candidates done
    // ---- resolve: shade (MODE 0) or export the visibility buffer (MODE 1) ----
#pragma unroll
    for (int by = 0; by < 2; ++by)
#pragma unroll
        for (int bx = 0; bx < 2; ++bx) {
            const int x = x0 + 8 * bx, r = r0 + 8 * by;
            if (r >= p.H || x >= p.W) continue;
            const int32_t f = fbest[2 * by + bx];
            if (MODE == 1 || p.vis) p.vis[((size_t)ib * p.H + r) * p.W + x] = f;
            if (MODE == 0) shade_pixel(p, recs, ib, x, r, px[bx], py[by], f);
        }
    TRACE_MARK();  // 7: stored
#ifdef DIRT_TRACE
    if (lane == 0 && g_trace_buf) {
        long long* o = g_trace_buf + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = tr_t[i];
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
    // also with F == 0: fill publishes the (all-zero) directory the raster kernel reads
    const dim3 grid((unsigned)g.nchunk, (unsigned)g.B);
    hipLaunchKernelGGL(setup_kernel, grid, dim3(256), 0, stream, g);
    hipLaunchKernelGGL(fill_kernel, grid, dim3(256), 0, stream, g);
    return hipGetLastError();
}

void chunking(int F, int& nchunk, int& chunk_faces)
{
    // <= 256 chunks of >= 256 faces: the count matrix stays small enough for every fill workgroup
    // to read all of it
    chunk_faces = 256;
    if ((long long)chunk_faces * 256 < F) chunk_faces = (F + 255) / 256;
    nchunk = F > 0 ? (F + chunk_faces - 1) / chunk_faces : 1;
}

BinGrid make_bin_grid(int H, int W)
{
    BinGrid g;
    g.shift = 7;
    while (((W + (1 << g.shift) - 1) >> g.shift) * ((H + (1 << g.shift) - 1) >> g.shift) > MAX_BINS) ++g.shift;
    g.bins_x = (W + (1 << g.shift) - 1) >> g.shift;
    g.bins_y = (H + (1 << g.shift) - 1) >> g.shift;
    return g;
}

hipError_t launch_raster(const RasterParams& p_in, int B, bool visibility_only, hipStream_t stream)
{
    if (B == 0) return hipSuccess;
    RasterParams p = p_in;
    p.tiles_x = (p.W + TILE_W - 1) / TILE_W;
    p.tiles_y = (p.H + TILE_H - 1) / TILE_H;
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)B);
    if (visibility_only)
        hipLaunchKernelGGL(raster_kernel<1>, grid, dim3(RTHREADS), 0, stream, p);
    else
        hipLaunchKernelGGL(raster_kernel<0>, grid, dim3(RTHREADS), 0, stream, p);
    return hipGetLastError();
}

}  // namespace dirt
