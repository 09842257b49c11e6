// dirt_raster.hip -- triangle set-up and the tiled visibility / shading kernels for gfx950.
//
// Replaces, for the whole batch in one launch each:
//   * the GL vertex pipeline + upload_vertices      (csrc/rasterise_grad_egl.cu:12-34)
//   * B x { glViewport, glScissor, glClear(DEPTH), glDrawElementsBaseVertex }
//                                                   (csrc/rasterise_egl.cpp:362-380)
//   * upload_background / download_pixels           (csrc/rasterise_egl.cu:10-38,65-91): there is no
//     RGBA32F atlas; tiles read `background` and write `pixels` in place, top row first.
//
// Structure of raster_kernel (one 256-thread workgroup = one 32x32 pixel tile of one scene):
//   scan   : the four waves stride over the scene's FaceBox array (8 B/face, coalesced) and append
//            the faces whose box touches the tile to an LDS list (wave-aggregated LDS atomic);
//   raster : wave w owns the 8-row band w of the tile as four 8x8 blocks, one pixel per lane.
//            Per block, lanes first test 64 list entries at a time against the block rectangle
//            (ballot -> 64-bit survivor mask); survivors are visited with a scalar bit-scan, their
//            FaceRec fetched with wave-uniform (scalar) loads so the nine f64 edge coefficients sit
//            in SGPRs; each lane evaluates the three edge functions at its pixel centre exactly as
//            the specification writes them, then depth, then a (z24, face) lexicographic min held
//            in registers -- no LDS or global atomics, and the result is independent of list order;
//   shade  : the winner's record is re-read per pixel, barycentrics and all C channels are
//            interpolated once, and the HWC pixel is written (background copied where uncovered).
#include "dirt_device.h"
#include "dirt_launch.h"

namespace dirt {

__global__ __launch_bounds__(256) void setup_kernel(const float* __restrict__ vertices,
                                                    const int32_t* __restrict__ faces, FaceRec* __restrict__ recs,
                                                    FaceBox* __restrict__ boxes, int B, int V, int F, int H, int W)
{
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (long long)B * F) return;
    const int ib = (int)(n / F);
    FaceRec rec;
    FaceBox box;
    const bool ok = setup_face(vertices + (size_t)ib * V * 4, V, faces + (size_t)n * 3, H, W, rec, box);
    if (ok) {
        recs[n] = rec;
    } else {
        recs[n].flags = 0;
        box.i_min = 32767; box.i_max = -32768; box.r_min = 32767; box.r_max = -32768;
    }
    boxes[n] = box;
}

constexpr int TILE = 32;        // tile edge in pixels
constexpr int BLK = 8;          // block edge: one wave = one 8x8 block at a time
constexpr int LIST_CAP = 2048;  // faces scanned (and at most listed) per round

struct ListEntry {
    int32_t face;
    int16_t i_min, i_max, r_min, r_max;
};

// Per-candidate work of one wave on one 8x8 block: coverage + depth + visibility update.
// `rec` is wave-uniform: the compiler keeps it in SGPRs.
__device__ __forceinline__ void raster_candidate(const FaceRec* __restrict__ rec, int face, double px, double py,
                                                 uint32_t& zbest, int32_t& fbest)
{
    const uint32_t flags = rec->flags;
    double Fk[3];
    edge_eval(rec->coef, px, py, Fk);
    const bool c0 = (Fk[0] >= 0.0) != ((flags & 1u) != 0);
    const bool c1 = (Fk[1] >= 0.0) != ((flags & 2u) != 0);
    const bool c2 = (Fk[2] >= 0.0) != ((flags & 4u) != 0);
    if (c0 && c1 && c2) {
        const double t = Fk[2] * rec->zs[2];
        const double zn = fma(Fk[0], rec->zs[0], fma(Fk[1], rec->zs[1], t));
        if (zn >= -1.0 && zn <= 1.0) {
            const uint32_t z24 = (uint32_t)rint(fma(zn, 8388607.5, 8388607.5));
            // GL_LESS against the stored depth; equal depth keeps the lower face index, which is
            // what drawing the faces in index order does (csrc/rasterise_egl.cpp:373-379).
            if (z24 < zbest || (z24 == zbest && face < fbest)) { zbest = z24; fbest = face; }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void raster_kernel(RasterParams p)
{
    __shared__ ListEntry s_list[LIST_CAP];
    __shared__ uint32_t s_count;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ib = blockIdx.y;
    const int tile = blockIdx.x;
    const int tx0 = (tile % p.tiles_x) * TILE;
    const int tr0 = (tile / p.tiles_x) * TILE;
    const int tx1 = tx0 + TILE - 1, tr1 = tr0 + TILE - 1;

    const FaceRec* __restrict__ recs = p.recs + (size_t)ib * p.F;
    const FaceBox* __restrict__ boxes = p.boxes + (size_t)ib * p.F;

    // this lane's pixel inside block `blk` of its wave's band: column bx(blk)+lx, row r
    const int lx = lane & 7, ly = lane >> 3;
    const int r = tr0 + wave * BLK + ly;
    const double py = (double)(p.H - 1 - r) + 0.5;
    const int br0 = tr0 + wave * BLK, br1 = br0 + BLK - 1;

    uint32_t zbest[4];
    int32_t fbest[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { zbest[k] = Z24_CLEAR; fbest[k] = -1; }  // -1: a tie with the cleared depth never wins

    for (int round = 0; round < p.F; round += LIST_CAP) {
        if (tid == 0) s_count = 0;
        __syncthreads();
        const int round_end = min(p.F, round + LIST_CAP);
        for (int base = round; base < round_end; base += 256) {
            const int f = base + tid;
            bool hit = false;
            FaceBox bb;
            if (f < round_end) {
                bb = boxes[f];
                hit = bb.i_min <= tx1 && bb.i_max >= tx0 && bb.r_min <= tr1 && bb.r_max >= tr0;
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                uint32_t off = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) off = atomicAdd(&s_count, (uint32_t)__popcll(m));
                off = __shfl(off, leader);
                if (hit) {
                    const uint32_t slot = off + __popcll(m & ((1ull << lane) - 1ull));
                    ListEntry e;
                    e.face = f; e.i_min = bb.i_min; e.i_max = bb.i_max; e.r_min = bb.r_min; e.r_max = bb.r_max;
                    s_list[slot] = e;
                }
            }
        }
        __syncthreads();
        const int n = (int)s_count;

#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int bx0 = tx0 + blk * BLK, bx1 = bx0 + BLK - 1;
            const double px = (double)(bx0 + lx) + 0.5;
            for (int cb = 0; cb < n; cb += 64) {
                const int idx = cb + lane;
                bool hit = false;
                int32_t myface = 0;
                if (idx < n) {
                    const ListEntry e = s_list[idx];
                    myface = e.face;
                    hit = e.i_min <= bx1 && e.i_max >= bx0 && e.r_min <= br1 && e.r_max >= br0;
                }
                unsigned long long m = __ballot(hit);
                while (m) {
                    const int k = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int face = __builtin_amdgcn_readlane(myface, k);
                    raster_candidate(recs + face, face, px, py, zbest[blk], fbest[blk]);
                }
            }
        }
        __syncthreads();
    }

    // ---- resolve: shade (MODE 0) or export the visibility buffer (MODE 1) ----
    if (r >= p.H) return;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int x = tx0 + blk * BLK + lx;
        if (x >= p.W) continue;
        const size_t pix = ((size_t)ib * p.H + r) * p.W + x;
        const int32_t f = fbest[blk];
        if (MODE == 1) {
            p.vis[pix] = f;
            continue;
        }
        const int C = p.C;
        float* __restrict__ out = p.pixels + pix * C;
        if (f < 0) {
            const float* __restrict__ bg = p.background + pix * C;
            if ((C & 3) == 0) {
                for (int c = 0; c < C; c += 4)
                    *reinterpret_cast<float4*>(out + c) = *reinterpret_cast<const float4*>(bg + c);
            } else {
                for (int c = 0; c < C; ++c) out[c] = bg[c];
            }
            continue;
        }
        const FaceRec* __restrict__ rec = recs + f;
        double cf[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) cf[k] = rec->coef[k];
        double Fk[3];
        edge_eval(cf, (double)x + 0.5, py, Fk);
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
}

// ------------------------------------------------------------------------------------------------

hipError_t launch_setup(const float* vertices, const int32_t* faces, FaceRec* recs, FaceBox* boxes, int B, int V,
                        int F, int H, int W, hipStream_t stream)
{
    const long long n = (long long)B * F;
    if (n == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(setup_kernel, dim3(grid), dim3(256), 0, stream, vertices, faces, recs, boxes, B, V, F, H, W);
    return hipGetLastError();
}

hipError_t launch_raster(const RasterParams& p, int B, bool visibility_only, hipStream_t stream)
{
    if (B == 0) return hipSuccess;
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)B);
    if (visibility_only)
        hipLaunchKernelGGL(raster_kernel<1>, grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(raster_kernel<0>, grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace dirt
