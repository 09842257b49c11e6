// dirt_raster_common.h -- what the tiled visibility / shading kernels share (dirt_raster.hip: rounds 1-5; dirt_forward.hip:
// the two-trip forward of round 6): the tile-local candidate record, the coverage / depth step of one candidate, the state
// export and the clearing side job.
#pragma once
#include "dirt_device.h"
#include "dirt_launch.h"

namespace dirt {

// A tile is 2 x 2 wave regions; a wave region is NB x NB blocks of 8 x 8 pixels (NB * NB pixels per lane).
// NB = 2 (32 x 32 tiles, one record fetch serves 256 pixels) is the normal shape; NB = 1 (16 x 16 tiles) gives four
// times as many workgroups for small frames, where 32 x 32 tiles would leave most of the 1024 SIMDs idle.
constexpr int RTHREADS = 256;     // 4 waves: one per region
constexpr int LIST_CAP = 1024;    // candidates listed per round
constexpr int SHADE_CAP = 96;     // candidates whose set-up record (and vertex colours) stay in LDS for the shading pass
struct alignas(8) ShadeRec {      // the part of a FaceRec the shading pass reads, 104 bytes
    double coef[9];
    double inv_det;
    uint32_t flags;
    int32_t vid[3];
    double pad;
};
static_assert(sizeof(ShadeRec) == 104, "ShadeRec is 26 dwords");

// What the coverage / depth loop reads per candidate: built once per (tile, candidate) by one lane when the
// candidate is staged in LDS, 80 bytes = five 16-byte LDS reads.
//
// The specification decides coverage by the SIGN of E_k = fma(a_k, px, fma(b_k, py, c_k)) in f64 (and a tie
// rule on exact zeros).  Here E_k is first evaluated in float32 in tile-local coordinates (dx = px - ox in
// 0..31, dy = py - oy in -31..0, c' = E_k(ox, oy) rounded from f64) together with a certified bound on
// |E32_k - E_k| over the tile:
//   |E32 - E| <= u32*(2*31|a| + 3*31|b| + 3|c'|) + 2^-51*Mg,  u32 = 2^-24,  Mg = |a|W + |b|H + |c|
//            <= 2^-22*(32|a32| + 32|b32| + |c32|) + 2^-49*Mg32 = bnd_k      (Mg32: Mg from the rounded values)
// With bound = max_k bnd_k:  min_k E32_k > bound: certainly inside;  min_k E32_k < -bound: certainly outside; anything
// else (a ~1e-5 pixel strip along an edge, or exactly on it; inf / NaN) takes the specification's f64 path,
// covered_exact().  Results are bit-identical to the specification at a fraction of the f64 work.
struct alignas(16) TileRec {
    float a[3], b[3], c[3];          //  0: tile-local float32 edge functions (true sign: inside = positive)
    float bound;                     // 36
    FaceBox box;                     // 40: (face-local records, make_local_rec: the face's pixel box; its top-left corner is the origin)
    double zp[3];                    // 48: depth plane scaled to the 24-bit range, global coordinates
    uint32_t flags;                  // 72
    int32_t face;                    // 76
};
static_assert(sizeof(TileRec) == 80, "TileRec is 80 bytes");

// One lane builds the TileRec of one candidate.  ox, oy: sample position of the tile's top-left pixel.
__device__ __forceinline__ void make_tile_rec(const FaceRec& rec, int face, double ox, double oy, float wf, float hf, TileRec* out)
{
    const uint32_t flags = rec.flags;
    float bound = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = rec.coef[3 * k], b = rec.coef[3 * k + 1], c = rec.coef[3 * k + 2];
        const double cl = fma(a, ox, fma(b, oy, c));
        const float sg = (flags & (1u << k)) ? -1.f : 1.f;  // undo the sign folding: E_k = sg * F_k
        const float a32 = (float)a, b32 = (float)b, c32 = (float)cl, cg = (float)c;
        const float mg = fabsf(a32) * wf + fabsf(b32) * hf + fabsf(cg);
        const float bnd = (0x1p-22f * (32.f * fabsf(a32) + 32.f * fabsf(b32) + fabsf(c32)) + 0x1p-49f * mg) * 1.0001f;
        bound = (bnd > bound || bnd != bnd) ? bnd : bound;   // a NaN bound stays: everything is then decided exactly
        out->a[k] = sg * a32; out->b[k] = sg * b32; out->c[k] = sg * c32;
    }
    out->bound = bound; out->box.i_min = 0; out->box.i_max = 0; out->box.r_min = 0; out->box.r_max = 0;
    out->zp[0] = rec.zp[0]; out->zp[1] = rec.zp[1]; out->zp[2] = rec.zp[2];
    out->flags = flags; out->face = face;
}

// The FACE-local form of the same record (round 6): built ONCE per face by the set-up kernel instead of once per (tile,
// candidate) by the raster kernel -- the origin is the sample of the top-left pixel of the face's own pixel box, so a tile
// evaluates E32_k = fma(a_k, x - box.i_min, fma(b_k, box.r_min - r, c'_k)) with exact small integers as offsets.  The bound
// takes the largest offset a tile can use instead of 32: samples are only ever tested in 8 x 8 blocks that the box touches,
// i.e. |x - i_min|, |r - r_min| <= D = max(box width, box height) + 7, and
//   |E32 - E| <= u32 * (2 D |a| + 3 D |b| + 3 |c'|) + 2^-51 Mg <= 2^-22 * (D |a32| + D |b32| + |c32|) + 2^-49 * Mg32
// (u32 = 2^-24; the derivation of make_tile_rec's bound with D for 32).  A 20-pixel face has D = 27 (a narrower undecided
// strip than a 32 x 32 tile's), a frame-filling one D ~ W: its strip is ~30 x wider, still ~1e-4 pixel; the undecided
// samples take the specification's f64 test either way, so results do not depend on D.
__device__ __forceinline__ void make_local_rec(const FaceRec& rec, int face, const FaceBox& box, int H, float wf, float hf, TileRec* out)
{
    const uint32_t flags = rec.flags;
    const double ox = (double)box.i_min + 0.5, oy = (double)(H - 1 - box.r_min) + 0.5;
    const float D = (float)(max(box.i_max - box.i_min, box.r_max - box.r_min) + 8);
    float bound = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = rec.coef[3 * k], b = rec.coef[3 * k + 1], c = rec.coef[3 * k + 2];
        const double cl = fma(a, ox, fma(b, oy, c));
        const float sg = (flags & (1u << k)) ? -1.f : 1.f;  // undo the sign folding: E_k = sg * F_k
        const float a32 = (float)a, b32 = (float)b, c32 = (float)cl, cg = (float)c;
        const float mg = fabsf(a32) * wf + fabsf(b32) * hf + fabsf(cg);
        const float bnd = (0x1p-22f * (D * fabsf(a32) + D * fabsf(b32) + fabsf(c32)) + 0x1p-49f * mg) * 1.0001f;
        bound = (bnd > bound || bnd != bnd) ? bnd : bound;   // a NaN bound stays: everything is then decided exactly
        out->a[k] = sg * a32; out->b[k] = sg * b32; out->c[k] = sg * c32;
    }
    out->bound = bound; out->box = box;
    out->zp[0] = rec.zp[0]; out->zp[1] = rec.zp[1]; out->zp[2] = rec.zp[2];
    out->flags = flags; out->face = face;
}

// Which of a wave's NB x NB blocks (bits NB * by + bx of `m4`) a candidate's TRIANGLE can touch, given the blocks its box
// touches: an edge function is linear, so its largest value over a block's 8 x 8 samples is at a corner sample; where that is
// below -bound -- the float32 form's certified error, so the exact value is negative too -- no sample of the block is inside
// that edge.  (dx_lo, dy_hi): the record's offsets (x - origin, origin row - r) of the top-left sample of block (0, 0).  One lane per
// candidate; ~15 instructions per block that save whole passes of the serial coverage loop (round 6: about every third
// block a box touches).  NaN coefficients or bound: no comparison holds, nothing is culled.
template <int NB, int NBY = NB>
__device__ __forceinline__ uint32_t cull_blocks(const TileRec& t, uint32_t m4, float dx_lo0, float dy_hi0)
{
    const float nb = -t.bound;
#pragma unroll
    for (int by = 0; by < NBY; ++by) {
        const float dy_hi = dy_hi0 - 8.f * (float)by, dy_lo = dy_hi - 7.f;
#pragma unroll
        for (int bx = 0; bx < NB; ++bx) {
            const float dx_lo = dx_lo0 + 8.f * (float)bx, dx_hi = dx_lo + 7.f;
            bool out = false;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = fmaf(t.a[k], t.a[k] > 0.f ? dx_hi : dx_lo, fmaf(t.b[k], t.b[k] > 0.f ? dy_hi : dy_lo, t.c[k]));
                out |= e < nb;
            }
            if (out) m4 &= ~(1u << (NB * by + bx));
        }
    }
    return m4;
}

// The specification's f64 coverage test for the samples the float filter cannot decide.  Rare; kept out of
// line (and reading the FaceRec from global memory) so that nothing of it is speculated into the main loop.
static __device__ __noinline__ bool covered_exact(const FaceRec* __restrict__ rec, double px, double py)
{
    const uint32_t flags = rec->flags;
    const double F0 = fma(rec->coef[0], px, fma(rec->coef[1], py, rec->coef[2]));
    const double F1 = fma(rec->coef[3], px, fma(rec->coef[4], py, rec->coef[5]));
    const double F2 = fma(rec->coef[6], px, fma(rec->coef[7], py, rec->coef[8]));
    return ((F0 >= 0.0) != ((flags & 1u) != 0)) && ((F1 >= 0.0) != ((flags & 2u) != 0)) && ((F2 >= 0.0) != ((flags & 4u) != 0));
}

// One candidate (list index ci) against the wave's NB x NB blocks (one pixel of each per lane); `m4` (wave-uniform) says
// which blocks its box touches.  dx, dy: tile-local sample coordinates; px, py: the sample positions.
//   coverage: float32 edge functions against the certified bound; the undecided samples take the specification's f64
//             test behind a wave-uniform branch; blocks the box misses, and blocks no sample of which is covered, are skipped;
//   depth   : q = fma(zA, px, fma(zB, py, zC)) in f64, as specified.  z24 = rint(q) and the depth clip 0 <= q <= 2^24-1 cost
//             one f64 add and one 64-bit integer compare: for 0 <= q < 2^32 the low word of q + 2^52 is rint(q) (round to
//             nearest even, like rint), and the bit patterns of non-negative doubles order like the numbers while every
//             negative one has the top bit set (q is never -0: its constant term fma(zC, S, S) cannot round to -0);
//   update  : GL_LESS against the stored depth; equal depth keeps the lower face index, which is what drawing the faces
//             in index order does (csrc/rasterise_egl.cpp:373-379): both as one unsigned compare of the 64-bit key
//             (z24 << 32 | face).  The stored key starts as (Z24_CLEAR << 32 | 0): no fragment at the cleared depth is
//             ever less.  Bitwise, not short-circuit, operators: one predicated update instead of nested divergent branches.
template <int NB, int NBY = NB>
__device__ __forceinline__ void raster_candidate(const TileRec& t, const FaceRec* __restrict__ recs, int ci, uint32_t m4, const float* dx,
                                                 const float* dy, const double* px, const double* py, unsigned long long* best,
                                                 int* cbest)
{
#pragma unroll
    for (int by = 0; by < NBY; ++by) {
        if (!((m4 >> (NB * by)) & ((1u << NB) - 1u))) continue;   // wave-uniform: the candidate's box misses this block row
        float trow[3];
        trow[0] = fmaf(t.b[0], dy[by], t.c[0]);
        trow[1] = fmaf(t.b[1], dy[by], t.c[1]);
        trow[2] = fmaf(t.b[2], dy[by], t.c[2]);
        const double qrow = fma(t.zp[1], py[by], t.zp[2]);
#pragma unroll
        for (int bx = 0; bx < NB; ++bx) {
            const int k = NB * by + bx;
            if (!((m4 >> k) & 1u)) continue;
            const float E0 = fmaf(t.a[0], dx[bx], trow[0]);
            const float E1 = fmaf(t.a[1], dx[bx], trow[1]);
            const float E2 = fmaf(t.a[2], dx[bx], trow[2]);
            const float m = fminf(fminf(E0, E1), E2);
            // (the predicates as wave-wide lane masks: what combines them is scalar work, and "no lane covered" is a scalar test)
            unsigned long long cov_m = __builtin_amdgcn_ballot_w64(m > t.bound);                       // certainly inside
            const unsigned long long unsure_m = __builtin_amdgcn_ballot_w64(!(m < -t.bound)) & ~cov_m;  // neither certainly inside nor outside (inf / NaN land here)
            if (__builtin_expect(unsure_m != 0ull, 0)) {
                bool c = false;
                if (__builtin_amdgcn_inverse_ballot_w64(unsure_m)) c = covered_exact(recs + t.face, px[bx], py[by]);
                cov_m |= __builtin_amdgcn_ballot_w64(c);
            }
            if (cov_m == 0ull) continue;
            const double q = fma(t.zp[0], px[bx], qrow);
            const uint32_t z24 = (uint32_t)__double_as_longlong(q + 4503599627370496.0);
            const unsigned long long in_range_m = __builtin_amdgcn_ballot_w64((unsigned long long)__double_as_longlong(q) <= 0x416FFFFFE0000000ull);   // 0 <= q <= 16777215
            // (z24, face) as ONE 64-bit key, z24 in the high word: GL_LESS and "equal depth keeps the lower face index" are a
            // single unsigned 64-bit compare (rounds 1-3: three 32-bit compares and their scalar combination per block)
            const unsigned long long key = ((unsigned long long)z24 << 32) | (unsigned long long)(uint32_t)t.face;
            const unsigned long long wins_m = cov_m & in_range_m & __builtin_amdgcn_ballot_w64(key < best[k]);
            const bool wins = __builtin_amdgcn_inverse_ballot_w64(wins_m);
            best[k] = wins ? key : best[k];
            cbest[k] = wins ? ci : cbest[k];
        }
    }
}

// Two candidates against the wave's one block (NB = 1: 16 x 16 tiles, one pixel per lane), the arithmetic of
// raster_candidate operation for operation, the two chains interleaved.  Small frames run ONE wave per SIMD: nothing hides
// a dependent instruction's latency but the wave's own independent work, and a single candidate is one chain of ~40
// instructions (437 clocks per candidate at K3-256); two candidates at a time share the branches and fill each other's
// gaps.  Update order is the list order (t0 before t1), as in the one-candidate loop.
__device__ __forceinline__ void raster_candidate_pair(const TileRec& t0, const TileRec& t1, const FaceRec* __restrict__ recs, int ci0, int ci1,
                                                      float dx, float dy, double px, double py, unsigned long long& best, int& cbest)
{
    const float E00 = fmaf(t0.a[0], dx, fmaf(t0.b[0], dy, t0.c[0])), E10 = fmaf(t1.a[0], dx, fmaf(t1.b[0], dy, t1.c[0]));
    const float E01 = fmaf(t0.a[1], dx, fmaf(t0.b[1], dy, t0.c[1])), E11 = fmaf(t1.a[1], dx, fmaf(t1.b[1], dy, t1.c[1]));
    const float E02 = fmaf(t0.a[2], dx, fmaf(t0.b[2], dy, t0.c[2])), E12 = fmaf(t1.a[2], dx, fmaf(t1.b[2], dy, t1.c[2]));
    const float m0 = fminf(fminf(E00, E01), E02), m1 = fminf(fminf(E10, E11), E12);
    unsigned long long cov0 = __builtin_amdgcn_ballot_w64(m0 > t0.bound), cov1 = __builtin_amdgcn_ballot_w64(m1 > t1.bound);
    const unsigned long long uns0 = __builtin_amdgcn_ballot_w64(!(m0 < -t0.bound)) & ~cov0, uns1 = __builtin_amdgcn_ballot_w64(!(m1 < -t1.bound)) & ~cov1;
    if (__builtin_expect((uns0 | uns1) != 0ull, 0)) {
        bool c0 = false, c1 = false;
        if (__builtin_amdgcn_inverse_ballot_w64(uns0)) c0 = covered_exact(recs + t0.face, px, py);
        if (__builtin_amdgcn_inverse_ballot_w64(uns1)) c1 = covered_exact(recs + t1.face, px, py);
        cov0 |= __builtin_amdgcn_ballot_w64(c0); cov1 |= __builtin_amdgcn_ballot_w64(c1);
    }
    if ((cov0 | cov1) == 0ull) return;
    const double q0 = fma(t0.zp[0], px, fma(t0.zp[1], py, t0.zp[2])), q1 = fma(t1.zp[0], px, fma(t1.zp[1], py, t1.zp[2]));
    const uint32_t z0 = (uint32_t)__double_as_longlong(q0 + 4503599627370496.0), z1 = (uint32_t)__double_as_longlong(q1 + 4503599627370496.0);
    const unsigned long long in0 = __builtin_amdgcn_ballot_w64((unsigned long long)__double_as_longlong(q0) <= 0x416FFFFFE0000000ull);
    const unsigned long long in1 = __builtin_amdgcn_ballot_w64((unsigned long long)__double_as_longlong(q1) <= 0x416FFFFFE0000000ull);
    const unsigned long long key0 = ((unsigned long long)z0 << 32) | (unsigned long long)(uint32_t)t0.face;
    const unsigned long long key1 = ((unsigned long long)z1 << 32) | (unsigned long long)(uint32_t)t1.face;
    const bool w0 = __builtin_amdgcn_inverse_ballot_w64(cov0 & in0 & __builtin_amdgcn_ballot_w64(key0 < best));
    const unsigned long long b1 = w0 ? key0 : best;
    const int c1i = w0 ? ci0 : cbest;
    const bool w1 = __builtin_amdgcn_inverse_ballot_w64(cov1 & in1 & __builtin_amdgcn_ballot_w64(key1 < b1));
    best = w1 ? key1 : b1;
    cbest = w1 ? ci1 : c1i;
}

// The backward pass's state of one pixel -- csrc/shaders.cpp:64-77: {clip_w, face} and two of the three barycentrics
// (encode_bary, dirt_device.h; the face index stands for the index triple) -- or the clear values of
// csrc/rasterise_grad_egl.cpp:442-445.
__device__ __forceinline__ void store_state(const RasterParams& p, size_t pix, bool has, float b0, float b1, float b2, float clip_w, int32_t face)
{
    p.state_a[pix] = make_float2(has ? clip_w : INFINITY, __int_as_float(face));
    p.state_b[pix] = has ? encode_bary(b0, b1, b2) : make_float2(-1.f, -1.f);
}

// One workgroup's share of a buffer to clear: `per` 16-byte units (the buffers are 16-byte aligned: [B,V,4] floats, the
// 256-byte aligned workspace regions), a dword tail for caller tensors whose size is not a multiple of 16.
template <int NTHREADS = RTHREADS>
__device__ __forceinline__ void zero_share(void* buf, size_t bytes, unsigned per, unsigned gwg, int tid)
{
    const size_t units = bytes / 16;
    const size_t i0 = (size_t)gwg * per;
    uint4* q = reinterpret_cast<uint4*>(buf);
    for (unsigned i = (unsigned)tid; i < per; i += NTHREADS)
        if (i0 + i < units) q[i0 + i] = make_uint4(0u, 0u, 0u, 0u);
    if (gwg == 0 && (size_t)tid < (bytes % 16) / 4) reinterpret_cast<uint32_t*>(buf)[units * 4 + tid] = 0u;
}

}  // namespace dirt
