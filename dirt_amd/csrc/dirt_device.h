// dirt_device.h -- device-side data layout and per-sample arithmetic shared by the gfx950 kernels.
//
// The arithmetic follows the numeric specification in DESIGN.md ("Numeric specification"), which
// the CPU oracle (oracle/dirt_oracle.c) restates independently: every step is an IEEE-754 basic
// operation evaluated in the written order (the library is built with -ffp-contract=off), so the
// two agree bit for bit on coverage, depth and interpolated colour.
//
// Reference semantics being replaced (paths relative to the reference repository):
//   triangle set-up / coverage / depth / interpolation : the GL pipeline driven by
//       csrc/rasterise_egl.cpp:362-380 and csrc/rasterise_grad_egl.cpp:432-456 with the shaders of
//       csrc/shaders.cpp:16-79;
//   vertex expansion                                   : upload_vertices, csrc/rasterise_grad_egl.cu:12-34.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dirt {

// One set-up triangle, 128 bytes, 128-byte aligned so a record is one L2 line and can be fetched
// with two wide scalar loads.  The first 100 bytes are what the coverage / depth loop needs.
//
// Sign folding: for an edge whose on-edge samples are EXCLUDED by the tie rule the three
// coefficients are stored negated (F_k = -E_k).  The coverage test then is a single `F_k >= 0`
// compare per edge, xor-ed with the edge's `excl` flag.
// (Round 6: what the SHADING pass reads -- coefficients, 1/|det|, flags, vertex indices -- is the first 96 bytes, six 16-byte
// pieces that raster_kernel_v2 copies into LDS by LDS-DMA while its coverage loop runs; the depth plane follows.)
struct alignas(128) FaceRec {
    double coef[9];   //   0: (a,b,c) of edges 0,1,2, sign-folded
    double inv_det;   //  72: 1/|det|
    uint32_t flags;   //  80: bit k = edge k folded (exclusive); bit 31 = valid
    int32_t vid[3];   //  84: vertex indices of the face
    double zp[3];     //  96: depth plane scaled to the 24-bit range: q = fma(zp[0], px, fma(zp[1], py, zp[2]))
    uint32_t pad0;    // 120
    uint32_t pad1;    // 124
};
static_assert(sizeof(FaceRec) == 128, "FaceRec must be 128 bytes");
constexpr int FACE_SHADE_BYTES = 96;   // the head of a FaceRec the shading pass reads

constexpr uint32_t FACE_VALID = 0x80000000u;

// Conservative pixel bounding box in tensor orientation (columns i, rows r, inclusive).
// A culled face has an empty box (i_min > i_max) that overlaps nothing.
struct alignas(8) FaceBox {
    int16_t i_min, i_max, r_min, r_max;
};
static_assert(sizeof(FaceBox) == 8, "FaceBox must be 8 bytes");

constexpr uint32_t Z24_CLEAR = 0x00FFFFFFu;  // glClear(GL_DEPTH_BUFFER_BIT) to 1.0 in a D24 buffer

// ---------------------------------------------------------------------------------------------
// Triangle set-up for one face (the GL primitive assembly + viewport transform).
// Returns false when the face is dropped (bad index, non-finite data, zero area, behind the eye,
// outside the depth range as a whole, or covering no pixel centre).
// (In two steps so that a caller can put other work between the requests and their use: face_fetch_indices, then
// face_fetch_vertices once the indices are there, then setup_face_from -- setup_kernel clears its buffers meanwhile.)
__device__ __forceinline__ void face_fetch_indices(const int32_t* __restrict__ face, int32_t (&idx)[3])
{
    idx[0] = face[0]; idx[1] = face[1]; idx[2] = face[2];
}
// a bad index reads vertex 0 instead (the face is dropped by setup_face_from)
__device__ __forceinline__ void face_fetch_vertices(const float* __restrict__ verts, int V, const int32_t (&idx)[3], float4 (&vv)[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k)
        vv[k] = V > 0 ? *reinterpret_cast<const float4*>(verts + (size_t)((uint32_t)idx[k] < (uint32_t)V ? idx[k] : 0) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ inline bool setup_face_from(const float4 (&vv)[3], const int32_t (&idx)[3], int V, int H, int W, FaceRec& rec, FaceBox& box);

__device__ inline bool setup_face(const float* __restrict__ verts, int V, const int32_t* __restrict__ face,
                                  int H, int W, FaceRec& rec, FaceBox& box)
{
    // the three indices, then the three vertices, are requested together (two memory latencies, not six)
    int32_t idx[3];
    float4 vv[3];
    face_fetch_indices(face, idx);
    face_fetch_vertices(verts, V, idx, vv);
    return setup_face_from(vv, idx, V, H, W, rec, box);
}

// The set-up of one face in two independent halves (setup_face_from = both; setup_kernel_v2 runs them on different waves):
//   setup_face_edges: index / finiteness checks, the edge functions' coefficients, det, the depth plane, the sign folding --
//                     everything of the record but the box;
//   setup_face_box  : the w > 0 count, the trivial depth rejection, the conservative pixel box.
// A face is set up iff both return true.  Each is the corresponding part of the former single function, operation for operation.
__device__ inline bool setup_face_edges(const float4 (&vv)[3], const int32_t (&idx)[3], int V, int H, int W, FaceRec& rec)
{
    double X[3], Y[3], Wc[3], Z[3];
    const int32_t i0 = idx[0], i1 = idx[1], i2 = idx[2];
    const bool index_ok = ((uint32_t)i0 < (uint32_t)V) & ((uint32_t)i1 < (uint32_t)V) & ((uint32_t)i2 < (uint32_t)V);
    if (V <= 0) return false;
    if (!index_ok) return false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t vi = k == 0 ? i0 : (k == 1 ? i1 : i2);
        const float4 v = vv[k];
        if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w))) return false;
        X[k] = ((double)v.x + (double)v.w) * (0.5 * (double)W);
        Y[k] = ((double)v.y + (double)v.w) * (0.5 * (double)H);
        Wc[k] = (double)v.w;
        Z[k] = (double)v.z;
        rec.vid[k] = vi;
    }
    double a[3], b[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        double m1, m2;
        m1 = Y[p] * Wc[q]; m2 = Wc[p] * Y[q]; a[k] = m1 - m2;
        m1 = Wc[p] * X[q]; m2 = X[p] * Wc[q]; b[k] = m1 - m2;
        m1 = X[p] * Y[q];  m2 = Y[p] * X[q];  c[k] = m1 - m2;
    }
    const double t0 = X[0] * a[0], t1 = Y[0] * b[0], t2 = Wc[0] * c[0];
    double det = (t0 + t1) + t2;
    if (!isfinite(det) || det == 0.0) return false;
    if (det < 0.0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[k] = -a[k]; b[k] = -b[k]; c[k] = -c[k]; }
        det = -det;
    }
    const double inv_det = 1.0 / det;
    if (!isfinite(inv_det)) return false;

    // depth plane from the unfolded coefficients (NDC depth = sum_k E_k * z_k / det is affine in the sample)
    double zs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) zs[k] = Z[k] * inv_det;
    {
        double m0, m1, m2;
        m0 = a[0] * zs[0]; m1 = a[1] * zs[1]; m2 = a[2] * zs[2]; rec.zp[0] = ((m0 + m1) + m2) * 8388607.5;
        m0 = b[0] * zs[0]; m1 = b[1] * zs[1]; m2 = b[2] * zs[2]; rec.zp[1] = ((m0 + m1) + m2) * 8388607.5;
        m0 = c[0] * zs[0]; m1 = c[1] * zs[1]; m2 = c[2] * zs[2]; rec.zp[2] = fma((m0 + m1) + m2, 8388607.5, 8388607.5);
    }
    uint32_t flags = FACE_VALID;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool incl = (a[k] > 0.0) || (a[k] == 0.0 && b[k] > 0.0);
        if (!incl) { a[k] = -a[k]; b[k] = -b[k]; c[k] = -c[k]; flags |= (1u << k); }
        rec.coef[3 * k + 0] = a[k]; rec.coef[3 * k + 1] = b[k]; rec.coef[3 * k + 2] = c[k];
    }
    rec.flags = flags; rec.pad0 = 0; rec.pad1 = 0;
    rec.inv_det = inv_det;
    return true;
}

// (Vertices that setup_face_edges rejects -- bad indices, non-finite components -- may reach this half: it then returns
// anything, without trapping; the face is dropped by the other half.)
__device__ inline bool setup_face_box(const float4 (&vv)[3], int H, int W, FaceBox& box)
{
    double X[3], Y[3], Wc[3], Z[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 v = vv[k];
        X[k] = ((double)v.x + (double)v.w) * (0.5 * (double)W);
        Y[k] = ((double)v.y + (double)v.w) * (0.5 * (double)H);
        Wc[k] = (double)v.w;
        Z[k] = (double)v.z;
    }
    const int npos = (Wc[0] > 0.0) + (Wc[1] > 0.0) + (Wc[2] > 0.0);
    if (npos == 0) return false;
    int i_min = 0, i_max = W - 1, j_min = 0, j_max = H - 1;
    if (npos == 3) {
        if (Z[0] > Wc[0] && Z[1] > Wc[1] && Z[2] > Wc[2]) return false;
        if (Z[0] < -Wc[0] && Z[1] < -Wc[1] && Z[2] < -Wc[2]) return false;
        // window-space box in float32: three reciprocals instead of six f64 divisions.  The box is only a
        // work-skipping device (coverage is decided by the edge functions), so it just has to be
        // conservative: float rounding (2^-22 relative on |x| <= 32768 px) is covered by the 1/32 px margin,
        // anything larger is outside the frame on that side anyway.
        float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float rw = 1.0f / (float)Wc[k];
            float xw = (float)X[k] * rw, yw = (float)Y[k] * rw;
            if (!(fabsf(xw) <= 32768.f)) xw = xw > 0.f ? 32768.f : -32768.f;  // also catches inf / NaN of tiny w
            if (!(fabsf(yw) <= 32768.f)) yw = yw > 0.f ? 32768.f : -32768.f;
            xmin = fminf(xmin, xw); xmax = fmaxf(xmax, xw);
            ymin = fminf(ymin, yw); ymax = fmaxf(ymax, yw);
        }
        if (!(xmin <= xmax) || !(ymin <= ymax)) { xmin = ymin = -32768.f; xmax = ymax = 32768.f; }  // NaN: keep everything
        const float d = 1.0f / 32.0f;
        float lo, hi;
        lo = ceilf(fmaxf(xmin - 0.5f - d, -1.0f)); hi = floorf(fminf(xmax - 0.5f + d, (float)W));
        if (lo > (float)i_min) i_min = (int)lo;
        if (hi < (float)i_max) i_max = (int)hi;
        lo = ceilf(fmaxf(ymin - 0.5f - d, -1.0f)); hi = floorf(fminf(ymax - 0.5f + d, (float)H));
        if (lo > (float)j_min) j_min = (int)lo;
        if (hi < (float)j_max) j_max = (int)hi;
    }
    if (i_min > i_max || j_min > j_max) return false;
    box.i_min = (int16_t)i_min; box.i_max = (int16_t)i_max;
    box.r_min = (int16_t)(H - 1 - j_max); box.r_max = (int16_t)(H - 1 - j_min);
    return true;
}

__device__ inline bool setup_face_from(const float4 (&vv)[3], const int32_t (&idx)[3], int V, int H, int W, FaceRec& rec, FaceBox& box)
{
    if (!setup_face_edges(vv, idx, V, H, W, rec)) return false;
    return setup_face_box(vv, H, W, box);
}

// Folded edge functions of one record at a sample.
__device__ inline void edge_eval(const double* coef, double px, double py, double F[3])
{
    F[0] = fma(coef[0], px, fma(coef[1], py, coef[2]));
    F[1] = fma(coef[3], px, fma(coef[4], py, coef[5]));
    F[2] = fma(coef[6], px, fma(coef[7], py, coef[8]));
}

// Perspective-correct barycentrics and clip-space w of the fragment (csrc/shaders.cpp:52-57,74).
__device__ inline void bary_eval(const double F[3], uint32_t flags, double inv_det, float b[3], float& clip_w)
{
    const double s0 = (flags & 1u) ? -inv_det : inv_det;
    const double s1 = (flags & 2u) ? -inv_det : inv_det;
    const double s2 = (flags & 4u) ? -inv_det : inv_det;
    const float l0 = (float)(F[0] * s0), l1 = (float)(F[1] * s1), l2 = (float)(F[2] * s2);
    const float s = (l0 + l1) + l2;
    const float r = 1.0f / s;
    b[0] = l0 * r; b[1] = l1 * r; b[2] = l2 * r;
    clip_w = r;
}

// The backward pass's per-pixel barycentrics travel as TWO floats (state plane B).  Which two: always the two that are
// not the largest -- the third is re-derived as (1 - p) - q, whose absolute error (half an ulp of 1) is then also a
// relative error of at most 2^-22 of the value (it is >= 1/3).  With a fixed pair (b0, b1) a sliver fragment's
// b2 ~ 1e-4 would come back with a relative error of 1e-3, which is what the per-element gradient tolerance
// (tests/parity.py) sees.  The choice is carried in the two SIGN bits (barycentrics of a covered sample are >= 0):
//   code 0: (p, q) = (b0, b1), b2 derived;  sign(p) set = code 1: (b1, b2), b0 derived;  sign(q) set = code 2: (b2, b0), b1 derived.
__device__ __forceinline__ float2 encode_bary(float b0, float b1, float b2)
{
    const bool omit2 = (b2 >= b0) & (b2 >= b1);
    const bool omit0 = !omit2 & (b0 >= b1);
    const bool omit1 = !omit2 & !omit0;
    const float p = omit2 ? b0 : (omit0 ? b1 : b2);
    const float q = omit2 ? b1 : (omit0 ? b2 : b0);
    return make_float2(__uint_as_float(__float_as_uint(p) | (omit0 ? 0x80000000u : 0u)),
                       __uint_as_float(__float_as_uint(q) | (omit1 ? 0x80000000u : 0u)));
}
__device__ __forceinline__ void decode_bary(float2 e, float (&b)[3])
{
    const bool sp = (int32_t)__float_as_uint(e.x) < 0, sq = (int32_t)__float_as_uint(e.y) < 0;
    const float P = fabsf(e.x), Q = fabsf(e.y);
    const float d = (1.f - P) - Q;
    b[0] = sq ? Q : (sp ? d : P);
    b[1] = sq ? d : (sp ? P : Q);
    b[2] = sq ? P : (sp ? Q : d);
}

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8); give every XCD a
// contiguous run of tiles (a band of tile rows) so neighbouring tiles, which share faces, hit the
// same L2, and the backward pass reads a band's visibility / fragments / pixels on the XCD
// that wrote them.  Bijective for any tile count.  Speed only; nothing depends on placement.
__device__ __forceinline__ int xcd_tile(int b, int ntiles)
{
    const int x = b & 7, j = b >> 3;
    const int q = ntiles >> 3, rem = ntiles & 7;
    return x * q + min(x, rem) + j;
}

// tile -> (column, row) of the tile grid without an integer division: q = (tile * magic) >> 32 with
// magic = floor(2^32 / tiles_x) + 1 is exact for tile * tiles_x < 2^32 (tiles: at most 2^18, tiles_x: at most 2^9); the
// launcher computes the magic number (tile_magic), the kernels pay one s_mul_hi_u32 instead of the ~25 scalar
// instructions of a division by a run-time value at the head of every wave.
// (tiles_x == 1 has no 32-bit magic number -- 2^32 + 1 -- and is marked by 0: the row is the tile index itself.)
inline uint32_t tile_magic(int tiles_x) { return tiles_x <= 1 ? 0u : (uint32_t)(0x100000000ull / (uint32_t)tiles_x) + 1u; }
__device__ __forceinline__ void tile_xy(int tile, int tiles_x, uint32_t magic, int& tx, int& ty)
{
    ty = magic ? (int)__umulhi((uint32_t)tile, magic) : tile;
    tx = tile - ty * tiles_x;
}

}  // namespace dirt
