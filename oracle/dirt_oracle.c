/*
 * dirt_oracle.c -- CPU restatement of the pmh47/dirt rasterise / rasterise_grad hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity checker for the HIP kernels in
 * dirt_amd/csrc and the "port" CPU baseline of bench.py.  Nothing under dirt_amd/ may
 * import, link or call it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * What it restates (all file:line are relative to /root/reference):
 *   forward  : the OpenGL draw issued by RasteriseOpGpu::Compute, csrc/rasterise_egl.cpp:362-380,
 *              with the GL state of csrc/rasterise_egl.cpp:196-214 (depth test LESS, no culling,
 *              no blending) and the pass-through shaders csrc/shaders.cpp:16-43; the background
 *              upload / pixel download with their vertical flip, csrc/rasterise_egl.cu:10-38,65-91.
 *   backward : the barycentric / index render of csrc/rasterise_grad_egl.cpp:432-456 with shaders
 *              csrc/shaders.cpp:45-79, then assemble_grads, csrc/rasterise_grad_egl.cu:93-236,
 *              line for line, and the channel grouping of dirt/rasterise_ops.py:86-108,132-177.
 *
 * PARITY STATUS.  The forward arithmetic of the reference lives in the NVIDIA OpenGL driver, which
 * is closed source, un-vendored and cannot run here.  The forward restatement therefore follows
 * the OpenGL 3.3 core specification (sections 2.13, 2.14, 3.6, 4.1.5) and is pinned by the only
 * known-answer test the reference has, tests/square_test.py:11-17,54-57 (see tests/test_oracle.py).
 * The reference has NO gradient test of any kind, but its gradient kernel compiles for the host:
 * oracle/_ref (oracle/make_ref.py) is csrc/rasterise_grad_egl.cu itself behind a TensorFlow / CUDA shim,
 * and with DIRT_ORACLE_FLAG_F32_SEQUENTIAL the backward restatement below equals it BIT FOR BIT in all
 * four outputs (tests/test_oracle_ref.py; committed vectors tests/golden/ref_grads.npz), given the
 * visibility surfaces; and the forward restatement equals the reference's own upload_background /
 * download_pixels (csrc/rasterise_egl.cu, same shim) around dirt_oracle_draw_gl, a draw that knows only GL
 * window coordinates: vertical flip, atlas tiling and channel packing are the reference's.  "parity
 * unpinned" therefore applies to what the GL driver does in between -- coverage, perspective-correct
 * interpolation, depth ordering beyond the square test: pinned to the GL specification, not to a
 * reference run.
 *
 * NUMERIC SPECIFICATION (shared, by specification and not by code, with the HIP kernels; see
 * DESIGN.md section "Numeric specification").  Every operation below is an IEEE-754 basic
 * operation (+,-,*,/,fma,rint,convert), evaluated in the order written, without contraction, so
 * that CPU and GPU agree bit for bit:
 *
 *   per vertex k of a face (x,y,z,w are the float32 clip coordinates):
 *       X_k = ((double)x + (double)w) * (0.5*W)      window-space homogeneous x (y-up, pixels)
 *       Y_k = ((double)y + (double)w) * (0.5*H)
 *       W_k = (double)w
 *   per edge k (opposite vertex k; p=(k+1)%3, q=(k+2)%3), (a,b,c)_k = v_p x v_q:
 *       a_k = Y_p*W_q - W_p*Y_q ;  b_k = W_p*X_q - X_p*W_q ;  c_k = X_p*Y_q - Y_p*X_q
 *   det = (X_0*a_0 + Y_0*b_0) + W_0*c_0 ; faces with det==0 or non-finite data are dropped; the
 *   nine coefficients are multiplied by sign(det) (both windings are drawn: culling is never
 *   enabled, csrc/rasterise_egl.cpp:213-214) ; inv_det = 1/|det|.
 *   per sample (pixel column i, GL row j = H-1-r; px=i+0.5, py=j+0.5):
 *       E_k   = fma(a_k, px, fma(b_k, py, c_k))                      (= lambda_k / w * |det|)
 *       inside iff for all k: E_k > 0 or (E_k == 0 and (a_k > 0 or (a_k == 0 and b_k > 0)))
 *               -- watertight tie rule: a sample on an edge belongs to the triangle whose
 *                  interior lies at larger x, or for horizontal edges at larger framebuffer row.
 *       q     = fma(qA, px, fma(qB, py, qC))                         window depth scaled to the 24-bit range:
 *               NDC depth zn = sum_k E_k*z_k/det is affine in (px,py); q = zn*S + S with S = 8388607.5
 *               = (2^24-1)/2 maps [-1,1] to [0, 2^24-1].  The plane is fixed at set-up time:
 *               zs_k = (double)z_k * inv_det ;
 *               zA = (a_0*zs_0 + a_1*zs_1) + a_2*zs_2 ; zB, zC likewise from b_k, c_k ;
 *               qA = zA*S ; qB = zB*S ; qC = fma(zC, S, S)
 *               the fragment is kept iff 0 <= q <= 16777215 (the depth clip -1<=zn<=1)
 *       z24   = (uint32) rint(q)                                     24-bit depth (D24S8 buffer,
 *                                                                    csrc/rasterise_egl.cpp:245)
 *       the fragment wins iff z24 < stored (GL_LESS, buffer cleared to 0xFFFFFF); faces are
 *       visited in index order so the earlier face wins ties.
 *   winner shading:
 *       l_k = (float)(E_k * inv_det) ; s = (l_0 + l_1) + l_2 ; r = 1.0f / s ;
 *       b_k = l_k * r  (perspective-correct barycentrics) ; clip_w = r ;
 *       colour_c = fmaf(b_2, col_2c, fmaf(b_1, col_1c, b_0 * col_0c)).
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fopenmp).
 */

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DIRT_ORACLE_FLAG_Q1_INTENDED 1u /* L1 over the real channels of a 1-channel group */
/* Accumulate exactly as ONE CUDA thread walking the reference's loops would: float32 adds, pixels in the
   order of CUDA_AXIS_KERNEL_LOOP(buffer_x){CUDA_AXIS_KERNEL_LOOP(buffer_y)} (csrc/rasterise_grad_egl.cu:
   106-107), one accumulator per channel group, groups then added in float32 (dirt/rasterise_ops.py:163).
   With it the oracle must equal oracle/_ref (the reference's kernel compiled for the host) BIT FOR BIT. */
#define DIRT_ORACLE_FLAG_F32_SEQUENTIAL 2u

typedef struct {
    double a[3], b[3], c[3];
    double inv_det;
    double z[3];  /* clip-space z of the three vertices */
    double zp[3]; /* scaled depth plane qA, qB, qC: q = fma(qA, px, fma(qB, py, qC)) */
    int32_t vid[3];
    int incl[3];
    int i_min, i_max, r_min, r_max; /* pixel-column range and tensor-row range, inclusive */
    int valid;
} OFace;

static int finite4(const float *v) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]); }

/* Triangle setup: the GL primitive assembly + viewport transform of one face. */
static void setup_face(const float *verts, int V, const int32_t *face, int H, int W, OFace *o)
{
    o->valid = 0;
    double X[3], Y[3], Wc[3];
    for (int k = 0; k < 3; ++k) {
        int32_t vi = face[k];
        if (vi < 0 || vi >= V) return;
        const float *v = verts + (size_t)vi * 4;
        if (!finite4(v)) return;
        X[k] = ((double)v[0] + (double)v[3]) * (0.5 * (double)W);
        Y[k] = ((double)v[1] + (double)v[3]) * (0.5 * (double)H);
        Wc[k] = (double)v[3];
        o->z[k] = (double)v[2];
        o->vid[k] = vi;
    }
    for (int k = 0; k < 3; ++k) {
        int p = (k + 1) % 3, q = (k + 2) % 3;
        double m1, m2;
        m1 = Y[p] * Wc[q]; m2 = Wc[p] * Y[q]; o->a[k] = m1 - m2;
        m1 = Wc[p] * X[q]; m2 = X[p] * Wc[q]; o->b[k] = m1 - m2;
        m1 = X[p] * Y[q]; m2 = Y[p] * X[q]; o->c[k] = m1 - m2;
    }
    double t0 = X[0] * o->a[0], t1 = Y[0] * o->b[0], t2 = Wc[0] * o->c[0];
    double det = (t0 + t1) + t2;
    if (!(isfinite(det)) || det == 0.0) return;
    if (det < 0.0) {
        for (int k = 0; k < 3; ++k) { o->a[k] = -o->a[k]; o->b[k] = -o->b[k]; o->c[k] = -o->c[k]; }
        det = -det;
    }
    o->inv_det = 1.0 / det;
    if (!isfinite(o->inv_det)) return;
    double zs[3];
    for (int k = 0; k < 3; ++k) {
        o->incl[k] = (o->a[k] > 0.0) || (o->a[k] == 0.0 && o->b[k] > 0.0);
        zs[k] = o->z[k] * o->inv_det;
    }
    {
        double m0, m1, m2;
        m0 = o->a[0] * zs[0]; m1 = o->a[1] * zs[1]; m2 = o->a[2] * zs[2]; o->zp[0] = ((m0 + m1) + m2) * 8388607.5;
        m0 = o->b[0] * zs[0]; m1 = o->b[1] * zs[1]; m2 = o->b[2] * zs[2]; o->zp[1] = ((m0 + m1) + m2) * 8388607.5;
        m0 = o->c[0] * zs[0]; m1 = o->c[1] * zs[1]; m2 = o->c[2] * zs[2]; o->zp[2] = fma((m0 + m1) + m2, 8388607.5, 8388607.5);
    }

    /* Conservative screen bounding box (only a work-skipping device: the edge test decides). */
    int npos = (Wc[0] > 0.0) + (Wc[1] > 0.0) + (Wc[2] > 0.0);
    if (npos == 0) return; /* entirely behind the eye: clipped away (GL spec 2.13) */
    int i_min = 0, i_max = W - 1, j_min = 0, j_max = H - 1;
    if (npos == 3) {
        /* depth clip, whole-triangle form of -w <= z <= w */
        if (o->z[0] > Wc[0] && o->z[1] > Wc[1] && o->z[2] > Wc[2]) return;
        if (o->z[0] < -Wc[0] && o->z[1] < -Wc[1] && o->z[2] < -Wc[2]) return;
        double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        for (int k = 0; k < 3; ++k) {
            double xw = X[k] / Wc[k], yw = Y[k] / Wc[k];
            xmin = fmin(xmin, xw); xmax = fmax(xmax, xw);
            ymin = fmin(ymin, yw); ymax = fmax(ymax, yw);
        }
        const double d = 1.0 / 1024.0;
        double lo, hi;
        lo = ceil(fmax(xmin - 0.5 - d, -1.0)); hi = floor(fmin(xmax - 0.5 + d, (double)W));
        if (lo > i_min) i_min = (int)lo;
        if (hi < i_max) i_max = (int)hi;
        lo = ceil(fmax(ymin - 0.5 - d, -1.0)); hi = floor(fmin(ymax - 0.5 + d, (double)H));
        if (lo > j_min) j_min = (int)lo;
        if (hi < j_max) j_max = (int)hi;
    }
    if (i_min > i_max || j_min > j_max) return;
    o->i_min = i_min; o->i_max = i_max;
    o->r_min = H - 1 - j_max; o->r_max = H - 1 - j_min;
    o->valid = 1;
}

static inline int sample_inside(const OFace *o, double px, double py, double E[3])
{
    for (int k = 0; k < 3; ++k) {
        E[k] = fma(o->a[k], px, fma(o->b[k], py, o->c[k]));
        if (!(E[k] > 0.0 || (E[k] == 0.0 && o->incl[k]))) return 0;
    }
    return 1;
}

static inline int sample_depth(const OFace *o, double px, double py, uint32_t *z24)
{
    double q = fma(o->zp[0], px, fma(o->zp[1], py, o->zp[2]));
    if (!(q >= 0.0 && q <= 16777215.0)) return 0;
    *z24 = (uint32_t)rint(q);
    return 1;
}

static inline void sample_bary(const OFace *o, const double E[3], float b[3], float *clip_w)
{
    float l0 = (float)(E[0] * o->inv_det), l1 = (float)(E[1] * o->inv_det), l2 = (float)(E[2] * o->inv_det);
    float s = (l0 + l1) + l2;
    float r = 1.0f / s;
    b[0] = l0 * r; b[1] = l1 * r; b[2] = l2 * r;
    *clip_w = r;
}

/*
 * Visibility of one scene: for every pixel (tensor orientation, top row first) the index of the
 * front-most face (or -1).  This is the z-buffered draw of csrc/rasterise_egl.cpp:371-379 /
 * csrc/rasterise_grad_egl.cpp:446-455.  Rows are processed in independent bands so that the CPU
 * baseline can use every core; within a band faces are visited in index order.
 */
static void scene_visibility(const OFace *faces, int F, int H, int W, int32_t *face_id)
{
    const int band = 8;
    int nbands = (H + band - 1) / band;
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < nbands; ++bi) {
        int r0 = bi * band, r1 = r0 + band - 1;
        if (r1 > H - 1) r1 = H - 1;
        uint32_t *zbuf = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)band * W);
        for (int n = 0; n < band * W; ++n) zbuf[n] = 0x00FFFFFFu; /* glClear(DEPTH) to 1.0 */
        for (int r = r0; r <= r1; ++r)
            for (int i = 0; i < W; ++i) face_id[(size_t)r * W + i] = -1;
        for (int f = 0; f < F; ++f) {
            const OFace *o = &faces[f];
            if (!o->valid || o->r_max < r0 || o->r_min > r1) continue;
            int ra = o->r_min > r0 ? o->r_min : r0, rb = o->r_max < r1 ? o->r_max : r1;
            for (int r = ra; r <= rb; ++r) {
                double py = (double)(H - 1 - r) + 0.5;
                for (int i = o->i_min; i <= o->i_max; ++i) {
                    double E[3];
                    uint32_t z24;
                    if (!sample_inside(o, (double)i + 0.5, py, E)) continue;
                    if (!sample_depth(o, (double)i + 0.5, py, &z24)) continue;
                    uint32_t *zb = &zbuf[(size_t)(r - r0) * W + i];
                    if (z24 < *zb) { *zb = z24; face_id[(size_t)r * W + i] = f; }
                }
            }
        }
        free(zbuf);
    }
}

static OFace *setup_scene(const float *verts, int V, const int32_t *faces, int F, int H, int W)
{
    OFace *of = (OFace *)malloc(sizeof(OFace) * (size_t)(F > 0 ? F : 1));
#pragma omp parallel for schedule(static)
    for (int f = 0; f < F; ++f) setup_face(verts, V, faces + (size_t)f * 3, H, W, &of[f]);
    return of;
}

static int check_dims(int B, int V, int F, int H, int W, int C)
{
    return B >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0 && C > 0;
}

/*
 * Forward: `Rasterise` (csrc/rasterise_egl.cpp:32-51,276-407) for any C >= 1.  The Python layer
 * of the reference splits C not in {1,3} into channel groups (dirt/rasterise_ops.py:86-108); the
 * forward result is group-invariant (same geometry, same winner), so all C channels are
 * interpolated from one visibility pass.
 */
int dirt_oracle_forward(const float *background, const float *vertices, const float *vertex_colors,
                        const int32_t *faces, float *pixels, int B, int V, int F, int H, int W, int C)
{
    if (!check_dims(B, V, F, H, W, C)) return -1;
    size_t P = (size_t)H * W;
    int32_t *fid = (int32_t *)malloc(sizeof(int32_t) * P);
    for (int ib = 0; ib < B; ++ib) {
        const float *verts = vertices + (size_t)ib * V * 4;
        const float *cols = vertex_colors + (size_t)ib * V * C;
        const float *bg = background + (size_t)ib * P * C;
        float *out = pixels + (size_t)ib * P * C;
        OFace *of = setup_scene(verts, V, faces + (size_t)ib * F * 3, F, H, W);
        scene_visibility(of, F, H, W, fid);
#pragma omp parallel for schedule(static)
        for (int r = 0; r < H; ++r) {
            double py = (double)(H - 1 - r) + 0.5;
            for (int i = 0; i < W; ++i) {
                size_t p = (size_t)r * W + i;
                int32_t f = fid[p];
                if (f < 0) { /* pixels start as the background: csrc/rasterise_egl.cpp:348-356 */
                    memcpy(out + p * C, bg + p * C, sizeof(float) * (size_t)C);
                    continue;
                }
                const OFace *o = &of[f];
                double E[3];
                float b[3], cw;
                sample_inside(o, (double)i + 0.5, py, E);
                sample_bary(o, E, b, &cw);
                const float *c0 = cols + (size_t)o->vid[0] * C, *c1 = cols + (size_t)o->vid[1] * C,
                            *c2 = cols + (size_t)o->vid[2] * C;
                for (int c = 0; c < C; ++c) out[p * C + c] = fmaf(b[2], c2[c], fmaf(b[1], c1[c], b[0] * c0[c]));
            }
        }
        free(of);
    }
    free(fid);
    return 0;
}

/*
 * Visibility "surfaces" of one scene in GL buffer orientation (row 0 = bottom), exactly what the
 * backward fragment shader writes (csrc/shaders.cpp:64-77) over the clear values of
 * csrc/rasterise_grad_egl.cpp:442-445: bary_w[y][x] = (b0,b1,b2,clip_w) or (-1,-1,-1,+inf);
 * index[y][x] = (i0,i1,i2) as floats or (-1,-1,-1).
 */
static void scene_surfaces(const OFace *of, int F, int H, int W, float *bary_w, float *index_f)
{
    size_t P = (size_t)H * W;
    int32_t *fid = (int32_t *)malloc(sizeof(int32_t) * P);
    scene_visibility(of, F, H, W, fid);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) { /* y = GL buffer row */
        int r = H - 1 - y;
        for (int x = 0; x < W; ++x) {
            size_t s = ((size_t)y * W + x);
            int32_t f = fid[(size_t)r * W + x];
            if (f < 0) {
                bary_w[s * 4 + 0] = bary_w[s * 4 + 1] = bary_w[s * 4 + 2] = -1.f;
                bary_w[s * 4 + 3] = INFINITY;
                index_f[s * 3 + 0] = index_f[s * 3 + 1] = index_f[s * 3 + 2] = -1.f;
            } else {
                const OFace *o = &of[f];
                double E[3];
                float b[3], cw;
                sample_inside(o, (double)x + 0.5, (double)y + 0.5, E);
                sample_bary(o, E, b, &cw);
                bary_w[s * 4 + 0] = b[0]; bary_w[s * 4 + 1] = b[1]; bary_w[s * 4 + 2] = b[2];
                bary_w[s * 4 + 3] = cw;
                for (int k = 0; k < 3; ++k) index_f[s * 3 + k] = (float)o->vid[k];
            }
        }
    }
    free(fid);
}

static inline void atomic_add_d(double *p, double v)
{
#pragma omp atomic
    *p += v;
}

/* One accumulation: `acc` is the sum (double; rounded to float32 after every add in the sequential
   mode -- a double add of two floats rounded to float IS the float add, 53 >= 2*24+2), `mass` (may be
   NULL) the L1 mass of the terms, the scale the per-element tolerance of the parity tests refers to. */
static inline void accumulate(double *acc, double *mass, size_t idx, float term, float term_mass, int seq)
{
    if (seq) {
        acc[idx] = (double)(float)(acc[idx] + (double)term);
        if (mass) mass[idx] += (double)term_mass;
    } else {
        atomic_add_d(&acc[idx], (double)term);
        if (mass) atomic_add_d(&mass[idx], (double)term_mass);
    }
}

/* The CANCELLATION scale of a position-gradient term: the same product with every difference the reference forms on
   the way -- the Scharr filter's (a + b) - c - d, sum_k b_k * vertex_k.x -- replaced by the sum of the magnitudes.  Where a
   term is the rounding residue of such a difference (a frame one pixel wide: every tap of the x filter is the same
   pixel, and ((a + b) - a) - b is +-ulp, not 0) its value is defined by the reference only up to a few ulps of THIS
   scale -- nvcc's own choice of fma contraction would change it -- and so is the parity tolerance (tests/parity.py). */
static inline void accumulate_cond(double *cond, size_t idx, double scale, int seq)
{
    if (!cond) return;
    if (seq) cond[idx] += scale;
    else atomic_add_d(&cond[idx], scale);
}

/*
 * assemble_grads for one scene and ONE channel group (csrc/rasterise_grad_egl.cu:93-236).
 * `pix` / `gpix` are the group's contiguous [B,H,W,G] slices (what TF hands the op after
 * dirt/rasterise_ops.py:156-157), G in {1,3}; `iib` selects the scene.  Gradients are accumulated
 * in double (the reference uses float atomics in unspecified order).
 * Variable names follow the CUDA source.
 */
static void assemble_grads_group(double *grad_vertices /*[V,4]*/, double *grad_vertex_colors /*[V,C] */,
                                 double *mass_vertices /*[V,4] or NULL*/, double *mass_vertex_colors /*[V,C] or NULL*/,
                                 double *cond_vertices /*[V,4] or NULL*/,
                                 float *grad_background /*[H,W,C] of this scene*/, float *debug_thingy /*[H,W,3] or NULL*/,
                                 const float *bary_w, const float *index_f, const float *pix, const float *gpix,
                                 const float *vertices /*[V,4] of this scene*/, int iib, int B, int H, int W, int G,
                                 int C, int c_begin, unsigned flags)
{
    const int frame_height = H, frame_width = W, channels = G;
    const size_t total = (size_t)B * H * W * G;
    const int seq = (flags & DIRT_ORACLE_FLAG_F32_SEQUENTIAL) != 0;
    const long n_pixels = (long)H * W;
#pragma omp parallel for schedule(static) if (!seq)
    for (long n = 0; n < n_pixels; ++n) {
        {
            /* parallel: row-major; sequential: buffer_x outer, buffer_y inner, as one thread of the reference walks */
            const int buffer_y = seq ? (int)(n % H) : (int)(n / W);
            const int buffer_x = seq ? (int)(n / H) : (int)(n % W);
            const int x_in_frame = buffer_x;
            const int y_in_frame = frame_height - 1 - buffer_y; /* :111 vertical flip */

            /* at(): csrc/rasterise_grad_egl.cu:113-124.  Always reads "channels" 0,1,2; for G==1
               the Eigen index arithmetic aliases channels 1,2 onto the next two floats of the
               flattened [B,H,W,1] tensor (quirk Q1); reads past the end are clamped to the last
               element here (undefined in the reference). */
            float sx[3], sy[3];
            float asx[3], asy[3]; /* the filters' cancellation scale: the same taps, magnitudes summed */
            {
                float t[3][3][3]; /* [oy+1][ox+1][ch] */
                for (int oy = -1; oy <= 1; ++oy)
                    for (int ox = -1; ox <= 1; ++ox) {
                        int ux = x_in_frame + ox, uy = y_in_frame - oy; /* :115-116 */
                        int cx = ux < 0 ? 0 : (ux > frame_width - 1 ? frame_width - 1 : ux);
                        int cy = uy < 0 ? 0 : (uy > frame_height - 1 ? frame_height - 1 : uy);
                        size_t base = (((size_t)iib * H + cy) * W + cx) * G;
                        for (int ch = 0; ch < 3; ++ch) {
                            size_t idx = base + ch;
                            if (G == 1 && (flags & DIRT_ORACLE_FLAG_Q1_INTENDED)) idx = base; /* unused below */
                            if (idx > total - 1) idx = total - 1;
                            t[oy + 1][ox + 1][ch] = pix[idx];
                        }
                    }
#define AT(ox, oy, ch) t[(oy) + 1][(ox) + 1][ch]
                for (int ch = 0; ch < 3; ++ch) { /* :126-127, negative-offset minus positive-offset */
                    float d1 = ((AT(-1, -1, ch) + AT(-1, +1, ch)) - AT(+1, -1, ch)) - AT(+1, +1, ch);
                    float d2 = AT(-1, 0, ch) - AT(+1, 0, ch);
                    float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
                    sx[ch] = m1 + m2;
                    d1 = ((AT(-1, -1, ch) + AT(+1, -1, ch)) - AT(-1, +1, ch)) - AT(+1, +1, ch);
                    d2 = AT(0, -1, ch) - AT(0, +1, ch);
                    m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
                    sy[ch] = m1 + m2;
                    float corners = (fabsf(AT(-1, -1, ch)) + fabsf(AT(-1, +1, ch))) + (fabsf(AT(+1, -1, ch)) + fabsf(AT(+1, +1, ch)));
                    asx[ch] = corners * (3.f / 32.f) + (fabsf(AT(-1, 0, ch)) + fabsf(AT(+1, 0, ch))) * (10.f / 32.f);
                    asy[ch] = corners * (3.f / 32.f) + (fabsf(AT(0, -1, ch)) + fabsf(AT(0, +1, ch))) * (10.f / 32.f);
                }
#undef AT
            }

            size_t s_here = (size_t)buffer_y * W + buffer_x;
            float barycentric[3] = {bary_w[s_here * 4 + 0], bary_w[s_here * 4 + 1], bary_w[s_here * 4 + 2]};
            float clip_w = bary_w[s_here * 4 + 3];
            float index_f3[3] = {index_f[s_here * 3 + 0], index_f[s_here * 3 + 1], index_f[s_here * 3 + 2]};

            size_t p_here = (size_t)y_in_frame * W + x_in_frame;
            const float *g_here = gpix + (((size_t)iib * H + y_in_frame) * W + x_in_frame) * G;

            /* colour gradients, :135-148 */
            if (barycentric[0] != -1.f) {
                for (int k = 0; k < 3; ++k) {
                    int vertex_index = (int)index_f3[k];
                    for (int channel = 0; channel < channels; ++channel) {
                        float color_grad = g_here[channel] * barycentric[k];
                        accumulate(grad_vertex_colors, mass_vertex_colors, (size_t)vertex_index * C + c_begin + channel,
                                   color_grad, fabsf(color_grad), seq);
                    }
                }
            } else {
                for (int channel = 0; channel < channels; ++channel)
                    grad_background[p_here * C + c_begin + channel] = g_here[channel];
            }

            if (debug_thingy) { /* :150-151, reads grad_pixels "channels" 1 and 2 (aliasing as Q1) */
                for (int ch = 1; ch <= 2; ++ch) {
                    size_t idx = (((size_t)iib * H + y_in_frame) * W + x_in_frame) * G + ch;
                    if (idx > total - 1) idx = total - 1;
                    debug_thingy[p_here * 3 + ch] = gpix[idx];
                }
            }

            /* dilation, :155-194 */
            if (x_in_frame > 0 && y_in_frame > 0 && x_in_frame < frame_width - 1 && y_in_frame < frame_height - 1) {
                int dilated = 0;
                float l1x, l1y;
                if (G == 1 && (flags & DIRT_ORACLE_FLAG_Q1_INTENDED)) {
                    l1x = fabsf(sx[0]); l1y = fabsf(sy[0]);
                } else {
                    l1x = (fabsf(sx[0]) + fabsf(sx[1])) + fabsf(sx[2]); /* Vec3::L1, :82-84 */
                    l1y = (fabsf(sy[0]) + fabsf(sy[1])) + fabsf(sy[2]);
                }
                int off_x = l1x > l1y ? 1 : 0, off_y = l1x > l1y ? 0 : 1; /* :185 */
                if ((x_in_frame + y_in_frame) % 2 == 1) { off_x = -off_x; off_y = -off_y; } /* :186-190 */
                for (int attempt = 0; attempt < 2 && !dilated; ++attempt) { /* :191-193 */
                    int ox = attempt == 0 ? off_x : -off_x, oy = attempt == 0 ? off_y : -off_y;
                    size_t s_off = (size_t)(buffer_y + oy) * W + (buffer_x + ox); /* :161-162, GL orientation */
                    const float *idx_off = index_f + s_off * 3;
                    const float *bw_off = bary_w + s_off * 4;
                    float clip_w_at_offset = bw_off[3];
                    int differs = idx_off[0] != index_f3[0] || idx_off[1] != index_f3[1] || idx_off[2] != index_f3[2];
                    if (idx_off[0] != -1.f && differs && clip_w > clip_w_at_offset) { /* :165 */
                        barycentric[0] = bw_off[0]; barycentric[1] = bw_off[1]; barycentric[2] = bw_off[2];
                        index_f3[0] = idx_off[0]; index_f3[1] = idx_off[1]; index_f3[2] = idx_off[2];
                        clip_w = clip_w_at_offset;
                        dilated = 1;
                        if (debug_thingy) debug_thingy[p_here * 3 + 0] = 1.e-2f; /* :172 */
                    }
                }
            }

            /* position gradients, :196-232 */
            if (barycentric[0] != -1.f) {
                const float width_f = (float)frame_width, height_f = (float)frame_height;
                float dL_dx = 0.f, dL_dy = 0.f;
                double c_dx = 0., c_dy = 0.; /* cancellation scales of dL_dx, dL_dy */
                for (int channel = 0; channel < channels; ++channel) {
                    float dL_dchannel = g_here[channel];
                    float m = dL_dchannel * sx[channel];
                    dL_dx = dL_dx + m;
                    m = dL_dchannel * sy[channel];
                    dL_dy = dL_dy + m;
                    c_dx += fabs((double)dL_dchannel) * asx[channel];
                    c_dy += fabs((double)dL_dchannel) * asy[channel];
                }
                float clip_x = 0.f, clip_y = 0.f;
                double c_clip_x = 0., c_clip_y = 0.;
                for (int k = 0; k < 3; ++k) {
                    int vertex_index = (int)index_f3[k];
                    float m = barycentric[k] * vertices[(size_t)vertex_index * 4 + 0];
                    clip_x = clip_x + m;
                    c_clip_x += fabs((double)m);
                    m = barycentric[k] * vertices[(size_t)vertex_index * 4 + 1];
                    clip_y = clip_y + m;
                    c_clip_y += fabs((double)m);
                }
                for (int k = 0; k < 3; ++k) {
                    float d_xview_by_xclip = (.5f * width_f) / clip_w;
                    float d_yview_by_yclip = (.5f * height_f) / clip_w;
                    float ww = clip_w * clip_w;
                    float d_xview_by_wclip = ((-.5f * width_f) * clip_x) / ww;
                    float d_yview_by_wclip = ((-.5f * height_f) * clip_y) / ww;
                    float dLx_b = dL_dx * barycentric[k];
                    float dLy_b = dL_dy * barycentric[k];
                    int vertex_index = (int)index_f3[k];
                    float gx = dLx_b * d_xview_by_xclip;
                    float gy = dLy_b * d_yview_by_yclip;
                    float gw1 = dLx_b * d_xview_by_wclip, gw2 = dLy_b * d_yview_by_wclip;
                    float gw = gw1 + gw2;
                    accumulate(grad_vertices, mass_vertices, (size_t)vertex_index * 4 + 0, gx, fabsf(gx), seq);
                    accumulate(grad_vertices, mass_vertices, (size_t)vertex_index * 4 + 1, gy, fabsf(gy), seq);
                    accumulate(grad_vertices, mass_vertices, (size_t)vertex_index * 4 + 3, gw, fabsf(gw1) + fabsf(gw2), seq);
                    if (cond_vertices) {
                        double bk = fabs((double)barycentric[k]), w1 = fabs((double)clip_w), w2 = w1 * w1;
                        accumulate_cond(cond_vertices, (size_t)vertex_index * 4 + 0, c_dx * bk * (.5 * width_f) / w1, seq);
                        accumulate_cond(cond_vertices, (size_t)vertex_index * 4 + 1, c_dy * bk * (.5 * height_f) / w1, seq);
                        accumulate_cond(cond_vertices, (size_t)vertex_index * 4 + 3,
                                        c_dx * bk * (.5 * width_f) * c_clip_x / w2 + c_dy * bk * (.5 * height_f) * c_clip_y / w2, seq);
                    }
                }
            }
        }
    }
}

/*
 * Backward: `RasteriseGrad` (csrc/rasterise_grad_egl.cpp:33-53,324-485) for any C >= 1, with
 * the channel grouping of dirt/rasterise_ops.py:145-165: groups of 3 while at least 3 channels
 * remain, then singles; grad_vertices summed over groups, the other two concatenated.
 * debug_thingy (optional, [B,H,W,3]) is that of the FIRST group.
 */
int dirt_oracle_backward_ex(const float *vertices, const int32_t *faces, const float *pixels, const float *grad_pixels,
                            float *grad_background, float *grad_vertices, float *grad_vertex_colors, float *debug_thingy,
                            float *mass_vertices /*[B,V,4] or NULL*/, float *mass_vertex_colors /*[B,V,C] or NULL*/,
                            float *cond_vertices /*[B,V,4] or NULL*/,
                            int B, int V, int F, int H, int W, int C, unsigned flags)
{
    if (!check_dims(B, V, F, H, W, C)) return -1;
    if (V > (1 << 24)) return -2; /* csrc/rasterise_grad_egl.cpp:399-405 */
    const int seq = (flags & DIRT_ORACLE_FLAG_F32_SEQUENTIAL) != 0;
    size_t P = (size_t)H * W;
    size_t nv = (size_t)B * V * 4, nvc = (size_t)B * V * C;
    /* launch_grad_assembly zeroes every output first: csrc/rasterise_grad_egl.cu:244-250 */
    memset(grad_background, 0, sizeof(float) * (size_t)B * P * C);
    if (debug_thingy) memset(debug_thingy, 0, sizeof(float) * (size_t)B * P * 3);
    double *gv = (double *)calloc(nv + 1, sizeof(double));
    double *gv_group = seq ? (double *)calloc(nv + 1, sizeof(double)) : gv; /* one op call's grad_vertices */
    double *gvc = (double *)calloc(nvc + 1, sizeof(double));
    double *mv = mass_vertices ? (double *)calloc(nv + 1, sizeof(double)) : NULL;
    double *mvc = mass_vertex_colors ? (double *)calloc(nvc + 1, sizeof(double)) : NULL;
    double *cv = cond_vertices ? (double *)calloc(nv + 1, sizeof(double)) : NULL;
    float *bary_w = (float *)malloc(sizeof(float) * P * 4);
    float *index_f = (float *)malloc(sizeof(float) * P * 3);
    float *pix_g = (float *)malloc(sizeof(float) * (size_t)B * P * 3);
    float *gpix_g = (float *)malloc(sizeof(float) * (size_t)B * P * 3);

    for (int c_begin = 0; c_begin < C;) {
        int G = (c_begin + 3 <= C) ? 3 : 1; /* dirt/rasterise_ops.py:148-152 */
        /* contiguous group slices over the whole batch: pixels[..., c_begin:c_begin+G] */
        for (size_t n = 0; n < (size_t)B * P; ++n)
            for (int ch = 0; ch < G; ++ch) {
                pix_g[n * G + ch] = pixels[n * C + c_begin + ch];
                gpix_g[n * G + ch] = grad_pixels[n * C + c_begin + ch];
            }
        if (seq) memset(gv_group, 0, sizeof(double) * nv);
        for (int ib = 0; ib < B; ++ib) {
            const float *verts = vertices + (size_t)ib * V * 4;
            OFace *of = setup_scene(verts, V, faces + (size_t)ib * F * 3, F, H, W);
            scene_surfaces(of, F, H, W, bary_w, index_f);
            assemble_grads_group(gv_group + (size_t)ib * V * 4, gvc + (size_t)ib * V * C,
                                 mv ? mv + (size_t)ib * V * 4 : NULL, mvc ? mvc + (size_t)ib * V * C : NULL,
                                 cv ? cv + (size_t)ib * V * 4 : NULL, grad_background + (size_t)ib * P * C,
                                 (debug_thingy && c_begin == 0) ? debug_thingy + (size_t)ib * P * 3 : NULL, bary_w, index_f,
                                 pix_g, gpix_g, verts, ib, B, H, W, G, C, c_begin, flags);
            free(of);
        }
        if (seq) /* `sum([result.grad_vertices ...])` in float32, dirt/rasterise_ops.py:163 */
            for (size_t n = 0; n < nv; ++n) gv[n] = (double)(float)(gv[n] + gv_group[n]);
        c_begin += G;
    }
    for (size_t n = 0; n < nv; ++n) grad_vertices[n] = (float)gv[n];
    for (size_t n = 0; n < nvc; ++n) grad_vertex_colors[n] = (float)gvc[n];
    if (mv) for (size_t n = 0; n < nv; ++n) mass_vertices[n] = (float)mv[n];
    if (mvc) for (size_t n = 0; n < nvc; ++n) mass_vertex_colors[n] = (float)mvc[n];
    if (cv) for (size_t n = 0; n < nv; ++n) cond_vertices[n] = (float)cv[n];
    if (seq) free(gv_group);
    free(gv); free(gvc); free(mv); free(mvc); free(cv); free(bary_w); free(index_f); free(pix_g); free(gpix_g);
    return 0;
}

int dirt_oracle_backward(const float *vertices, const int32_t *faces, const float *pixels, const float *grad_pixels,
                         float *grad_background, float *grad_vertices, float *grad_vertex_colors, float *debug_thingy,
                         int B, int V, int F, int H, int W, int C, unsigned flags)
{
    return dirt_oracle_backward_ex(vertices, faces, pixels, grad_pixels, grad_background, grad_vertices, grad_vertex_colors,
                                   debug_thingy, NULL, NULL, NULL, B, V, F, H, W, C, flags);
}

/*
 * Test helper: the GL draw of ONE scene in the framebuffer's OWN frame (csrc/rasterise_egl.cpp:362-380 with the
 * pass-through shaders csrc/shaders.cpp:16-43): RGBA32F atlas [atlas_h][atlas_w][4], row 0 = the bottom row, viewport
 * (frame_x, frame_y, W, H); a depth buffer cleared to 1.0 for the call; faces in index order, GL_LESS; the fragment
 * shader writes vec4(colour, 1) with a 1-channel attribute's missing components 0 (csrc/rasterise_egl.cpp:208).
 * Nothing here knows about tensor rows: every sample is addressed by its window coordinates (i + 0.5, j + 0.5), and no
 * bounding box is used.  oracle/ref.py puts the reference's own upload_background / download_pixels around it
 * (csrc/rasterise_egl.cu, compiled for the host) and tests/test_oracle_ref.py requires the result to equal
 * dirt_oracle_forward bit for bit: that pins the vertical flip and the atlas tiling of the forward restatement to the
 * reference's code.  C in {1, 3}.
 */
int dirt_oracle_draw_gl(const float *vertices, const float *vertex_colors, const int32_t *faces, float *atlas,
                        int V, int F, int H, int W, int C, int atlas_w, int atlas_h, int frame_x, int frame_y)
{
    if (!check_dims(1, V, F, H, W, C) || (C != 1 && C != 3)) return -1;
    if (frame_x < 0 || frame_y < 0 || frame_x + W > atlas_w || frame_y + H > atlas_h) return -2;
    OFace *of = setup_scene(vertices, V, faces, F, H, W);
    uint32_t *zbuf = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)H * W);
    for (size_t n = 0; n < (size_t)H * W; ++n) zbuf[n] = 0x00FFFFFFu; /* glClear(GL_DEPTH_BUFFER_BIT) */
    for (int f = 0; f < F; ++f) {
        const OFace *o = &of[f];
        if (!o->valid) continue;
        const float *c0 = vertex_colors + (size_t)o->vid[0] * C, *c1 = vertex_colors + (size_t)o->vid[1] * C,
                    *c2 = vertex_colors + (size_t)o->vid[2] * C;
        for (int j = 0; j < H; ++j)       /* GL window row, bottom first */
            for (int i = 0; i < W; ++i) { /* GL window column */
                double E[3];
                uint32_t z24;
                double px = (double)i + 0.5, py = (double)j + 0.5;
                if (!sample_inside(o, px, py, E)) continue;
                if (!sample_depth(o, px, py, &z24)) continue;
                uint32_t *zb = &zbuf[(size_t)j * W + i];
                if (!(z24 < *zb)) continue;
                *zb = z24;
                float b[3], cw;
                sample_bary(o, E, b, &cw);
                float *texel = atlas + (((size_t)(frame_y + j)) * atlas_w + (frame_x + i)) * 4;
                for (int c = 0; c < 3; ++c) texel[c] = c < C ? fmaf(b[2], c2[c], fmaf(b[1], c1[c], b[0] * c0[c])) : 0.f;
                texel[3] = 1.f;
            }
    }
    free(zbuf);
    free(of);
    return 0;
}

/*
 * Test helper: visibility of one scene as arrays in tensor orientation.
 *   face_id [H,W] int32 (-1 none); bary [H,W,3]; clip_w [H,W] (+inf none); z24 is not exported.
 */
int dirt_oracle_visibility(const float *vertices, const int32_t *faces, int32_t *face_id, float *bary, float *clip_w,
                           int V, int F, int H, int W)
{
    if (!check_dims(1, V, F, H, W, 1)) return -1;
    OFace *of = setup_scene(vertices, V, faces, F, H, W);
    scene_visibility(of, F, H, W, face_id);
    for (int r = 0; r < H; ++r)
        for (int i = 0; i < W; ++i) {
            size_t p = (size_t)r * W + i;
            int32_t f = face_id[p];
            if (f < 0) {
                if (bary) bary[p * 3] = bary[p * 3 + 1] = bary[p * 3 + 2] = -1.f;
                if (clip_w) clip_w[p] = INFINITY;
            } else {
                double E[3];
                float b[3], cw;
                sample_inside(&of[f], (double)i + 0.5, (double)(H - 1 - r) + 0.5, E);
                sample_bary(&of[f], E, b, &cw);
                if (bary) { bary[p * 3] = b[0]; bary[p * 3 + 1] = b[1]; bary[p * 3 + 2] = b[2]; }
                if (clip_w) clip_w[p] = cw;
            }
        }
    free(of);
    return 0;
}

int dirt_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void dirt_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
