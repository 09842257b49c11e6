// dirt_grad.hip -- gradient assembly kernel for gfx950.
//
// Replaces assemble_grads / launch_grad_assembly (csrc/rasterise_grad_egl.cu:93-278) and, by
// evaluating every channel group of dirt/rasterise_ops.py:145-165 inside one launch, the N
// per-group RasteriseGrad ops (and N GL re-draws) the reference issues for C not in {1,3}.
//
// Inputs: the visibility buffer (front-most face per pixel, written by raster_kernel<1>) instead of
// the reference's two RGBA32F surfaces; barycentrics and clip-w of a pixel are recomputed from the
// face's set-up record exactly as the forward pass computes them.
//
// The reference issues up to 3C+9 global float atomics per covered pixel (:140,228-230).  Here one
// 256-thread workgroup owns a 32x32 tile (wave w = the 8-row band w, four 8x8 blocks, one pixel per
// lane, as in raster_kernel) and accumulates per-face partial sums in LDS:
//   * the faces that receive gradient in the tile get a slot in a small LDS hash table (LDS CAS);
//   * each value is first summed over the 4 lanes of a 4x1 pixel quad with two DPP quad_perm adds
//     when the quad targets one face (the common case), then added with ds_add_f32 to one of
//     COPIES replicas of the slot's accumulator (replicas spread same-address lanes over banks);
//   * at the end the replicas are summed and ONE global atomic per (face, vertex, component) is
//     issued for the whole tile.  Faces that do not fit the table fall back to direct atomics.
// Variable names in the per-pixel arithmetic follow the CUDA source.
#include "dirt_device.h"
#include "dirt_launch.h"
#include "../../include/dirt_hip.h"

namespace dirt {

constexpr int GT = 32;                 // tile edge
constexpr int VW = GT + 2;             // visibility tile with a 1-pixel halo
constexpr int COPIES = 4;              // accumulator replicas per (slot, value)
constexpr int MAX_SLOTS = 128;         // hash table capacity (LDS)
constexpr int CH = 4;                  // colour channels accumulated per pass
constexpr int NVAL = 9 + 3 * CH;       // 9 position values (3 vertices x {x,y,w}) + 3 x CH colour values

struct Frag {
    float b[3];
    float w;
    int32_t vid[3];
};

// (barycentric, clip_w, indices) of face `f` at pixel (x, r): what the backward fragment shader
// writes (csrc/shaders.cpp:64-77).
__device__ inline Frag frag_eval(const FaceRec* __restrict__ recs, int f, int x, int r, int H)
{
    const FaceRec* __restrict__ rec = recs + f;
    double cf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k] = rec->coef[k];
    double Fk[3];
    edge_eval(cf, (double)x + 0.5, (double)(H - 1 - r) + 0.5, Fk);
    Frag o;
    bary_eval(Fk, rec->flags, rec->inv_det, o.b, o.w);
    o.vid[0] = rec->vid[0]; o.vid[1] = rec->vid[1]; o.vid[2] = rec->vid[2];
    return o;
}

__device__ __forceinline__ float quad_sum(float v)
{
    // v_add_f32 with DPP quad_perm: lanes 4q..4q+3 all end up with the sum of the quad
    float t = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // [1,0,3,2]
    v = v + t;
    t = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));        // [2,3,0,1]
    return v + t;
}

// Open-addressing insert of `face` into the tile's slot table; returns the slot or -1 when full.
__device__ inline int slot_insert(int32_t* keys, int nslots, int face)
{
    uint32_t h = ((uint32_t)face * 2654435761u) % (uint32_t)nslots;
    for (int probe = 0; probe < nslots; ++probe) {
        const int32_t cur = reinterpret_cast<volatile int32_t*>(keys)[h];
        if (cur == face) return (int)h;
        if (cur == -1) {
            const int32_t prev = atomicCAS(&keys[h], -1, face);
            if (prev == -1 || prev == face) return (int)h;
        }
        h = (h + 1 == (uint32_t)nslots) ? 0u : h + 1;
    }
    return -1;
}

// Accumulation context of one lane for one target face.
struct Target {
    int slot;       // LDS slot, -1 = none (no contribution), -2 = table full: global atomics
    bool uniform;   // the lane's 4x1 quad targets one slot
    bool leader;    // first lane of the quad
    int copy;       // accumulator replica
};

__device__ __forceinline__ Target make_target(int slot, int lane)
{
    Target t;
    t.slot = slot;
    const int s0 = __builtin_amdgcn_mov_dpp(slot, 0x00, 0xF, 0xF, true);  // quad_perm [0,0,0,0]
    const unsigned long long m = __ballot(slot == s0);
    t.uniform = ((m >> (lane & ~3)) & 0xFull) == 0xFull && slot != -2;
    t.leader = (lane & 3) == 0;
    t.copy = (lane >> 2) & (COPIES - 1);
    return t;
}

// Add `v` (0 for lanes that contribute nothing) to value `idx` of the lane's target slot.
// Must be called by all 64 lanes of the wave (DPP).
__device__ __forceinline__ void lds_accumulate(float* acc, const Target& t, int idx, float v)
{
    const float q = quad_sum(v);
    const float out = t.uniform ? q : v;
    const bool active = t.slot >= 0 && (t.uniform ? t.leader : true);
    if (active && out != 0.f) atomicAdd(&acc[(t.slot * NVAL + idx) * COPIES + t.copy], out);
}

__global__ __launch_bounds__(256) void grad_kernel(GradParams p)
{
    extern __shared__ __align__(16) float s_acc[];  // [nslots][NVAL][COPIES]
    __shared__ int32_t s_vis[VW * VW];
    __shared__ int32_t s_key[MAX_SLOTS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int iib = blockIdx.y;
    const int tile = blockIdx.x;
    const int tx0 = (tile % p.tiles_x) * GT;
    const int tr0 = (tile / p.tiles_x) * GT;
    const int H = p.H, W = p.W, C = p.C;
    const int nslots = p.nslots;
    const size_t frame = (size_t)H * W;
    const size_t total_pix = (size_t)p.B * frame;

    const FaceRec* __restrict__ recs = p.recs + (size_t)iib * p.F;
    const int32_t* __restrict__ vis = p.vis + (size_t)iib * frame;
    const float* __restrict__ vertices = p.vertices + (size_t)iib * p.V * 4;
    float* __restrict__ grad_vertices = p.grad_vertices + (size_t)iib * p.V * 4;
    float* __restrict__ grad_vertex_colors = p.grad_vertex_colors + (size_t)iib * p.V * C;

    // ---- init: slot table, accumulators, visibility tile with halo (clamped reads; the halo is
    //      only consulted for interior pixels, whose neighbours are inside the frame) ----
    for (int i = tid; i < nslots; i += 256) s_key[i] = -1;
    for (int i = tid; i < nslots * NVAL * COPIES; i += 256) s_acc[i] = 0.f;
    for (int i = tid; i < VW * VW; i += 256) {
        const int vy = i / VW, vx = i - vy * VW;
        const int rr = min(max(tr0 + vy - 1, 0), H - 1), xx = min(max(tx0 + vx - 1, 0), W - 1);
        s_vis[i] = vis[(size_t)rr * W + xx];
    }
    __syncthreads();

    const int lx = lane & 7, ly = lane >> 3;
    const int y_in_frame = tr0 + wave * 8 + ly;  // tensor row (top row first)
    const bool q1_intended = (p.flags & DIRT_FLAG_Q1_INTENDED) != 0;
    const float width_f = (float)W, height_f = (float)H;

    for (int c0 = 0; c0 < C; c0 += CH) {
        const int nch = min(CH, C - c0);
        // groups whose first channel lies in this colour pass are evaluated in this pass
#pragma unroll 1
        for (int blk = 0; blk < 4; ++blk) {
            const int x_in_frame = tx0 + blk * 8 + lx;
            const bool inside = x_in_frame < W && y_in_frame < H;
            const int xs = min(x_in_frame, W - 1), ys = min(y_in_frame, H - 1);  // safe addresses for idle lanes
            const size_t pix = (size_t)iib * frame + (size_t)ys * W + xs;
            const int vpos = (wave * 8 + ly + 1) * VW + (blk * 8 + lx + 1);
            const float* __restrict__ g_here = p.grad_pixels + pix * C;

            const int32_t face_here = inside ? s_vis[vpos] : -1;
            Frag here;
            if (face_here >= 0) {
                here = frag_eval(recs, face_here, xs, ys, H);
            } else {  // clear values, csrc/rasterise_grad_egl.cpp:442-445
                here.b[0] = here.b[1] = here.b[2] = -1.f;
                here.w = INFINITY;
                here.vid[0] = here.vid[1] = here.vid[2] = -1;
            }
            int slot_here = -1;
            if (face_here >= 0) {
                slot_here = slot_insert(s_key, nslots, face_here);
                if (slot_here < 0) slot_here = -2;
            }
            const Target t_here = make_target(slot_here, lane);

            // ---- colour / background gradients, csrc/rasterise_grad_egl.cu:135-148 ----
            float gch[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) gch[c] = (c < nch) ? g_here[c0 + c] : 0.f;
            if (inside) {
                float* gb = p.grad_background + pix * C + c0;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c < nch) gb[c] = face_here >= 0 ? 0.f : gch[c];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float color_grad = (face_here >= 0 && c < nch) ? gch[c] * here.b[k] : 0.f;
                    lds_accumulate(s_acc, t_here, 9 + k * CH + c, color_grad);
                    if (slot_here == -2 && c < nch)
                        atomicAdd(&grad_vertex_colors[(size_t)here.vid[k] * C + c0 + c], color_grad);
                }
            }

            // ---- channel groups starting in [c0, c0+nch): Scharr, dilation, position gradients ----
            const bool interior = inside && x_in_frame > 0 && y_in_frame > 0 && x_in_frame < W - 1 && y_in_frame < H - 1;
            for (int c_begin = 0; c_begin < C;) {
                const int G = (c_begin + 3 <= C) ? 3 : 1;  // dirt/rasterise_ops.py:148-152
                if (c_begin < c0 || c_begin >= c0 + CH) { c_begin += G; continue; }
                const bool alias = (G == 1) && !q1_intended;  // quirk Q1

                // 3x3 neighbourhood of `pixels`, edge clamped: at(), csrc/rasterise_grad_egl.cu:113-124
                float sx[3], sy[3];
                {
                    float t[3][3][3];
#pragma unroll
                    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
                        for (int ox = -1; ox <= 1; ++ox) {
                            const int cx = max(0, min(W - 1, xs + ox));
                            const int cy = max(0, min(H - 1, ys - oy));
                            const size_t n = (size_t)iib * frame + (size_t)cy * W + cx;
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) {
                                float v = 0.f;
                                if (G == 3) {
                                    v = p.pixels[n * C + c_begin + ch];
                                } else if (ch == 0) {
                                    v = p.pixels[n * C + c_begin];
                                } else if (alias) {
                                    size_t m = n + ch;
                                    if (m > total_pix - 1) m = total_pix - 1;
                                    v = p.pixels[m * C + c_begin];
                                }
                                t[oy + 1][ox + 1][ch] = v;
                            }
                        }
#define AT(ox, oy, ch) t[(oy) + 1][(ox) + 1][ch]
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {  // :126-127
                        float d1 = ((AT(-1, -1, ch) + AT(-1, +1, ch)) - AT(+1, -1, ch)) - AT(+1, +1, ch);
                        float d2 = AT(-1, 0, ch) - AT(+1, 0, ch);
                        float m1 = d1 * (3.f / 32.f), m2 = d2 * (10.f / 32.f);
                        sx[ch] = m1 + m2;
                        d1 = ((AT(-1, -1, ch) + AT(+1, -1, ch)) - AT(-1, +1, ch)) - AT(+1, +1, ch);
                        d2 = AT(0, -1, ch) - AT(0, +1, ch);
                        m1 = d1 * (3.f / 32.f); m2 = d2 * (10.f / 32.f);
                        sy[ch] = m1 + m2;
                    }
#undef AT
                }

                Frag cur = here;
                int face_cur = face_here;
                bool dilated = false;
                if (interior) {  // :155-194
                    float l1x, l1y;
                    if (G == 1 && q1_intended) {
                        l1x = fabsf(sx[0]); l1y = fabsf(sy[0]);
                    } else {
                        l1x = (fabsf(sx[0]) + fabsf(sx[1])) + fabsf(sx[2]);
                        l1y = (fabsf(sy[0]) + fabsf(sy[1])) + fabsf(sy[2]);
                    }
                    int off_x = l1x > l1y ? 1 : 0, off_y = l1x > l1y ? 0 : 1;
                    if (((x_in_frame + y_in_frame) & 1) == 1) { off_x = -off_x; off_y = -off_y; }
                    for (int attempt = 0; attempt < 2 && !dilated; ++attempt) {
                        const int ox = attempt == 0 ? off_x : -off_x, oy = attempt == 0 ? off_y : -off_y;
                        // the reference offsets in GL buffer orientation (y up): tensor row = y_in_frame - oy
                        const int32_t face_off = s_vis[vpos - oy * VW + ox];
                        if (face_off >= 0 && face_off != face_cur) {
                            const Frag off = frag_eval(recs, face_off, x_in_frame + ox, y_in_frame - oy, H);
                            const bool differs =
                                off.vid[0] != cur.vid[0] || off.vid[1] != cur.vid[1] || off.vid[2] != cur.vid[2];
                            if (differs && cur.w > off.w) {  // :165
                                cur = off;
                                face_cur = face_off;
                                dilated = true;
                            }
                        }
                    }
                }

                if (p.debug_thingy && c_begin == 0 && inside) {  // :150-151,172
                    float* dbg = p.debug_thingy + pix * 3;
                    dbg[0] = dilated ? 1.e-2f : 0.f;
                    for (int ch = 1; ch <= 2; ++ch) {
                        size_t m = pix * G + ch;
                        if (m > total_pix * G - 1) m = total_pix * G - 1;
                        dbg[ch] = p.grad_pixels[(m / G) * C + c_begin + (m % G)];  // element m of the [B,H,W,G] slice
                    }
                }

                // position gradients, :196-232 (zero contribution where nothing is covered)
                const bool covered = inside && face_cur >= 0;
                int slot_cur = -1;
                if (covered) {
                    slot_cur = (face_cur == face_here) ? slot_here : slot_insert(s_key, nslots, face_cur);
                    if (slot_cur == -1) slot_cur = -2;
                }
                const Target t_cur = make_target(slot_cur, lane);
                float dL_dx = 0.f, dL_dy = 0.f;
                for (int channel = 0; channel < G; ++channel) {
                    const float dL_dchannel = g_here[c_begin + channel];
                    float m = dL_dchannel * sx[channel];
                    dL_dx = dL_dx + m;
                    m = dL_dchannel * sy[channel];
                    dL_dy = dL_dy + m;
                }
                float clip_x = 0.f, clip_y = 0.f;
                if (covered) {
                    for (int k = 0; k < 3; ++k) {
                        const float2 vxy = *reinterpret_cast<const float2*>(vertices + (size_t)cur.vid[k] * 4);
                        float m = cur.b[k] * vxy.x;
                        clip_x = clip_x + m;
                        m = cur.b[k] * vxy.y;
                        clip_y = clip_y + m;
                    }
                }
                const float clip_w = cur.w;
                const float d_xview_by_xclip = (.5f * width_f) / clip_w;
                const float d_yview_by_yclip = (.5f * height_f) / clip_w;
                const float ww = clip_w * clip_w;
                const float d_xview_by_wclip = ((-.5f * width_f) * clip_x) / ww;
                const float d_yview_by_wclip = ((-.5f * height_f) * clip_y) / ww;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float dLx_b = dL_dx * cur.b[k];
                    const float dLy_b = dL_dy * cur.b[k];
                    const float gx = covered ? dLx_b * d_xview_by_xclip : 0.f;
                    const float gy = covered ? dLy_b * d_yview_by_yclip : 0.f;
                    const float gw1 = dLx_b * d_xview_by_wclip, gw2 = dLy_b * d_yview_by_wclip;
                    const float gw = covered ? gw1 + gw2 : 0.f;
                    lds_accumulate(s_acc, t_cur, k * 3 + 0, gx);
                    lds_accumulate(s_acc, t_cur, k * 3 + 1, gy);
                    lds_accumulate(s_acc, t_cur, k * 3 + 2, gw);
                    if (slot_cur == -2) {
                        float* gv = grad_vertices + (size_t)cur.vid[k] * 4;
                        atomicAdd(gv + 0, gx);
                        atomicAdd(gv + 1, gy);
                        atomicAdd(gv + 3, gw);
                    }
                }
                c_begin += G;
            }
        }
        __syncthreads();

        // ---- flush this pass: colour values of channels [c0, c0+nch); position values after the last pass ----
        const bool last = c0 + CH >= C;
        for (int e = tid; e < nslots * NVAL; e += 256) {
            const int slot = e / NVAL, v = e - slot * NVAL;
            const int32_t face = s_key[slot];
            if (face < 0) continue;
            if (v < 9 && !last) continue;
            float* a = &s_acc[(size_t)e * COPIES];
            float sum = 0.f;
#pragma unroll
            for (int cp = 0; cp < COPIES; ++cp) sum += a[cp];
            if (v >= 9) {
#pragma unroll
                for (int cp = 0; cp < COPIES; ++cp) a[cp] = 0.f;
            }
            if (sum == 0.f) continue;
            if (v < 9) {
                const int k = v / 3, comp = v - k * 3;
                atomicAdd(&grad_vertices[(size_t)recs[face].vid[k] * 4 + (comp == 2 ? 3 : comp)], sum);
            } else {
                const int k = (v - 9) / CH, c = (v - 9) - k * CH;
                if (c < nch) atomicAdd(&grad_vertex_colors[(size_t)recs[face].vid[k] * C + c0 + c], sum);
            }
        }
        __syncthreads();
    }
}

hipError_t launch_grad(const GradParams& p_in, hipStream_t stream)
{
    if (p_in.B == 0) return hipSuccess;
    GradParams p = p_in;
    p.tiles_x = (p.W + GT - 1) / GT;
    p.tiles_y = (p.H + GT - 1) / GT;
    p.nslots = MAX_SLOTS;
    const size_t shmem = (size_t)p.nslots * NVAL * COPIES * sizeof(float);
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)p.B);
    hipLaunchKernelGGL(grad_kernel, grid, dim3(256), shmem, stream, p);
    return hipGetLastError();
}

}  // namespace dirt
