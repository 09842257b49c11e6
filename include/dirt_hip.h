/*
 * dirt_hip.h -- C ABI of libdirt_hip.so, the MI355X (gfx950) replacement for the two TensorFlow
 * custom ops of pmh47/dirt.  Plain C, raw device pointers and sizes; no torch / TF / HIP types.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository root).  The reference binds its ops through the TF op registry
 * (dirt/rasterise_ops.py:5-10: tf.load_op_library('librasterise.so')); a maintainer binds this
 * library with ctypes -- see INTEGRATION.md for the stub.
 *
 * Conventions shared by every call
 *   - all tensors are dense, row-major, float32 except `faces` (int32); device pointers, 16-byte
 *     aligned (torch / hipMalloc allocations are);
 *   - background / pixels / grad_* images are [B,H,W,C], top row first (README.md:183);
 *     vertices [B,V,4] are OpenGL clip-space (x,y,z,w); vertex_colors [B,V,C]; faces [B,F,3];
 *   - `workspace` is caller-owned device scratch of at least dirt_workspace_bytes(...) bytes,
 *     16-byte aligned; the library keeps no DEVICE state between calls.  Calls on different workspaces are independent
 *     (any threads, any streams).  ONE workspace is a single-stream object: the calls that share it -- a forward with
 *     DIRT_FLAG_KEEP_STATE and the backward calls with DIRT_FLAG_REUSE_STATE that consume it -- must be enqueued on one
 *     stream (or be ordered by the caller's own events): nothing in the library serialises two streams on one workspace;
 *   - the one piece of HOST state: per workspace address, which gradient buffers the last forward left cleared (see
 *     DIRT_FLAG_OUTPUTS_CLEARED / DIRT_FLAG_DENSE_FROM_STATE).  It is consulted at ENQUEUE time and knows pointers, not
 *     contents -- so the "skip the clearing launch" paths additionally require that (i) forward and backward are captured
 *     TOGETHER when a HIP graph is recorded (a backward captured alone replays without its forward's clear and would
 *     accumulate), (ii) the cleared tensors stay allocated between the two calls (a caching allocator handing the same
 *     addresses to other tensors in between is indistinguishable), and (iii) nothing writes to them in between.  The
 *     Python wrapper keeps the tensors on the state object, which guarantees (ii) and (iii); a caller that cannot, omits
 *     the flags and pays one clearing launch.  Any call that rebuilds a workspace (a forward, a stateless backward,
 *     dirt_rasterise_visibility) forgets what was cleared in it;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); work is enqueued
 *     asynchronously on it and the call returns without synchronising;
 *   - the current HIP device must be the one that owns the pointers;
 *   - return value 0 = success, <0 = DIRT_E_* below; dirt_last_error() describes the last
 *     failure on the calling thread.  The library never aborts the process.
 */
#ifndef DIRT_HIP_H
#define DIRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIRT_ABI_VERSION 4

/* error codes */
#define DIRT_OK 0
#define DIRT_E_INVALID_ARGUMENT (-1) /* bad sizes / null pointers: the OP_REQUIRES checks of
                                        csrc/rasterise_egl.cpp:301-316, csrc/rasterise_grad_egl.cpp:349-377
                                        and the CHECKs of csrc/hwc.h:27-28 */
#define DIRT_E_TOO_MANY_VERTICES (-2) /* V > 2^24: csrc/rasterise_grad_egl.cpp:399-405 */
#define DIRT_E_WORKSPACE (-3)         /* workspace NULL / too small / misaligned */
#define DIRT_E_HIP (-4)               /* a HIP runtime call failed (the reference LOG(FATAL)s) */

/* flags (bitwise or) */
#define DIRT_FLAG_Q1_INTENDED 1u /* backward: for 1-channel groups take the Scharr L1 over the one
                                    real channel instead of reproducing the reference's out-of-range
                                    channel reads (csrc/rasterise_grad_egl.cu:119-123,185; SURVEY.md
                                    App. A.3 quirk Q1).  Default (0) reproduces the reference. */

#define DIRT_FLAG_KEEP_STATE 2u  /* forward: also leave, in `workspace`, the state the backward pass needs
                                    (per-face set-up records and the per-pixel front-most face). */
#define DIRT_FLAG_REUSE_STATE 4u /* backward: `workspace` is the very buffer a forward call with
                                    DIRT_FLAG_KEEP_STATE filled for the same vertices, faces and sizes, and
                                    nothing has written to it since: skip triangle set-up and the visibility
                                    render.  The reference re-renders in RasteriseGrad and notes the
                                    alternative itself (csrc/rasterise_grad_egl.cpp:446-447: "It may or may
                                    not more efficient to render these in the forward pass and return them
                                    in separate outputs").  The state is caller-owned memory; the library
                                    stays stateless.  Results are identical with or without the flag. */
#define DIRT_FLAG_DENSE_FROM_STATE 8u /* backward, with DIRT_FLAG_REUSE_STATE and dense caller tensors for grad_vertices /
                                    grad_vertex_colors: sum the vertex gradients in the accumulators the forward pass
                                    cleared inside the state (one interleaved row per vertex: what the float atomics are
                                    fastest on) and copy them out into the caller's dense [B,V,4] / [B,V,C] tensors with one
                                    more launch -- what csrc/rasterise_grad_egl.cpp:381-391 allocates as the op's outputs.
                                    Re-entrant since ABI 3: the library knows (host-side, per workspace address) whether the
                                    accumulators are still as the forward left them and clears them itself when they are
                                    not, so a second backward call over one forward returns that call's gradients, not the
                                    sum of both.  Results agree to summation order. */
#define DIRT_FLAG_OUTPUTS_CLEARED 0x10u /* backward, with DIRT_FLAG_REUSE_STATE: grad_vertices / grad_vertex_colors are the
                                    very tensors dirt_rasterise_forward_train was given with this workspace, and nothing
                                    has written to them since: skip the launch that clears them (the reference's
                                    cudaMemsetAsync, csrc/rasterise_grad_egl.cu:244-250, happened inside the forward's
                                    launch).  Checked against the library's host-side record of that forward: if the
                                    record does not name these pointers, or a backward call has consumed it already, the
                                    outputs are cleared as without the flag -- never added onto. */

#define DIRT_FLAG_TILES_LARGE 0x200u /* pin the forward / visibility kernels' tile shape instead of letting the library
                                       choose it from the frame size and the face density: 32x32 pixel tiles ... */
#define DIRT_FLAG_TILES_SMALL 0x400u /* ... or 16x16.  Results do not depend on the shape (pixels and visibility bit for
                                       bit); for tests.  (The gradient kernel's shapes: DIRT_FLAG_GRAD_*.)  BOTH bits:
                                       32x32 tiles rendered by EIGHT half-size waves each, the shape the library takes by
                                       itself for launches of at most 2048 such tiles (DIRT_FLAG_TILES_LARGE alone pins four
                                       waves per tile); where that shape does not exist -- meshes of more than 16 384 faces,
                                       channel counts other than 1, 3, 4, the visibility pass -- the pair means 32x32. */
#define DIRT_FLAG_GRAD_ROWS 0x1000u  /* pin the gradient kernel's face-loop shape instead of letting the library choose by
                                        frame size: every 8x8 block of a wave walks its own faces ... */
#define DIRT_FLAG_GRAD_PAIRS 0x2000u /* ... or pairs of blocks share a face (fewer float atomics).  Results agree to
                                        summation order (how the parity tests cover both) */
#define DIRT_FLAG_GRAD_SMALL 0x4000u /* ... or the small-frame gradient kernel: one pixel per lane on 16x16 tiles
                                        (channel counts 1, 3, 4; chosen by the library when tiles x scenes of the call --
                                        32x32 tiles -- are at most 256 and the mesh has at least 96 faces per tile, counted
                                        over ALL faces of a scene, culled and off-screen ones included).  Same results to
                                        summation order */
#define DIRT_FLAG_GRAD_PX2 0x8000u   /* ... or the two-pixels-per-lane gradient kernel on 32x16 tiles (channel counts 1, 3, 4:
                                        twice the waves at half the instruction chain each; chosen by the library for frames
                                        of more than 256 and fewer than 1024 32x32 tiles -- 3 channels: of more than 256 --
                                        where it measured faster).  Same results to summation order */
#define DIRT_FLAG_GRAD_PX4 0x10000u  /* ... or the four-pixels-per-lane kernel (rows or pairs by the library's own rule) where
                                        the library would choose the two-pixels-per-lane one */
#define DIRT_FLAG_GRAD_STREAM 0x20000u /* ... or the streaming four-pixels-per-lane kernel (4 channels, W and H multiples of 32, no
                                        debug_thingy: wave-private tiles filled by LDS-DMA while the previous slice is worked on;
                                        never the library's own choice -- it measured slower, profiles/EXPERIMENTS.md round 6);
                                        ignored where it does not apply.  Same results to summation order */
#define DIRT_FLAG_SHARED_FACES 0x800u /* `faces` is one [F,3] topology shared by all B scenes instead of [B,F,3] (the
                                        TODO of csrc/rasterise_egl.cpp:314).  Same flag on the
                                        forward, visibility and backward calls of one scene batch. */
#define DIRT_FLAG_PROFILE 0x100u /* record a HIP-event pair around every kernel this call launches (on the
                                    call's stream); read the totals with dirt_profile_read.  Replaces the
                                    reference's compile-time TIME_SECTIONS wall-clock prints
                                    (csrc/rasterise_egl.cpp:398-405, csrc/rasterise_grad_egl.cpp:479-483). */

/* Limits of this implementation (the reference's limit is the GL max texture size of its atlas). */
#define DIRT_MAX_DIM 16384

/* Returns DIRT_ABI_VERSION of the loaded library. */
int dirt_abi_version(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char *dirt_last_error(void);

/*
 * Scratch bytes needed by dirt_rasterise_forward / dirt_rasterise_backward for these sizes
 * (replaces the grow-only GL buffers / framebuffer atlas the reference caches per thread,
 * csrc/rasterise_egl.cpp:325-346, csrc/rasterise_grad_egl.cpp:407-425).  Returns 0 on invalid sizes.
 */
size_t dirt_workspace_bytes(int B, int V, int F, int H, int W, int C);

/*
 * Forward.  Replaces the `Rasterise` op: REGISTER_OP csrc/rasterise_egl.cpp:32-51 and
 * RasteriseOpGpu::Compute csrc/rasterise_egl.cpp:276-407, as called from
 * dirt/rasterise_ops.py:81-85,98-105.  Unlike the reference op, C may be any value >= 1: the
 * result equals the reference's channel-grouped evaluation (dirt/rasterise_ops.py:86-108).
 *   in : background [B,H,W,C], vertices [B,V,4], vertex_colors [B,V,C], faces [B,F,3]
 *   out: pixels [B,H,W,C]
 * B, V or F may be 0 (pixels = background when there is nothing to draw).
 */
int dirt_rasterise_forward(const float *background, const float *vertices, const float *vertex_colors,
                           const int32_t *faces, float *pixels, int B, int V, int F, int H, int W, int C,
                           void *workspace, size_t workspace_bytes, unsigned flags, void *stream);

/*
 * Forward of a training step: dirt_rasterise_forward with DIRT_FLAG_KEEP_STATE that, in the same launch, also clears the
 * DENSE gradient tensors the backward call will be given -- the RasteriseGrad op's outputs grad_vertices [B,V,4] and
 * grad_vertex_colors [B,V,C] (csrc/rasterise_grad_egl.cpp:381-391), which the reference clears with cudaMemsetAsync at the
 * start of its gradient op (csrc/rasterise_grad_egl.cu:244-250).  A following dirt_rasterise_backward with
 * DIRT_FLAG_REUSE_STATE | DIRT_FLAG_OUTPUTS_CLEARED on the same workspace and the same two tensors then is ONE launch that
 * adds straight into the op's contract outputs: no clearing launch, no copy out of the state.  (The state's own
 * interleaved accumulators are NOT cleared by this call.)
 */
int dirt_rasterise_forward_train(const float *background, const float *vertices, const float *vertex_colors,
                                 const int32_t *faces, float *pixels, float *grad_vertices, float *grad_vertex_colors,
                                 int B, int V, int F, int H, int W, int C, void *workspace, size_t workspace_bytes,
                                 unsigned flags, void *stream);

/*
 * Backward.  Replaces the `RasteriseGrad` op: REGISTER_OP csrc/rasterise_grad_egl.cpp:33-53,
 * Compute csrc/rasterise_grad_egl.cpp:324-485 and launch_grad_assembly / assemble_grads
 * csrc/rasterise_grad_egl.cu:93-278, as called from dirt/rasterise_ops.py:113-118,154-160; for C
 * not in {1,3} it reproduces _rasterise_grad_multichannel (dirt/rasterise_ops.py:132-177):
 * channel groups of 3 then 1s, grad_vertices summed over groups, the others concatenated.
 *   in : vertices [B,V,4], faces [B,F,3], pixels [B,H,W,C] (the forward output), grad_pixels [B,H,W,C]
 *   out: grad_background [B,H,W,C], grad_vertices [B,V,4], grad_vertex_colors [B,V,C],
 *        debug_thingy [B,H,W,3] or NULL (the reference's 4th, diagnostic output, of the first
 *        channel group; csrc/rasterise_grad_egl.cu:150-151,172)
 * All outputs are fully written (zero where the reference's memsets leave zero,
 * csrc/rasterise_grad_egl.cu:244-250).  Float atomics make grad_vertices / grad_vertex_colors
 * order-dependent in the last bits, as in the reference.
 */
int dirt_rasterise_backward(const float *vertices, const int32_t *faces, const float *pixels,
                            const float *grad_pixels, float *grad_background, float *grad_vertices,
                            float *grad_vertex_colors, float *debug_thingy, int B, int V, int F, int H, int W,
                            int C, void *workspace, size_t workspace_bytes, unsigned flags, void *stream);

/*
 * Visibility only (no reference counterpart as an op; it is the render pass of
 * csrc/rasterise_grad_egl.cpp:432-456 exposed for tests and for the deferred-shading row):
 *   out: face_id [B,H,W] int32, index of the front-most face or -1.
 */
int dirt_rasterise_visibility(const float *vertices, const int32_t *faces, int32_t *face_id, int B, int V,
                              int F, int H, int W, void *workspace, size_t workspace_bytes, unsigned flags,
                              void *stream);

/*
 * With DIRT_FLAG_KEEP_STATE the forward pass also clears the gradient accumulators inside the workspace.  This
 * returns their addresses and ROW STRIDES (in floats): a backward call with DIRT_FLAG_REUSE_STATE whose grad_vertices /
 * grad_vertex_colors ARE these pointers accumulates straight into them and needs no clearing launch (the reference
 * clears its outputs with four cudaMemsetAsync, csrc/rasterise_grad_egl.cu:244-250).  Any other output pointers work
 * too; they are dense ([B,V,4] and [B,V,C]) and are cleared first.
 *   The two accumulators are INTERLEAVED: one row of S = 4 + C (rounded up to a multiple of 4) floats per vertex,
 * {x, y, z, w, c0 .. cC-1}, so both strides are S and *grad_vertex_colors == *grad_vertices + 4: a face adds all of a
 * vertex's values to one row, which is what the memory system's float atomics are priced by (tools/atomic_bench.hip).
 * View them as strided tensors ([B,V,4] with strides (S V, S, 1) ...).  The layout depends on C: use the state's
 * accumulators only with the channel count of the forward call that cleared them.
 */
int dirt_state_grad_buffers(void *workspace, size_t workspace_bytes, int B, int V, int F, int H, int W, int C,
                            float **grad_vertices, float **grad_vertex_colors, int *grad_vertices_row_stride,
                            int *grad_vertex_colors_row_stride);

/*
 * Texture look-up of a deferred shader, fused.  Replaces the TensorFlow composition
 * `sample_texture(texture, uvs_to_pixel_indices(uvs, shape, mode), filter)` of the reference's samples/textured.py:16-61
 * as used by its shader_fn (samples/textured.py:116-141): (u, v) with (0, 0) at the TOP-LEFT of the image -> repeat
 * (uvs % 1) or clamp -> scaled by the texture size -> bilinear blend of the four neighbours (fraction of the index, no
 * half-texel shift) or nearest.  Float32, the reference's operation order; where its gather would read row Ht / column Wt
 * the last texel is used.
 *   texture [Ht,Wt,Ct]; uvs: n pairs (u, v) `uv_stride` >= 2 floats apart -- read in place from a G-buffer [H,W,C] with
 *   uv_stride = C and the pointer at the u channel; out [n,Ct].
 * Backward: grad_out [n,Ct] -> grad_texture [Ht,Wt,Ct] (cleared by the call, then accumulated with float atomics) and
 * grad_uvs (n pairs `grad_uv_stride` apart; may be NULL).
 * dirt_texture_last_error(): thread-local description of the last failure of these two calls.
 */
#define DIRT_TEX_CLAMP 1u   /* mode 'clamp' instead of 'repeat' (samples/textured.py:21-24) */
#define DIRT_TEX_NEAREST 2u /* mode 'nearest' instead of 'bilinear' (samples/textured.py:31-33) */
int dirt_texture_sample_forward(const float *texture, const float *uvs, float *out, long long n, int Ht, int Wt, int Ct,
                                int uv_stride, unsigned flags, void *stream);
int dirt_texture_sample_backward(const float *texture, const float *uvs, const float *grad_out, float *grad_texture,
                                 float *grad_uvs, long long n, int Ht, int Wt, int Ct, int uv_stride, int grad_uv_stride,
                                 unsigned flags, void *stream);
/* The same gradient for look-ups that form an IMAGE -- rows x cols pixels, row-major, n = rows * cols pairs (a G-buffer slice
 * [H, W, 2]; batches stack their rows): the kernel then works on 16 x 16-pixel tiles, sums a tile's contributions in an LDS copy
 * of the texture patch they fall into and sends every texel of the patch to memory once -- instead of 4 Ct float atomics
 * per pixel (what `gather_nd`'s gradient in the reference and dirt_texture_sample_backward's flat runs of 256 amount to where
 * neighbouring pixels share texels).  Same results to summation order. */
int dirt_texture_sample_backward_image(const float *texture, const float *uvs, const float *grad_out, float *grad_texture,
                                       float *grad_uvs, long long rows, long long cols, int Ht, int Wt, int Ct, int uv_stride,
                                       int grad_uv_stride, unsigned flags, void *stream);
const char *dirt_texture_last_error(void);

/*
 * Per-kernel timing (host-side state only).  Slots are the library's kernels; dirt_profile_count()
 * returns how many there are, dirt_profile_name(i) their names.  dirt_profile_read waits for the
 * recorded events of calls made with DIRT_FLAG_PROFILE on this thread, adds them to the running
 * totals and returns total milliseconds and launch count of slot i; dirt_profile_reset clears the
 * totals.  Returns 0 or DIRT_E_*.
 */
int dirt_profile_count(void);
const char *dirt_profile_name(int slot);
int dirt_profile_read(int slot, double *total_ms, long long *launches);
int dirt_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* DIRT_HIP_H */
