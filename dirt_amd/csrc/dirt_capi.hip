// dirt_capi.hip -- the C ABI of libdirt_hip.so (see include/dirt_hip.h for the contract and the
// reference interfaces each entry point replaces).  Host code only: argument validation with the
// reference's error conditions, workspace carving, and kernel launches on the caller's stream.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <iterator>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/dirt_hip.h"
#include "dirt_launch.h"
#include "dirt_raster_common.h"

namespace {

thread_local char g_last_error[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

// ---- optional per-kernel HIP-event timing (DIRT_FLAG_PROFILE) ----------------------------------
enum Slot { SLOT_GEOMETRY = 0, SLOT_RASTER_FWD, SLOT_RASTER_VIS, SLOT_GRAD, SLOT_COUNT };
const char* const kSlotNames[SLOT_COUNT] = {"setup_kernel", "raster_kernel<shade>",
                                            "raster_kernel<visibility>", "grad_kernel"};
struct Pending {
    hipEvent_t a, b;
    int slot;
};
struct Profile {
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[SLOT_COUNT] = {0};
    long long launches[SLOT_COUNT] = {0};
};
thread_local Profile g_prof;

struct Scope {  // records an event pair around the launches issued during its lifetime
    bool on;
    hipStream_t stream;
    Pending p;
    static hipEvent_t get()
    {
        if (!g_prof.pool.empty()) { hipEvent_t e = g_prof.pool.back(); g_prof.pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return e;
    }
    Scope(bool enabled, int slot, hipStream_t s) : on(enabled && g_prof.pending.size() < 65536), stream(s)
    {
        if (!on) return;
        p.slot = slot; p.a = get(); p.b = get();
        if (!p.a || !p.b) {  // out of events: this launch goes untimed, the work itself is unaffected
            if (p.a) g_prof.pool.push_back(p.a);
            if (p.b) g_prof.pool.push_back(p.b);
            on = false;
            return;
        }
        (void)hipEventRecord(p.a, stream);
    }
    ~Scope()
    {
        if (!on) return;
        (void)hipEventRecord(p.b, stream);
        g_prof.pending.push_back(p);
    }
};

void drain_profile()
{
    for (const Pending& p : g_prof.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_prof.total_ms[p.slot] += ms;
            g_prof.launches[p.slot] += 1;
        }
        g_prof.pool.push_back(p.a);
        g_prof.pool.push_back(p.b);
    }
    g_prof.pending.clear();
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- which gradient outputs a forward call has pre-cleared (host-side bookkeeping only) ---------------------------
// A KEEP_STATE forward clears, inside its launch, the buffers the backward pass will add into: the interleaved accumulators
// of the state, or (dirt_rasterise_forward_train) the caller's dense grad_vertices / grad_vertex_colors.  That is valid for
// ONE backward call.  Instead of trusting the caller with that (round 4: a second backward with DIRT_FLAG_DENSE_FROM_STATE
// silently returned doubled gradients), the library remembers per workspace address what is still "armed": a backward call
// that finds its outputs armed consumes them and launches nothing to clear; one that does not -- a second backward over the
// same forward, other tensors than the forward was given, a record that was evicted -- clears them itself with one more
// launch.  Either way every backward call returns the gradients of that call alone.  No device state; the table is bounded.
struct Armed {
    const void* gv = nullptr;    // dense outputs cleared by the forward, still untouched
    const void* gvc = nullptr;
    bool acc = false;            // the state's own accumulators are cleared and untouched
    unsigned long long tick = 0;
};
std::mutex g_armed_mu;
std::unordered_map<const void*, Armed> g_armed;
unsigned long long g_armed_tick = 0;
constexpr size_t ARMED_CAP = 4096;

void armed_set(const void* ws, const void* gv, const void* gvc, bool acc)
{
    std::lock_guard<std::mutex> lock(g_armed_mu);
    if (!gv && !gvc && !acc) { g_armed.erase(ws); return; }
    if (g_armed.size() >= ARMED_CAP && !g_armed.count(ws)) {   // drop the older half: a lost record only costs a clearing launch
        const unsigned long long cut = g_armed_tick - ARMED_CAP / 2;
        for (auto it = g_armed.begin(); it != g_armed.end();) it = it->second.tick < cut ? g_armed.erase(it) : std::next(it);
    }
    Armed a; a.gv = gv; a.gvc = gvc; a.acc = acc; a.tick = ++g_armed_tick;
    g_armed[ws] = a;
}
bool armed_take_dense(const void* ws, const void* gv, const void* gvc)
{
    std::lock_guard<std::mutex> lock(g_armed_mu);
    auto it = g_armed.find(ws);
    if (it == g_armed.end() || !gv || it->second.gv != gv || it->second.gvc != gvc) return false;
    it->second.gv = it->second.gvc = nullptr;
    if (!it->second.acc) g_armed.erase(it);
    return true;
}
bool armed_take_acc(const void* ws)
{
    std::lock_guard<std::mutex> lock(g_armed_mu);
    auto it = g_armed.find(ws);
    if (it == g_armed.end() || !it->second.acc) return false;
    it->second.acc = false;
    if (!it->second.gv) g_armed.erase(it);
    return true;
}

struct Workspace {
    size_t recs_off, boxes_off, cells_off, entries_off, lrecs_off, crecs_off, state_a_off, state_b_off, gv_off, gvc_off, total;
    int acc_stride;     // the two gradient accumulators share rows of this many floats
};

// Layout: [FaceRec x B*F | FaceBox x B*F | chunk x bin directory | per-chunk BinEntry segments (5 per face) |
//          float2 per-pixel state {clip_w, face} x B*H*W | float2 barycentric pair (encode_bary) x B*H*W | gradient accumulators:
//          float x B*V*(4 + C rounded up to 4): a vertex's position and colour gradients in one row]
Workspace carve(int B, int V, int F, int H, int W, int C)
{
    Workspace w;
    size_t off = 0;
    w.recs_off = off;    off = align_up(off + (size_t)B * F * sizeof(dirt::FaceRec), 256);
    w.boxes_off = off;   off = align_up(off + (size_t)B * F * sizeof(dirt::FaceBox), 256);
    int nchunk, chunk_faces;
    dirt::chunking(F, nchunk, chunk_faces);
    const bool masked = dirt::directory_is_masked(chunk_faces);
    w.cells_off = off;   off = align_up(off + (size_t)B * nchunk * ((masked ? dirt::MAX_BINS_MASKED + 1 : dirt::MAX_BINS) + 1) * sizeof(dirt::BinCell), 256);
    w.entries_off = off; off = align_up(off + (size_t)B * nchunk * 5 * (size_t)chunk_faces * sizeof(dirt::BinEntry), 256);
    // (masked directory, setup_kernel_v2: per face its face-local coverage record and its three vertex colours as float4s)
    w.lrecs_off = off;   off = align_up(off + (masked ? (size_t)B * F * sizeof(dirt::TileRec) : 0), 256);
    w.crecs_off = off;   off = align_up(off + (masked ? (size_t)B * F * 3 * sizeof(float4) : 0), 256);
    w.state_a_off = off; off = align_up(off + (size_t)B * H * W * sizeof(float2), 256);
    w.state_b_off = off; off = align_up(off + (size_t)B * H * W * sizeof(float2), 256);
    // gradient accumulators of the backward pass, pre-cleared by a KEEP_STATE forward (dirt_state_grad_buffers):
    // interleaved, one row {x, y, z, w, c0 .. cC-1} of acc_stride = 4 + C (rounded up to a multiple of 4) floats per vertex
    w.acc_stride = (4 + C + 3) / 4 * 4;
    w.gv_off = off;  w.gvc_off = off + 4 * sizeof(float);
    off = align_up(off + (size_t)B * V * w.acc_stride * sizeof(float), 256);
    w.total = off + 256;
    return w;
}

struct Carved {
    dirt::FaceRec* recs;
    dirt::FaceBox* boxes;
    dirt::BinCell* cells;
    dirt::BinEntry* entries;
    dirt::TileRec* lrecs;
    float4* crecs;
    float2* state_a;
    float2* state_b;
    float* gv;
    float* gvc;
};

int check_sizes(const char* who, int B, int V, int F, int H, int W, int C)
{
    if (B < 0 || V < 0 || F < 0)
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: negative batch / vertex / face count (B=%d V=%d F=%d)", who, B, V, F);
    if (H <= 0 || W <= 0)  // CHECK(width > 0 && height > 0), csrc/hwc.h:28
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: height and width must be positive (H=%d W=%d)", who, H, W);
    if (C <= 0)  // the reference op requires C in {1,3} (csrc/hwc.h:27); its Python layer accepts C > 0
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: channels must be positive (C=%d)", who, C);
    if (H > DIRT_MAX_DIM || W > DIRT_MAX_DIM)
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: frame larger than %d pixels (H=%d W=%d)", who, DIRT_MAX_DIM, H, W);
    if (B > 65535) return fail(DIRT_E_INVALID_ARGUMENT, "%s: batch larger than 65535 (B=%d)", who, B);
    return DIRT_OK;
}

int check_workspace(const char* who, const Workspace& w, void* workspace, size_t bytes)
{
    if (!workspace) return fail(DIRT_E_WORKSPACE, "%s: workspace is NULL", who);
    if (((uintptr_t)workspace & 15u) != 0) return fail(DIRT_E_WORKSPACE, "%s: workspace is not 16-byte aligned", who);
    if (bytes < w.total)
        return fail(DIRT_E_WORKSPACE, "%s: workspace too small (%zu bytes given, %zu needed)", who, bytes, w.total);
    return DIRT_OK;
}

// Kernels use 16-byte loads / stores on these tensors (float4 vertices, HWC pixels with C % 4 == 0).
int check_aligned(const char* who, std::initializer_list<const void*> ptrs)
{
    for (const void* q : ptrs)
        if ((reinterpret_cast<uintptr_t>(q) & 15u) != 0)
            return fail(DIRT_E_INVALID_ARGUMENT, "%s: tensor pointers must be 16-byte aligned", who);
    return DIRT_OK;
}

#define HIP_TRY(who, expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(DIRT_E_HIP, "%s: %s failed: %s", who, #expr, hipGetErrorString(_e)); \
    } while (0)

inline char* base256(void* workspace)
{
    // FaceRec needs 128-byte alignment; the caller guarantees 16.
    return reinterpret_cast<char*>(align_up((size_t)(uintptr_t)workspace, 256));
}

Carved carved(void* workspace, const Workspace& w)
{
    char* ws = base256(workspace);
    Carved c;
    c.recs = reinterpret_cast<dirt::FaceRec*>(ws + w.recs_off);
    c.boxes = reinterpret_cast<dirt::FaceBox*>(ws + w.boxes_off);
    c.cells = reinterpret_cast<dirt::BinCell*>(ws + w.cells_off);
    c.entries = reinterpret_cast<dirt::BinEntry*>(ws + w.entries_off);
    c.lrecs = reinterpret_cast<dirt::TileRec*>(ws + w.lrecs_off);
    c.crecs = reinterpret_cast<float4*>(ws + w.crecs_off);
    c.state_a = reinterpret_cast<float2*>(ws + w.state_a_off);
    c.state_b = reinterpret_cast<float2*>(ws + w.state_b_off);
    c.gv = reinterpret_cast<float*>(ws + w.gv_off);
    c.gvc = reinterpret_cast<float*>(ws + w.gvc_off);
    return c;
}

dirt::GeomParams geom_params(const Carved& c, const float* vertices, const int32_t* faces, int B, int V, int F, int H,
                             int W, unsigned flags, const float* vertex_colors = nullptr, int C = 0)
{
    dirt::GeomParams g;
    g.shared_faces = (flags & DIRT_FLAG_SHARED_FACES) ? 1 : 0;
    g.vertices = vertices; g.faces = faces; g.recs = c.recs; g.boxes = c.boxes; g.cells = c.cells;
    g.entries = c.entries;
    dirt::chunking(F, g.nchunk, g.chunk_faces);
    g.masked = dirt::directory_is_masked(g.chunk_faces) ? 1 : 0;
    g.B = B; g.V = V; g.F = F; g.H = H; g.W = W;
    g.grid = dirt::make_bin_grid(H, W, g.nchunk, g.masked != 0, dirt::raster_tile_choice(H, W, B, flags) == 16 ? 4 : 5);
    g.v2_only = 0;
    g.lrecs = g.masked ? c.lrecs : nullptr;
    // the faces' vertex colours ride along for the forward pass of 1 / 3 / 4-channel images (raster_kernel_v2)
    const bool colours = g.masked && vertex_colors != nullptr && V > 0 && (C == 1 || C == 3 || C == 4);
    g.crecs = colours ? c.crecs : nullptr;
    g.vertex_colors = colours ? vertex_colors : nullptr;
    g.C = C;
    return g;
}

dirt::RasterParams raster_params(const Carved& c, const dirt::GeomParams& g, int C, unsigned flags)
{
    dirt::RasterParams p;
    p.flags = flags;
    p.recs = c.recs; p.cells = c.cells; p.entries = c.entries; p.nchunk = g.nchunk; p.chunk_faces = g.chunk_faces; p.masked = g.masked;
    p.lrecs = g.lrecs; p.crecs = g.crecs;
    p.background = nullptr; p.vertex_colors = nullptr; p.pixels = nullptr; p.vis = nullptr; p.state_a = nullptr; p.state_b = nullptr;
    p.V = g.V; p.F = g.F; p.H = g.H; p.W = g.W; p.C = C;
    p.grid = g.grid; p.tiles_x = 0; p.tiles_y = 0;
    p.zero_b = nullptr; p.zero_b_bytes = 0; p.zero_c = nullptr; p.zero_c_bytes = 0;
    return p;
}

}  // namespace

extern "C" {

int dirt_abi_version(void) { return DIRT_ABI_VERSION; }

const char* dirt_last_error(void) { return g_last_error; }

size_t dirt_workspace_bytes(int B, int V, int F, int H, int W, int C)
{
    if (check_sizes("dirt_workspace_bytes", B, V, F, H, W, C) != DIRT_OK) return 0;
    return carve(B, V, F, H, W, C).total;
}

static int forward_impl(const char* who, const float* background, const float* vertices, const float* vertex_colors,
                        const int32_t* faces, float* pixels, float* grad_vertices, float* grad_vertex_colors, int B, int V, int F,
                        int H, int W, int C, void* workspace, size_t workspace_bytes, unsigned flags, void* stream_)
{
    int rc = check_sizes(who, B, V, F, H, W, C);
    if (rc) return rc;
    if (B == 0) return DIRT_OK;
    if (!background || !pixels) return fail(DIRT_E_INVALID_ARGUMENT, "%s: background / pixels is NULL", who);
    if ((V > 0 && (!vertices || !vertex_colors)) || (F > 0 && !faces))
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: vertices / vertex_colors / faces is NULL", who);
    rc = check_aligned(who, {background, vertices, vertex_colors, faces, pixels, grad_vertices, grad_vertex_colors});
    if (rc) return rc;
    const Workspace w = carve(B, V, F, H, W, C);
    rc = check_workspace(who, w, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const Carved c = carved(workspace, w);
    const bool prof = (flags & DIRT_FLAG_PROFILE) != 0;
    dirt::GeomParams g = geom_params(c, vertices, faces, B, V, F, H, W, flags, vertex_colors, C);
    dirt::RasterParams p = raster_params(c, g, C, flags);
    p.background = background; p.vertex_colors = vertex_colors; p.pixels = pixels;
    g.v2_only = dirt::raster_v2_applies(p, B, false) ? 1 : 0;   // (set-up then skips what only dirt_raster.hip's kernels read)
    {
        Scope sc(prof, SLOT_GEOMETRY, stream);
        HIP_TRY(who, dirt::launch_geometry(g, stream));
    }
    const bool dense = grad_vertices != nullptr && grad_vertex_colors != nullptr && V > 0;
    if (flags & DIRT_FLAG_KEEP_STATE) {
        p.state_a = c.state_a; p.state_b = c.state_b;
        // pre-clear what the backward pass will add into, a side job of the raster kernel's workgroups: the caller's dense
        // gradient tensors (dirt_rasterise_forward_train), or the interleaved accumulators inside the state
        if (dense) {
            p.zero_b = grad_vertices;      p.zero_b_bytes = sizeof(float) * (size_t)B * V * 4;
            p.zero_c = grad_vertex_colors; p.zero_c_bytes = sizeof(float) * (size_t)B * V * C;
        } else {
            p.zero_b = c.gv;  p.zero_b_bytes = sizeof(float) * (size_t)B * V * w.acc_stride;
        }
    }
    {
        Scope sc(prof, SLOT_RASTER_FWD, stream);
        HIP_TRY(who, dirt::launch_raster(p, B, false, stream));
    }
    if (flags & DIRT_FLAG_KEEP_STATE) armed_set(workspace, dense ? grad_vertices : nullptr, dense ? grad_vertex_colors : nullptr, !dense);
    else armed_set(workspace, nullptr, nullptr, false);
    g_last_error[0] = 0;
    return DIRT_OK;
}

int dirt_rasterise_forward(const float* background, const float* vertices, const float* vertex_colors,
                           const int32_t* faces, float* pixels, int B, int V, int F, int H, int W, int C,
                           void* workspace, size_t workspace_bytes, unsigned flags, void* stream_)
{
    return forward_impl("dirt_rasterise_forward", background, vertices, vertex_colors, faces, pixels, nullptr, nullptr, B, V, F, H, W, C,
                        workspace, workspace_bytes, flags, stream_);
}

int dirt_rasterise_forward_train(const float* background, const float* vertices, const float* vertex_colors,
                                 const int32_t* faces, float* pixels, float* grad_vertices, float* grad_vertex_colors,
                                 int B, int V, int F, int H, int W, int C, void* workspace, size_t workspace_bytes,
                                 unsigned flags, void* stream_)
{
    const char* who = "dirt_rasterise_forward_train";
    if (B > 0 && V > 0 && (!grad_vertices || !grad_vertex_colors))
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: grad_vertices / grad_vertex_colors is NULL", who);
    return forward_impl(who, background, vertices, vertex_colors, faces, pixels, grad_vertices, grad_vertex_colors, B, V, F, H, W, C,
                        workspace, workspace_bytes, flags | DIRT_FLAG_KEEP_STATE, stream_);
}

int dirt_rasterise_visibility(const float* vertices, const int32_t* faces, int32_t* face_id, int B, int V, int F,
                              int H, int W, void* workspace, size_t workspace_bytes, unsigned flags, void* stream_)
{
    const char* who = "dirt_rasterise_visibility";
    int rc = check_sizes(who, B, V, F, H, W, 1);
    if (rc) return rc;
    if (B == 0) return DIRT_OK;
    if (!face_id) return fail(DIRT_E_INVALID_ARGUMENT, "%s: face_id is NULL", who);
    if ((V > 0 && !vertices) || (F > 0 && !faces))
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: vertices / faces is NULL", who);
    rc = check_aligned(who, {vertices, faces, face_id});
    if (rc) return rc;
    const Workspace w = carve(B, V, F, H, W, 1);
    rc = check_workspace(who, w, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const Carved c = carved(workspace, w);
    const bool prof = (flags & DIRT_FLAG_PROFILE) != 0;
    dirt::GeomParams g = geom_params(c, vertices, faces, B, V, F, H, W, flags);
    armed_set(workspace, nullptr, nullptr, false);   // the workspace is rebuilt: whatever a forward left cleared in it is forgotten
    dirt::RasterParams p = raster_params(c, g, 1, flags);
    p.vis = face_id;
    g.v2_only = dirt::raster_v2_applies(p, B, true) ? 1 : 0;
    {
        Scope sc(prof, SLOT_GEOMETRY, stream);
        HIP_TRY(who, dirt::launch_geometry(g, stream));
    }
    {
        Scope sc(prof, SLOT_RASTER_VIS, stream);
        HIP_TRY(who, dirt::launch_raster(p, B, true, stream));
    }
    g_last_error[0] = 0;
    return DIRT_OK;
}

int dirt_rasterise_backward(const float* vertices, const int32_t* faces, const float* pixels,
                            const float* grad_pixels, float* grad_background, float* grad_vertices,
                            float* grad_vertex_colors, float* debug_thingy, int B, int V, int F, int H, int W, int C,
                            void* workspace, size_t workspace_bytes, unsigned flags, void* stream_)
{
    const char* who = "dirt_rasterise_backward";
    int rc = check_sizes(who, B, V, F, H, W, C);
    if (rc) return rc;
    if (V > (1 << 24))  // csrc/rasterise_grad_egl.cpp:399-405
        return fail(DIRT_E_TOO_MANY_VERTICES, "%s: supports a maximum of %d vertices, vs. %d passed", who, 1 << 24, V);
    if (B == 0) return DIRT_OK;
    if (!pixels || !grad_pixels || !grad_background)
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: pixels / grad_pixels / grad_background is NULL", who);
    if ((V > 0 && (!vertices || !grad_vertices || !grad_vertex_colors)) || (F > 0 && !faces))
        return fail(DIRT_E_INVALID_ARGUMENT, "%s: vertices / faces / grad_vertices / grad_vertex_colors is NULL", who);
    rc = check_aligned(who, {vertices, faces, pixels, grad_pixels, grad_background, grad_vertices, grad_vertex_colors});
    if (rc) return rc;
    const Workspace w = carve(B, V, F, H, W, C);
    rc = check_workspace(who, w, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const Carved c = carved(workspace, w);
    const bool prof = (flags & DIRT_FLAG_PROFILE) != 0;
    // the cudaMemsetAsync x4 of csrc/rasterise_grad_egl.cu:244-250: grad_vertices / grad_vertex_colors
    // are cleared by one launch; grad_background and debug_thingy are fully written by the gradient
    // kernel instead
    // the caller's outputs are dense unless they are the state's own accumulators (dirt_state_grad_buffers)
    // (... or, DIRT_FLAG_DENSE_FROM_STATE, dense tensors that receive a copy of those accumulators after the kernel)
    const bool own_outputs = (flags & DIRT_FLAG_REUSE_STATE) && grad_vertices == c.gv && grad_vertex_colors == c.gvc;
    const bool unpack = (flags & DIRT_FLAG_REUSE_STATE) && (flags & DIRT_FLAG_DENSE_FROM_STATE) && !own_outputs;
    const bool state_outputs = own_outputs || unpack;
    if (flags & DIRT_FLAG_REUSE_STATE) {
        // records + visibility were left in this workspace by the forward pass; so were cleared gradient accumulators (or
        // cleared dense outputs: dirt_rasterise_forward_train).  Whatever this call adds into is taken if it is still armed
        // (cleared by that forward, untouched since) and cleared here otherwise: every call returns its own gradients.
        if (state_outputs) {
            if (!armed_take_acc(workspace)) {
                Scope sc(prof, SLOT_GEOMETRY, stream);
                HIP_TRY(who, dirt::launch_zero(c.gv, sizeof(float) * (size_t)B * V * w.acc_stride, nullptr, 0, stream));
            }
        } else if (!((flags & DIRT_FLAG_OUTPUTS_CLEARED) && armed_take_dense(workspace, grad_vertices, grad_vertex_colors))) {
            Scope sc(prof, SLOT_GEOMETRY, stream);
            HIP_TRY(who, dirt::launch_zero(grad_vertices, sizeof(float) * (size_t)B * V * 4, grad_vertex_colors,
                                           sizeof(float) * (size_t)B * V * C, stream));
        }
    } else {
        armed_set(workspace, nullptr, nullptr, false);   // the state is rebuilt below; its accumulators are not cleared
        dirt::GeomParams g = geom_params(c, vertices, faces, B, V, F, H, W, flags);
        dirt::RasterParams rp = raster_params(c, g, C, flags);
        g.v2_only = dirt::raster_v2_applies(rp, B, true) ? 1 : 0;
        {
            Scope sc(prof, SLOT_GEOMETRY, stream);
            HIP_TRY(who, dirt::launch_geometry(g, stream));
        }
        rp.state_a = c.state_a; rp.state_b = c.state_b;
        rp.zero_b = grad_vertices;      rp.zero_b_bytes = sizeof(float) * (size_t)B * V * 4;
        rp.zero_c = grad_vertex_colors; rp.zero_c_bytes = sizeof(float) * (size_t)B * V * C;
        {
            Scope sc(prof, SLOT_RASTER_VIS, stream);
            HIP_TRY(who, dirt::launch_raster(rp, B, true, stream));
        }
    }
    dirt::GradParams gp;
    gp.state_a = c.state_a; gp.state_b = c.state_b; gp.faces = faces; gp.shared_faces = (flags & DIRT_FLAG_SHARED_FACES) ? 1 : 0;
    gp.pixels = pixels; gp.grad_pixels = grad_pixels;
    gp.grad_background = grad_background; gp.grad_vertices = unpack ? c.gv : grad_vertices;
    gp.grad_vertex_colors = unpack ? c.gvc : grad_vertex_colors; gp.debug_thingy = debug_thingy;
    gp.gv_stride = state_outputs ? w.acc_stride : 4;
    gp.gvc_stride = state_outputs ? w.acc_stride : C;
    gp.B = B; gp.V = V; gp.F = F; gp.H = H; gp.W = W; gp.C = C; gp.flags = flags;
    {
        Scope sc(prof, SLOT_GRAD, stream);
        HIP_TRY(who, dirt::launch_grad(gp, stream));
        if (unpack) HIP_TRY(who, dirt::launch_unpack(c.gv, c.gvc, w.acc_stride, grad_vertices, grad_vertex_colors, C, (size_t)B * V, stream));
    }
    g_last_error[0] = 0;
    return DIRT_OK;
}

int dirt_state_grad_buffers(void* workspace, size_t workspace_bytes, int B, int V, int F, int H, int W, int C,
                            float** grad_vertices, float** grad_vertex_colors, int* grad_vertices_row_stride,
                            int* grad_vertex_colors_row_stride)
{
    const char* who = "dirt_state_grad_buffers";
    int rc = check_sizes(who, B, V, F, H, W, C);
    if (rc) return rc;
    const Workspace w = carve(B, V, F, H, W, C);
    rc = check_workspace(who, w, workspace, workspace_bytes);
    if (rc) return rc;
    const Carved c = carved(workspace, w);
    if (grad_vertices) *grad_vertices = c.gv;
    if (grad_vertex_colors) *grad_vertex_colors = c.gvc;
    if (grad_vertices_row_stride) *grad_vertices_row_stride = w.acc_stride;
    if (grad_vertex_colors_row_stride) *grad_vertex_colors_row_stride = w.acc_stride;
    return DIRT_OK;
}

int dirt_profile_count(void) { return SLOT_COUNT; }

const char* dirt_profile_name(int slot) { return (slot >= 0 && slot < SLOT_COUNT) ? kSlotNames[slot] : ""; }

int dirt_profile_read(int slot, double* total_ms, long long* launches)
{
    if (slot < 0 || slot >= SLOT_COUNT) return fail(DIRT_E_INVALID_ARGUMENT, "dirt_profile_read: bad slot %d", slot);
    drain_profile();
    if (total_ms) *total_ms = g_prof.total_ms[slot];
    if (launches) *launches = g_prof.launches[slot];
    return DIRT_OK;
}

int dirt_profile_reset(void)
{
    drain_profile();
    for (int i = 0; i < SLOT_COUNT; ++i) { g_prof.total_ms[i] = 0; g_prof.launches[i] = 0; }
    return DIRT_OK;
}

}  // extern "C"
