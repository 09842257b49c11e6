// ubench.hip -- issue cost of the gfx950 instructions the kernels of this library are built from.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o tools/_bin/ubench
// Every probe runs 8 independent copies of one instruction back to back (so dependent-issue latency is not what is
// measured) RxU times, on (a) one wave alone and (b) four waves per SIMD on every CU (the occupancy of the raster and
// gradient kernels), and prints shader clocks (s_memtime) per instruction per wave and, for (b), per SIMD: the latter is
// what a kernel that is bound by instruction issue pays per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define R 64   // loop trips
#define U 8    // asm blocks per trip (each 8 instructions)

#define REGS float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float& a6, float& a7, float& b0, float& b1
#define OPS8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)

struct K_add      { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0)); } };
struct K_fma      { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };
struct K_cndmask_vcc { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc"); } };
struct K_cndmask_sgpr { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20", "s21"); } };
struct K_cmp_vcc  { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_cmp_eq_u32 vcc, %0, %8\n v_cmp_eq_u32 vcc, %1, %8\n v_cmp_eq_u32 vcc, %2, %8\n v_cmp_eq_u32 vcc, %3, %8\n v_cmp_eq_u32 vcc, %4, %8\n v_cmp_eq_u32 vcc, %5, %8\n v_cmp_eq_u32 vcc, %6, %8\n v_cmp_eq_u32 vcc, %7, %8"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc"); } };
struct K_cmp_sgpr { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_cmp_eq_u32_e64 s[20:21], %0, %8\n v_cmp_eq_u32_e64 s[22:23], %1, %8\n v_cmp_eq_u32_e64 s[24:25], %2, %8\n v_cmp_eq_u32_e64 s[26:27], %3, %8\n v_cmp_eq_u32_e64 s[20:21], %4, %8\n v_cmp_eq_u32_e64 s[22:23], %5, %8\n v_cmp_eq_u32_e64 s[24:25], %6, %8\n v_cmp_eq_u32_e64 s[26:27], %7, %8"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); } };
struct K_min_u32  { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0)); } };
struct K_min3_u32 { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n v_min3_u32 %3, %3, %8, %9\n v_min3_u32 %4, %4, %8, %9\n v_min3_u32 %5, %5, %8, %9\n v_min3_u32 %6, %6, %8, %9\n v_min3_u32 %7, %7, %8, %9"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };
#define DPP8(ins, ctl) \
    ins " %0, %8, %8 " ctl "\n " ins " %1, %8, %8 " ctl "\n " ins " %2, %8, %8 " ctl "\n " ins " %3, %8, %8 " ctl "\n " \
    ins " %4, %9, %9 " ctl "\n " ins " %5, %9, %9 " ctl "\n " ins " %6, %9, %9 " ctl "\n " ins " %7, %9, %9 " ctl
#define DPPK(name, ins, ctl) struct name { static __device__ __forceinline__ void run(REGS) { asm volatile(DPP8(ins, ctl) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };
DPPK(K_dpp_ror8, "v_add_f32_dpp", "row_ror:8 row_mask:0xf bank_mask:0xf")
DPPK(K_dpp_ror8_bank, "v_add_f32_dpp", "row_ror:8 row_mask:0xf bank_mask:0xc")
DPPK(K_dpp_quad, "v_add_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DPPK(K_dpp_hmirror, "v_add_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
DPPK(K_dpp_mirror, "v_add_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
DPPK(K_dpp_shl4, "v_add_f32_dpp", "row_shl:4 row_mask:0xf bank_mask:0x5")
DPPK(K_dpp_min, "v_min_u32_dpp", "row_ror:4 row_mask:0xf bank_mask:0xf")
struct K_dpp_mov { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_mov_b32_dpp %0, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %4, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 row_ror:8 row_mask:0xf bank_mask:0xf"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };
struct K_permlane16 { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };
struct K_permlane32 { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };
struct K_readlane { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); } };
struct K_rcp { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0)); } };
struct K_mul_lo { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0)); } };
struct K_mad_u24 { static __device__ __forceinline__ void run(REGS) { asm volatile(
    "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };

// packed fp32 and f64: register pairs
typedef float f2 __attribute__((ext_vector_type(2)));
#define REGS2 f2& a0, f2& a1, f2& a2, f2& a3, f2& a4, f2& a5, f2& a6, f2& a7, f2& b0, f2& b1
#define PK8(ins) \
    ins " %0, %0, %8, %9\n " ins " %1, %1, %8, %9\n " ins " %2, %2, %8, %9\n " ins " %3, %3, %8, %9\n " \
    ins " %4, %4, %8, %9\n " ins " %5, %5, %8, %9\n " ins " %6, %6, %8, %9\n " ins " %7, %7, %8, %9"
#define PK8_2(ins) \
    ins " %0, %0, %8\n " ins " %1, %1, %8\n " ins " %2, %2, %8\n " ins " %3, %3, %8\n " \
    ins " %4, %4, %8\n " ins " %5, %5, %8\n " ins " %6, %6, %8\n " ins " %7, %7, %8"
#define PAIRK(name, body) struct name { static __device__ __forceinline__ void run(REGS2) { asm volatile(body \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); } };
PAIRK(K_pk_fma, PK8("v_pk_fma_f32"))
PAIRK(K_pk_add, PK8_2("v_pk_add_f32"))
PAIRK(K_pk_mul, PK8_2("v_pk_mul_f32"))
PAIRK(K_fma_f64, PK8("v_fma_f64"))
PAIRK(K_add_f64, PK8_2("v_add_f64"))
PAIRK(K_mul_f64, PK8_2("v_mul_f64"))
PAIRK(K_lshl_add_u64, "v_lshl_add_u64 %0, %0, 2, %8\n v_lshl_add_u64 %1, %1, 2, %8\n v_lshl_add_u64 %2, %2, 2, %8\n v_lshl_add_u64 %3, %3, 2, %8\n v_lshl_add_u64 %4, %4, 2, %8\n v_lshl_add_u64 %5, %5, 2, %8\n v_lshl_add_u64 %6, %6, 2, %8\n v_lshl_add_u64 %7, %7, 2, %8")
PAIRK(K_pk_mov, PK8_2("v_pk_mov_b32"))

template <class KT, class T>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc)
{
    T a[8], b[2];
    for (int i = 0; i < 8; ++i) a[i] = (T)((float)threadIdx.x * 0.001f + (float)i);
    b[0] = (T)1.0001f; b[1] = (T)0.5f;
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    asm volatile("" : "+v"(b[0]), "+v"(b[1]));
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) KT::run(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], b[0], b[1]);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    T s = a[0];
    for (int i = 1; i < 8; ++i) s += a[i];
    float sf; if constexpr (sizeof(T) == 8) sf = s.x + s.y; else sf = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sf;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- LDS probes: 8 LDS instructions per block ----
template <int KIND>
__global__ __launch_bounds__(256) void lds_probe(float* out, long long* cyc)
{
    __shared__ __align__(16) float s[8192];   // 32 KB: four workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 256) s[i] = 0.f;
    __syncthreads();
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    float acc = 0.f;
    float* base = s + wave * 2048;
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    const uint32_t lbase = (uint32_t)(uintptr_t)base;   // LDS byte address
    typedef float v4 __attribute__((ext_vector_type(4)));
    typedef float v2 __attribute__((ext_vector_type(2)));
    v4 q4 = {1.f, 2.f, 3.f, 4.f}; v2 q2 = {1.f, 2.f}; float q1 = 1.f;
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { v4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(lbase + 16 * lane + 1024 * u)); asm volatile("" :: "v"(q)); }
            else if (KIND == 1) { asm volatile("ds_write_b128 %0, %1" :: "v"(lbase + 16 * lane + 1024 * u), "v"(q4) : "memory"); }
            else if (KIND == 2) { asm volatile("ds_add_f32 %0, %1" :: "v"(lbase + 4 * lane + 256 * u), "v"(q1) : "memory"); }
            else if (KIND == 3) { asm volatile("ds_add_f32 %0, %1" :: "v"(lbase + 4 * (lane & 15) + 256 * u), "v"(q1) : "memory"); }
            else if (KIND == 4) { asm volatile("ds_add_f32 %0, %1" :: "v"(lbase + 256 * u), "v"(q1) : "memory"); }
            else if (KIND == 5) { if (lane < 24) asm volatile("ds_add_f32 %0, %1" :: "v"(lbase + 4 * lane + 256 * u), "v"(q1) : "memory"); }
            else if (KIND == 6) { float q; asm volatile("ds_read_b32 %0, %1" : "=v"(q) : "v"(lbase + 4 * lane + 256 * u)); asm volatile("" :: "v"(q)); }
            else if (KIND == 7) { v2 q; asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(lbase + 8 * lane + 512 * u)); asm volatile("" :: "v"(q)); }
            else if (KIND == 8) { float q; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(q) : "v"((lane ^ 5) << 2), "v"(q1)); asm volatile("" :: "v"(q)); }
            else if (KIND == 9) { asm volatile("ds_write_b32 %0, %1" :: "v"(lbase + 4 * lane + 256 * u), "v"(q1) : "memory"); }
            else if (KIND == 10) { asm volatile("ds_write_b64 %0, %1" :: "v"(lbase + 8 * lane + 512 * u), "v"(q2) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * blockDim.x + tid] = acc + s[tid];
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static float* g_out; static long long* g_cyc;

template <class F>
void report(const char* name, F launch, int n_instr_per_trip)
{
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg == 0 ? 1 : 1024, threads = cfg == 0 ? 64 : 256;
        launch(blocks, threads); launch(blocks, threads);
        hipDeviceSynchronize();
        const int nw = blocks * threads / 64;
        std::vector<long long> h(nw);
        hipMemcpy(h.data(), g_cyc, nw * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        const double per = s / nw / ((double)R * n_instr_per_trip);
        if (cfg == 0) printf("%-34s 1 wave: %6.2f clk/instr", name, per);
        else printf("   4 waves/SIMD: %6.2f clk/instr/wave = %5.2f clk/instr/SIMD\n", per, per / 4.0);
    }
}

template <class KT, class T> void run(const char* name)
{
    report(name, [](int b, int t) { hipLaunchKernelGGL((probe<KT, T>), dim3(b), dim3(t), 0, 0, g_out, g_cyc); }, U * 8);
}
template <int KIND> void run_lds(const char* name)
{
    report(name, [](int b, int t) { hipLaunchKernelGGL((lds_probe<KIND>), dim3(b), dim3(t), 0, 0, g_out, g_cyc); }, 8);
}

int main()
{
    hipMalloc(&g_out, 4 * 1024 * 256 * 2); hipMalloc(&g_cyc, 8 * 4096);
    run<K_add, float>("v_add_f32");
    run<K_fma, float>("v_fma_f32");
    run<K_pk_fma, f2>("v_pk_fma_f32");
    run<K_pk_add, f2>("v_pk_add_f32");
    run<K_pk_mul, f2>("v_pk_mul_f32");
    run<K_pk_mov, f2>("v_pk_mov_b32");
    run<K_cndmask_vcc, float>("v_cndmask_b32 (vcc)");
    run<K_cndmask_sgpr, float>("v_cndmask_b32_e64 (sgpr)");
    run<K_cmp_vcc, float>("v_cmp_eq_u32 -> vcc");
    run<K_cmp_sgpr, float>("v_cmp_eq_u32_e64 -> sgpr");
    run<K_min_u32, float>("v_min_u32");
    run<K_min3_u32, float>("v_min3_u32");
    run<K_dpp_ror8, float>("v_add_f32_dpp row_ror:8");
    run<K_dpp_ror8_bank, float>("v_add_f32_dpp row_ror:8 bank 0xc");
    run<K_dpp_quad, float>("v_add_f32_dpp quad_perm");
    run<K_dpp_hmirror, float>("v_add_f32_dpp row_half_mirror");
    run<K_dpp_mirror, float>("v_add_f32_dpp row_mirror");
    run<K_dpp_shl4, float>("v_add_f32_dpp row_shl:4 bank 0x5");
    run<K_dpp_min, float>("v_min_u32_dpp row_ror:4");
    run<K_dpp_mov, float>("v_mov_b32_dpp row_ror:8");
    run<K_permlane16, float>("v_permlane16_swap_b32");
    run<K_permlane32, float>("v_permlane32_swap_b32");
    run<K_readlane, float>("v_readlane_b32");
    run<K_rcp, float>("v_rcp_f32");
    run<K_mul_lo, float>("v_mul_lo_u32");
    run<K_mad_u24, float>("v_mad_u32_u24");
    run<K_lshl_add_u64, f2>("v_lshl_add_u64");
    run<K_fma_f64, f2>("v_fma_f64");
    run<K_add_f64, f2>("v_add_f64");
    run<K_mul_f64, f2>("v_mul_f64");
    run_lds<0>("ds_read_b128 contiguous");
    run_lds<7>("ds_read_b64 contiguous");
    run_lds<6>("ds_read_b32 contiguous");
    run_lds<1>("ds_write_b128 contiguous");
    run_lds<10>("ds_write_b64 contiguous");
    run_lds<9>("ds_write_b32 contiguous");
    run_lds<2>("ds_add_f32 64 distinct");
    run_lds<3>("ds_add_f32 16 addr x 4 lanes");
    run_lds<4>("ds_add_f32 1 addr x 64 lanes");
    run_lds<5>("ds_add_f32 24 lanes distinct");
    run_lds<8>("ds_bpermute_b32");
    return 0;
}
