// dirt_reduce.h -- the cross-lane reduction of the gradient kernel's face loop (gfx950 DPP), kept apart so that
// tools/reduce_test.hip can check the lane mapping on the device.
//
// A wave is four DPP ROWS of 16 lanes; in dirt_grad.hip a row holds one 8 x 8 pixel block.  row_reduce_scatter<N> sums
// N (16 or 24) per-lane values over the 16 lanes of every row at once, with a transposing butterfly: at each of the
// four levels a lane is paired with the lane that differs in one lane bit and value i with value i + (half of what is
// left); the lane keeps one of the two values, sends the other to its partner and adds what it receives, so registers
// halve while lanes specialise.  Lane bits 3 and 2 select DPP BANKS (groups of four lanes), so those levels are two
// bank-masked v_add_f32_dpp per pair and no selects (inline assembly: the masked form has no builtin); bits 1 and 0 are
// quad permutations (two selects and one DPP add per pair).  N = 24: 24 + 12 + 9 + 4 = 49 instructions (+ 4 s_nop) for the totals
// of 24 values x 4 rows; a plain DPP row sum of every value would be 96 + the selects.
//
// Where the totals end up (row_value_of_lane): with b_k = bit k of the lane,
//     N = 16:  d0 = total of value  b0 + 2 b1 + 4 b2 + 8 b3                         (one value per lane)
//     N = 24:  d0 = total of value  b0 + 3 b1 + 6 b2 + 12 b3,
//              d1 = total of value  2 + 3 b1 + 6 b2 + 12 b3   (in both lanes of a b0 pair: the even lane uses it)
//     N = 32:  d0 = total of value  b0 + 4 b1 + 8 b2 + 16 b3,  d1 = total of value d0's + 2
#pragma once
#include <hip/hip_runtime.h>

namespace dirt {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}

// keep `lo` (lower lane of the pair) or `hi` (upper lane), add what the partner sends
template <int CTRL>
__device__ __forceinline__ float pack_pair(float lo, float hi, bool upper)
{
    const float keep = upper ? hi : lo, send = upper ? lo : hi;
    return keep + dpp_mov<CTRL>(send);
}

// One butterfly level whose "upper" lanes are whole DPP banks, for NP pairs (lo[i], hi[i]) at once, IN PLACE in lo[]: a single
// asm block, so that the wait states a DPP operand needs behind the VALU instruction that wrote it (two) are paid once
// per level -- one leading s_nop; inside the block every source was written at least two instructions earlier -- and not
// once per pair (rounds 2-3: 18 s_nop per reduction, 154 per wave at K3).  The trailing s_nop covers whatever DPP
// instruction the compiler places behind the block (it does not look inside inline assembly).  The in/out operands are
// EARLY-CLOBBER ("+&v"): lo[i] is overwritten while hi[j] of later pairs is still to be read, so no hi[j] may share a register
// with a lo[i] -- which the compiler would otherwise be free to arrange when both hold the same value (a padding zero in both
// halves: tools/reduce_test.hip has that case).
// bit 3 (lanes 8-15 of a row = banks 2, 3): lo + partner's lo everywhere, then hi + partner's hi in the upper banks.
// bit 2 (banks 1, 3 are the upper lanes): banks 0, 2 take lo + lo of the lane four above (row_shl:4), banks 1, 3 take
// hi + hi of the lane four below (row_shr:4).
// (16 pairs: the 32-value reduction of the two-triple pass shape)
__device__ __forceinline__ void level_bit3(float (&lo)[16], const float (&hi)[16])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %4, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %5, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %6, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %7, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %8, %24, %24 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %9, %25, %25 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %10, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %10, %26, %26 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %11, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %11, %27, %27 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %12, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %12, %28, %28 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %13, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %13, %29, %29 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %14, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %14, %30, %30 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %15, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %15, %31, %31 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5]), "+&v"(lo[6]), "+&v"(lo[7]), "+&v"(lo[8]), "+&v"(lo[9]), "+&v"(lo[10]), "+&v"(lo[11]), "+&v"(lo[12]), "+&v"(lo[13]), "+&v"(lo[14]), "+&v"(lo[15])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]), "v"(hi[8]), "v"(hi[9]), "v"(hi[10]), "v"(hi[11]), "v"(hi[12]), "v"(hi[13]), "v"(hi[14]), "v"(hi[15]));
}
__device__ __forceinline__ void level_bit3(float (&lo)[12], const float (&hi)[12])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %5, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %6, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %7, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %8, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %9, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %10, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %10, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %11, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %11, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5]), "+&v"(lo[6]), "+&v"(lo[7]), "+&v"(lo[8]), "+&v"(lo[9]), "+&v"(lo[10]), "+&v"(lo[11])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]), "v"(hi[8]), "v"(hi[9]), "v"(hi[10]), "v"(hi[11]));
}
__device__ __forceinline__ void level_bit3(float (&lo)[8], const float (&hi)[8])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5]), "+&v"(lo[6]), "+&v"(lo[7])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]));
}
__device__ __forceinline__ void level_bit3(float (&lo)[6], const float (&hi)[6])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %3, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %4, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %5, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]));
}
__device__ __forceinline__ void level_bit3(float (&lo)[4], const float (&hi)[4])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
}
__device__ __forceinline__ void level_bit2(float (&lo)[16], const float (&hi)[16])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %16, %16 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %17, %17 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %18, %18 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %3, %19, %19 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %4, %20, %20 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %5, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %5, %21, %21 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %6, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %6, %22, %22 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %7, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %7, %23, %23 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %8, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %8, %24, %24 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %9, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %9, %25, %25 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %10, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %10, %26, %26 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %11, %11, %11 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %11, %27, %27 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %12, %12, %12 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %12, %28, %28 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %13, %13, %13 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %13, %29, %29 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %14, %14, %14 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %14, %30, %30 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %15, %15, %15 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %15, %31, %31 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5]), "+&v"(lo[6]), "+&v"(lo[7]), "+&v"(lo[8]), "+&v"(lo[9]), "+&v"(lo[10]), "+&v"(lo[11]), "+&v"(lo[12]), "+&v"(lo[13]), "+&v"(lo[14]), "+&v"(lo[15])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]), "v"(hi[8]), "v"(hi[9]), "v"(hi[10]), "v"(hi[11]), "v"(hi[12]), "v"(hi[13]), "v"(hi[14]), "v"(hi[15]));
}
__device__ __forceinline__ void level_bit2(float (&lo)[12], const float (&hi)[12])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %3, %15, %15 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %4, %16, %16 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %5, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %5, %17, %17 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %6, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %6, %18, %18 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %7, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %7, %19, %19 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %8, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %8, %20, %20 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %9, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %9, %21, %21 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %10, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %10, %22, %22 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %11, %11, %11 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %11, %23, %23 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5]), "+&v"(lo[6]), "+&v"(lo[7]), "+&v"(lo[8]), "+&v"(lo[9]), "+&v"(lo[10]), "+&v"(lo[11])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]), "v"(hi[8]), "v"(hi[9]), "v"(hi[10]), "v"(hi[11]));
}
__device__ __forceinline__ void level_bit2(float (&lo)[8], const float (&hi)[8])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %3, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %4, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %5, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %5, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %6, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %6, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %7, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %7, %15, %15 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5]), "+&v"(lo[6]), "+&v"(lo[7])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]));
}
__device__ __forceinline__ void level_bit2(float (&lo)[6], const float (&hi)[6])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %3, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %4, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %5, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %5, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3]), "+&v"(lo[4]), "+&v"(lo[5])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]));
}
__device__ __forceinline__ void level_bit2(float (&lo)[4], const float (&hi)[4])
{
    asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1"
        : "+&v"(lo[0]), "+&v"(lo[1]), "+&v"(lo[2]), "+&v"(lo[3])
        : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
}

template <int N>
__device__ __forceinline__ void row_reduce_scatter(const float* val, int lane, float& d0, float& d1)
{
    static_assert(N == 16 || N == 24 || N == 32, "16, 24 or 32 values");
    constexpr int QUAD_XOR2 = 0x4E /* [2,3,0,1] */, QUAD_XOR1 = 0xB1 /* [1,0,3,2] */;
    const bool u1 = (lane & 2) != 0, u0 = (lane & 1) != 0;
    float a[N / 2], ah[N / 2], b[N / 4], bh[N / 4], c[N / 8];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { a[i] = val[i]; ah[i] = val[i + N / 2]; }
    level_bit3(a, ah);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { b[i] = a[i]; bh[i] = a[i + N / 4]; }
    level_bit2(b, bh);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) c[i] = pack_pair<QUAD_XOR2>(b[i], b[i + N / 8], u1);
    d0 = pack_pair<QUAD_XOR1>(c[0], c[1], u0);
    d1 = 0.f;
    if (N == 24) d1 = c[2] + dpp_mov<QUAD_XOR1>(c[2]);
    if (N == 32) d1 = pack_pair<QUAD_XOR1>(c[2], c[3], u0);
}

// The values whose row totals row_reduce_scatter<N> leaves in `lane`: v0 (d0) and v1 (d1; -1: none).
template <int N>
__device__ __forceinline__ void row_value_of_lane(int lane, int& v0, int& v1)
{
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
    if (N == 16) { v0 = b0 + 2 * b1 + 4 * b2 + 8 * b3; v1 = -1; }
    else if (N == 24) { v0 = b0 + 3 * b1 + 6 * b2 + 12 * b3; v1 = b0 ? -1 : 2 + 3 * b1 + 6 * b2 + 12 * b3; }
    else { v0 = b0 + 4 * b1 + 8 * b2 + 16 * b3; v1 = v0 + 2; }
}

}  // namespace dirt
