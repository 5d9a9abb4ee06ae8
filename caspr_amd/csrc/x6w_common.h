// x6w_common.h -- shared by the two kernels that keep 256 accumulators per lane in the accumulator file BY HAND
// (ode_bf16x6w.hip: the 128-point CNF sampling kernel; gemm_bf16x6w.hip: the 128-point x 512-channel pointwise conv):
// v_mfma_f32_32x32x16_bf16 on a literal a[..] range, v_accvgpr moves, one scheduling slot per MFMA, the exact three-way bf16 split
// in micro-steps.  Both files are compiled with -mllvm -amdgpu-mfma-vgpr-form (hipcc's own MFMAs must stay out of the AGPRs) and
// -fno-slp-vectorize (no packed f32 VALU beside MFMAs); every statement that writes an AGPR names all of a0..a255 as clobbered.
#pragma once
#include <type_traits>

#include "ode_x6.h"
#include "ode_x6w_agprs.h"

#define XW_INL __attribute__((always_inline))
#define XW_FENCE __builtin_amdgcn_sched_barrier(0)

template <int I, int N, class F>
__device__ __forceinline__ void xw_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        xw_for<I + 1, N>(f);
    }
}

// ---- the hand-managed accumulator file -------------------------------------------------------------------------------
// acc tile T += A * B.  NOPS: two wait states in front, for an operand a VALU instruction may just have written (hipcc does
// not see an MFMA in the statement and pads nothing)
template <int T, bool NOPS>
__device__ __forceinline__ void xw_mfma_a(bf16x8 af, bf16x8 bf)
{
    if constexpr (NOPS)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(af), "v"(bf), "i"(16 * T), "i"(16 * T + 15) : XW_ACLOB);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(af), "v"(bf), "i"(16 * T), "i"(16 * T + 15) : XW_ACLOB);
}
template <int N>
__device__ __forceinline__ float xw_acc_rd()
{
    float x;
#if XW_EXP & 256
    asm volatile("v_mov_b32 %0, 1.0" : "=v"(x));
#else
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(N));
#endif
    return x;
}
template <int N>
__device__ __forceinline__ void xw_acc_wr(float x)
{
#if XW_EXP & 256
    asm volatile("" : : "v"(x));
#else
    asm volatile("v_accvgpr_write_b32 a%c1, %0" : : "v"(x), "i"(N) : XW_ACLOB);
#endif
}
template <int N>
__device__ __forceinline__ void xw_acc_zero()
{
    asm volatile("v_accvgpr_write_b32 a%c0, 0" : : "i"(N) : XW_ACLOB);
}

// ---- the gated softplus + exact three-way split of a value pair, in micro-steps (one per scheduling slot) ----------------
struct XwPair {
    float x0, x1, u0, u1;
    unsigned p1, p2;
    float r0, r1;
};
__device__ __forceinline__ void xw_sp1(XwPair &p)     // u = 2^(-|x| log2 e)
{
#if XW_EXP & 128
    p.u0 = fabsf(p.x0) * -1.44269504088896341f * p.x1;
    p.u1 = fabsf(p.x1) * -1.44269504088896341f * p.x0;
#else
    p.u0 = __builtin_amdgcn_exp2f(fabsf(p.x0) * -1.44269504088896341f);
    p.u1 = __builtin_amdgcn_exp2f(fabsf(p.x1) * -1.44269504088896341f);
#endif
}
__device__ __forceinline__ void xw_sp2(XwPair &p)     // u = log2(1 + u)
{
#if XW_EXP & 128
    p.u0 = (1.0f + p.u0) * p.x1;
    p.u1 = (1.0f + p.u1) * p.x0;
#else
    p.u0 = __builtin_amdgcn_logf(1.0f + p.u0);
    p.u1 = __builtin_amdgcn_logf(1.0f + p.u1);
#endif
}
__device__ __forceinline__ void xw_sp3(XwPair &p)     // x = max(x, 0) + ln 2 * u  == softplus_fast(x)
{
    p.x0 = fmaxf(p.x0, 0.0f) + 0.69314718055994531f * p.u0;
    p.x1 = fmaxf(p.x1, 0.0f) + 0.69314718055994531f * p.u1;
}
// the exact split of xc_split_pair in three steps, with SCALAR subtractions: a v_pk_add_f32 costs ~11 matrix-pipe cycles more
// than the two v_sub_f32 it replaces when it sits beside MFMAs (MI355X guide, "price of one filler"); the file is compiled
// with -fno-slp-vectorize so that hipcc does not re-pack them
__device__ __forceinline__ unsigned xw_cvt_pk(float lo, float hi)
{
    const xc_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, xc_bf16x2));
}
__device__ __forceinline__ void xw_split1(XwPair &p)  // first plane + remainder
{
    p.p1 = xw_cvt_pk(p.x0, p.x1);
    p.r0 = p.x0 - __uint_as_float(p.p1 << 16);
    p.r1 = p.x1 - __uint_as_float(p.p1 & 0xffff0000u);
}
__device__ __forceinline__ void xw_split2(XwPair &p)
{
    p.p2 = xw_cvt_pk(p.r0, p.r1);
    p.r0 = p.r0 - __uint_as_float(p.p2 << 16);
    p.r1 = p.r1 - __uint_as_float(p.p2 & 0xffff0000u);
}
__device__ __forceinline__ void xw_split3(const XwPair &p, u32x4 (&bw)[3], int q)
{
    bw[0][q] = p.p1;
    bw[1][q] = p.p2;
    bw[2][q] = xw_cvt_pk(p.r0, p.r1);
}


// ---- cross-lane reductions over the 32 columns of a 32 x 32 accumulator tile, for 16 values per lane at once ---------------------------
// In the D layout a lane (j = lane & 31, h = lane >> 5) holds, per tile, 16 rows (r & 3) + 8 (r >> 2) + 4 h of ONE column j: a per-row
// statistic over the columns is a reduction over the 32 lanes of a half-wave for each of the 16 registers.  Done value by value that is
// 5 cross-lane steps x 16 (and the 16-lane -> 32-lane step was a ds_bpermute).  Here the 16 values are reduced TOGETHER, the register
// count halving at every level -- a transposition folded into the reduction:
//   S   v_permlane16_swap (gfx950) of the pair (x[2i], x[2i+1]): afterwards one register holds both 16-lane rows of x[2i] in row 0 and of
//       x[2i+1] in row 1 (vdst | src), so vdst (+) src is the 2-row combination of x[2i] in the even rows and of x[2i+1] in the odd ones;
//   R8  y (+) row_ror:8 of itself for both registers, then banks 0,1 of a row keep y[2i]'s, banks 2,3 take y[2i+1]'s (masked DPP move);
//   R4  z[2i] (+) its lane + 4 in banks 0,2, z[2i+1] (+) its lane - 4 in banks 1,3;
//   Q   two quad_perm steps.
// 38 instructions instead of 80 + 16 LDS crossbar operations.  Result: two registers w[0], w[1]; lane 32 h + 16 p + 4 b + q (q: four
// copies) of w[i] holds the statistic of register index r = 8 i + 4 (b & 1) + 2 (b >> 1) + p, i.e. of row (r & 3) + 8 (r >> 2) + 4 h.
__device__ __forceinline__ void xw_swap16(float &a, float &b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
template <int BANKS>
__device__ __forceinline__ float xw_bank_sel(float keep, float take)      // `take` in the banks of BANKS, `keep` elsewhere
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(keep), __float_as_uint(take), 0xE4, 0xf, BANKS, false));
}
struct XwAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct XwMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct XwMin { __device__ __forceinline__ float operator()(float a, float b) const { return fminf(a, b); } };
struct XwFirst { __device__ __forceinline__ float operator()(float a, float) const { return a; } };      // values already equal in all lanes: transposition only
template <class OP>
__device__ __forceinline__ void xw_treduce16(float (&x)[16], float (&w)[2], OP op)
{
    constexpr bool PICK = std::is_same<OP, XwFirst>::value;
    float y[8], z[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = x[2 * i], b = x[2 * i + 1];
        xw_swap16(a, b);
        y[i] = op(a, b);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (PICK)
            z[i] = xw_bank_sel<0xC>(y[2 * i], y[2 * i + 1]);
        else         // each register with its own lane + 8 first (row_ror:8), THEN banks 0,1 from y[2i], banks 2,3 from y[2i+1]
            z[i] = xw_bank_sel<0xC>(op(y[2 * i], dpp_mov<0x128>(y[2 * i])), op(y[2 * i + 1], dpp_mov<0x128>(y[2 * i + 1])));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (PICK) {
            w[i] = xw_bank_sel<0xA>(z[2 * i], z[2 * i + 1]);
        } else {
            const float t0 = op(z[2 * i], dpp_mov<0x12C>(z[2 * i]));             // row_ror:12: lane L reads lane L + 4 (valid in banks 0, 2)
            const float t1 = op(z[2 * i + 1], dpp_mov<0x124>(z[2 * i + 1]));     // row_ror:4:  lane L reads lane L - 4 (valid in banks 1, 3)
            float t = xw_bank_sel<0xA>(t0, t1);
            t = op(t, dpp_mov<0x4E>(t));
            w[i] = op(t, dpp_mov<0xB1>(t));
        }
    }
}
// x (+) the same lane of the other 16-lane row of its half-wave, in both rows (after a 16-lane all-reduce: the 32-lane total everywhere)
__device__ __forceinline__ float xw_rows_add(float x)
{
    float a = x, b = x;
    xw_swap16(a, b);
    return a + b;
}
