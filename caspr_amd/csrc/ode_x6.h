// ode_x6.h -- shared between the two bf16x6 point-CNF kernels (ode_bf16x6.hip: 64 points per workgroup, 16x16x32 MFMA,
// sampling + divergence variants; ode_bf16x6w.hip: 128 points per workgroup, 32x32x16 MFMA, sampling).
#pragma once
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define XC_H 512

__device__ __forceinline__ void xc_split(float x, float &h1, float &h2, float &h3)
{
    h1 = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - h1;
    h2 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float r2 = r1 - h2;
    h3 = __uint_as_float(__float_as_uint(r2) & 0xffff0000u);
}
// the same exact split for a pair of values with the hardware round-to-nearest conversion (v_cvt_pk_bf16_f32): the three
// packed words are the pair's entries of the three planes.  Exact as well: the remainder after rounding 24 bits to 8 has
// at most 15 significant bits, after the second rounding at most 7.  4.5 VALU operations per value instead of 5.5.
typedef __bf16 xc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float xc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xc_split_pair(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3)
{
    xc_f32x2 v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, xc_bf16x2));
    xc_f32x2 h = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    v = v - h;
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, xc_bf16x2));
    h = (xc_f32x2){__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xffff0000u)};
    v = v - h;
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, xc_bf16x2));
}
__device__ __forceinline__ unsigned xc_pack(float lo, float hi) { return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u); }

struct CnfX6Args {
    const float *y_in, *hyper, *tcol, *w0, *b0, *b1, *b2, *w3, *b3, *mbn_in, *mbn_out;
    const float *e, *logp_in;     // DIV only: Hutchinson noise (BT,n,3), initial log-density (BT,n) or NULL
    float *logp_out;              // DIV only
    const unsigned char *w1x, *w2x;
    float *y_out;
    int ldh, n, steps, reverse;
    float t_end;
    unsigned long long *trace;   // debug build: s_memtime stamps of workgroup (0,0), thread 0, RK4 step 0 / stage 1
    int diag;                    // debug build: timing experiments (CASPR_X6_DIAG)
};

// ---- the 128-point kernel (ode_bf16x6w.hip)
#define XW_PTS 128
#define XW_FRAG 1024                      // one A fragment of v_mfma_f32_32x32x16_bf16: 64 lanes x 16 B
#define XW_PIECE (2 * 4 * 3 * XW_FRAG)    // 24 KB: [k-step 2][row tile 4][plane 3][fragment]
#define XW_PACK (4L * 16 * XW_PIECE)      // one hidden layer: [row quarter 4][k chunk 16][piece] = 1.5 MB
int caspr_cnf_x6w_launch(const CnfX6Args &a, int BT, hipStream_t stream) __attribute__((visibility("hidden")));
int caspr_cnf_x6w_pack(const float *w, int ldw, unsigned char *out, hipStream_t stream) __attribute__((visibility("hidden")));
