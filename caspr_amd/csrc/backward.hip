// backward.hip -- training tier, part 1: gradient kernels of the pointwise conv -> GroupNorm -> ReLU blocks
// (reached in the reference through autograd from train_utils.py:173).
//
//   data gradient   dX = dY . W            : the forward conv1x1 kernel with the transposed packed weight (no new code)
//   weight gradient dW = dY^T . in(X)      : conv1x1_wgrad_kernel  (f32 MFMA, contraction over points, split into
//                                            point slabs, fixed-order slab reduction -> deterministic)
//   bias gradient   db = colsum(dY)        : rides along in the weight-gradient kernel's k-tile-0 blocks
//   GroupNorm(+ReLU) backward              : gn_bwd_partial / finalize / apply
#include <stdlib.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------
// weight gradient.  dW[co][k] = sum over rows r = (b,p) of dY[r][co] * in(X[r][k]).
// Block tile 128 (co) x 128 (k), 2x2 waves of 64x64, 32 rows per LDS stage (both operands row-major
// [row][channel] with a 16-float pad: the b32 fragment reads of rows r and r+1 then fall on disjoint banks).
// MFMA 16x16x4: A[i = co][kk = row], B[kk = row][j = k].
// ---------------------------------------------------------------------------------------------
#define WG_T 128
#define WG_ROWS 32
#define WG_LD (WG_T + 16)

__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const float *__restrict__ dY, int lddy,
                                                            const float *__restrict__ X, int ldx,
                                                            const float *__restrict__ in_scale,
                                                            const float *__restrict__ in_shift, int in_relu,
                                                            int relu_from, long R, int P, int Cin, int Cout,
                                                            long rows_per_slab, float *__restrict__ part,
                                                            float *__restrict__ bpart)
{
    __shared__ __attribute__((aligned(16))) float sA[WG_ROWS * WG_LD], sB[WG_ROWS * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, j = lane & 15;
    const int co0 = blockIdx.x * WG_T, k0 = blockIdx.y * WG_T;
    const int slab = blockIdx.z;
    const long r_beg = (long)slab * rows_per_slab;
    const long r_end = (r_beg + rows_per_slab) < R ? (r_beg + rows_per_slab) : R;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: 32 rows x 128 channels = 1024 float4 per operand, 4 per thread: row = f >> 5, c4 = f & 31.
    // Software pipeline: the global loads of stage s+1 are issued before the MFMAs of stage s and stay in flight
    // under them (sched_barrier pins the issue point -- hipcc otherwise sinks the loads to their first use).
    f32x4 va[4], vb[4];
    f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_stage = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            const int row = f >> 5, c = (f & 31) * 4;
            const long r = r0 + row;
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r < r_end) {
                if (co0 + c < Cout) a = ld4(dY + r * lddy + co0 + c);   // lddy >= roundup4(Cout)
                if (k0 + c < Cin) b = ld4(X + r * ldx + k0 + c);
            }
            va[i] = a;
            vb[i] = b;
        }
    };
    // interior tiles (all 128 channels of both operands valid, all 32 rows inside the slab) skip every mask
    const bool full_cols = co0 + WG_T <= Cout && k0 + WG_T <= Cin;
    auto store_stage = [&](long r0) {   // masking and the fused input transform run here, when the data has arrived
        const bool interior = full_cols && r0 + WG_ROWS <= r_end;   // block-uniform
        // batch entry of the stage's first row, once per stage (a 64-bit division per float4 was the cost of the fused path)
        long bi0 = 0;
        int rem0 = 0;
        if (in_scale) {
            bi0 = r0 / P;
            rem0 = (int)(r0 - bi0 * P);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            const int row = f >> 5, c = (f & 31) * 4;
            const long r = r0 + row;
            f32x4 a = va[i], b = vb[i];
            const int k = k0 + c;
            if (in_scale && (interior || (r < r_end && k < Cin))) {
                long bi = bi0;
                int pr = rem0 + row;
                while (pr >= P) { pr -= P; ++bi; }
                const f32x4 s4 = ld4(in_scale + bi * Cin + k), t4 = ld4(in_shift + bi * Cin + k);
                b = b * s4 + t4;
                if (in_relu && k >= relu_from) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = b[q] > 0.f ? b[q] : 0.f;
                }
            }
            if (!interior) {
                if (r < r_end) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (co0 + c + q >= Cout) a[q] = 0.f;
                        if (k + q >= Cin) b[q] = 0.f;
                    }
                } else {
                    a = (f32x4){0.f, 0.f, 0.f, 0.f};
                    b = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            bsum = bsum + a;   // bias gradient: column sums of dY ride along (used by the k-tile-0 blocks only)
            st4(&sA[row * WG_LD + c], a);
            st4(&sB[row * WG_LD + c], b);
        }
    };
    load_stage(r_beg);
    for (long r0 = r_beg; r0 < r_end; r0 += WG_ROWS) {
        __syncthreads();   // previous stage fully consumed
        store_stage(r0);
        __syncthreads();
        if (r0 + WG_ROWS < r_end) load_stage(r0 + WG_ROWS);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < WG_ROWS / 4; ++ks) {
            float af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = sA[(ks * 4 + g) * WG_LD + wm * 64 + mi * 16 + j];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = sB[(ks * 4 + g) * WG_LD + wn * 64 + ni * 16 + j];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(af[mi], bf[ni], acc[mi][ni]);
        }
    }
    // bias-gradient partial of this slab: thread (row phase tid >> 5, column quad tid & 31) summed rows phase, phase+8, ..
    // of its quad; combine the 8 phases in a fixed order through LDS (sA is free after the last MFMA)
    if (bpart && blockIdx.y == 0) {
        __syncthreads();
        st4(&sA[(tid >> 5) * WG_LD + (tid & 31) * 4], bsum);
        __syncthreads();
        if (tid < WG_T && co0 + tid < Cout) {
            float t = 0.f;
#pragma unroll
            for (int ph = 0; ph < 8; ++ph) t += sA[ph * WG_LD + tid];
            bpart[(long)slab * Cout + co0 + tid] = t;
        }
    }
    // partial slab: part[slab][co][k]  (D fragment: row = co = 4g + r, col = k = j)
    float *pp = part + (long)slab * Cout * Cin;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 64 + mi * 16 + 4 * g + r, k = k0 + wn * 64 + ni * 16 + j;
                if (co < Cout && k < Cin) pp[(long)co * Cin + k] = acc[mi][ni][r];
            }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient with the products on the bf16 matrix pipe in the exact three-way split of gemm_bf16x6.hip (both
// operands are activations here, so both are split on the way to LDS).  Same tile (128 x 128, 2 x 2 waves of 64 x 64, 32
// rows per stage), same slab decomposition and fixed-order reduction.
//   A bf16 MFMA operand wants, per lane, 8 consecutive CONTRACTION indices (rows r) of one channel -- the row-major [row][channel]
//   data the wrong way round.  The LDS image is therefore kept row-major, as 16-channel subtiles [32 rows][16 channels] of
//   bf16 (1 KB + 32 B pad each, three planes per operand), and read with ds_read_b64_tr_b16: a 16-lane group reads a 4-row x
//   16-channel block and every lane receives the 4 rows of ITS channel.  Lane group g takes rows 4g..4g+3 and 16+4g..16+4g+3 as
//   its 8 contraction slots -- the same permutation of the 32 rows for both operands, which is all a contraction needs -- so
//   every read is 512 contiguous bytes (conflict-free), and the 8-byte stores of the staging pass (4 channels of one row) hit 64
//   distinct banks per 32 lanes thanks to the 32-byte pad between subtiles.
// ---------------------------------------------------------------------------------------------
typedef short wg6_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg6_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wg6_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wg6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg6_f32x2 __attribute__((ext_vector_type(2)));
#define WG6_SUB (32 * 32 + 32)          // bytes of a [32][16] bf16 subtile + pad

// x0, x1 -> their entries of the three planes (exact: round-to-nearest conversion, exact remainders; see gemm_bf16x6.hip)
__device__ __forceinline__ void wg6_split_pair(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3)
{
    wg6_f32x2 v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg6_bf16x2));
    wg6_f32x2 h = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    v = v - h;
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg6_bf16x2));
    h = (wg6_f32x2){__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xffff0000u)};
    v = v - h;
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg6_bf16x2));
}

// NW = 4: the 128 x 128 tile of the f32 kernel (2 x 2 waves of 64 x 64, two workgroups per CU); NW = 8: 256 x 256 (2 x 4 waves of
// 128 x 64, one workgroup per CU): every staged element then meets 256 instead of 128 channels of the other operand -- half the
// split arithmetic and half the re-reads of dY and X per product -- at the price of more padding on widths like 1600 (7 x 256).
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv1x1_wgrad_bf16x6_kernel(const float *__restrict__ dY, int lddy,
                                                                      const float *__restrict__ X, int ldx,
                                                                      const float *__restrict__ in_scale,
                                                                      const float *__restrict__ in_shift, int in_relu,
                                                                      int relu_from, long R, int P, int Cin, int Cout,
                                                                      long rows_per_slab, float *__restrict__ part,
                                                                      float *__restrict__ bpart)
{
    constexpr int NT = 64 * NW, T = 32 * NW;          // threads; tile edge (channels of either operand)
    constexpr int MI = NW == 4 ? 4 : 8;                // row tiles per wave
    constexpr int PLANE = (T / 16) * WG6_SUB, OPER = 3 * PLANE;
    constexpr int CQ = T / 4, LD = T + 16;             // float4 per staged row; row stride of the bias scratch
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * OPER];
    unsigned char *sA = lds, *sB = lds + OPER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;
    const int g = lane >> 4, j = lane & 15;
    const int co0 = blockIdx.x * T, k0 = blockIdx.y * T;
    const int slab = blockIdx.z;
    const long r_beg = (long)slab * rows_per_slab;
    const long r_end = (r_beg + rows_per_slab) < R ? (r_beg + rows_per_slab) : R;

    f32x4 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 va[4], vb[4];
    f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_stage = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + NT * i;
            const int row = f / CQ, c = (f % CQ) * 4;
            const long r = r0 + row;
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r < r_end) {
                if (co0 + c < Cout) a = ld4(dY + r * lddy + co0 + c);   // lddy >= roundup4(Cout)
                if (k0 + c < Cin) b = ld4(X + r * ldx + k0 + c);
            }
            va[i] = a;
            vb[i] = b;
        }
    };
    const bool full_cols = co0 + T <= Cout && k0 + T <= Cin;
    auto put = [&](unsigned char *base, int row, int c, const f32x4 &v) __attribute__((always_inline)) {
        unsigned a1, a2, a3, b1, b2, b3;
        wg6_split_pair(v[0], v[1], a1, a2, a3);
        wg6_split_pair(v[2], v[3], b1, b2, b3);
        unsigned char *dst = base + (c >> 4) * WG6_SUB + row * 32 + (c & 15) * 2;
        *(wg6_u32x2 *)(dst) = (wg6_u32x2){a1, b1};
        *(wg6_u32x2 *)(dst + PLANE) = (wg6_u32x2){a2, b2};
        *(wg6_u32x2 *)(dst + 2 * PLANE) = (wg6_u32x2){a3, b3};
    };
    auto store_stage = [&](long r0) {   // masking, the fused input transform and the split run here, when the data has arrived
        const bool interior = full_cols && r0 + WG_ROWS <= r_end;   // block-uniform
        long bi0 = 0;
        int rem0 = 0;
        if (in_scale) {
            bi0 = r0 / P;
            rem0 = (int)(r0 - bi0 * P);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + NT * i;
            const int row = f / CQ, c = (f % CQ) * 4;
            const long r = r0 + row;
            f32x4 a = va[i], b = vb[i];
            const int k = k0 + c;
            if (in_scale && (interior || (r < r_end && k < Cin))) {
                long bi = bi0;
                int pr = rem0 + row;
                while (pr >= P) { pr -= P; ++bi; }
                const f32x4 s4 = ld4(in_scale + bi * Cin + k), t4 = ld4(in_shift + bi * Cin + k);
                b = b * s4 + t4;
                if (in_relu && k >= relu_from) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = b[q] > 0.f ? b[q] : 0.f;
                }
            }
            if (!interior) {
                if (r < r_end) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (co0 + c + q >= Cout) a[q] = 0.f;
                        if (k + q >= Cin) b[q] = 0.f;
                    }
                } else {
                    a = (f32x4){0.f, 0.f, 0.f, 0.f};
                    b = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            bsum = bsum + a;
            put(sA, row, c, a);
            put(sB, row, c, b);
        }
    };
    // fragment of subtile `sub` of an operand plane: lane (g, i) supplies the address of 4 channels of row 4g + (i >> 2) (then
    // 16 + ...), and receives rows 4g .. 4g+3 (16+4g ..) of channel i
    auto frag = [&](const unsigned char *plane, int sub) __attribute__((always_inline)) -> wg6_bf16x8 {
        const unsigned char *p = plane + sub * WG6_SUB + (4 * g + (j >> 2)) * 32 + (j & 3) * 8;
        const wg6_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg6_s16x4 *)(p));
        const wg6_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg6_s16x4 *)(p + 16 * 32));
        return (wg6_bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    load_stage(r_beg);
    for (long r0 = r_beg; r0 < r_end; r0 += WG_ROWS) {
        __syncthreads();   // previous stage fully consumed
        store_stage(r0);
        __syncthreads();
        if (r0 + WG_ROWS < r_end) load_stage(r0 + WG_ROWS);
        __builtin_amdgcn_sched_barrier(0);
        wg6_bf16x8 bfr[3][4];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bfr[pl][ni] = frag(sB + pl * PLANE, wn * 4 + ni);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            wg6_bf16x8 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[pl] = frag(sA + pl * PLANE, wm * MI + mi);
            // smallest terms first (plane 0 = leading part)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bfr[0][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bfr[1][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[2][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bfr[0][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[1][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[0][ni], acc[mi][ni], 0, 0, 0);
        }
    }
    if (bpart && blockIdx.y == 0) {
        float *sF = (float *)lds;
        __syncthreads();
        st4(&sF[(tid / CQ) * LD + (tid % CQ) * 4], bsum);
        __syncthreads();
        if (tid < T && co0 + tid < Cout) {
            float t = 0.f;
#pragma unroll
            for (int ph = 0; ph < 8; ++ph) t += sF[ph * LD + tid];
            bpart[(long)slab * Cout + co0 + tid] = t;
        }
    }
    float *pp = part + (long)slab * Cout * Cin;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (16 * MI) + mi * 16 + 4 * g + r, k = k0 + wn * 64 + ni * 16 + j;
                if (co < Cout && k < Cin) pp[(long)co * Cin + k] = acc[mi][ni][r];
            }
}

// out[i] (+)= sum_s part[s][i]  in a fixed order.  Few slabs: one thread per element walks them (coalesced over i).
// Many slabs (small weights of the set-abstraction MLPs, up to 4096 slabs): 32 lanes per element each take every 32nd
// slab, then a fixed xor-shuffle tree -- a single thread walking thousands of slabs was a 0.6 ms latency chain.
__global__ void slab_reduce_kernel(const float *__restrict__ part, long n, int S, int accumulate, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int k = 0; k < S; ++k) s += part[(long)k * n + i];
    out[i] = s;
}

__global__ __launch_bounds__(256) void slab_reduce_wide_kernel(const float *__restrict__ part, long n, int S, int accumulate,
                                                               float *__restrict__ out)
{
    const int l32 = threadIdx.x & 31;
    const long i = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    float s = 0.f;
    if (i < n)
        for (int k = l32; k < S; k += 32) s += part[(long)k * n + i];
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, 32);
    if (i < n && l32 == 0) out[i] = (accumulate ? out[i] : 0.f) + s;
}

static void launch_slab_reduce(const float *part, long n, int S, int accumulate, float *out, hipStream_t st)
{
    // many slabs AND few elements: 32 lanes per element (its lanes read addresses n floats apart); with 64K+ elements one thread per
    // element keeps every load coalesced and there are enough threads in flight to cover the S dependent adds
    if (S > 64 && n < 65536)
        slab_reduce_wide_kernel<<<dim3((unsigned)((n + 7) / 8)), dim3(256), 0, st>>>(part, n, S, accumulate, out);
    else
        slab_reduce_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(part, n, S, accumulate, out);
}

// Weight gradient of a layer with up to four outputs (the CNF's 512 -> 3 output layer): dW[j][c] = sum_r dY[r][j] X[r][c] is one
// pass over X -- a thread keeps four channels x four outputs, a wave reads whole rows (dY's row is a scalar load), the four waves
// of a workgroup take every fourth row of the slab.  The 128 x 128 tile kernel spent a full tile of products on three live
// rows (228 us per call at cfg-3; this one is bound by reading X).  Same slab partials, same fixed-order reduction.
__global__ __launch_bounds__(256) void conv1x1_wgrad_skinny_kernel(const float *__restrict__ dY, int lddy, const float *__restrict__ X, int ldx,
                                                                   long R, int Cin, int Cout, long rows_per_slab, float *__restrict__ part)
{
    __shared__ float s_red[3][64][16];
    const int lane = threadIdx.x & 63;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 256 + lane * 4;
    const int slab = blockIdx.y;
    const long r_beg = (long)slab * rows_per_slab;
    const long r_end = (r_beg + rows_per_slab) < R ? (r_beg + rows_per_slab) : R;
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
    const bool ok = c < Cin;             // Cin % 4 == 0
#pragma unroll 4
    for (long r = r_beg + sub; r < r_end; r += 4) {
        const f32x4 d = ld4(dY + r * lddy);                                  // wave-uniform row (lddy >= 4, zero past Cout)
        const f32x4 x = ok ? ld4(X + r * ldx + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[j][q] = fmaf(d[j], x[q], acc[j][q]);
    }
    if (sub > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_red[sub - 1][lane][4 * j + q] = acc[j][q];
    }
    __syncthreads();
    if (sub == 0 && ok) {
        float *pp = part + (long)slab * Cout * Cin;
        for (int j = 0; j < Cout; ++j) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (acc[j][q] + s_red[0][lane][4 * j + q]) + (s_red[1][lane][4 * j + q] + s_red[2][lane][4 * j + q]);
            st4(pp + (long)j * Cin + c, v);
        }
    }
}

// number of row slabs: enough workgroups (tiles x slabs ~ 2048) to fill 256 CUs even when the weight is one tile,
// at least 512 rows per slab, at most 4096 slabs
// ---------------------------------------------------------------------------------------------
// weight gradient of a NARROW conv over many rows (the training encoder's set-abstraction MLPs: 1.3 - 2.6 M grouped rows, 9 - 131
// inputs, 16 - 128 outputs).  The call is a stream -- (Cin + Cout) floats per row against a few hundred MFMA cycles -- and the
// 128 x 128 tile above spent it staging padding through LDS (0.4 - 0.8 ms per call, 1 - 2 TB/s of useful traffic).  Here no LDS stage
// at all: the f32 MFMA 16x16x4 contracts over 4 ROWS, its A operand (lane (g, j): dY[row 4t + g][16 ta + j]) and B operand
// (X[row 4t + g][16 tb + j]) are single floats of a 64-byte piece of a row, loaded straight from global memory, U steps (16 U rows per
// workgroup: the four waves take consecutive 4-row steps) in flight before the first MFMA.  A wave owns NTA x NTB output tiles
// (<= 16: 64 accumulators; wider shapes stay on the LDS tile).  The bias gradient rides along on the VALU (the lane's own
// dY values).  Waves are combined in wave order through LDS, slabs in slab order by slab_reduce: deterministic.
// ---------------------------------------------------------------------------------------------
template <int NTA, int NTB, int U>
__global__ __launch_bounds__(256) void conv1x1_wgrad_narrow_kernel(const float *__restrict__ dY, int lddy, const float *__restrict__ X,
                                                                   int ldx, long R, int Cin, int Cout, long rows_per_slab,
                                                                   float *__restrict__ part, float *__restrict__ bpart)
{
    __shared__ __attribute__((aligned(16))) float red[NTA * NTB * 256];
    __shared__ float bred[4][NTA * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int slab = blockIdx.x, tb0 = blockIdx.y * NTB;
    const long r_beg = (long)slab * rows_per_slab;
    const long r_end = (r_beg + rows_per_slab) < R ? (r_beg + rows_per_slab) : R;

    bool oka[NTA], okb[NTB];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) oka[ta] = 16 * ta + j < Cout;
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) okb[tb] = 16 * (tb0 + tb) + j < Cin;
    f32x4 acc[NTA][NTB];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[ta][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[NTA];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) bsum[ta] = 0.f;

    for (long r0 = r_beg; r0 < r_end; r0 += 16 * U) {
        float a[U][NTA], b[U][NTB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = r0 + 16 * u + 4 * wave + g;
            const bool ok = row < r_end;
            const float *pa = dY + row * lddy + j;
            const float *pb = X + row * ldx + 16 * tb0 + j;
#pragma unroll
            for (int ta = 0; ta < NTA; ++ta) a[u][ta] = (ok && oka[ta]) ? pa[16 * ta] : 0.f;
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) b[u][tb] = (ok && okb[tb]) ? pb[16 * tb] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ta = 0; ta < NTA; ++ta) {
                bsum[ta] += a[u][ta];
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb) acc[ta][tb] = mfma16(a[u][ta], b[u][tb], acc[ta][tb]);
            }
    }

    // bias gradient: the four row phases g of a wave, then the four waves, in a fixed order
    if (bpart && blockIdx.y == 0) {
#pragma unroll
        for (int ta = 0; ta < NTA; ++ta) {
            float t = bsum[ta];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (g == 0) bred[wave][16 * ta + j] = t;
        }
    }
    // the four waves' tiles: waves 1..3 hand theirs to wave 0 one after the other
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb) st4(&red[((ta * NTB + tb) * 64 + lane) * 4], acc[ta][tb]);
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb) acc[ta][tb] = acc[ta][tb] + ld4(&red[((ta * NTB + tb) * 64 + lane) * 4]);
        }
    }
    if (wave != 0) return;
    if (bpart && blockIdx.y == 0) {
#pragma unroll
        for (int ta = 0; ta < NTA; ++ta)
            if (g == 0 && oka[ta]) bpart[(long)slab * Cout + 16 * ta + j] = ((bred[0][16 * ta + j] + bred[1][16 * ta + j]) + bred[2][16 * ta + j]) + bred[3][16 * ta + j];
    }
    float *pp = part + (long)slab * Cout * Cin;
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * ta + 4 * g + r, k = 16 * (tb0 + tb) + j;
                if (co < Cout && k < Cin) pp[(long)co * Cin + k] = acc[ta][tb][r];
            }
}

// -> true if the narrow kernel took the call
static bool wgrad_narrow_launch(hipStream_t st, const float *dY, int lddy, const float *X, int ldx, long R, int Cin, int Cout, int S, float *part,
                                float *bpart)
{
    const int tA = ceil_div(Cout, 16), tB = ceil_div(Cin, 16);
    const int nta = tA > 4 ? 8 : tA > 2 ? 4 : tA;
    int ntb = tB > 4 ? 8 : tB > 2 ? 4 : tB;
    // measured (cfg-3 step): with more than 16 tiles per wave (280 registers, one workgroup per CU) or a second pass over dY the 128 x 128
    // LDS tile is the faster one again (96 -> 128: 0.40 vs 0.22 ms, 131 -> 64: 0.37 vs 0.31, 64 -> 96: 0.20 vs 0.18)
    if (nta * ntb > 16 || ntb < tB) return false;
    const long rps = ((R + S - 1) / S + 127) / 128 * 128;   // a multiple of 16 U for both U
    dim3 grid(S, ceil_div(tB, ntb));
#define WN(A, B_, U_) conv1x1_wgrad_narrow_kernel<A, B_, U_><<<grid, dim3(256), 0, st>>>(dY, lddy, X, ldx, R, Cin, Cout, rps, part, bpart)
#define WN_ROW(A)                                     \
    if (nta == A) {                                   \
        if (ntb == 1) WN(A, 1, 8);                    \
        else if (ntb == 2) WN(A, 2, 8);               \
        else if (ntb == 4) WN(A, (A * 4 <= 16 ? 4 : 1), 8); \
        else WN(A, (A * 8 <= 16 ? 8 : 1), 4);         \
        return true;                                  \
    }
    WN_ROW(1) WN_ROW(2) WN_ROW(4) WN_ROW(8)
#undef WN_ROW
#undef WN
    return false;
}

static int pick_slabs(long R, int Cin, int Cout)
{
    const long tiles = (long)ceil_div(Cout, WG_T) * ceil_div(Cin, WG_T);
    long s = (2048 + tiles - 1) / tiles;
    const long by_rows = (R + 511) / 512;
    if (s > by_rows) s = by_rows;
    if (s > 4096) s = 4096;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" long caspr_wgrad_ws_bytes(long R, int Cin, int Cout)
{
    return (long)pick_slabs(R, Cin, Cout) * ((long)Cout * Cin + Cout) * 4 + 256;
}

static void wgrad_launch(int kind, dim3 grid, hipStream_t st, const float *dY, int lddy, const float *X, int ldx, const float *in_scale,
                         const float *in_shift, int in_relu, int in_relu_from, long R, int P, int Cin, int Cout, long rps, float *part,
                         float *bpart)
{
    if (kind == 2)
        conv1x1_wgrad_bf16x6_kernel<8><<<grid, dim3(512), 0, st>>>(dY, lddy, X, ldx, in_scale, in_shift, in_relu, in_relu_from, R, P, Cin, Cout, rps,
                                                                   part, bpart);
    else if (kind == 1)
        conv1x1_wgrad_bf16x6_kernel<4><<<grid, dim3(256), 0, st>>>(dY, lddy, X, ldx, in_scale, in_shift, in_relu, in_relu_from, R, P, Cin, Cout, rps,
                                                                   part, bpart);
    else
        conv1x1_wgrad_kernel<<<grid, dim3(256), 0, st>>>(dY, lddy, X, ldx, in_scale, in_shift, in_relu, in_relu_from, R, P, Cin, Cout, rps, part,
                                                         bpart);
}

static int wgrad_impl(bool bf16x6, const float *dY, int lddy, const float *X, int ldx, const float *in_scale, const float *in_shift, int in_relu,
                      int in_relu_from, int B, int P, int Cin, int Cout, float *dW, float *dbias, int accumulate, void *ws, long ws_bytes,
                      void *stream)
{
    CASPR_REQUIRE(dY && X && dW && ws && B > 0 && P > 0 && Cin > 0 && Cout > 0, "conv1x1_wgrad: bad arguments");
    CASPR_REQUIRE(lddy % 4 == 0 && lddy >= ((Cout + 3) & ~3) && ldx % 4 == 0 && ldx >= ((Cin + 3) & ~3),
                  "conv1x1_wgrad: row strides must be multiples of 4 and cover the channels (lddy=%d ldx=%d)", lddy, ldx);
    CASPR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (in_scale == nullptr || Cin % 4 == 0),
                  "conv1x1_wgrad: in_scale/in_shift must be given together and need Cin %% 4 == 0");
    const long R = (long)B * P;
    CASPR_REQUIRE(ws_bytes >= caspr_wgrad_ws_bytes(R, Cin, Cout), "conv1x1_wgrad: workspace too small");
    int S = pick_slabs(R, Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)ws;
    if (Cout <= 4 && Cin % 4 == 0 && Cin >= 256 && !in_scale && !dbias && lddy >= 4) {
        // dY rows are read four wide: columns past Cout must be zero (the callers pad gradients with zeros, _pad4)
        const long rps4 = ((R + S - 1) / S + 3) / 4 * 4;
        conv1x1_wgrad_skinny_kernel<<<dim3(ceil_div(Cin, 256), S), dim3(256), 0, st>>>(dY, lddy, X, ldx, R, Cin, Cout, rps4, part);
        launch_slab_reduce(part, (long)Cout * Cin, S, accumulate, dW, st);
        CASPR_CHECK_LAUNCH("conv1x1_wgrad");
        return CASPR_OK;
    }
    // narrow convs over many rows (both matrix modes: the f32 products are at least as exact as the split's)
    if (!in_scale && Cout <= 128 && Cin <= 160 && R >= 65536) {
        if (S > 1024) S = 1024;
        float *bp = dbias ? part + (long)S * ((long)Cout * Cin) : nullptr;
        if (wgrad_narrow_launch(st, dY, lddy, X, ldx, R, Cin, Cout, S, part, bp)) {
            launch_slab_reduce(part, (long)Cout * Cin, S, accumulate, dW, st);
            if (dbias) launch_slab_reduce(bp, Cout, S, accumulate, dbias, st);
            CASPR_CHECK_LAUNCH("conv1x1_wgrad(narrow)");
            return CASPR_OK;
        }
    }
    // bf16x6: the 256 x 256 tile where both widths fill it reasonably (padded area at most 20 % above the 128-tile's), else 128 x 128
    int kind = 0, T = WG_T;
    if (bf16x6) {
        const double a128 = (double)ceil_div(Cout, 128) * ceil_div(Cin, 128) * 128.0 * 128.0;
        const double a256 = (double)ceil_div(Cout, 256) * ceil_div(Cin, 256) * 256.0 * 256.0;
        kind = (Cout >= 256 && Cin >= 256 && a256 <= 1.2 * a128) ? 2 : 1;
        const int force = CASPR_DEBUG_ENV_INT("CASPR_WGRAD_TILE");   // debug build: 128 / 256
        if (force == 128) kind = 1;
        if (force == 256) kind = 2;
        T = kind == 2 ? 256 : 128;
        if (kind == 2) {
            // one 512-thread workgroup per CU: ONE round of workgroups (tiles x slabs just under 256) instead of the 128-tile's
            // 2048 -- half the partial sums to write and to reduce (64 instead of 128 slabs of 1 MB on the CNF's 512 x 512 layers)
            const long t256 = (long)ceil_div(Cout, 256) * ceil_div(Cin, 256);
            long s1 = t256 >= 256 ? 1 : 256 / t256;
            if (s1 < S) S = (int)s1;
        }
    }
    const long rps = ((R + S - 1) / S + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    wgrad_launch(kind, dim3(ceil_div(Cout, T), ceil_div(Cin, T), S), st, dY, lddy, X, ldx, in_scale, in_shift, in_relu, in_relu_from, R, P, Cin,
                 Cout, rps, part, dbias ? part + (long)S * ((long)Cout * Cin) : nullptr);
    const long n = (long)Cout * Cin;
    launch_slab_reduce(part, n, S, accumulate, dW, st);
    if (dbias) {
        float *bpart = part + (long)S * n;
        launch_slab_reduce(bpart, Cout, S, accumulate, dbias, st);
    }
    CASPR_CHECK_LAUNCH("conv1x1_wgrad");
    return CASPR_OK;
}

extern "C" int caspr_conv1x1_wgrad_f32(const float *dY, int lddy, const float *X, int ldx, const float *in_scale,
                                       const float *in_shift, int in_relu, int in_relu_from, int B, int P, int Cin,
                                       int Cout, float *dW, float *dbias, int accumulate, void *ws, long ws_bytes,
                                       void *stream)
{
    return wgrad_impl(false, dY, lddy, X, ldx, in_scale, in_shift, in_relu, in_relu_from, B, P, Cin, Cout, dW, dbias, accumulate, ws, ws_bytes, stream);
}

// the same contract with the products in the exact bf16 three-way split (conv1x1_wgrad_bf16x6_kernel); workspace as above
extern "C" int caspr_conv1x1_wgrad_bf16x6_f32(const float *dY, int lddy, const float *X, int ldx, const float *in_scale,
                                              const float *in_shift, int in_relu, int in_relu_from, int B, int P, int Cin,
                                              int Cout, float *dW, float *dbias, int accumulate, void *ws, long ws_bytes,
                                              void *stream)
{
    return wgrad_impl(true, dY, lddy, X, ldx, in_scale, in_shift, in_relu, in_relu_from, B, P, Cin, Cout, dW, dbias, accumulate, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm (+ReLU) backward.  Forward: xh = (y - mean) * rstd ; a_pre = gamma*xh + beta ; a = relu?(a_pre).
// Given da = dL/da:  g = da * [a_pre > 0]      (relu)        dgamma_c += sum g*xh     dbeta_c += sum g
//   per (b, group):  s1 = sum_c gamma_c * sum_p g ,  s2 = sum_c gamma_c * sum_p g*xh ,  n = cpg * P
//   dy = rstd * ( g*gamma - (s1 + xh*s2) / n )
// Pass 1: per (b, c, point split) partial sums of g and g*xh in f64.  Pass 2: combine (fixed order) -> s1, s2 per
// (b, group), dgamma / dbeta.  Pass 3: elementwise dy, written over da.
// ---------------------------------------------------------------------------------------------
#define GB_SPLIT 1024

__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float *__restrict__ Y, int ldy,
                                                             const float *__restrict__ dA, int ldd, int P, int C, int G,
                                                             const float *__restrict__ mean, const float *__restrict__ rstd,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             int relu, const float *__restrict__ dMax,
                                                             const int32_t *__restrict__ aMax, double *__restrict__ part)
{
    __shared__ double s_g[256 * 4], s_gx[256 * 4];
    const int grp = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const long b = blockIdx.z;
    const int cpg = C / G, Q4 = cpg >> 2, TP = 256 / Q4;
    const int tq = threadIdx.x % Q4, tp = threadIdx.x / Q4;
    const int pbeg = s * GB_SPLIT, pend = (pbeg + GB_SPLIT) < P ? (pbeg + GB_SPLIT) : P;
    const float mu = mean[b * G + grp], rs = rstd[b * G + grp];
    double sg[4] = {0, 0, 0, 0}, sgx[4] = {0, 0, 0, 0};
    const int c0 = grp * cpg + tq * 4;
    if (tp < TP) {
        const f32x4 ga = ld4(gamma + c0), be = ld4(beta + c0);
        f32x4 dm = (f32x4){0.f, 0.f, 0.f, 0.f};
        int am[4] = {-1, -1, -1, -1};
        if (dMax) {
            dm = ld4(dMax + b * C + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = aMax[b * C + c0 + q];
        }
        for (int p = pbeg + tp; p < pend; p += TP) {
            const f32x4 y = ld4(Y + (b * P + p) * ldy + c0);
            const f32x4 d = dA ? ld4(dA + (b * P + p) * ldd + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xh = (y[q] - mu) * rs;
                float gq = (relu && !(ga[q] * xh + be[q] > 0.f)) ? 0.f : d[q];
                if (p == am[q]) gq += dm[q];   // the max over points is taken before the ReLU (tpointnet2.py:100,111)
                sg[q] += (double)gq;
                sgx[q] += (double)gq * (double)xh;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s_g[threadIdx.x * 4 + q] = sg[q];
        s_gx[threadIdx.x * 4 + q] = sgx[q];
    }
    __syncthreads();
    if (threadIdx.x < cpg) {
        const int c = threadIdx.x, q4 = c >> 2, q = c & 3;
        double a = 0.0, bx = 0.0;
        for (int r = 0; r < TP; ++r) {
            a += s_g[(r * Q4 + q4) * 4 + q];
            bx += s_gx[(r * Q4 + q4) * 4 + q];
        }
        double *o = part + ((b * C + grp * cpg + c) * S + s) * 2;
        o[0] = a;
        o[1] = bx;
    }
}

__global__ void gn_bwd_finalize_kernel(const double *__restrict__ part, int B, int C, int G, int S,
                                       const float *__restrict__ gamma, double *__restrict__ chan, float *__restrict__ s12)
{
    // one thread per (b, group): s1, s2 ; also per-(b,c) totals into `chan` for the dgamma/dbeta pass below
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int cpg = C / G;
    if (t < B * G) {
        const int b = t / G, grp = t % G;
        double s1 = 0.0, s2 = 0.0;
        for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) {
            double a = 0.0, bx = 0.0;
            for (int s = 0; s < S; ++s) {
                a += part[(((long)b * C + c) * S + s) * 2 + 0];
                bx += part[(((long)b * C + c) * S + s) * 2 + 1];
            }
            chan[((long)b * C + c) * 2 + 0] = a;
            chan[((long)b * C + c) * 2 + 1] = bx;
            s1 += (double)gamma[c] * a;
            s2 += (double)gamma[c] * bx;
        }
        s12[t * 2 + 0] = (float)s1;
        s12[t * 2 + 1] = (float)s2;
    }
}

__global__ void gn_bwd_param_kernel(const double *__restrict__ chan, int B, int C, float *__restrict__ dgamma,
                                    float *__restrict__ dbeta, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, bx = 0.0;
    for (int b = 0; b < B; ++b) {
        a += chan[((long)b * C + c) * 2 + 0];
        bx += chan[((long)b * C + c) * 2 + 1];
    }
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a;
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)bx;
}

__global__ void gn_bwd_apply_kernel(const float *__restrict__ Y, int ldy, const float *dA, int ldd, float *dY, int lddy,
                                    int P, int C, int G, const float *__restrict__ mean, const float *__restrict__ rstd,
                                    const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                    const float *__restrict__ dMax, const int32_t *__restrict__ aMax,
                                    const float *__restrict__ s12, long total4)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const int C4 = C >> 2;
    const int c0 = (int)(t % C4) * 4;
    const long row = t / C4;
    const long b = row / P;
    const int cpg = C / G;
    const f32x4 y = ld4(Y + row * ldy + c0);
    f32x4 d = dA ? ld4(dA + row * ldd + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 ga = ld4(gamma + c0), be = ld4(beta + c0);
    const int p = (int)(row - b * P);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int grp = (c0 + q) / cpg;
        const float mu = mean[b * G + grp], rs = rstd[b * G + grp];
        const float xh = (y[q] - mu) * rs;
        float gq = (relu && !(ga[q] * xh + be[q] > 0.f)) ? 0.f : d[q];
        if (dMax && p == aMax[b * C + c0 + q]) gq += dMax[b * C + c0 + q];
        const float inv_n = 1.0f / ((float)cpg * (float)P);
        d[q] = rs * (gq * ga[q] - (s12[(b * G + grp) * 2 + 0] + xh * s12[(b * G + grp) * 2 + 1]) * inv_n);
    }
    st4(dY + row * lddy + c0, d);
}

extern "C" long caspr_gn_bwd_ws_bytes(long B, int P, int C, int G)
{
    const long S = (P + GB_SPLIT - 1) / GB_SPLIT;
    return B * C * S * 16 + B * C * 16 + B * G * 8 + 256;
}

extern "C" int caspr_gn_bwd_f32(const float *Y, int ldy, const float *dA, int ldd, const float *dMax, const int32_t *aMax,
                                float *dY, int lddy, long B, int P, int C, int G, const float *mean, const float *rstd,
                                const float *gamma, const float *beta, int relu, float *dgamma, float *dbeta, int accumulate,
                                void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE((dA || dMax) && (dMax == nullptr) == (aMax == nullptr), "gn_bwd: give dA and/or dMax + aMax");
    CASPR_REQUIRE(lddy % 4 == 0 && lddy >= C, "gn_bwd: lddy=%d must be a multiple of 4 and >= C", lddy);
    CASPR_REQUIRE(Y && dY && mean && rstd && gamma && beta && dgamma && dbeta && ws, "gn_bwd: null pointer");
    CASPR_REQUIRE(C % G == 0 && (C / G) % 4 == 0 && (C / G) <= 256 && ldy % 4 == 0 && ldd % 4 == 0 && ldy >= C && (!dA || ldd >= C),
                  "gn_bwd: C/G=%d must be a multiple of 4 (<= 256) and strides multiples of 4", C / G);
    CASPR_REQUIRE(ws_bytes >= caspr_gn_bwd_ws_bytes(B, P, C, G), "gn_bwd: workspace too small");
    CASPR_REQUIRE(B <= 65535, "gn_bwd: B=%ld > 65535 (split the call)", B);
    const int S = ceil_div(P, GB_SPLIT);
    double *part = (double *)ws;
    double *chan = part + B * C * S * 2;
    float *s12 = (float *)(chan + B * C * 2);
    hipStream_t st = (hipStream_t)stream;
    gn_bwd_partial_kernel<<<dim3(G, S, (unsigned)B), dim3(256), 0, st>>>(Y, ldy, dA, ldd, P, C, G, mean, rstd, gamma, beta, relu, dMax, aMax, part);
    gn_bwd_finalize_kernel<<<dim3((unsigned)((B * G + 255) / 256)), dim3(256), 0, st>>>(part, (int)B, C, G, S, gamma, chan, s12);
    gn_bwd_param_kernel<<<dim3(ceil_div(C, 256)), dim3(256), 0, st>>>(chan, (int)B, C, dgamma, dbeta, accumulate);
    const long total4 = B * P * (C / 4);
    gn_bwd_apply_kernel<<<dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st>>>(Y, ldy, dA, ldd, dY, lddy, P, C, G, mean, rstd, gamma, beta, relu, dMax, aMax, s12, total4);
    CASPR_CHECK_LAUNCH("gn_bwd");
    return CASPR_OK;
}
