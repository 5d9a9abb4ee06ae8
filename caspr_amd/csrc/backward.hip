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
    if (S > 64)
        slab_reduce_wide_kernel<<<dim3((unsigned)((n + 7) / 8)), dim3(256), 0, st>>>(part, n, S, accumulate, out);
    else
        slab_reduce_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(part, n, S, accumulate, out);
}

// number of row slabs: enough workgroups (tiles x slabs ~ 2048) to fill 256 CUs even when the weight is one tile,
// at least 512 rows per slab, at most 4096 slabs
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

extern "C" int caspr_conv1x1_wgrad_f32(const float *dY, int lddy, const float *X, int ldx, const float *in_scale,
                                       const float *in_shift, int in_relu, int in_relu_from, int B, int P, int Cin,
                                       int Cout, float *dW, float *dbias, int accumulate, void *ws, long ws_bytes,
                                       void *stream)
{
    CASPR_REQUIRE(dY && X && dW && ws && B > 0 && P > 0 && Cin > 0 && Cout > 0, "conv1x1_wgrad: bad arguments");
    CASPR_REQUIRE(lddy % 4 == 0 && lddy >= ((Cout + 3) & ~3) && ldx % 4 == 0 && ldx >= ((Cin + 3) & ~3),
                  "conv1x1_wgrad: row strides must be multiples of 4 and cover the channels (lddy=%d ldx=%d)", lddy, ldx);
    CASPR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (in_scale == nullptr || Cin % 4 == 0),
                  "conv1x1_wgrad: in_scale/in_shift must be given together and need Cin %% 4 == 0");
    const long R = (long)B * P;
    CASPR_REQUIRE(ws_bytes >= caspr_wgrad_ws_bytes(R, Cin, Cout), "conv1x1_wgrad: workspace too small");
    const int S = pick_slabs(R, Cin, Cout);
    const long rps = ((R + S - 1) / S + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)ws;
    conv1x1_wgrad_kernel<<<dim3(ceil_div(Cout, WG_T), ceil_div(Cin, WG_T), S), dim3(256), 0, st>>>(
        dY, lddy, X, ldx, in_scale, in_shift, in_relu, in_relu_from, R, P, Cin, Cout, rps, part, dbias ? part + (long)S * ((long)Cout * Cin) : nullptr);
    const long n = (long)Cout * Cin;
    launch_slab_reduce(part, n, S, accumulate, dW, st);
    if (dbias) {
        float *bpart = part + (long)S * n;
        launch_slab_reduce(bpart, Cout, S, accumulate, dbias, st);
    }
    CASPR_CHECK_LAUNCH("conv1x1_wgrad");
    return CASPR_OK;
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
