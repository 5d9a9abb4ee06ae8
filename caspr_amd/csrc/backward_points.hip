// backward_points.hip -- training tier, part 2: the point-set side of the encoder's backward pass
// (Kaolin's *_grad scatter-adds and the per-neighbourhood MLP) and the row-materialised
// set-abstraction forward used in training (activations are kept for the backward pass instead of living in LDS).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// three_interpolate backward (autograd node of pointnet2.py:519):
//   dFeat[b, idx[b,i,k], c] += weight[b,i,k] * dOut[b,i,c]       one wave per fine point, float atomics
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_interp_bwd_kernel(const float *__restrict__ dOut, int ldo,
                                                               const int32_t *__restrict__ idx,
                                                               const float *__restrict__ weight, int m, int n, int C,
                                                               float *__restrict__ dFeat, int ldf, long rows)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long b = row / n;
    const float *d = dOut + row * ldo;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = idx[row * 3 + k];
        const float w = weight[row * 3 + k];
        float *f = dFeat + (b * m + i) * ldf;
        for (int c = lane; c < C; c += 64) atomicAdd(f + c, w * d[c]);
    }
}

extern "C" int caspr_three_interp_bwd_f32(const float *dOut, int ldo, const int32_t *idx, const float *weight, int B,
                                          int m, int n, int C, float *dFeat, int ldf, void *stream)
{
    CASPR_REQUIRE(dOut && idx && weight && dFeat && B > 0 && m > 0 && n > 0 && C > 0 && ldo >= C && ldf >= C,
                  "three_interp_bwd: bad arguments");
    const long rows = (long)B * n;
    three_interp_bwd_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        dOut, ldo, idx, weight, m, n, C, dFeat, ldf, rows);
    CASPR_CHECK_LAUNCH("three_interp_bwd");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Grouping as rows (training layout of pointnet2.py:391,397-399): one row per (b, centre, sample),
//   G[(b*M+j)*ns+s, :] = [ xyz[b,i]-new_xyz[b,j] (3) | feat[b,i,0:C] | 0 pad ]   i = idx[b,j,s]
// in the reference's own channel order (xyz first), so weight gradients need no permutation.
// Backward: dFeat[b,i,c] += dG[row, 3+c]   (float atomics; xyz carries no gradient).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_rows_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                         const float *__restrict__ feat, int ldf,
                                                         const int32_t *__restrict__ idx, int n, int M, int C, int ns,
                                                         int centred, int feat_kind, float *__restrict__ G, int ldg, long rows)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long bj = row / ns;
    const long b = bj / M;
    const int i = idx[row];
    const float *px = xyz + (b * n + i) * 3, *pc = new_xyz + bj * 3;
    const float *f = feat ? feat + (b * n + i) * ldf : nullptr;
    float *o = G + row * ldg;
    if (!centred) {
        for (int c = lane; c < ldg; c += 64) {
            float v = 0.f;
            if (c < 3) v = px[c] - pc[c];
            else if (c - 3 < C) v = f[c - 3];
            o[c] = v;
        }
        return;
    }
    // centred on the neighbourhood's sample 0 (see caspr_group_rows_f32 / sa_small_kernel in sa_mlp.hip)
    const int i0 = idx[bj * ns];
    const float *q = xyz + (b * n + i0) * 3;
    const float *f0 = feat ? feat + (b * n + i0) * ldf : nullptr;
    const float x = px[0], y = px[1], z = px[2], x0 = q[0], y0 = q[1], z0 = q[2];
    const float dx = x - x0, dy = y - y0, dz = z - z0;
    for (int c = lane; c < ldg; c += 64) {
        float v = 0.f;
        if (c < 3) v = c == 0 ? dx : (c == 1 ? dy : dz);
        else if (c - 3 < C) {
            const int k = c - 3;
            if (feat_kind) {
                const int kq = (feat_kind & CASPR_FEAT_QUAD) ? k : k + 3;    // position in [x2 y2 z2 xz xy zy]
                v = kq == 0 ? dx * (x + x0) : kq == 1 ? dy * (y + y0) : kq == 2 ? dz * (z + z0)
                  : kq == 3 ? dx * z + x0 * dz : kq == 4 ? dx * y + x0 * dy : dz * y + z0 * dy;
            } else {
                v = f[k] - f0[k];
            }
        }
        o[c] = v;
    }
}

__global__ __launch_bounds__(256) void group_rows_bwd_kernel(const float *__restrict__ dG, int ldg,
                                                             const int32_t *__restrict__ idx, int n, int M, int C,
                                                             int ns, float *__restrict__ dFeat, int ldf, long rows)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long b = row / ((long)ns * M);
    const int i = idx[row];
    const float *d = dG + row * ldg + 3;
    float *f = dFeat + (b * n + i) * ldf;
    for (int c = lane; c < C; c += 64) atomicAdd(f + c, d[c]);
}

extern "C" int caspr_group_rows_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                                    const int32_t *idx, int B, int n, int M, int C, int ns, int centred, int feat_kind,
                                    float *G, int ldg, void *stream)
{
    CASPR_REQUIRE(xyz && new_xyz && idx && G && B > 0 && n > 0 && M > 0 && ns > 0 && C >= 0, "group_rows: bad arguments");
    const int want_c = ((feat_kind & CASPR_FEAT_QUAD) ? 3 : 0) + ((feat_kind & CASPR_FEAT_PAIRS) ? 3 : 0);
    CASPR_REQUIRE(feat_kind == 0 || (centred && feat_kind > 0 && feat_kind <= 3 && C == want_c), "group_rows: feat_kind=%d needs centred rows and C=%d", feat_kind, want_c);
    CASPR_REQUIRE((C == 0 || (feat && ldf >= C)) && ldg >= 3 + C, "group_rows: feat/ldf/ldg inconsistent with C=%d", C);
    const long rows = (long)B * M * ns;
    group_rows_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(xyz, new_xyz, feat, ldf, idx, n, M, C,
                                                                                               ns, centred, feat_kind, G, ldg, rows);
    CASPR_CHECK_LAUNCH("group_rows");
    return CASPR_OK;
}

// The first layer of a point MLP as rows when its feature part was PRE-AGGREGATED per source point (csrc/sa_mlp.hip: pre-aggregated first
// layer; the row-materialised form of the coarsest level): Y1[(b*M+j)*ns+s, c] = pre[b, i, c] + wx[c] . (xyz[b,i] - new_xyz[b,j]) + bias[c].
__global__ __launch_bounds__(256) void group_rows_pre_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                             const float *__restrict__ pre, int ldp, const int32_t *__restrict__ idx, int n,
                                                             int M, int C1, int ns, const float *__restrict__ wx,
                                                             const float *__restrict__ bias, float *__restrict__ Y, int ldy, long rows)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long bj = row / ns;
    const long b = bj / M;
    const int i = idx[row];
    const float *px = xyz + (b * n + i) * 3, *pc = new_xyz + bj * 3;
    const float d0 = px[0] - pc[0], d1 = px[1] - pc[1], d2 = px[2] - pc[2];      // the grouper's f32 subtraction (pointnet2.py:391-398)
    const float *f = pre + (b * n + i) * ldp;
    float *o = Y + row * ldy;
    for (int c = lane * 4; c < C1; c += 256) {      // host: C1 % 4 == 0
        const f32x4 v = ld4(f + c), b4 = ld4(bias + c);
        const f32x4 w0 = ld4(wx + c * 3), w1 = ld4(wx + c * 3 + 4), w2 = ld4(wx + c * 3 + 8);   // rows c .. c + 3 of (C1, 3)
        const float wr[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = (fmaf(wr[3 * q + 2], d2, fmaf(wr[3 * q + 1], d1, wr[3 * q] * d0)) + v[q]) + b4[q];
        st4(o + c, r);
    }
}

extern "C" int caspr_group_rows_pre_f32(const float *xyz, const float *new_xyz, const float *pre, int ldp, const int32_t *idx, int B, int n,
                                        int M, int C1, int ns, const float *wx, const float *bias, float *Y, int ldy, void *stream)
{
    CASPR_REQUIRE(xyz && new_xyz && pre && idx && wx && bias && Y && B > 0 && n > 0 && M > 0 && ns > 0 && C1 > 0, "group_rows_pre: bad arguments");
    CASPR_REQUIRE(C1 % 4 == 0 && ldp % 4 == 0 && ldp >= C1 && ldy % 4 == 0 && ldy >= C1 && ((uintptr_t)pre % 16) == 0 && ((uintptr_t)wx % 16) == 0 &&
                  ((uintptr_t)bias % 16) == 0 && ((uintptr_t)Y % 16) == 0,
                  "group_rows_pre: C1=%d ldp=%d ldy=%d must be multiples of 4 (16-byte rows)", C1, ldp, ldy);
    const long rows = (long)B * M * ns;
    group_rows_pre_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(xyz, new_xyz, pre, ldp, idx, n, M, C1, ns, wx,
                                                                                                   bias, Y, ldy, rows);
    CASPR_CHECK_LAUNCH("group_rows_pre");
    return CASPR_OK;
}

extern "C" int caspr_group_rows_bwd_f32(const float *dG, int ldg, const int32_t *idx, int B, int n, int M, int C, int ns,
                                        float *dFeat, int ldf, void *stream)
{
    CASPR_REQUIRE(dG && idx && dFeat && B > 0 && n > 0 && M > 0 && ns > 0 && C > 0 && ldg >= 3 + C && ldf >= C,
                  "group_rows_bwd: bad arguments");
    const long rows = (long)B * M * ns;
    group_rows_bwd_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(dG, ldg, idx, n, M, C, ns, dFeat,
                                                                                                   ldf, rows);
    CASPR_CHECK_LAUNCH("group_rows_bwd");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(16) over small row groups (one neighbourhood = ns rows, pointnet2.py:649-703): statistics per
// (neighbourhood, group) over cpg channels x ns rows.  One wave per neighbourhood; lane = (group g = lane & 15,
// row phase sub = lane >> 4); sums in f64; sub phases combined with two xor shuffles.
//   forward : A = relu?(gamma*(y-mean)*rstd + beta)  (dense)   or, for the last layer of the point MLP,
//             out[nb, c] = max over rows, arg[nb, c] = first row attaining it          (pointnet2.py:701)
//   backward: dY = rstd*(g*gamma - (s1 + xh*s2)/n), g = relu-masked dA  or  (row == arg ? dMax : 0);
//             dgamma/dbeta: per-lane f32 accumulators over the wave's neighbourhoods -> per-block partials ->
//             fixed-order f64 combine (deterministic).
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ double shfl_xor_d(double v, int m)
{
    return __shfl_xor(v, m, 64);
}

template <int CPG>
__global__ __launch_bounds__(256) void gn_rows_fwd_kernel(const float *__restrict__ Y, int ldy, long NB, int ns, int C,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float eps, int relu, float *__restrict__ A, int lda,
                                                          float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                          float *__restrict__ maxout, int ldm, int32_t *__restrict__ arg)
{
    const int lane = threadIdx.x & 63, g = lane & 15, sub = lane >> 4;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int c0 = g * CPG;
    float ga[CPG], be[CPG];
#pragma unroll
    for (int q = 0; q < CPG; ++q) { ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q]; }
    const double inv_n = 1.0 / ((double)CPG * ns);
    for (long nb = wave0; nb < NB; nb += nwaves) {
        const float *y = Y + nb * ns * ldy + c0;
        double s = 0.0, sq = 0.0;
        for (int p = sub; p < ns; p += 4) {
#pragma unroll
            for (int q = 0; q < CPG; ++q) {
                const double v = (double)y[(long)p * ldy + q];
                s += v;
                sq += v * v;
            }
        }
        s += shfl_xor_d(s, 16); sq += shfl_xor_d(sq, 16);
        s += shfl_xor_d(s, 32); sq += shfl_xor_d(sq, 32);
        const double mu = s * inv_n;
        double var = sq * inv_n - mu * mu;
        var = var < 0.0 ? 0.0 : var;
        const double rs = 1.0 / sqrt(var + (double)eps);
        const float muf = (float)mu, rsf = (float)rs;
        if (sub == 0) { mean_out[nb * 16 + g] = muf; rstd_out[nb * 16 + g] = rsf; }
        if (!maxout) {
            float *a = A + nb * ns * lda + c0;
            for (int p = sub; p < ns; p += 4) {
#pragma unroll
                for (int q = 0; q < CPG; ++q) {
                    float v = ga[q] * ((y[(long)p * ldy + q] - muf) * rsf) + be[q];
                    a[(long)p * lda + q] = (relu && !(v > 0.f)) ? 0.f : v;
                }
            }
        } else {
            float best[CPG];
            int bi[CPG];
#pragma unroll
            for (int q = 0; q < CPG; ++q) { best[q] = -INFINITY; bi[q] = 0x7fffffff; }
            for (int p = sub; p < ns; p += 4) {
#pragma unroll
                for (int q = 0; q < CPG; ++q) {
                    float v = ga[q] * ((y[(long)p * ldy + q] - muf) * rsf) + be[q];
                    v = (relu && !(v > 0.f)) ? 0.f : v;
                    if (v > best[q]) { best[q] = v; bi[q] = p; }
                }
            }
#pragma unroll
            for (int q = 0; q < CPG; ++q) {
#pragma unroll
                for (int m = 16; m <= 32; m <<= 1) {
                    const float ov = __shfl_xor(best[q], m, 64);
                    const int oi = __shfl_xor(bi[q], m, 64);
                    if (ov > best[q] || (ov == best[q] && oi < bi[q])) { best[q] = ov; bi[q] = oi; }
                }
                if (sub == 0) {
                    maxout[nb * ldm + c0 + q] = best[q];
                    arg[nb * C + c0 + q] = bi[q];
                }
            }
        }
    }
}

template <int CPG>
__global__ __launch_bounds__(256) void gn_rows_bwd_kernel(const float *__restrict__ Y, int ldy, long NB, int ns, int C,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          int relu, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                          const float *__restrict__ dA, int lda,
                                                          const float *__restrict__ dMax, int ldm, const int32_t *__restrict__ arg,
                                                          float *__restrict__ dY, int lddy, float *__restrict__ part)
{
    __shared__ float s_acc[4][16 * CPG * 2];
    const int lane = threadIdx.x & 63, g = lane & 15, sub = lane >> 4, wave = threadIdx.x >> 6;
    const long wave0 = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    const int c0 = g * CPG;
    float ga[CPG], be[CPG], acc_g[CPG], acc_gx[CPG];
#pragma unroll
    for (int q = 0; q < CPG; ++q) { ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q]; acc_g[q] = 0.f; acc_gx[q] = 0.f; }
    const float inv_n = 1.0f / ((float)CPG * (float)ns);
    for (long nb = wave0; nb < NB; nb += nwaves) {
        const float *y = Y + nb * ns * ldy + c0;
        const float mu = mean[nb * 16 + g], rs = rstd[nb * 16 + g];
        float dm[CPG];
        int am[CPG];
        if (dMax) {
#pragma unroll
            for (int q = 0; q < CPG; ++q) { dm[q] = dMax[nb * ldm + c0 + q]; am[q] = arg[nb * C + c0 + q]; }
        }
        float s1 = 0.f, s2 = 0.f;
        for (int p = sub; p < ns; p += 4) {
#pragma unroll
            for (int q = 0; q < CPG; ++q) {
                const float xh = (y[(long)p * ldy + q] - mu) * rs;
                float gq = dMax ? (p == am[q] ? dm[q] : 0.f) : dA[(nb * ns + p) * lda + c0 + q];
                if (relu && !(ga[q] * xh + be[q] > 0.f)) gq = 0.f;
                acc_g[q] += gq;
                acc_gx[q] += gq * xh;
                s1 += gq * ga[q];
                s2 += gq * ga[q] * xh;
            }
        }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        float *dy = dY + nb * ns * lddy + c0;
        for (int p = sub; p < ns; p += 4) {
#pragma unroll
            for (int q = 0; q < CPG; ++q) {
                const float xh = (y[(long)p * ldy + q] - mu) * rs;
                float gq = dMax ? (p == am[q] ? dm[q] : 0.f) : dA[(nb * ns + p) * lda + c0 + q];
                if (relu && !(ga[q] * xh + be[q] > 0.f)) gq = 0.f;
                dy[(long)p * lddy + q] = rs * (gq * ga[q] - (s1 + xh * s2) * inv_n);
            }
        }
    }
    // wave partials: combine the 4 row phases, then the 4 waves, in a fixed order
#pragma unroll
    for (int q = 0; q < CPG; ++q) {
        acc_g[q] += __shfl_xor(acc_g[q], 16, 64); acc_gx[q] += __shfl_xor(acc_gx[q], 16, 64);
        acc_g[q] += __shfl_xor(acc_g[q], 32, 64); acc_gx[q] += __shfl_xor(acc_gx[q], 32, 64);
        if (sub == 0) {
            s_acc[wave][(c0 + q) * 2 + 0] = acc_g[q];
            s_acc[wave][(c0 + q) * 2 + 1] = acc_gx[q];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 16 * CPG * 2; t += 256)
        part[(long)blockIdx.x * (16 * CPG * 2) + t] = (s_acc[0][t] + s_acc[1][t]) + (s_acc[2][t] + s_acc[3][t]);
}

// one workgroup per channel: 256 threads take every 256th block partial (f64), LDS tree in a fixed order
__global__ __launch_bounds__(256) void gn_rows_param_kernel(const float *__restrict__ part, int nblocks, int C,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate)
{
    __shared__ double sa[256], sb[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double a = 0.0, bx = 0.0;
    for (int k = t; k < nblocks; k += 256) {
        a += (double)part[(long)k * C * 2 + c * 2 + 0];
        bx += (double)part[(long)k * C * 2 + c * 2 + 1];
    }
    sa[t] = a;
    sb[t] = bx;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) { sa[t] += sa[t + w]; sb[t] += sb[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sa[0];
        dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sb[0];
    }
}

#define GN_ROWS_BLOCKS 2048

template <int CPG>
static void launch_rows_fwd(const float *Y, int ldy, long NB, int ns, int C, const float *gamma, const float *beta, float eps,
                            int relu, float *A, int lda, float *mean, float *rstd, float *maxout, int ldm, int32_t *arg,
                            hipStream_t st)
{
    const long want = (NB + 3) / 4;
    const int blocks = (int)(want < 8192 ? want : 8192);
    gn_rows_fwd_kernel<CPG><<<dim3(blocks), dim3(256), 0, st>>>(Y, ldy, NB, ns, C, gamma, beta, eps, relu, A, lda, mean, rstd,
                                                                maxout, ldm, arg);
}

extern "C" int caspr_gn_rows_f32(const float *Y, int ldy, long NB, int ns, int C, const float *gamma, const float *beta,
                                 float eps, int relu, float *A, int lda, float *mean, float *rstd, float *maxout, int ldm,
                                 int32_t *arg, void *stream)
{
    CASPR_REQUIRE(Y && gamma && beta && mean && rstd && NB > 0 && ns > 0 && C > 0 && ldy >= C, "gn_rows: bad arguments");
    CASPR_REQUIRE((maxout != nullptr) == (arg != nullptr) && (maxout ? ldm >= C : (A && lda >= C)),
                  "gn_rows: give either A (dense activations) or maxout + arg");
    CASPR_REQUIRE(C % 16 == 0, "gn_rows: C=%d must be a multiple of the 16 groups", C);
    hipStream_t st = (hipStream_t)stream;
#define CASE(K) case K: launch_rows_fwd<K>(Y, ldy, NB, ns, C, gamma, beta, eps, relu, A, lda, mean, rstd, maxout, ldm, arg, st); break;
    switch (C / 16) {
        CASE(1) CASE(2) CASE(4) CASE(6) CASE(8) CASE(16) CASE(32)
    default:
        caspr_set_error("gn_rows: C/16=%d not instantiated (1,2,4,6,8,16,32)", C / 16);
        return CASPR_EINVAL;
    }
#undef CASE
    CASPR_CHECK_LAUNCH("gn_rows");
    return CASPR_OK;
}

extern "C" long caspr_gn_rows_bwd_ws_bytes(int C) { return (long)GN_ROWS_BLOCKS * C * 2 * 4 + 256; }

extern "C" int caspr_gn_rows_bwd_f32(const float *Y, int ldy, long NB, int ns, int C, const float *gamma,
                                     const float *beta, int relu, const float *mean, const float *rstd, const float *dA,
                                     int lda, const float *dMax, int ldm, const int32_t *arg, float *dY, int lddy,
                                     float *dgamma, float *dbeta, int accumulate, void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(Y && gamma && beta && mean && rstd && dY && dgamma && dbeta && ws && NB > 0 && ns > 0 && C > 0 && ldy >= C && lddy >= C,
                  "gn_rows_bwd: bad arguments");
    CASPR_REQUIRE((dA != nullptr) != (dMax != nullptr) && (dMax ? (arg && ldm >= C) : lda >= C),
                  "gn_rows_bwd: give either dA (dense) or dMax + arg");
    CASPR_REQUIRE(C % 16 == 0 && ws_bytes >= caspr_gn_rows_bwd_ws_bytes(C), "gn_rows_bwd: C %% 16 != 0 or workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long want = (NB + 3) / 4;
    const int blocks = (int)(want < GN_ROWS_BLOCKS ? want : GN_ROWS_BLOCKS);
    float *part = (float *)ws;
#define CASE(K) case K: gn_rows_bwd_kernel<K><<<dim3(blocks), dim3(256), 0, st>>>(Y, ldy, NB, ns, C, gamma, beta, relu, mean, rstd, dA, lda, dMax, ldm, arg, dY, lddy, part); break;
    switch (C / 16) {
        CASE(1) CASE(2) CASE(4) CASE(6) CASE(8) CASE(16) CASE(32)
    default:
        caspr_set_error("gn_rows_bwd: C/16=%d not instantiated (1,2,4,6,8,16,32)", C / 16);
        return CASPR_EINVAL;
    }
#undef CASE
    gn_rows_param_kernel<<<dim3(C), dim3(256), 0, st>>>(part, blocks, C, dgamma, dbeta, accumulate);
    CASPR_CHECK_LAUNCH("gn_rows_bwd");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// per-batch column sums  out[b,c] = sum_p A[b,p,c]   (gradient of the per-sequence bias that carries the tiled
// global feature through the head's first conv, tpointnet2.py:96-99)   -- fixed-order: one thread per column
// quad walks a 256-row stripe, stripes combined in order.
// ---------------------------------------------------------------------------------------------
#define CSB_SPLIT 512
__global__ __launch_bounds__(256) void colsum_batched_kernel(const float *__restrict__ A, int ld, int P, int C,
                                                             float *__restrict__ part)
{
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int s = blockIdx.y, S = gridDim.y;
    const long b = blockIdx.z;
    const int pbeg = s * CSB_SPLIT, pend = (pbeg + CSB_SPLIT) < P ? (pbeg + CSB_SPLIT) : P;
    double acc = 0.0;
    if (c < C)
        for (int p = pbeg + sub; p < pend; p += 4) acc += (double)A[(b * P + p) * ld + c];
    red[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0 && c < C)
        part[(b * S + s) * C + c] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

__global__ void colsum_batched_final_kernel(const float *__restrict__ part, long BC, int C, int S, float *__restrict__ out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= BC) return;
    const long b = t / C;
    const int c = (int)(t % C);
    double acc = 0.0;
    for (int s = 0; s < S; ++s) acc += (double)part[(b * S + s) * C + c];
    out[t] = (float)acc;
}

extern "C" long caspr_colsum_ws_bytes(long B, int P, int C) { return B * ((P + CSB_SPLIT - 1) / CSB_SPLIT) * C * 4 + 256; }

extern "C" int caspr_colsum_batched_f32(const float *A, int ld, int B, int P, int C, float *out, void *ws, long ws_bytes,
                                        void *stream)
{
    CASPR_REQUIRE(A && out && ws && B > 0 && B <= 65535 && P > 0 && C > 0 && ld >= C, "colsum_batched: bad arguments");
    CASPR_REQUIRE(ws_bytes >= caspr_colsum_ws_bytes(B, P, C), "colsum_batched: workspace too small");
    const int S = ceil_div(P, CSB_SPLIT);
    hipStream_t st = (hipStream_t)stream;
    colsum_batched_kernel<<<dim3(ceil_div(C, 64), S, B), dim3(256), 0, st>>>(A, ld, P, C, (float *)ws);
    const long BC = (long)B * C;
    colsum_batched_final_kernel<<<dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, st>>>((const float *)ws, BC, C, S, out);
    CASPR_CHECK_LAUNCH("colsum_batched");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// arg-max over points of the normalised feature  v = y*scale + shift  (first index on ties), the index
// torch.max records for its backward at tpointnet2.py:111 and pointnet.py:42.  Two passes over 1024-point splits.
// ---------------------------------------------------------------------------------------------
#define AM_SPLIT 1024
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float *__restrict__ Y, int ldy, int P, int C,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             float *__restrict__ pval, int32_t *__restrict__ pidx)
{
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, s = blockIdx.y, S = gridDim.y;
    const long b = blockIdx.z;
    const int pbeg = s * AM_SPLIT, pend = (pbeg + AM_SPLIT) < P ? (pbeg + AM_SPLIT) : P;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (c < C) {
        const float sc = scale[b * C + c], sh = shift[b * C + c];
        for (int p = pbeg + sub; p < pend; p += 4) {
            const float v = fmaf(Y[(b * P + p) * ldy + c], sc, sh);
            if (v > best) { best = v; bi = p; }
        }
    }
    sv[sub][cl] = best;
    si[sub][cl] = bi;
    __syncthreads();
    if (sub == 0 && c < C) {
        for (int k = 1; k < 4; ++k)
            if (sv[k][cl] > best || (sv[k][cl] == best && si[k][cl] < bi)) { best = sv[k][cl]; bi = si[k][cl]; }
        pval[(b * C + c) * S + s] = best;
        pidx[(b * C + c) * S + s] = bi;
    }
}

__global__ void argmax_final_kernel(const float *__restrict__ pval, const int32_t *__restrict__ pidx, long BC, int S,
                                    int32_t *__restrict__ out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= BC) return;
    float best = -INFINITY;
    int bi = 0;
    for (int s = 0; s < S; ++s) {
        const float v = pval[t * S + s];
        if (v > best) { best = v; bi = pidx[t * S + s]; }   // splits are visited in point order: ties keep the first
    }
    out[t] = bi;
}

extern "C" long caspr_argmax_ws_bytes(long B, int P, int C) { return B * C * ((P + AM_SPLIT - 1) / AM_SPLIT) * 8 + 256; }

extern "C" int caspr_argmax_points_f32(const float *Y, int ldy, int B, int P, int C, const float *scale, const float *shift,
                                       int32_t *out, void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(Y && scale && shift && out && ws && B > 0 && B <= 65535 && P > 0 && C > 0 && ldy >= C, "argmax_points: bad arguments");
    CASPR_REQUIRE(ws_bytes >= caspr_argmax_ws_bytes(B, P, C), "argmax_points: workspace too small");
    const int S = ceil_div(P, AM_SPLIT);
    float *pval = (float *)ws;
    int32_t *pidx = (int32_t *)(pval + (long)B * C * S);
    hipStream_t st = (hipStream_t)stream;
    argmax_partial_kernel<<<dim3(ceil_div(C, 64), S, B), dim3(256), 0, st>>>(Y, ldy, P, C, scale, shift, pval, pidx);
    const long BC = (long)B * C;
    argmax_final_kernel<<<dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, st>>>(pval, pidx, BC, S, out);
    CASPR_CHECK_LAUNCH("argmax_points");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Deterministic scatter-add = gather over precomputed segments (CSR): dst[t, c] (+)= sum_{e in seg(t)} w[e] * src[row[e], col0+c]
// with the entries of every segment in ascending source order.  The training path uses this for the backward of
// three_interpolate and of the grouper instead of the float-atomic kernels above: gradients become bit-reproducible.
// One wave per target row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_sum_kernel(const float *__restrict__ src, int lds, int col0,
                                                          const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_row,
                                                          const float *__restrict__ seg_w, int C, float *__restrict__ dst, int ldd,
                                                          int accumulate, long targets)
{
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= targets) return;
    const int e0 = seg_start[t], e1 = seg_start[t + 1];
    float *d = dst + t * ldd;
    for (int c = lane; c < C; c += 64) {
        float acc = accumulate ? d[c] : 0.f;
        for (int e = e0; e < e1; ++e) {
            const float v = src[(long)seg_row[e] * lds + col0 + c];
            acc += seg_w ? seg_w[e] * v : v;
        }
        d[c] = acc;
    }
}

extern "C" int caspr_segment_sum_f32(const float *src, int lds, int col0, const int32_t *seg_start, const int32_t *seg_row,
                                     const float *seg_w, long targets, int C, float *dst, int ldd, int accumulate, void *stream)
{
    CASPR_REQUIRE(src && seg_start && seg_row && dst && targets > 0 && C > 0 && col0 >= 0 && lds >= col0 + C && ldd >= C,
                  "segment_sum: bad arguments");
    segment_sum_kernel<<<dim3((unsigned)((targets + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(src, lds, col0, seg_start, seg_row, seg_w,
                                                                                                   C, dst, ldd, accumulate, targets);
    CASPR_CHECK_LAUNCH("segment_sum");
    return CASPR_OK;
}
