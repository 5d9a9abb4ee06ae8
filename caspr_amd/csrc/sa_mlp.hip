// sa_mlp.hip -- fused set-abstraction scale: grouping + per-neighbourhood point MLP for gfx950.
// Replaces PointNet2GroupingLayer + PointNetFeatureExtractor (models/pointnet2.py:391-409,649-703).
//
// One 256-thread workgroup owns NCOL = NCEN*NS columns (NCEN centres x NS neighbours).  The
// gathered neighbourhood (xyz - centre || features) is staged ONCE in LDS as an MFMA B-tile; the
// three conv(k=1) layers run as f32 MFMA 16x16x4 chains whose D fragments are written straight
// back as the next layer's B-tile; GroupNorm(16) statistics are per neighbourhood (pointnet2.py:642
// -> per column block of NS), two-pass in LDS; the final max over the NS samples is fused.  The
// (B,M,C+3,ns) grouped tensor (22.7 MB / frame at N=2048) is never written to HBM.
#include "common.h"

struct SaLayer {
    const float *wp, *bias, *gamma, *beta;
    int cout;  // multiple of 16
    int kc;    // packed 16-wide K chunks (even)
};

struct SaArgs {
    const float *xyz, *new_xyz, *feat;
    const int32_t *idx;
    int ldf, n, M, C;
    SaLayer L[3];
    float *out;
    int ldo, out_off;
    int rowsA, rowsB;  // kq rows of the two ping-pong B-tiles
};

template <int NS, int NCOL>
__global__ __launch_bounds__(256) void sa_mlp_kernel(SaArgs a)
{
    constexpr int NCEN = NCOL / NS;
    constexpr int CT = NCOL / 16;
    constexpr int NSTAT = NCEN * 16;
    constexpr int TPS = 256 / NSTAT;  // threads per (centre, group) statistic: 4, 8 or 16
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bufA = smem;
    float *bufB = bufA + a.rowsA * NCOL * 4;
    double *s_mean = reinterpret_cast<double *>(bufB + a.rowsB * NCOL * 4);  // [NSTAT] (f64: see the statistics pass)
    float *s_rstd = reinterpret_cast<float *>(s_mean + NSTAT);              // [NSTAT]
    float *s_cen = s_rstd + NSTAT;              // [NCEN*4]
    int *s_idx = reinterpret_cast<int *>(s_cen + NCEN * 4);  // [NCOL]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * NCEN;

    // ---- neighbour indices + centres
    if (tid < NCOL) {
        const int cen = tid / NS, s = tid % NS;
        const int m = (m0 + cen) < a.M ? (m0 + cen) : (a.M - 1);
        s_idx[tid] = a.idx[((long)b * a.M + m) * NS + s];
    }
    if (tid < NCEN * 3) {
        const int cen = tid / 3, d = tid % 3;
        const int m = (m0 + cen) < a.M ? (m0 + cen) : (a.M - 1);
        s_cen[cen * 4 + d] = a.new_xyz[((long)b * a.M + m) * 3 + d];
    }
    __syncthreads();

    // ---- gather: K order = [feat (C, padded to C4) | dx dy dz 0 | zeros ...]
    {
        const int C4 = (a.C + 3) & ~3;
        const int nkq = a.L[0].kc * 4;
        const int qfeat = C4 >> 2;
        for (int it = tid; it < NCOL * nkq; it += 256) {
            const int kq = it % nkq, col = it / nkq;
            const int k = s_idx[col];
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (kq < qfeat) {
                v = ld4(a.feat + ((long)b * a.n + k) * a.ldf + kq * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (kq * 4 + q >= a.C) v[q] = 0.f;
            } else if (kq == qfeat) {
                const float *p = a.xyz + ((long)b * a.n + k) * 3;
                const float *c = s_cen + (col / NS) * 4;
                v[0] = p[0] - c[0];
                v[1] = p[1] - c[1];
                v[2] = p[2] - c[2];
            }
            st4(bufA + btile_off(kq, col, NCOL), v);
        }
    }
    __syncthreads();

    float *bin = bufA, *bout = bufB;
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        const SaLayer L = a.L[l];
        const int RT = L.cout >> 4;
        // ---- MFMA: work items (row tile, column group)
        if (RT >= 4) {
            for (int rt = wave; rt < RT; rt += 4) {
                f32x4 acc[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float *wrow = L.wp + ((long)rt * L.kc) * 256 + lane * 4;
                f32x4 af = ld4(wrow);
                for (int kc = 0; kc < L.kc; ++kc) {
                    f32x4 an = af;
                    if (kc + 1 < L.kc) an = ld4(wrow + (long)(kc + 1) * 256);
                    f32x4 bf[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) bf[ct] = ld4(bin + btile_off(kc * 4 + g, ct * 16 + j, NCOL));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma16(af[q], bf[ct][q], acc[ct]);
                    af = an;
                }
                f32x4 bias4 = ld4(L.bias + rt * 16 + 4 * g);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) st4(bout + btile_off(rt * 4 + g, ct * 16 + j, NCOL), acc[ct] + bias4);
            }
        } else {
            for (int item = wave; item < RT * CT; item += 4) {
                const int rt = item / CT, ct = item % CT;
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float *wrow = L.wp + ((long)rt * L.kc) * 256 + lane * 4;
                for (int kc = 0; kc < L.kc; ++kc) {
                    const f32x4 af = ld4(wrow + (long)kc * 256);
                    const f32x4 bf = ld4(bin + btile_off(kc * 4 + g, ct * 16 + j, NCOL));
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = mfma16(af[q], bf[q], acc);
                }
                f32x4 bias4 = ld4(L.bias + rt * 16 + 4 * g);
                st4(bout + btile_off(rt * 4 + g, ct * 16 + j, NCOL), acc + bias4);
            }
        }
        __syncthreads();

        // ---- GroupNorm statistics per (centre, group): two-pass over cpg*NS elements in LDS, in f64.
        // Most first-level neighbourhoods are padded with duplicates of one or two points; GroupNorm then
        // divides near-zero deviations by sqrt(var + 1e-5) -> up to 316x amplification of any rounding in
        // the mean.  f64 sums make mean/(x - mean) exact for such groups (the reference's f32 path is not).
        const int cpg = L.cout >> 4;
        {
            const int stat = tid / TPS, sub = tid % TPS;
            const int cen = stat >> 4, grp = stat & 15;
            const int cnt = cpg * NS;
            double s = 0.0;
            for (int e = sub; e < cnt; e += TPS) {
                const int co = grp * cpg + e / NS, col = cen * NS + e % NS;
                s += (double)bout[btile_off(co >> 2, col, NCOL) + (co & 3)];
            }
#pragma unroll
            for (int off = TPS >> 1; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            const double mean = s / (double)cnt;
            double v = 0.0;
            for (int e = sub; e < cnt; e += TPS) {
                const int co = grp * cpg + e / NS, col = cen * NS + e % NS;
                const double d = (double)bout[btile_off(co >> 2, col, NCOL) + (co & 3)] - mean;
                v += d * d;
            }
#pragma unroll
            for (int off = TPS >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (sub == 0) {
                s_mean[stat] = mean;
                s_rstd[stat] = (float)(1.0 / sqrt(v / (double)cnt + 1e-5));
            }
        }
        __syncthreads();

        if (l < 2) {
            // ---- normalise + ReLU in place; zero the K padding rows of the next layer's operand
            const int nq = L.cout >> 2;
            const int nq_pad = a.L[l + 1].kc * 4;
            for (int it = tid; it < NCOL * nq_pad; it += 256) {
                const int col = it % NCOL, kq = it / NCOL;
                float *p = bout + btile_off(kq, col, NCOL);
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kq < nq) {
                    v = ld4(p);
                    const int cen = col / NS;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = kq * 4 + q;
                        const int st = cen * 16 + co / cpg;
                        // (x - mean) first: exact for the near-constant neighbourhoods where rstd -> 1/sqrt(eps)
                        const float y = (float)((double)v[q] - s_mean[st]) * (s_rstd[st] * L.gamma[co]) + L.beta[co];
                        v[q] = y > 0.f ? y : 0.f;
                    }
                }
                st4(p, v);
            }
            __syncthreads();
            float *t = bin;
            bin = bout;
            bout = t;
        } else {
            // ---- last layer: GroupNorm (no ReLU) then max over the NS samples (pointnet2.py:690-698)
            for (int it = tid; it < NCEN * L.cout; it += 256) {
                const int co = it % L.cout, cen = it / L.cout;
                if (m0 + cen >= a.M) continue;
                const int st = cen * 16 + co / cpg;
                const float sc = s_rstd[st] * L.gamma[co], be = L.beta[co];
                const double mean = s_mean[st];
                float mx = -INFINITY;
                for (int s = 0; s < NS; ++s) {
                    const float y = (float)((double)bout[btile_off(co >> 2, cen * NS + s, NCOL) + (co & 3)] - mean) * sc + be;
                    mx = y > mx ? y : mx;
                }
                a.out[((long)b * a.M + m0 + cen) * a.ldo + a.out_off + co] = mx;
            }
        }
    }
}

template <int NS, int NCOL>
static int launch_sa(const SaArgs &a, int B, size_t shmem, hipStream_t st)
{
    auto kern = sa_mlp_kernel<NS, NCOL>;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) {
            caspr_set_error("sa_mlp_max: hipFuncSetAttribute(%zu) failed: %s", shmem, hipGetErrorString(e));
            return CASPR_ELAUNCH;
        }
    }
    constexpr int NCEN = NCOL / NS;
    kern<<<dim3(ceil_div(a.M, NCEN), B), dim3(256), shmem, st>>>(a);
    return CASPR_OK;
}

extern "C" int caspr_sa_mlp_max_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                                    const int32_t *idx, int B, int n, int M, int C, int ns, const float *w1p,
                                    const float *b1, const float *g1, const float *be1, int C1, const float *w2p,
                                    const float *b2, const float *g2, const float *be2, int C2, const float *w3p,
                                    const float *b3, const float *g3, const float *be3, int C3, float *out, int ldo,
                                    int out_off, void *stream)
{
    CASPR_REQUIRE(xyz && new_xyz && idx && out && w1p && w2p && w3p && b1 && b2 && b3 && g1 && g2 && g3 && be1 && be2 && be3,
                  "sa_mlp_max: null pointer");
    CASPR_REQUIRE(C == 0 || (feat && ldf % 4 == 0 && ldf >= ((C + 3) & ~3)), "sa_mlp_max: feat/ldf invalid (C=%d ldf=%d)", C, ldf);
    CASPR_REQUIRE(ns == 16 || ns == 32, "sa_mlp_max: ns=%d unsupported (16 or 32)", ns);
    CASPR_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0 && C3 % 16 == 0 && C1 > 0 && C2 > 0 && C3 > 0,
                  "sa_mlp_max: layer widths must be multiples of 16 (%d,%d,%d)", C1, C2, C3);
    CASPR_REQUIRE(B > 0 && B <= 65535 && n > 0 && M > 0 && ldo >= out_off + C3, "sa_mlp_max: bad sizes");
    SaArgs a;
    a.xyz = xyz; a.new_xyz = new_xyz; a.feat = feat; a.idx = idx;
    a.ldf = ldf; a.n = n; a.M = M; a.C = C;
    const int K0 = ((C + 3) & ~3) + 3;
    a.L[0] = {w1p, b1, g1, be1, C1, 2 * ((K0 + 31) / 32)};
    a.L[1] = {w2p, b2, g2, be2, C2, 2 * ((C1 + 31) / 32)};
    a.L[2] = {w3p, b3, g3, be3, C3, 2 * ((C2 + 31) / 32)};
    a.out = out; a.ldo = ldo; a.out_off = out_off;
    const int rA = a.L[0].kc * 4 > a.L[2].kc * 4 ? a.L[0].kc * 4 : a.L[2].kc * 4;
    const int rB0 = a.L[1].kc * 4, rB1 = C3 / 4;
    a.rowsA = rA > C2 / 4 ? rA : C2 / 4;
    a.rowsB = rB0 > rB1 ? rB0 : rB1;
    if (a.rowsB < C1 / 4) a.rowsB = C1 / 4;
    const int bigK = (K0 > 160) || (C3 > 128);
    const int ncol = bigK ? 32 : 64;
    const size_t shmem = (size_t)(a.rowsA + a.rowsB) * ncol * 16 + (3 * 64 + 16 + 64) * 4 + 64;
    CASPR_REQUIRE(shmem <= 160 * 1024, "sa_mlp_max: needs %zu bytes of LDS (> 160 KiB)", shmem);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (ns == 16 && ncol == 64) rc = launch_sa<16, 64>(a, B, shmem, st);
    else if (ns == 32 && ncol == 64) rc = launch_sa<32, 64>(a, B, shmem, st);
    else if (ns == 16 && ncol == 32) rc = launch_sa<16, 32>(a, B, shmem, st);
    else rc = launch_sa<32, 32>(a, B, shmem, st);
    if (rc != CASPR_OK) return rc;
    CASPR_CHECK_LAUNCH("sa_mlp_max");
    return CASPR_OK;
}
