// sa_mlp.hip -- fused set-abstraction scale: grouping + per-neighbourhood point MLP for gfx950.
// Replaces PointNet2GroupingLayer + PointNetFeatureExtractor (models/pointnet2.py:391-409,649-703).
//
// One 256-thread workgroup owns NCOL = NCEN*NS columns (NCEN centres x NS neighbours).  The
// gathered neighbourhood (xyz - centre || features) is staged ONCE in LDS as an MFMA B-tile; the
// three conv(k=1) layers run as f32 MFMA 16x16x4 chains whose D fragments are written straight
// back as the next layer's B-tile; GroupNorm(16) statistics are per neighbourhood (pointnet2.py:642
// -> per column block of NS), two-pass in LDS; the final max over the NS samples is fused.  The
// (B,M,C+3,ns) grouped tensor (22.7 MB / frame at N=2048) is never written to HBM.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

struct SaLayer {
    const float *wp, *bias, *gamma, *beta;
    int cout;  // multiple of 16
    int kc;    // packed 16-wide K chunks (even)
    int kcr;   // chunks that carry inputs (<= kc; the pack's trailing all-zero chunk is neither staged nor multiplied by the LDS kernel)
};

struct SaArgs {
    const float *xyz, *new_xyz, *feat;
    const int32_t *idx;
    int ldf, n, M, C;
    int feat_kind;     // CASPR_FEAT_QUAD | CASPR_FEAT_PAIRS: feat = quadratic augmentation of xyz (caspr_prep_input_f32)
    int lo_in, lo_out; // CASPR_FEAT_LO_IN: feat rows carry low parts at column ldf / 2; CASPR_FEAT_LO_OUT: the low part of the output goes to column ldo / 2 + out_off
    int repair_kmax;   // balls of 1 .. repair_kmax distinct samples are evaluated by sa_repair_f64_kernel behind the register kernel (0: none)
    const int32_t *order, *count;   // the register kernel's list of the OTHER balls, per cloud (sa_list_kernel); NULL: every ball, in order
    SaLayer L[3];
    float *out;
    int ldo, out_off;
    int rowsA, rowsB;  // kq rows of the two ping-pong B-tiles
    int split_last;    // LDS kernel: the last layer in two halves of cout / 2 channels (8 GroupNorm groups each) through a half-size tile
    const float *pre;  // LDS kernel, pre-aggregated first layer: pre[b, point, 0:C1] = W_f . feat of every SOURCE point (ldp floats per row) ...
    int ldp;
    const float *wx;   // ... and the layer's three coordinate columns, (C1, 3) row-major: y_1 = pre[sample] + wx . (p - centre) + bias
    unsigned long long *trace;   // debug stamps (NULL in production)
};

#ifdef CASPR_DEBUG_HOOKS
#define SAM_STAMP(i) if (a.trace && blockIdx.x == 3 && blockIdx.y == 0 && threadIdx.x == 0) a.trace[i] = __builtin_amdgcn_s_memtime();
#else
#define SAM_STAMP(i)
#endif
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
    float *s_d = reinterpret_cast<float *>(s_idx + NCOL);    // [NCOL*4]  p - centre of every column (pre-aggregated first layer)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * NCEN;

    SAM_STAMP(0)
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

    if (a.pre) {
        // ---- PRE-AGGREGATED FIRST LAYER.  Layer 1 is linear in front of its GroupNorm and its feature part does not depend on the centre:
        // W [p - c ; f] = W_x (p - c) + W_f f, and W_f f is a property of the SOURCE point -- computed once per point by a plain conv over
        // the level's n points (caspr_sa_mlp_max_pre_f32's `pre`) instead of once per (centre, sample) pair: 8-16 times fewer rows at the
        // third level, 4-8 at the fourth, and the gather moves C1 floats per sample instead of C + 3.  The tile this pass fills IS layer 1's
        // output (pre-GroupNorm), so the layer's MFMA phase is skipped.
        if (tid < NCOL) {
            const float *p = a.xyz + ((long)b * a.n + s_idx[tid]) * 3;
            const float *c = s_cen + (tid / NS) * 4;
            s_d[tid * 4 + 0] = p[0] - c[0];          // the grouper's f32 subtraction (pointnet2.py:391-398)
            s_d[tid * 4 + 1] = p[1] - c[1];
            s_d[tid * 4 + 2] = p[2] - c[2];
        }
        __syncthreads();
        const int nq1 = a.L[0].cout >> 2;
        constexpr int GU = 4;
        const int nit = NCOL * nq1;
        for (int base = tid; base < nit; base += 256 * GU) {
            f32x4 v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int it = base + 256 * u < nit ? base + 256 * u : nit - 1;   // clamped: the store below is guarded
                const int kq = it % nq1, col = it / nq1;
                v[u] = ld4(a.pre + ((long)b * a.n + s_idx[col]) * a.ldp + kq * 4);
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int it = base + 256 * u;
                if (it >= nit) continue;
                const int kq = it % nq1, col = it / nq1;
                const f32x4 w0 = ld4(a.wx + kq * 12), w1 = ld4(a.wx + kq * 12 + 4), w2 = ld4(a.wx + kq * 12 + 8);   // rows 4 kq .. 4 kq + 3 of (C1, 3)
                const f32x4 bias4 = ld4(a.L[0].bias + kq * 4);
                const float d0 = s_d[col * 4 + 0], d1 = s_d[col * 4 + 1], d2 = s_d[col * 4 + 2];
                const float wr[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
                f32x4 r;
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = (fmaf(wr[3 * q + 2], d2, fmaf(wr[3 * q + 1], d1, wr[3 * q] * d0)) + v[u][q]) + bias4[q];
                st4(bufB + btile_off(kq, col, NCOL), r);
            }
        }
    } else
    // ---- gather: K order = [feat (C, padded to C4) | dx dy dz 0 | zeros ...]
    {
        const int C4 = (a.C + 3) & ~3;
        const int nkq = a.L[0].kcr * 4;
        const int qfeat = C4 >> 2;
        // feature quads: four independent rows in flight per thread, no branch around the loads -- with one load per iteration
        // the loop was a chain of dependent L2 round trips (16.5k of the kernel's 60k cycles at 64 + 3 input channels,
        // tools/sa_mlp_phase_trace.py)
        constexpr int GU = 4;
        const int nfeat = NCOL * qfeat;
        for (int base = tid; base < nfeat; base += 256 * GU) {
            f32x4 v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int it = base + 256 * u < nfeat ? base + 256 * u : nfeat - 1;   // clamped: the store below is guarded
                const int kq = it % qfeat, col = it / qfeat;
                v[u] = ld4(a.feat + ((long)b * a.n + s_idx[col]) * a.ldf + kq * 4);
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int it = base + 256 * u;
                const int kq = it % qfeat, col = it / qfeat;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] = kq * 4 + q < a.C ? v[u][q] : 0.f;
                if (it < nfeat) st4(bufA + btile_off(kq, col, NCOL), v[u]);
            }
        }
        // [dx dy dz 0] and the zero quads behind it
        const int nrest = nkq - qfeat;
        for (int it = tid; it < NCOL * nrest; it += 256) {
            const int kq = qfeat + it % nrest, col = it / nrest;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (kq == qfeat) {
                const float *p = a.xyz + ((long)b * a.n + s_idx[col]) * 3;
                const float *c = s_cen + (col / NS) * 4;
                v[0] = p[0] - c[0];
                v[1] = p[1] - c[1];
                v[2] = p[2] - c[2];
            }
            st4(bufA + btile_off(kq, col, NCOL), v);
        }
    }
    __syncthreads();

    SAM_STAMP(1)
    float *bin = bufA, *bout = bufB;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        // passes 0, 1: layers 1, 2; pass 2: layer 3 (or its first half); pass 3: layer 3's second half (split_last only)
        const int l = pass < 2 ? pass : 2;
        const int half = pass == 3 ? 1 : 0;
        if (pass == 3 && !a.split_last) break;
        const bool halved = l == 2 && a.split_last;
        const SaLayer L = a.L[l];
        const int RTall = L.cout >> 4;
        const int RT = halved ? RTall >> 1 : RTall;          // row tiles of this pass
        const int rt0 = halved ? half * RT : 0;              // ... starting at
        // ---- MFMA: work items (row tile, column group)
        if (pass == 0 && a.pre) {
            // (the gather filled this pass's output tile)
        } else if (RT >= 4) {
            for (int rtl = wave; rtl < RT; rtl += 4) {
                const int rt = rt0 + rtl;
                f32x4 acc[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float *wrow = L.wp + ((long)rt * L.kc) * 256 + lane * 4;
                // weight fragments two chunks ahead (two static registers, K chunk counts are even; clamped re-loads at the end
                // keep the loop free of branches -- with a branch hipcc waits for ALL outstanding loads at every chunk)
                const int klast = L.kcr - 1;
                f32x4 aq[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) aq[d] = ld4(wrow + (long)(d < klast ? d : klast) * 256);
                auto step = [&](int d, int kc) __attribute__((always_inline)) {
                    const f32x4 af = aq[d];
                    const int ka = kc + 4 < klast ? kc + 4 : klast;
                    aq[d] = ld4(wrow + (long)ka * 256);
                    f32x4 bf[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) bf[ct] = ld4(bin + btile_off(kc * 4 + g, ct * 16 + j, NCOL));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma16(af[q], bf[ct][q], acc[ct]);
                };
                const int k4 = L.kcr & ~3;
                for (int kc = 0; kc < k4; kc += 4) {
                    step(0, kc);
                    step(1, kc + 1);
                    step(2, kc + 2);
                    step(3, kc + 3);
                }
                // the tail: one to three chunks (the run count of layer 1 may be odd: its pack's last chunk is all padding)
                if (k4 < L.kcr) step(0, k4);
                if (k4 + 1 < L.kcr) step(1, k4 + 1);
                if (k4 + 2 < L.kcr) step(2, k4 + 2);
                f32x4 bias4 = ld4(L.bias + rt * 16 + 4 * g);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) st4(bout + btile_off(rtl * 4 + g, ct * 16 + j, NCOL), acc[ct] + bias4);
            }
        } else {
            for (int item = wave; item < RT * CT; item += 4) {
                const int rt = item / CT, ct = item % CT;
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float *wrow = L.wp + ((long)rt * L.kc) * 256 + lane * 4;
                for (int kc = 0; kc < L.kcr; ++kc) {
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
        SAM_STAMP(2 + 3 * l)

        // ---- GroupNorm statistics per (centre, group): two-pass over cpg*NS elements in LDS, in f64.
        // Most first-level neighbourhoods are padded with duplicates of one or two points; GroupNorm then
        // divides near-zero deviations by sqrt(var + 1e-5) -> up to 316x amplification of any rounding in
        // the mean.  f64 sums make mean/(x - mean) exact for such groups (the reference's f32 path is not).
        const int cpg = L.cout >> 4;
        const float inv_cpg = 1.0f / (float)cpg;
        const int g0 = halved ? 8 * half : 0, g1 = halved ? g0 + 8 : 16;      // GroupNorm groups this pass's tile holds
        const int q0 = (g0 * cpg) >> 2;                                       // ... whose first channel quad is row 0 of the tile
        {
            const int stat = tid / TPS, sub = tid % TPS;
            const int cen = stat >> 4, grp = stat & 15;
            const int cnt = cpg * NS;
            const bool mine = grp >= g0 && grp < g1;      // (a halved pass leaves the lanes of the other half's groups idle)
            // One pass in f64 (sum and sum of squares: with 53 bits the cancellation in E[x^2] - mean^2 is ~1e-14 absolute,
            // nothing next to eps = 1e-5), 16-byte reads where a group is made of whole channel quads: the scalar two-pass
            // version was a quarter to a third of this kernel's time (4-way bank conflicts on 16-byte-strided b32 reads).
            double s = 0.0, ss = 0.0;
            if (!mine) {
            } else if ((cpg & 3) == 0) {
                const int nq = cpg >> 2;
                for (int e = sub; e < nq * NS; e += TPS) {
                    const f32x4 x4 = ld4(bout + btile_off(grp * nq - q0 + e / NS, cen * NS + e % NS, NCOL));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double d = (double)x4[q];
                        s += d;
                        ss += d * d;
                    }
                }
            } else {
                // groups that are not whole channel quads (96 channels: 6 per group): 16-byte reads of every quad the group
                // touches, entries outside it masked -- the scalar form of this loop (4-way bank conflicts on 16-byte-strided
                // b32 reads) was 6.9k of the 68k cycles of the 64-96-128 scale
                const int c_lo = grp * cpg, c_hi = c_lo + cpg;
                const int q_lo = c_lo >> 2, nq = ((c_hi - 1) >> 2) - q_lo + 1;
                for (int e = sub; e < nq * NS; e += TPS) {
                    const int kq = q_lo + e / NS;
                    const f32x4 x4 = ld4(bout + btile_off(kq, cen * NS + e % NS, NCOL));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = kq * 4 + q;
                        const double d = (co >= c_lo && co < c_hi) ? (double)x4[q] : 0.0;
                        s += d;
                        ss += d * d;
                    }
                }
            }
            s = row_allreduce_add<TPS>(s);
            ss = row_allreduce_add<TPS>(ss);
            const double mean = s / (double)cnt;
            double v = ss / (double)cnt - mean * mean;
            v = v > 0.0 ? v * (double)cnt : 0.0;
            if (sub == 0 && mine) {
                s_mean[stat] = mean;
                s_rstd[stat] = __builtin_amdgcn_rsqf((float)(v / (double)cnt) + 1e-5f);
            }
        }
        __syncthreads();
        SAM_STAMP(3 + 3 * l)

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
                    const f32x4 ga = ld4(L.gamma + kq * 4), be = ld4(L.beta + kq * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = kq * 4 + q;
                        const int st = cen * 16 + (int)(((float)co + 0.5f) * inv_cpg);   // co / cpg without the integer division (co < 512, cpg <= 32: exact)
                        // (x - mean) first: exact for the near-constant neighbourhoods where rstd -> 1/sqrt(eps)
                        const float y = (float)((double)v[q] - s_mean[st]) * (s_rstd[st] * ga[q]) + be[q];
                        v[q] = y > 0.f ? y : 0.f;
                    }
                }
                st4(p, v);
            }
            __syncthreads();
            SAM_STAMP(4 + 3 * l)
            float *t = bin;
            bin = bout;
            bout = t;
        } else {
            // ---- last layer: GroupNorm (no ReLU) then max over the NS samples (pointnet2.py:690-698)
            // (four lanes per channel quad with 16-byte reads and a cross-lane max measured SLOWER than this scalar form: 5.9k vs 5.1k cycles)
            const int cpass = RT << 4, c0 = rt0 << 4;          // channels of this pass: c0 .. c0 + cpass - 1
            const float inv_cout = 1.0f / (float)cpass;
            for (int it = tid; it < NCEN * cpass; it += 256) {
                const int cen = (int)(((float)it + 0.5f) * inv_cout), cl = it - cen * cpass, co = c0 + cl;   // it < NCEN * cout <= 2048: exact
                if (m0 + cen >= a.M) continue;
                const int st = cen * 16 + (int)(((float)co + 0.5f) * inv_cpg);
                const float sc = s_rstd[st] * L.gamma[co], be = L.beta[co];
                const double mean = s_mean[st];
                float mx = -INFINITY;
                for (int s = 0; s < NS; ++s) {
                    const float y = (float)((double)bout[btile_off(cl >> 2, cen * NS + s, NCOL) + (cl & 3)] - mean) * sc + be;
                    mx = y > mx ? y : mx;
                }
                a.out[((long)b * a.M + m0 + cen) * a.ldo + a.out_off + co] = mx;
            }
            if (halved && half == 0) __syncthreads();          // the second half's products overwrite the tile this pass read
            SAM_STAMP(10)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Register-resident variant for the narrow levels (all three widths <= 64: SA1 and SA2, 327,680 + 163,840
// neighbourhoods per 160-frame batch).  The LDS kernel above is barrier-latency bound there (8 MFMA tiles of
// work between each pair of __syncthreads).  Here ONE WAVE owns 64 columns (4 centres x 16 or 2 centres x 32
// neighbours) from gather to max, with no LDS and no barrier:
//   * the gathered operand goes global -> VGPR in MFMA B-fragment shape (lane (g,j) loads the float4
//     k = 16kc+4g..+3 of its column's neighbour row);
//   * the D fragment of row tile mt IS the next layer's B fragment of chunk kc = mt (common.h), so the three
//     layers chain through registers;
//   * GroupNorm groups are 1, 2 or 4 channels wide = registers of one lane; their statistics reduce over the
//     16 lanes (x 1 or 2 column tiles) of the neighbourhood with xor-shuffles, in f64 (one pass: sum, sum of squares);
//   * max over the neighbourhood = the same shuffle pattern; lanes j == 0 store 4 channels (16 bytes) each.
// EVERY layer runs on CENTRED activations (round 4; round 2 centred layer 1 only).  A neighbourhood is carried as one REFERENCE
// column -- its sample 0, absolute values -- and the DEVIATIONS d_s = a_s - a_0 of the other samples; sample 0's own deviation is
// zero, so the reference rides in ITS column of the MFMA B operand: no extra tile, and y_s = W a_s + b = (W a_0 + b) + W d_s comes
// out of the same products (column 0: mu = W a_0, the others W d_s, each accurate to ITS OWN magnitude).  Why: on sparse clouds
// most neighbourhoods are padded with copies of the first hit plus a few near-duplicates, the per-neighbourhood GroupNorm keeps only
// the variation inside the neighbourhood and scales it by 1 / sqrt(var + 1e-5) <= 316 -- three times in a row.  In the plain form
// W a_s = (large common part) + (small variation) carries the f32 rounding of the common part, which every GroupNorm then amplifies:
// 2.7e-4 / 6.2e-4 on the outputs of the two 16-sample scales (round 3: the f32 CPU reference is 4.7e-4 / 6.2e-4 from f64 itself).
// In the centred form the error of every quantity is relative to the VARIATION: the first level's quadratic features are formed from
// exact coordinate differences (a b - a0 b0 = (a - a0) b + a0 (b - b0)), the statistics add mu back in f64, the normalised deviation is
// d_s * rstd * gamma (no subtraction), ReLU acts on (n_0, n_0 + delta_s) and returns a deviation again, the final max is n_0 + max(0,
// max_s delta_s).  An exact reformulation of pointnet2.py:677-698, not an approximation; exact duplicates stay exactly zero throughout.
// THE REFERENCE COLUMN IS CARRIED IN f64 (round 4, second half).  What the centred form could not fix: a neighbourhood that is ALL copies
// of one point has zero deviations, its output is the GroupNorm chain of the reference alone, and where two channels of a group differ by
// ~1e-4 (variance 1e-7 < eps) the f32 ACCUMULATION error of mu = W a_0 + b (1e-7 of its partial sums) is multiplied by rstd = 314, twice
// in a row: 4e-4 / 6e-4 on the two 16-sample scales, in this kernel as in the reference's own f32 arithmetic.  So mu is NOT taken from
// column 0 of the MFMA any more: every lane multiplies ITS weight fragment (A layout: row r = lane & 15, k = 16 kc + 4 (lane >> 4) + q)
// with the reference activations of its k in f64 on the vector pipe (v_fma_f64 runs at the f32 rate on CDNA: 1/16 .. 1/32 of the layer's
// products), the four k-rows are summed by permlane swaps, a 2 KB LDS window per wave turns row r into the D layout's rows 4 g + e, and
// GroupNorm normalises the reference in f64 (rstd: v_rsq_f32 + one Newton step) and hands a_0 to the next layer in f64 through the same
// window.  No barrier: the window is the wave's own.
// ---------------------------------------------------------------------------------------------
// sum of x over the four 16-lane rows of the wave (the same lane j of each row), in every lane: the 16-lane rows of a half by
// v_permlane16_swap, the two halves by v_permlane32_swap (gfx950), on both dwords of the double
__device__ __forceinline__ double sa_rows_allreduce(double x)
{
    unsigned lo = (unsigned)__builtin_bit_cast(unsigned long long, x), hi = (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32);
    auto both = [](unsigned l, unsigned h) { return __builtin_bit_cast(double, ((unsigned long long)h << 32) | l); };
    {
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        x = both(rl[0], rh[0]) + both(rl[1], rh[1]);
    }
    lo = (unsigned)__builtin_bit_cast(unsigned long long, x);
    hi = (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32);
    {
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        x = both(rl[0], rh[0]) + both(rl[1], rh[1]);
    }
    return x;
}

// (The second launch bound = resident waves per SIMD hipcc allocates registers for -- it changes the register target even where no limit
// binds.  Measured per variant with the f64 reference column in, ms per cfg-2 launch without / with a bound: <16,16,16,32> 0.52 / 0.45 (bound 3:
// 158 registers instead of 184), <32,32,32,64> 1.24 + 0.64 / 0.97 + 0.70 for its two levels (bound 2: 167 instead of 204), <16,32,32,64>
// 0.52 / 0.55 (240 / 197: no bound).  Hence one body and three entry points.)
// Between a lane's write to the wave's own LDS window (s_a0 / s_mu) and another lane's read of it: LDS operations of one wave issue in
// order, so no hardware barrier is needed -- but the COMPILER must keep the ds_write in front of the ds_read; the fence pair + the wave
// barrier (a scheduling fence, no instruction) make that a contract instead of an observation (round-4 advice).
#ifndef SA_EXP
#define SA_EXP 0     // timing experiments (tools/sa_repair_cost.py --lib): 1 no window sync, 2 no low-part pass
#endif
__device__ __forceinline__ void sa_window_sync()
{
#if !(SA_EXP & 1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// mcen[cen]: the neighbourhood (centre index) of this wave's centre slot cen, -1 for an empty slot (wave-uniform): see sa_small_entry
// SA_F32REF = 1 (round 5, second part): the reference column mu = W a_0 is read from the MFMA's own column 0 (f32, row_newbcast:0) again, and
// the f64 side computation described above (per-lane f64 products between the MFMAs, permlane row sums, the wave's LDS window, a_0 handed on in
// f64) is compiled out.  It was introduced in round 4 for neighbourhoods that are all copies of one point or nearly so -- since round 5 exactly
// those (1 .. 8 / 1 .. 4 distinct samples) are re-evaluated entirely in f64 by sa_repair_f64_kernel, and with the per-cloud list this kernel does
// not even compute them.  On the balls it still computes the two forms agree to the last digits of every check (tests/test_hip_parity.py, all flat
// at 1e-5: isolated scales 1.6e-6 / 2.8e-6 / 4.4e-6 either way, z0 on sparse cars 4.5e-6 vs 4.4e-6, cfg-5 T-NOCS 7.6e-7 vs 6.4e-7), the kernels
// need 117-143 registers instead of 134-220 and no LDS: first level's 32-sample scale 1.11 -> 0.97 ms.  SA_F32REF = 0 restores the f64 column.
#ifndef SA_F32REF
#define SA_F32REF 1
#endif
template <int NS, int C1, int C2, int C3>
__device__ __forceinline__ void sa_small_body(const SaArgs &a, const int (&mcen)[64 / NS])
{
    constexpr int CT = 4;                 // 64 columns per wave
    constexpr int TPC = NS / 16;          // column tiles per centre
    constexpr int NCEN = CT / TPC;        // centres per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.y;
    // the wave's own LDS window of the f64 reference column: a_0 of the layer input [centre][k], mu of the layer output [row tile][centre][row]
    __shared__ double s_a0[4][4][64], s_mu[4][4][4][16];
    double (&a0s)[4][64] = s_a0[wave];
    double (&mus)[4][4][16] = s_mu[wave];
    const int ar = lane & 15, ag = lane >> 4;          // A-layout coordinates of this lane: weight row ar of a row tile, k quad ag of a chunk
#ifdef CASPR_DEBUG_HOOKS
#define SA_STAMP(i) if (a.trace && blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0) a.trace[i] = __builtin_amdgcn_s_memtime();
#else
#define SA_STAMP(i)
#endif
    SA_STAMP(0)

    // ---- per column tile: neighbour row + centre
    int nrow[CT];
    float cx[CT], cy[CT], cz[CT];
    bool cval[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int cen = ct / TPC;
        cval[ct] = mcen[cen] >= 0;
        const int m = cval[ct] ? mcen[cen] : mcen[0];      // an empty slot computes slot 0's neighbourhood once more and stores nothing
        const int s = (ct % TPC) * 16 + j;
        nrow[ct] = a.idx[((long)b * a.M + m) * NS + s];
        const float *c = a.new_xyz + ((long)b * a.M + m) * 3;
        cx[ct] = c[0]; cy[ct] = c[1]; cz[ct] = c[2];
    }
    const int C4 = (a.C + 3) & ~3;
    // reference sample of each neighbourhood: its sample 0 (wave-uniform row index)
    int rrow[NCEN];
#pragma unroll
    for (int cen = 0; cen < NCEN; ++cen) rrow[cen] = __builtin_amdgcn_readlane(nrow[cen * TPC], 0);
    const bool refl = j == 0;                   // column 0 of a neighbourhood's first tile carries its reference (sample 0, absolute)
    SA_STAMP(1)

    // ---- layer 1: K order = [feat (C, padded to C4) | dx dy dz 0 | zeros]
    f32x4 h1[C1 / 16][CT];
#pragma unroll
    for (int rt = 0; rt < C1 / 16; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) h1[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int KC0 = a.L[0].kc;
    // input quad k..k+3 of cloud row `row` as layer 1 sees it: [feat | p - centre | 0]
    auto in_quad = [&](int row, int k, float ccx, float ccy, float ccz) -> f32x4 {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (k < C4) {
            v = ld4(a.feat + ((long)b * a.n + row) * a.ldf + k);
        } else if (k == C4) {
            const float *p = a.xyz + ((long)b * a.n + row) * 3;
            v[0] = p[0] - ccx;
            v[1] = p[1] - ccy;
            v[2] = p[2] - ccz;
        }
        return v;
    };
    // centred input quad k..k+3 of row `row` against the reference row `r0` when the features are the quadratic augmentation
    // of the coordinates: every entry is a difference of products of NEARBY coordinates, formed from the exact coordinate
    // differences -- a b - a0 b0 = (a - a0) b + a0 (b - b0), a^2 - a0^2 = (a - a0)(a + a0) -- so its error is relative to the
    // variation inside the ball, not to the (100x larger) absolute feature value
    auto aug_quad = [&](int row, int r0, int k) -> f32x4 {
        const float *p = a.xyz + ((long)b * a.n + row) * 3, *q = a.xyz + ((long)b * a.n + r0) * 3;
        const float x = p[0], y = p[1], z = p[2], x0 = q[0], y0 = q[1], z0 = q[2];
        const float dx = x - x0, dy = y - y0, dz = z - z0;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int c = 0;
        if (a.feat_kind & CASPR_FEAT_QUAD) {
            f[0] = dx * (x + x0);
            f[1] = dy * (y + y0);
            f[2] = dz * (z + z0);
            c = 3;
        }
        if (a.feat_kind & CASPR_FEAT_PAIRS) {
            f[c + 0] = dx * z + x0 * dz;   // xz
            f[c + 1] = dx * y + x0 * dy;   // xy
            f[c + 2] = dz * y + z0 * dy;   // zy
        }
        if (k == C4) return (f32x4){dx, dy, dz, 0.f};
        return k == 0 ? (f32x4){f[0], f[1], f[2], f[3]} : (k == 4 ? (f32x4){f[4], f[5], f[6], f[7]} : (f32x4){0.f, 0.f, 0.f, 0.f});
    };
    // B fragments of chunk kc straight from global memory: deviations from the neighbourhood's sample 0, and in sample 0's own column
    // (lane j = 0 of the neighbourhood's first tile, whose deviation is zero) the reference itself
    auto gather = [&](f32x4(&bf)[CT], int kc) {
        const int k = kc * 16 + 4 * g;
        f32x4 ref[NCEN];
#pragma unroll
        for (int cen = 0; cen < NCEN; ++cen) ref[cen] = in_quad(rrow[cen], k, cx[cen * TPC], cy[cen * TPC], cz[cen * TPC]);
        if (a.feat_kind) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bf[ct] = aug_quad(nrow[ct], rrow[ct / TPC], k);
        } else {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bf[ct] = in_quad(nrow[ct], k, cx[ct], cy[ct], cz[ct]) - ref[ct / TPC];
        }
#pragma unroll
        for (int cen = 0; cen < NCEN; ++cen)
            if (refl) bf[cen * TPC] = ref[cen];
    };
    auto mask = [&](f32x4(&bf)[CT], int kc) {       // zero the padding lanes of the feature quad that straddles C
        const int k = kc * 16 + 4 * g;
        if (k < C4 && k + 3 >= a.C) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k + q >= a.C) bf[ct][q] = 0.f;
        }
    };
    double P1[C1 / 16][NCEN];                        // f64 partial of mu = W a_0 (this lane's weight row ar, its k quad of every chunk)
#pragma unroll
    for (int rt = 0; rt < C1 / 16; ++rt)
#pragma unroll
        for (int cen = 0; cen < NCEN; ++cen) P1[rt][cen] = 0.0;
    auto mma1 = [&](const f32x4(&bf)[CT], const f32x4(&af)[C1 / 16]) {
        // the reference quads of this lane's k (exact f32 inputs) ride in column 0 of each neighbourhood's first tile: row broadcast; the
        // f64 products sit BETWEEN the MFMAs of the chunk (behind them, fenced, the widest variant lost a quarter: 0.44 -> 0.55 ms)
        double rq[NCEN][4];
        if (!SA_F32REF) {
#pragma unroll
            for (int cen = 0; cen < NCEN; ++cen)
#pragma unroll
                for (int q = 0; q < 4; ++q) rq[cen][q] = (double)dpp_mov<0x150>(bf[cen * TPC][q]);      // row_newbcast:0
        }
#pragma unroll
        for (int rt = 0; rt < C1 / 16; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) h1[rt][ct] = mfma16(af[rt][q], bf[ct][q], h1[rt][ct]);
                if (!SA_F32REF) {
                    const double wq = (double)af[rt][q];
#pragma unroll
                    for (int cen = 0; cen < NCEN; ++cen) P1[rt][cen] = __builtin_fma(wq, rq[cen][q], P1[rt][cen]);
                }
            }
    };
    auto load_a1 = [&](f32x4(&af)[C1 / 16], int kc) {
#pragma unroll
        for (int rt = 0; rt < C1 / 16; ++rt) af[rt] = ld4(a.L[0].wp + (((long)rt * KC0 + kc) * 64 + lane) * 4);
    };
    {   // two register sets: chunk kc+1's gather + weights are in flight while chunk kc multiplies.  The pack holds an EVEN number
        // of 16-wide chunks (KC0); only those that carry inputs are gathered and multiplied -- 1 of 2 at the first level (9 + 3
        // inputs), 7 of 8 at the second (96 + 3): the all-zero chunk was a fifth of the first level's MFMAs
        const int kcu = (C4 + 4 + 15) >> 4;
        f32x4 b0[CT], b1[CT], w0[C1 / 16], w1[C1 / 16];
        gather(b0, 0);
        load_a1(w0, 0);
        for (int kc = 0; kc + 1 < kcu; kc += 2) {
            gather(b1, kc + 1);
            load_a1(w1, kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            mask(b0, kc);
            mma1(b0, w0);
            if (kc + 2 < kcu) {
                gather(b0, kc + 2);
                load_a1(w0, kc + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            mask(b1, kc + 1);
            mma1(b1, w1);
        }
        if (kcu & 1) {            // the last chunk of an odd count sits in set 0
            mask(b0, kcu - 1);
            mma1(b0, w0);
        }
    }
    // the four k-rows of the partials summed, then row ar of tile rt -> the window (every lane row writes the same value)
    auto publish_mu = [&](auto &P, auto RTc) {
        constexpr int RT = decltype(RTc)::value;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int cen = 0; cen < NCEN; ++cen) mus[rt][cen][ar] = sa_rows_allreduce(P[rt][cen]);
    };
    if (!SA_F32REF) {
        publish_mu(P1, std::integral_constant<int, C1 / 16>{});
        sa_window_sync();
    }

    // bias + GroupNorm(16) per neighbourhood on a register-resident layer output in the centred form: on entry column 0 of the
    // neighbourhood's first tile holds W a_0 (lane j = 0), every other column W d_s.  FINAL = false: ReLU, and the output is written back
    // in the same form (a_0 | a_s - a_0); FINAL = true (no activation, pointnet2.py:686-698): the max over the samples is stored.
    auto norm = [&](auto &h, auto RTc, const SaLayer &L, auto FINALc) {
        constexpr int RT = decltype(RTc)::value;
        constexpr bool FINAL = decltype(FINALc)::value;
        constexpr int CPG = RT;            // channels per group = (16*RT)/16
        static_assert(CPG == 1 || CPG == 2 || CPG == 4, "register GroupNorm handles widths 16, 32, 64");
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4 bias4 = ld4(L.bias + rt * 16 + 4 * g);
            const f32x4 ga = ld4(L.gamma + rt * 16 + 4 * g), be = ld4(L.beta + rt * 16 + 4 * g);
#pragma unroll
            for (int cen = 0; cen < NCEN; ++cen) {
                // mu = W a_0 + bias of this neighbourhood (rows 4g..4g+3) in f64 from the window; sample 0's deviation is zero (column 0 of
                // the MFMA carried W a_0 in f32: not used)
                __builtin_amdgcn_sched_barrier(0);      // one (row tile, centre) at a time: hoisted, the window reads of all of them cost 128 registers
                double mu[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (SA_F32REF) mu[e] = (double)dpp_mov<0x150>(h[rt][cen * TPC][e]) + (double)bias4[e];      // column 0 of the MFMA: W a_0 in f32
                    else mu[e] = mus[rt][cen][4 * g + e] + (double)bias4[e];
                    if (refl) h[rt][cen * TPC][e] = 0.f;
                }
                f32x4 mx = (f32x4){0.f, 0.f, 0.f, 0.f}, mlo = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sg = 0; sg < 4 / CPG; ++sg) {
                    double s1 = 0.0, s2 = 0.0;
#pragma unroll
                    for (int t = 0; t < TPC; ++t)
#pragma unroll
                        for (int r = 0; r < CPG; ++r) {
                            // one-channel groups: the constant cancels exactly, leave it out (sigma can be ~1e-2 of it)
                            const double o = CPG > 1 ? mu[sg * CPG + r] : 0.0;
                            const double x = (double)h[rt][cen * TPC + t][sg * CPG + r] + o;
                            s1 += x;
                            s2 += x * x;
                        }
                    s1 = row_allreduce_add<16>(s1);
                    s2 = row_allreduce_add<16>(s2);
                    constexpr double inv = 1.0 / (double)(CPG * NS);
                    const double mean = s1 * inv;
                    double var = s2 * inv - mean * mean;      // f64: exact enough even when var << mean^2
                    var = var < 0.0 ? 0.0 : var;
                    const float rstd = __builtin_amdgcn_rsqf((float)var + 1e-5f);
                    // the reference needs rstd beyond f32: one Newton step of r <- r (1.5 - 0.5 v r^2) in f64
                    double r64 = (double)rstd;
                    r64 = r64 * (1.5 - 0.5 * (var + 1e-5) * r64 * r64);
#pragma unroll
                    for (int r = 0; r < CPG; ++r) {
                        const int e = sg * CPG + r;
                        const float sc = rstd * ga[e];
                        const double n64 = ((CPG > 1 ? mu[e] : 0.0) - mean) * (r64 * (double)ga[e]) + (double)be[e];      // the reference, normalised
                        const float n0 = (float)n64;
                        if (FINAL) {
                            float dm = 0.f;
#pragma unroll
                            for (int t = 0; t < TPC; ++t) dm = fmaxf(dm, h[rt][cen * TPC + t][e] * sc);
                            dm = row_allreduce_max<16>(dm);
                            mx[e] = n0 + dm;
                            // what the f32 output lacks of (f64 reference + largest deviation): stored next to it when the next level asks
                            mlo[e] = (float)((n64 + (double)dm) - (double)mx[e]);
                        } else {
                            const double a64 = n64 > 0.0 ? n64 : 0.0;
                            const float a0 = (float)a64;
                            if (!SA_F32REF) a0s[cen][16 * rt + 4 * g + e] = a64;          // the next layer's reference input, in f64 (every lane of the row
                                                                          // writes the same value: no branch for hipcc to sink 32 doubles into)
#pragma unroll
                            for (int t = 0; t < TPC; ++t) {
                                const float dl = h[rt][cen * TPC + t][e] * sc;       // normalised deviation: no subtraction
                                const float ns_ = n0 + dl;
                                float d = n0 > 0.f ? (ns_ > 0.f ? dl : -n0) : (ns_ > 0.f ? ns_ : 0.f);
                                if (t == 0 && refl) d = a0;
                                h[rt][cen * TPC + t][e] = d;
                            }
                        }
                    }
                }
                if (FINAL && refl && cval[cen * TPC]) {
                    st4(a.out + ((long)b * a.M + mcen[cen]) * a.ldo + a.out_off + rt * 16 + 4 * g, mx);
                    if (a.lo_out) st4(a.out + ((long)b * a.M + mcen[cen]) * a.ldo + (a.ldo >> 1) + a.out_off + rt * 16 + 4 * g, mlo);
                }
            }
        }
    };
    // next layer from a register-resident input: chunk kc of the K loop = row tile kc of the input
    auto layer = [&](auto &hout, auto RTo, const auto &hin, auto RTi, const SaLayer &L) {
        constexpr int RO = decltype(RTo)::value, RI = decltype(RTi)::value;
        f32x4 af[RO][RI];   // all weight fragments of the layer first (L1/L2 hits), then the MFMAs back to back
#pragma unroll
        for (int rt = 0; rt < RO; ++rt)
#pragma unroll
            for (int kc = 0; kc < RI; ++kc) af[rt][kc] = ld4(L.wp + (((long)rt * L.kc + kc) * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RO; ++rt) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) hout[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < RI; ++kc)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) hout[rt][ct] = mfma16(af[rt][kc][q], hin[kc][ct][q], hout[rt][ct]);
        }
        // mu = W a_0 of the reference column in f64: this lane's weight row ar against the a_0 of its k quads (the window), the four
        // k-rows summed, row ar published for norm()
        if (SA_F32REF) return;
#pragma unroll
        for (int cen = 0; cen < NCEN; ++cen) {          // one centre at a time: RO doubles live, not RO x NCEN
            double P[RO];
#pragma unroll
            for (int rt = 0; rt < RO; ++rt) P[rt] = 0.0;
#pragma unroll
            for (int kc = 0; kc < RI; ++kc) {
                double av[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) av[q] = a0s[cen][16 * kc + 4 * ag + q];
#pragma unroll
                for (int rt = 0; rt < RO; ++rt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) P[rt] = __builtin_fma((double)af[rt][kc][q], av[q], P[rt]);
            }
#pragma unroll
            for (int rt = 0; rt < RO; ++rt) mus[rt][cen][ar] = sa_rows_allreduce(P[rt]);
        }
        sa_window_sync();          // mus published -> norm() of this layer reads rows 4 g + e
    };
    using I1 = std::integral_constant<int, C1 / 16>;
    using I2 = std::integral_constant<int, C2 / 16>;
    using I3 = std::integral_constant<int, C3 / 16>;
    SA_STAMP(2)
    norm(h1, I1{}, a.L[0], std::false_type{});
    sa_window_sync();              // a0s of layer 1 written by every row's lanes -> layer 2 reads the k quads of other lanes
    SA_STAMP(3)
    f32x4 h2[C2 / 16][CT];
    layer(h2, I2{}, h1, I1{}, a.L[1]);
    SA_STAMP(4)
    norm(h2, I2{}, a.L[1], std::false_type{});
    sa_window_sync();
    SA_STAMP(5)
    f32x4 h3[C3 / 16][CT];
    layer(h3, I3{}, h2, I2{}, a.L[2]);
    SA_STAMP(6)
    norm(h3, I3{}, a.L[2], std::true_type{});      // + max over the NS samples of each centre (pointnet2.py:690-698), stored
    SA_STAMP(7)
    SA_STAMP(8)
}

// WHICH NEIGHBOURHOODS THE REGISTER KERNEL COMPUTES.  With the f64 re-evaluation behind it (repair_kmax > 0) every neighbourhood of
// 1 .. repair_kmax distinct samples is overwritten by sa_repair_f64_kernel; round 5 skipped a wave only when ALL of its centres were such
// balls (most waves of the first level's 16-sample scale, few elsewhere).  Now sa_list_kernel lists, per cloud and in ascending order,
// the neighbourhoods that kernel will NOT overwrite -- more than repair_kmax distinct samples, or an index row that is not in ball
// query's layout: the very test of sa_row_distinct -- into the caller's workspace (a.order: B x M entries, a.count: B), and the register
// kernel's waves take their centres from that list: no MFMA runs for a result the f64 kernel replaces (29 % of the first level's
// 32-sample scale, 25 % of the second level's 16-sample scale on the car clouds).  A neighbourhood's arithmetic does not depend on the
// slot it is computed in, so the outputs are the same bits as with the identity list (test_sa_register_kernel_is_slot_invariant).
// (A list per WORKGROUP, surveyed inside the kernel, was measured first: the partial last round of a 32-ball span and the loop's
// registers -- 168 -> 194 -- gave back more than the skipped balls saved on the second level.)
__global__ __launch_bounds__(256) void sa_list_kernel(const int32_t *__restrict__ idx, int M, int ns, int kmax, int32_t *__restrict__ order,
                                                      int32_t *__restrict__ count)
{
    // (256 threads: a 1024-thread workgroup needs half a CU's wave slots at once, and beside the index chain's kernels on the other
    // stream it waited for them -- 55 us on average inside the step against 6 us alone, profiles/r05b_kernel_stats.txt)
    __shared__ int s_wave[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;                                   // kept neighbourhoods so far (uniform)
    for (int m0 = 0; m0 < M; m0 += 256) {
        const int m = m0 + tid;
        bool big = false;
        if (m < M) {
            const int32_t *row = idx + ((long)b * M + m) * ns;
            const int first = row[0];
            unsigned diff = 0u;
            for (int q = 0; q < ns; q += 4) {
                const int4 v = *reinterpret_cast<const int4 *>(row + q);
                diff |= (v.x != first ? 1u : 0u) << q;
                diff |= (v.y != first ? 2u : 0u) << q;
                diff |= (v.z != first ? 4u : 0u) << q;
                diff |= (v.w != first ? 8u : 0u) << q;
            }
            const int K = __builtin_popcount(diff) + 1;
            const bool layout = diff == (unsigned)((1ull << K) - 2ull);      // the other hits occupy exactly the entries 1 .. K - 1
            big = !(layout && K <= kmax);
        }
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(big);
        if (lane == 0) s_wave[wave] = __builtin_popcountll(mk);
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            off += w < wave ? s_wave[w] : 0;
            tot += s_wave[w];
        }
        if (big) order[(long)b * M + off + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = m;
        base += tot;
        __syncthreads();                            // s_wave is rewritten by the next round
    }
    if (tid == 0) count[b] = base;
}

template <int NS, int C1, int C2, int C3>
__device__ __forceinline__ void sa_small_entry(const SaArgs &a)
{
    constexpr int NCEN = 64 / NS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int slot0 = (blockIdx.x * 4 + wave) * NCEN;
    const int n = a.order ? a.count[b] : a.M;
    if (slot0 >= n) return;               // wave-uniform; no barriers in this kernel
    int mcen[NCEN];
#pragma unroll
    for (int cen = 0; cen < NCEN; ++cen) {
        const int at = slot0 + cen < n ? slot0 + cen : slot0;
        const int m = a.order ? a.order[(long)b * a.M + at] : at;
        mcen[cen] = slot0 + cen < n ? __builtin_amdgcn_readfirstlane(m) : -1;
    }
    sa_small_body<NS, C1, C2, C3>(a, mcen);
}

template <int NS, int C1, int C2, int C3>
__global__ __launch_bounds__(256) void sa_small_kernel(SaArgs a) { sa_small_entry<NS, C1, C2, C3>(a); }
template <int NS, int C1, int C2, int C3>
__global__ __launch_bounds__(256, 2) void sa_small_kernel_w2(SaArgs a) { sa_small_entry<NS, C1, C2, C3>(a); }
template <int NS, int C1, int C2, int C3>
__global__ __launch_bounds__(256, 3) void sa_small_kernel_w3(SaArgs a) { sa_small_entry<NS, C1, C2, C3>(a); }

// ---------------------------------------------------------------------------------------------
// SMALL NEIGHBOURHOODS IN f64 (round 5).  What the centred form with an f64 reference column could not reach: a ball that holds 2..4
// distinct points has 1..3 distinct deviation columns, and a GroupNorm group of one or two channels in which ALL of them happen to cancel
// (|W d| ~ 1e-3 out of partial sums of magnitude ~4) has a variance below eps: rstd -> 316 multiplies the f32 ACCUMULATION error of the
// MFMA's W d (4e-7), three layers in a row -- 4e-4 on the isolated 16-sample scale of level 0, 3.5e-5 on cfg-5's T-NOCS, in this build as
// in the reference's own f32 arithmetic.  With K distinct points the coincidence needs K - 1 independent cancellations, so the tail dies
// quickly with K; the cases that matter are exactly the cheap ones.  This kernel runs BEHIND the register kernel and re-evaluates every
// neighbourhood with 2 <= K <= KMAX distinct samples entirely in f64 on the vector pipe (v_fma_f64 runs at the f32 rate on CDNA): the K
// distinct columns only, the statistics weighted with the columns' multiplicities (ball query pads with copies of the first hit:
// multiplicity ns - K + 1 for column 0, 1 for the others), and overwrites the register kernel's output rows.  16 lanes own one
// neighbourhood: lane r the output rows 16 rt + r of every row tile (GroupNorm groups of 1 / 2 / 4 channels = 1 / 2 / 4 adjacent lanes);
// the activations of a layer pass through a wave-private LDS window (f64), the weights come from the same A-pack the MFMA kernel reads.
// K = 1 (all copies of one point) is a single column here (instantiation <1, 1>): the register kernel's reference column is f64 too,
// but its INPUT is f32 -- the first level's quadratic features as caspr_prep_input_f32 rounds them (x^2 ~ 6 carries 4e-7), the second
// level's features as the first stored them -- and a two-channel group whose channels differ by ~1e-4 multiplies that rounding by up
// to 158 (4e-4 on cfg-5's clouds); here the features are the exact f64 products of the coordinates (feat_kind) or hi + lo (lo_in).
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int sa_row_bcast_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

// (ball of row `row`: K = number of distinct samples when the row has ball query's layout -- entry 0 the first hit, entries 1..K-1 the
// other hits, then copies of entry 0 -- else 0; the 16 lanes of a DPP row look at one neighbourhood)
__device__ __forceinline__ int sa_row_distinct(const int32_t *row, int ns, int r, int &i0, int &first)
{
    i0 = row[r];
    const int i1 = ns > 16 ? row[16 + r] : 0;
    first = sa_row_bcast_i32<0x150>(i0);        // row_newbcast:0
    int cnt = (i0 != first ? 1 : 0) + ((ns > 16 && i1 != first) ? 1 : 0);
    cnt += sa_row_bcast_i32<0xB1>(cnt);
    cnt += sa_row_bcast_i32<0x4E>(cnt);
    cnt += sa_row_bcast_i32<0x141>(cnt);
    cnt += sa_row_bcast_i32<0x140>(cnt);
    const int K = cnt + 1;
    int bad = ((i0 != first) != (r >= 1 && r < K) ? 1 : 0) + ((ns > 16 && ((i1 != first) != (16 + r < K))) ? 1 : 0);
    bad += sa_row_bcast_i32<0xB1>(bad);
    bad += sa_row_bcast_i32<0x4E>(bad);
    bad += sa_row_bcast_i32<0x141>(bad);
    bad += sa_row_bcast_i32<0x140>(bad);
    return bad == 0 ? K : 0;
}

// KLO <= K <= KHI distinct samples.  A workgroup surveys 128 neighbourhoods, lists the ones in its range in LDS, and its 16 lane groups
// walk that list: on dense clouds the kernel is the index read, on sparse ones the lane groups stay busy whatever the pattern of
// small balls (with a fixed neighbourhood per lane group a wave paid for four whenever one of them was small).
template <int KLO, int KHI, int WMAX>
__global__ __launch_bounds__(256, KHI > 4 ? 3 : 4) void sa_repair_f64_kernel(SaArgs a, int ns)
{
    constexpr int KMAX = KHI;
    constexpr int CACT = 32;                              // widest first / second layer: what passes through the LDS window
    constexpr int RTA = CACT / 16;                        // their row tiles, all kept in registers at once (RTA x KMAX doubles)
    constexpr int NBW = 128;                              // neighbourhoods surveyed per workgroup
    constexpr int STR = KMAX * CACT + 2;                  // + 2 doubles: the 16 windows of a workgroup start in different banks
    // WMAX: floats of the three A-packs together (2048 / 4096 at the first level's two scales, 7168 at the second level's): the LDS a
    // workgroup takes decides how many neighbourhoods a CU keeps in flight, and this kernel is all latency
    __shared__ double s_act[16 * STR];
    __shared__ __attribute__((aligned(16))) float s_w[WMAX];
    __shared__ float s_par[3][3][64];                     // [layer][bias | gamma | beta][channel]
    __shared__ int s_list[NBW];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, r = tid & 15, nb = tid >> 4;
    const int b = blockIdx.y;
    double *act = s_act + nb * STR;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // (unrolled: the eight index rows of a lane group are eight independent loads, not a chain of eight round trips)
#pragma unroll
    for (int i = 0; i < NBW / 16; ++i) {
        const int m = blockIdx.x * NBW + nb + 16 * i;
        const int mm = m < a.M ? m : a.M - 1;
        int i0, first;
        const int K = sa_row_distinct(a.idx + ((long)b * a.M + mm) * ns, ns, r, i0, first);
        if (m < a.M && K >= KLO && K <= KHI && r == 0) s_list[atomicAdd(&s_cnt, 1)] = m;
    }
    __syncthreads();
    const int nlist = s_cnt;
    if (nlist == 0) return;                               // dense clouds: the kernel was the index read
    const int C4 = (a.C + 3) & ~3;
    // The three A-packs (the ones the MFMA kernel reads) and the per-channel parameters, once per workgroup into LDS: a neighbourhood
    // is a chain of ~20 dependent weight reads, and out of L2 each of them was a microsecond
    int woff[3];
    {
        int o = 0;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            woff[l] = o;
            const int nf = (a.L[l].cout >> 4) * a.L[l].kc * 256;
            for (int i = tid * 4; i < nf; i += 1024) st4(s_w + o + i, ld4(a.L[l].wp + i));
            o += nf;
            if (tid < a.L[l].cout) {
                s_par[l][0][tid] = a.L[l].bias[tid];
                s_par[l][1][tid] = a.L[l].gamma[tid];
                s_par[l][2][tid] = a.L[l].beta[tid];
            }
        }
    }
    __syncthreads();
    // the weight quad W[row 16 rt + r][k = 4 kq .. 4 kq + 3] of this lane
    auto wquad = [&](int l, const SaLayer &L, int rt, int kq) -> f32x4 {
        return ld4(s_w + woff[l] + ((rt * L.kc + (kq >> 2)) * 64 + (kq & 3) * 16 + r) * 4);
    };
    // bias + GroupNorm(16) of row tile rt (channels per group = RT = adjacent lanes of the tile) with the columns' multiplicities, two
    // passes in f64; on return y holds the normalised values (no activation)
    auto norm_tile = [&](double (&y)[KMAX], int l, int rt, int RT, const double (&mult)[KMAX]) {
        const int ch = rt * 16 + r;
        const double bias = (double)s_par[l][0][ch], ga = (double)s_par[l][1][ch], be = (double)s_par[l][2][ch];
        const double inv_cnt = 1.0 / (double)(RT * ns);
        double s1 = 0.0;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            y[c] += bias;
            s1 += mult[c] * y[c];
        }
        if (RT >= 2) s1 += dpp_mov<0xB1>(s1);
        if (RT >= 4) s1 += dpp_mov<0x4E>(s1);
        const double mean = s1 * inv_cnt;
        double s2 = 0.0;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            const double d = y[c] - mean;
            s2 += mult[c] * d * d;
        }
        if (RT >= 2) s2 += dpp_mov<0xB1>(s2);
        if (RT >= 4) s2 += dpp_mov<0x4E>(s2);
        // 1 / sqrt(var + eps): v_rsq_f64 (~27 bits) + two Newton steps r <- r (1.5 - 0.5 v r^2) (54+ bits) -- a dozen f64 operations
        // where the correctly rounded sqrt + division are ~100 dependent ones, three to twelve times per neighbourhood
        const double vv = s2 * inv_cnt + 1e-5;
        double rstd = __builtin_amdgcn_rsq(vv);
        rstd = rstd * (1.5 - 0.5 * vv * rstd * rstd);
        rstd = rstd * (1.5 - 0.5 * vv * rstd * rstd);
#pragma unroll
        for (int c = 0; c < KMAX; ++c) y[c] = (y[c] - mean) * (rstd * ga) + be;
    };
#pragma unroll 1
    for (int li = nb; li < nlist; li += 16) {
        const int m = s_list[li];
        int i0, first;
        const int K = sa_row_distinct(a.idx + ((long)b * a.M + m) * ns, ns, r, i0, first);
        int rows[KMAX];
        rows[0] = first;
        if (KMAX > 1) rows[1] = sa_row_bcast_i32<0x151>(i0);
        if (KMAX > 2) rows[2] = sa_row_bcast_i32<0x152>(i0);
        if (KMAX > 3) rows[3] = sa_row_bcast_i32<0x153>(i0);
        if (KMAX > 4) rows[4] = sa_row_bcast_i32<0x154>(i0);
        if (KMAX > 5) rows[5] = sa_row_bcast_i32<0x155>(i0);
        if (KMAX > 6) rows[6] = sa_row_bcast_i32<0x156>(i0);
        if (KMAX > 7) rows[7] = sa_row_bcast_i32<0x157>(i0);
#pragma unroll
        for (int c = 0; c < KMAX; ++c) rows[c] = c < K ? rows[c] : first;
        double mult[KMAX];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) mult[c] = c == 0 ? (double)(ns - K + 1) : (c < K ? 1.0 : 0.0);

        // ---- layer 1 from global memory: K order [feat (C, padded to C4) | dx dy dz 0]; both row tiles at once (the inputs are loaded once)
        double y[RTA][KMAX];
        {
            const SaLayer L = a.L[0];
            const int RT = L.cout >> 4;
#pragma unroll
            for (int rt = 0; rt < RTA; ++rt)
#pragma unroll
                for (int c = 0; c < KMAX; ++c) y[rt][c] = 0.0;
            const float *cen = a.new_xyz + ((long)b * a.M + m) * 3;
            const float ccx = cen[0], ccy = cen[1], ccz = cen[2];
            const int nkq = (C4 >> 2) + 1;
            for (int kq = 0; kq < nkq; ++kq) {
                f32x4 w[RTA];
#pragma unroll
                for (int rt = 0; rt < RTA; ++rt) w[rt] = wquad(0, L, rt < RT ? rt : 0, kq);
                // one column at a time (its four inputs live only across its own products: the register budget decides how many
                // neighbourhoods a CU keeps in flight, and this kernel is all latency)
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    double x[4];
                    const float *p = a.xyz + ((long)b * a.n + rows[c]) * 3;
                    if (kq * 4 == C4) {
                        x[0] = (double)(p[0] - ccx);              // the grouper's f32 subtraction (pointnet2.py:391-398)
                        x[1] = (double)(p[1] - ccy);
                        x[2] = (double)(p[2] - ccz);
                        x[3] = 0.0;
                    } else if (a.feat_kind) {
                        // the first level's quadratic augmentation (tpointnet2.py:79-90) from the coordinates, in f64
                        const double px = (double)p[0], py = (double)p[1], pz = (double)p[2];
                        const bool qd = a.feat_kind & CASPR_FEAT_QUAD, both = qd && (a.feat_kind & CASPR_FEAT_PAIRS);
                        const double xz = px * pz, xy = px * py, zy = pz * py;
                        const double f0 = qd ? px * px : xz, f1 = qd ? py * py : xy, f2 = qd ? pz * pz : zy;
                        x[0] = kq == 0 ? f0 : (both ? xy : 0.0);
                        x[1] = kq == 0 ? f1 : (both ? zy : 0.0);
                        x[2] = kq == 0 ? f2 : 0.0;
                        x[3] = kq == 0 ? (both ? xz : 0.0) : 0.0;
                    } else {
                        const float *fr = a.feat + ((long)b * a.n + rows[c]) * a.ldf + kq * 4;
                        const f32x4 v = ld4(fr);
                        const f32x4 vl = a.lo_in ? ld4(fr + (a.ldf >> 1)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] = kq * 4 + q < a.C ? (double)v[q] + (double)vl[q] : 0.0;
                    }
#pragma unroll
                    for (int rt = 0; rt < RTA; ++rt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[rt][c] = __builtin_fma((double)w[rt][q], x[q], y[rt][c]);
                }
            }
        }
#pragma unroll 1
        for (int l = 0; l < 2; ++l) {
            const SaLayer L = a.L[l];
            const int RT = L.cout >> 4;
            if (l == 1) {
                // ---- layer 2 from the window (written by layer 1 below)
                const int nkq = a.L[0].cout >> 2;
#pragma unroll
                for (int rt = 0; rt < RTA; ++rt)
#pragma unroll
                    for (int c = 0; c < KMAX; ++c) y[rt][c] = 0.0;
                for (int kq = 0; kq < nkq; ++kq) {
                    f32x4 w[RTA];
#pragma unroll
                    for (int rt = 0; rt < RTA; ++rt) w[rt] = wquad(1, L, rt < RT ? rt : 0, kq);
#pragma unroll
                    for (int c = 0; c < KMAX; ++c) {
                        double x[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] = act[c * CACT + kq * 4 + q];
#pragma unroll
                        for (int rt = 0; rt < RTA; ++rt)
#pragma unroll
                            for (int q = 0; q < 4; ++q) y[rt][c] = __builtin_fma((double)w[rt][q], x[q], y[rt][c]);
                    }
                }
                // every lane of the neighbourhood has read the window before layer 2's activations overwrite it
                sa_window_sync();
            }
#pragma unroll
            for (int rt = 0; rt < RTA; ++rt) {
                if (rt < RT) {
                    norm_tile(y[rt], l, rt, RT, mult);
#pragma unroll
                    for (int c = 0; c < KMAX; ++c) act[c * CACT + rt * 16 + r] = y[rt][c] > 0.0 ? y[rt][c] : 0.0;      // ReLU
                }
            }
            sa_window_sync();
        }
        // ---- layer 3 one row tile at a time (up to four: kept together they were half of this kernel's registers), GroupNorm without
        // activation, max over the columns (each occurs at least once), stored as hi (+ lo)
        {
            const SaLayer L = a.L[2];
            const int RT = L.cout >> 4;
            const int nkq = a.L[1].cout >> 2;
#pragma unroll 1
            for (int rt = 0; rt < RT; ++rt) {
                double y3[KMAX];
#pragma unroll
                for (int c = 0; c < KMAX; ++c) y3[c] = 0.0;
                for (int kq = 0; kq < nkq; ++kq) {
                    const f32x4 w = wquad(2, L, rt, kq);
#pragma unroll
                    for (int c = 0; c < KMAX; ++c) {
                        double x[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] = act[c * CACT + kq * 4 + q];
#pragma unroll
                        for (int q = 0; q < 4; ++q) y3[c] = __builtin_fma((double)w[q], x[q], y3[c]);
                    }
                }
                norm_tile(y3, 2, rt, RT, mult);
                double mx = y3[0];
#pragma unroll
                for (int c = 1; c < KMAX; ++c) mx = (c < K && y3[c] > mx) ? y3[c] : mx;
                float *o = a.out + ((long)b * a.M + m) * a.ldo + a.out_off + rt * 16 + r;
                const float hi = (float)mx;
                o[0] = hi;
                if (a.lo_out) o[a.ldo >> 1] = (float)(mx - (double)hi);
            }
        }
        sa_window_sync();          // the window is free for this lane group's next neighbourhood
    }
}

template <int NS, int NCOL>
static int launch_sa(const SaArgs &a, int B, size_t shmem, hipStream_t st)
{
    auto kern = sa_mlp_kernel<NS, NCOL>;
    if (shmem > 64 * 1024) {
        // the opt-in is an upper bound: ask for the device maximum once per (kernel, device) so that later, larger layers need no second call
        static CasprLdsOptIn optin;
        const hipError_t e = caspr_lds_opt_in(optin, (const void *)kern, 160 * 1024);
        if (e != hipSuccess) {
            caspr_set_error("sa_mlp_max: hipFuncSetAttribute(%zu) failed: %s", shmem, hipGetErrorString(e));
            return CASPR_ELAUNCH;
        }
    }
    constexpr int NCEN = NCOL / NS;
    kern<<<dim3(ceil_div(a.M, NCEN), B), dim3(256), shmem, st>>>(a);
    return CASPR_OK;
}

#ifdef CASPR_DEBUG_HOOKS
static unsigned long long *g_sa_trace = nullptr;
extern "C" void caspr_debug_set_sa_trace(unsigned long long *dev_buf) { g_sa_trace = dev_buf; }   // debug build only
#endif

static int sa_mlp_max_impl(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                           const int32_t *idx, int B, int n, int M, int C, int ns, int feat_kind, const float *w1p,
                           const float *b1, const float *g1, const float *be1, int C1, const float *w2p,
                           const float *b2, const float *g2, const float *be2, int C2, const float *w3p,
                           const float *b3, const float *g3, const float *be3, int C3, float *out, int ldo,
                           int out_off, int32_t *workspace, void *stream, const float *pre = nullptr, int ldp = 0, const float *wx = nullptr)
{
    CASPR_REQUIRE(xyz && new_xyz && idx && out && (w1p || pre) && w2p && w3p && b1 && b2 && b3 && g1 && g2 && g3 && be1 && be2 && be3,
                  "sa_mlp_max: null pointer");
    CASPR_REQUIRE(!pre || (wx && C == 0 && !feat && feat_kind == 0 && C1 >= 64 && ldp % 4 == 0 && ldp >= C1 && ((uintptr_t)pre % 16) == 0 && ((uintptr_t)wx % 16) == 0),
                  "sa_mlp_max_pre: needs wx, no feat / C / feat_kind, a first layer of >= 64 channels (the LDS kernel's shapes), pre rows of ldp %% 4 == 0 >= C1 floats");
    CASPR_REQUIRE(C == 0 || (feat && ldf % 4 == 0 && ldf >= ((C + 3) & ~3)), "sa_mlp_max: feat/ldf invalid (C=%d ldf=%d)", C, ldf);
    CASPR_REQUIRE(ns == 16 || ns == 32, "sa_mlp_max: ns=%d unsupported (16 or 32)", ns);
    CASPR_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0 && C3 % 16 == 0 && C1 > 0 && C2 > 0 && C3 > 0,
                  "sa_mlp_max: layer widths must be multiples of 16 (%d,%d,%d)", C1, C2, C3);
    CASPR_REQUIRE(B > 0 && B <= 65535 && n > 0 && M > 0 && ldo >= out_off + C3, "sa_mlp_max: bad sizes");
    SaArgs a;
    a.xyz = xyz; a.new_xyz = new_xyz; a.feat = feat; a.idx = idx;
    a.ldf = ldf; a.n = n; a.M = M; a.C = C;
    const int want_c = ((feat_kind & CASPR_FEAT_QUAD) ? 3 : 0) + ((feat_kind & CASPR_FEAT_PAIRS) ? 3 : 0);
    const int aug = feat_kind & (CASPR_FEAT_QUAD | CASPR_FEAT_PAIRS);
    CASPR_REQUIRE(feat_kind >= 0 && feat_kind < 64 && (aug == 0 || C == want_c), "sa_mlp_max: feat_kind=%d does not describe C=%d channels", feat_kind, C);
    const bool only_mfma = (feat_kind & CASPR_SA_ONLY_MFMA) != 0, only_f64 = (feat_kind & CASPR_SA_ONLY_F64) != 0;
    CASPR_REQUIRE(!(only_mfma && only_f64), "sa_mlp_max: CASPR_SA_ONLY_MFMA and CASPR_SA_ONLY_F64 exclude each other");
    a.feat_kind = aug;
    a.lo_in = (feat_kind & CASPR_FEAT_LO_IN) ? 1 : 0;
    a.lo_out = (feat_kind & CASPR_FEAT_LO_OUT) ? 1 : 0;
    CASPR_REQUIRE(!a.lo_in || (aug == 0 && feat && ldf % 8 == 0 && ldf / 2 >= ((C + 3) & ~3)), "sa_mlp_max: CASPR_FEAT_LO_IN needs feat rows of [C | low parts of C] (ldf=%d, C=%d)", ldf, C);
    CASPR_REQUIRE(!a.lo_out || (ldo % 8 == 0 && ldo / 2 >= out_off + C3), "sa_mlp_max: CASPR_FEAT_LO_OUT needs out rows of [channels | their low parts] (ldo=%d)", ldo);
    const int K0 = ((C + 3) & ~3) + 3;
    a.L[0] = {w1p, b1, g1, be1, C1, 2 * ((K0 + 31) / 32), (K0 + 1 + 15) / 16};      // inputs: K0 + 1 = [feat (C, padded to 4) | dx dy dz 0]
    a.L[1] = {w2p, b2, g2, be2, C2, 2 * ((C1 + 31) / 32), 2 * ((C1 + 31) / 32)};
    a.L[2] = {w3p, b3, g3, be3, C3, 2 * ((C2 + 31) / 32), 2 * ((C2 + 31) / 32)};
    a.out = out; a.ldo = ldo; a.out_off = out_off;
    a.pre = pre; a.ldp = ldp; a.wx = wx;
    a.trace = nullptr;
    CASPR_IF_DEBUG(a.trace = g_sa_trace;)
    const int rA = a.L[0].kcr * 4 > a.L[2].kc * 4 ? a.L[0].kcr * 4 : a.L[2].kc * 4;
    const int rB0 = a.L[1].kc * 4 > C1 / 4 ? a.L[1].kc * 4 : C1 / 4;
    a.rowsA = rA > C2 / 4 ? rA : C2 / 4;
    // the last layer's output is the widest thing tile B ever holds: in two halves of 8 GroupNorm groups each (C3 a multiple of 32,
    // >= 4 row tiles per half so that every wave has one) the tile needs half the rows -- 70 -> 52 KB at the fourth level: three
    // workgroups per CU instead of two
    // (only where it buys a workgroup: the third level -- 37 KB unsplit, four per CU -- lost 0.04 ms to the extra barriers and idle
    // statistics lanes; the fourth went 0.536 -> 0.510 and 1.035 -> 0.995 ms, A/B in one box)
    const size_t unsplit = (size_t)(a.rowsA + (rB0 > C3 / 4 ? rB0 : C3 / 4)) * 32 * 16;
    a.split_last = (C3 / 4 > rB0 && C3 % 128 == 0 && unsplit > 56 * 1024) ? 1 : 0;
    const int rB1 = a.split_last ? C3 / 8 : C3 / 4;
    a.rowsB = rB0 > rB1 ? rB0 : rB1;
    hipStream_t st = (hipStream_t)stream;
    const bool no_small = CASPR_DEBUG_ENV_INT("CASPR_SA_NO_SMALL") != 0;   // debug build only: force the LDS kernel
    a.repair_kmax = 0;
    if (!no_small && ldo % 4 == 0 && out_off % 4 == 0 && ((uintptr_t)out % 16) == 0) {
        {   // which balls the f64 re-evaluation behind the register kernel will take (the register kernel skips waves made of those only)
            const int rk_ = CASPR_DEBUG_ENV_INT("CASPR_SA_REPAIR_K");
            const int wf_ = ((C1 >> 4) * a.L[0].kc + (C2 >> 4) * a.L[1].kc + (C3 >> 4) * a.L[2].kc) * 256;
            const bool small_shape = (C1 == 16 || C1 == 32) && C1 == C2 && C3 == 2 * C1;
            const int wide16 = CASPR_DEBUG_ENV_INT("CASPR_SA_WIDE_NS16_ONLY");      // experiment: 5..8 at the 16-sample scale only
            if (small_shape && rk_ >= 0 && wf_ <= 7168) a.repair_kmax = ((rk_ == 0 || rk_ >= 8) && a.feat_kind != 0 && wf_ <= 4096 && !(wide16 && ns != 16)) ? 8 : 4;
        }
        const int cpb = 4 * (64 / ns);   // centres per 256-thread block (4 waves x 64 columns)
        dim3 grid(ceil_div(M, cpb), B);
        const bool small_shape_ = (C1 == 16 || C1 == 32) && C1 == C2 && C3 == 2 * C1;
        a.order = a.count = nullptr;
        if (workspace && a.repair_kmax > 0 && small_shape_ && !CASPR_DEBUG_ENV_INT("CASPR_SA_NO_LIST")) {
            // the neighbourhoods the f64 kernel will not overwrite, listed per cloud: the register kernel computes those only (sa_list_kernel)
            if (!only_f64) sa_list_kernel<<<dim3(B), dim3(256), 0, st>>>(idx, M, ns, a.repair_kmax, workspace + B, workspace);
            a.order = workspace + B;
            a.count = workspace;
        }
        // The two halves of the call separately (CASPR_SA_ONLY_MFMA / _ONLY_F64: the caller runs them on two streams).  They write disjoint
        // output rows only when the register kernel works from the list -- without it that kernel stores every row and the f64 kernel must
        // come behind it.
        CASPR_REQUIRE(!(only_mfma || only_f64) || a.repair_kmax == 0 || a.order != nullptr,
                      "sa_mlp_max: CASPR_SA_ONLY_MFMA / _ONLY_F64 need the workspace (the list that makes the two halves' output rows disjoint)");
        bool done = true;
        if (only_f64) done = small_shape_;
        else if (C1 == 16 && C2 == 16 && C3 == 32 && ns == 16) sa_small_kernel_w3<16, 16, 16, 32><<<grid, dim3(256), 0, st>>>(a);
        else if (C1 == 16 && C2 == 16 && C3 == 32 && ns == 32) sa_small_kernel_w3<32, 16, 16, 32><<<grid, dim3(256), 0, st>>>(a);
        else if (C1 == 32 && C2 == 32 && C3 == 64 && ns == 16) sa_small_kernel<16, 32, 32, 64><<<grid, dim3(256), 0, st>>>(a);
        else if (C1 == 32 && C2 == 32 && C3 == 64 && ns == 32) sa_small_kernel_w2<32, 32, 32, 64><<<grid, dim3(256), 0, st>>>(a);
        else done = false;
        if (done) {
            CASPR_CHECK_LAUNCH("sa_mlp_max(small)");
            // neighbourhoods of 2..8 distinct samples once more, in f64 (see sa_repair_f64_kernel); debug build: CASPR_SA_REPAIR_K=-1 skips it, 4 stops at 4
            // (the window of sa_repair_f64_kernel holds first / second layers of <= 32 channels: every shape of this branch)
            // 5..8 at the FIRST level only (feat_kind set): its outputs are what the second level's small balls amplify; the second
            // level's own outputs go to the wide levels, whose groups of >= 4 channels do not (tools/sa_repair_sweep.py)
            const int wfloats = ((C1 >> 4) * a.L[0].kc + (C2 >> 4) * a.L[1].kc + (C3 >> 4) * a.L[2].kc) * 256;      // the kernel keeps the three packs in LDS
            CASPR_REQUIRE(a.repair_kmax > 0 || CASPR_DEBUG_ENV_INT("CASPR_SA_REPAIR_K") < 0, "sa_mlp_max: the f64 re-evaluation keeps %d weight floats in LDS (> its window)", wfloats);
            const dim3 rgrid(ceil_div(M, 128), B);
            if (a.repair_kmax > 0 && !only_mfma) {
                const bool wide = a.repair_kmax > 4;
#define SA_REPAIR(W)                                                                                          \
    do {                                                                                                      \
        sa_repair_f64_kernel<1, 1, W><<<rgrid, dim3(256), 0, st>>>(a, ns); /* one point: the chain of its features alone, f64 from exact / hi + lo inputs */ \
        sa_repair_f64_kernel<2, 4, W><<<rgrid, dim3(256), 0, st>>>(a, ns);                                    \
        if (wide) sa_repair_f64_kernel<5, 8, (W > 4096 ? 4096 : W)><<<rgrid, dim3(256), 0, st>>>(a, ns);      \
    } while (0)
                if (wfloats <= 2048) SA_REPAIR(2048);
                else if (wfloats <= 4096) SA_REPAIR(4096);
                else SA_REPAIR(7168);
#undef SA_REPAIR
                CASPR_CHECK_LAUNCH("sa_mlp_max(repair)");
            }
            return CASPR_OK;
        }
    }
    CASPR_REQUIRE(!a.lo_in && !a.lo_out, "sa_mlp_max: low parts (CASPR_FEAT_LO_IN / _OUT) exist for the register kernel's shapes only (widths <= 64)");
    if (only_f64) return CASPR_OK;        // the LDS kernel's shapes have no f64 half
    // 32 columns per workgroup from 128 input channels up (round 5, second part: the third level too -- 37 KB of LDS, four workgroups per
    // CU instead of two of 75 KB: 0.505 -> 0.474 and 1.219 -> 1.187 ms for its two scales, A/B in one box)
    const int bigK = (K0 > 128) || (C3 > 128) || pre != nullptr;
    const int ncol = bigK ? 32 : 64;
    const size_t shmem = (size_t)(a.rowsA + a.rowsB) * ncol * 16 + (3 * 64 + 16 + 64) * 4 + 64 + 64 * 4 * 4;     // + s_d
    CASPR_REQUIRE(shmem <= 160 * 1024, "sa_mlp_max: needs %zu bytes of LDS (> 160 KiB)", shmem);
    int rc;
    if (ns == 16 && ncol == 64) rc = launch_sa<16, 64>(a, B, shmem, st);
    else if (ns == 32 && ncol == 64) rc = launch_sa<32, 64>(a, B, shmem, st);
    else if (ns == 16 && ncol == 32) rc = launch_sa<16, 32>(a, B, shmem, st);
    else rc = launch_sa<32, 32>(a, B, shmem, st);
    if (rc != CASPR_OK) return rc;
    CASPR_CHECK_LAUNCH("sa_mlp_max");
    return CASPR_OK;
}

extern "C" int caspr_sa_mlp_max_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                                    const int32_t *idx, int B, int n, int M, int C, int ns, int feat_kind, const float *w1p,
                                    const float *b1, const float *g1, const float *be1, int C1, const float *w2p,
                                    const float *b2, const float *g2, const float *be2, int C2, const float *w3p,
                                    const float *b3, const float *g3, const float *be3, int C3, float *out, int ldo,
                                    int out_off, void *stream)
{
    return sa_mlp_max_impl(xyz, new_xyz, feat, ldf, idx, B, n, M, C, ns, feat_kind, w1p, b1, g1, be1, C1, w2p, b2, g2, be2, C2, w3p, b3, g3,
                           be3, C3, out, ldo, out_off, nullptr, stream);
}

extern "C" long caspr_sa_mlp_max_workspace_ints(int B, int M) { return (long)B * ((long)M + 1); }

extern "C" int caspr_sa_mlp_max_ws_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                                       const int32_t *idx, int B, int n, int M, int C, int ns, int feat_kind, const float *w1p,
                                       const float *b1, const float *g1, const float *be1, int C1, const float *w2p,
                                       const float *b2, const float *g2, const float *be2, int C2, const float *w3p,
                                       const float *b3, const float *g3, const float *be3, int C3, float *out, int ldo,
                                       int out_off, int32_t *workspace, void *stream)
{
    return sa_mlp_max_impl(xyz, new_xyz, feat, ldf, idx, B, n, M, C, ns, feat_kind, w1p, b1, g1, be1, C1, w2p, b2, g2, be2, C2, w3p, b3, g3,
                           be3, C3, out, ldo, out_off, workspace, stream);
}

// The LDS kernel's shapes with the first layer PRE-AGGREGATED (see sa_mlp_kernel): pre (B, n, ldp) = W_f . feat over the level's source points
// (a plain conv, no bias: caspr_conv1x1_*), wx1 (C1, 3) row-major = the layer's coordinate columns, b1 / g1 / be1 its bias and GroupNorm.
extern "C" int caspr_sa_mlp_max_pre_f32(const float *xyz, const float *new_xyz, const float *pre, int ldp, const int32_t *idx, int B, int n,
                                        int M, int ns, const float *wx1, const float *b1, const float *g1, const float *be1, int C1,
                                        const float *w2p, const float *b2, const float *g2, const float *be2, int C2, const float *w3p,
                                        const float *b3, const float *g3, const float *be3, int C3, float *out, int ldo, int out_off,
                                        void *stream)
{
    CASPR_REQUIRE(pre && wx1, "sa_mlp_max_pre: null pointer");
    return sa_mlp_max_impl(xyz, new_xyz, nullptr, 0, idx, B, n, M, 0, ns, 0, nullptr, b1, g1, be1, C1, w2p, b2, g2, be2, C2, w3p, b3, g3, be3, C3,
                           out, ldo, out_off, nullptr, stream, pre, ldp, wx1);
}
