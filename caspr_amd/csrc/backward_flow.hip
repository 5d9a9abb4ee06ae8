// backward_flow.hip -- training tier, part 3: the gated softplus layer of the CNF's ODE function on value AND tangent
// rows (ConcatSquashLinear + Softplus, diffeq_layers.py:83-90 / odefunc.py:98-105; the Hutchinson divergence
// e^T (df/dy) e of odefunc.py:13-31 is carried as a forward-mode tangent), forward and backward, fused:
//
//   rows [0,R)  : value   zv = Z[r]     a  = (zv + b) * g[f] + beta[f]      H[r]   = softplus(a)
//   rows [R,2R) : tangent zt = Z[R+r]   ad = zt * g[f]                      H[R+r] = sigmoid(a) * ad       f = r / n (frame)
//
// backward, given dH:   da  = dHv*s + dHt*s*(1-s)*ad      dad = dHt*s            (s = sigmoid(a))
//                       dZv = da*g      dZt = dad*g
//                       dg[f]    = sum_{r in f} da*(zv+b) + dad*zt         dbeta[f] = sum_{r in f} da
// Only Z is kept between the two passes; a, ad, s are recomputed.  The per-frame sums are taken by the one workgroup
// that owns (frame, 64 channels): fixed order, no atomics.
//
// Row layout (round 3): `blk` = R is the layout above; blk = 32 interleaves, per 64 rows, 32 value rows and the tangent rows of
// the SAME 32 points -- value row of point p = (p / blk) 2 blk + p % blk, tangent row = value row + blk -- which is what lets
// the conv kernel apply this layer in its epilogue (conv1x1_bf16x6_kernel, act mode 2: a lane's column tiles 0, 1 are values,
// 2, 3 their tangents).  Every entry point takes blk; all tensors of one solve share it.
#include "common.h"

// value row of point p; the tangent row is + toff (= blk).  shift < 0: blk == R (rows [0,R) | [R,2R))
__device__ __forceinline__ long cnf_vrow(long p, int shift) { return shift < 0 ? p : (((p >> shift) << (shift + 1)) | (p & ((1L << shift) - 1))); }
static int cnf_blk_shift(long R, long blk)   // -1 for blk == R, log2 otherwise, -2 if unsupported
{
    if (blk == R) return -1;
    for (int sft = 0; sft < 30; ++sft)
        if ((1L << sft) == blk) return R % blk == 0 ? sft : -2;
    return -2;
}

__global__ __launch_bounds__(256) void cnf_act_fwd_kernel(const float *__restrict__ Z, int ldz, const float *__restrict__ b,
                                                          const float *__restrict__ gate, const float *__restrict__ beta,
                                                          long R, int n, int C, float *__restrict__ H, int ldh, int shift, long toff)
{
    const int C4 = C >> 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * C4) return;
    const long pt = t / C4, r = cnf_vrow(pt, shift);
    const int c = (int)(t % C4) * 4;
    const long f = pt / n;
    const f32x4 zv = ld4(Z + r * ldz + c), zt = ld4(Z + (toff + r) * ldz + c);
    const f32x4 bb = ld4(b + c), g = ld4(gate + f * C + c), be = ld4(beta + f * C + c);
    f32x4 hv, ht;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = (zv[q] + bb[q]) * g[q] + be[q];
        hv[q] = softplus_f(a);
        ht[q] = sigmoid_f(a) * (zt[q] * g[q]);
    }
    st4(H + r * ldh + c, hv);
    st4(H + (toff + r) * ldh + c, ht);
}

// dH == NULL: the layer feeds the 3-channel output layer directly (odefunc.py:103, no activation behind it) and its dH is that
// layer's data gradient dH[r][c] = sum_j dZo[r][j] Wo[j][c] -- three FMAs per element, formed here instead of being written by a
// K = 3 conv (335 MB per evaluation at cfg-3) and read back.
__global__ __launch_bounds__(256) void cnf_act_bwd_kernel(const float *__restrict__ Z, int ldz, const float *__restrict__ b,
                                                          const float *__restrict__ gate, const float *__restrict__ beta,
                                                          const float *__restrict__ dH, int ldd, long R, int n, int C,
                                                          float *__restrict__ dZ, int lddz, float *__restrict__ dgate,
                                                          float *__restrict__ dbeta, int shift, long toff,
                                                          const float *__restrict__ dZo, int ldo, const float *__restrict__ Wo, int ldw)
{
    __shared__ float s_g[4][64], s_b[4][64];
    const int cl = threadIdx.x & 63;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + cl;
    const long f = blockIdx.y;
    float acc_g = 0.f, acc_b = 0.f;
    if (c < C) {
        const float bb = b[c], g = gate[f * C + c], be = beta[f * C + c];
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        if (!dH) { w0 = Wo[c]; w1 = Wo[ldw + c]; w2 = Wo[2 * ldw + c]; }
        for (int p = sub; p < n; p += 4) {
            const long r = cnf_vrow(f * n + p, shift);
            const float zv = Z[r * ldz + c], zt = Z[(toff + r) * ldz + c];
            float dhv, dht;
            if (dH) {
                dhv = dH[r * ldd + c];
                dht = dH[(toff + r) * ldd + c];
            } else {
                const float *ov = dZo + r * ldo, *ot = dZo + (toff + r) * ldo;      // wave-uniform rows
                dhv = (ov[0] * w0 + ov[1] * w1) + ov[2] * w2;
                dht = (ot[0] * w0 + ot[1] * w1) + ot[2] * w2;
            }
            const float a = (zv + bb) * g + be, ad = zt * g;
            const float s = sigmoid_fast(a);
            const float da = dhv * s + dht * (s * (1.0f - s)) * ad;
            const float dad = dht * s;
            dZ[r * lddz + c] = da * g;
            dZ[(toff + r) * lddz + c] = dad * g;
            acc_g += da * (zv + bb) + dad * zt;
            acc_b += da;
        }
    }
    s_g[sub][cl] = acc_g;
    s_b[sub][cl] = acc_b;
    __syncthreads();
    if (sub == 0 && c < C) {
        dgate[f * C + c] = (s_g[0][cl] + s_g[1][cl]) + (s_g[2][cl] + s_g[3][cl]);
        dbeta[f * C + c] = (s_b[0][cl] + s_b[1][cl]) + (s_b[2][cl] + s_b[3][cl]);
    }
}

extern "C" int caspr_cnf_act_f32(const float *Z, int ldz, const float *b, const float *gate, const float *beta, long R,
                                 int n, int C, long blk, float *H, int ldh, void *stream)
{
    const int shift = cnf_blk_shift(R, blk);
    CASPR_REQUIRE(shift >= -1, "cnf_act: blk=%ld must be R or a power of two that divides R=%ld", blk, R);
    CASPR_REQUIRE(Z && b && gate && beta && H && R > 0 && n > 0 && R % n == 0 && C > 0 && C % 4 == 0 && ldz % 4 == 0 && ldh % 4 == 0 &&
                      ldz >= C && ldh >= C,
                  "cnf_act: bad arguments (C=%d must be a multiple of 4, R a multiple of n)", C);
    const long total = R * (C / 4);
    cnf_act_fwd_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(Z, ldz, b, gate, beta, R, n, C, H, ldh, shift, blk);
    CASPR_CHECK_LAUNCH("cnf_act");
    return CASPR_OK;
}

static int cnf_act_bwd_launch(const float *Z, int ldz, const float *b, const float *gate, const float *beta, const float *dH, int ldd,
                              const float *dZo, int ldo, const float *Wo, int ldw, long R, int n, int C, long blk, float *dZ, int lddz,
                              float *dgate, float *dbeta, void *stream)
{
    CASPR_REQUIRE(Z && b && gate && beta && (dH || (dZo && Wo)) && dZ && dgate && dbeta && R > 0 && n > 0 && R % n == 0 && C > 0 && ldz >= C &&
                      (!dH || ldd >= C) && lddz >= C,
                  "cnf_act_bwd: bad arguments");
    const int shift = cnf_blk_shift(R, blk);
    CASPR_REQUIRE(shift >= -1, "cnf_act_bwd: blk=%ld must be R or a power of two that divides R=%ld", blk, R);
    const long frames = R / n;
    CASPR_REQUIRE(frames <= 65535, "cnf_act_bwd: %ld frames > 65535", frames);
    cnf_act_bwd_kernel<<<dim3(ceil_div(C, 64), (unsigned)frames), dim3(256), 0, (hipStream_t)stream>>>(Z, ldz, b, gate, beta, dH, ldd, R, n, C, dZ,
                                                                                                       lddz, dgate, dbeta, shift, blk, dZo, ldo, Wo, ldw);
    CASPR_CHECK_LAUNCH("cnf_act_bwd");
    return CASPR_OK;
}

extern "C" int caspr_cnf_act_bwd_f32(const float *Z, int ldz, const float *b, const float *gate, const float *beta,
                                     const float *dH, int ldd, long R, int n, int C, long blk, float *dZ, int lddz, float *dgate,
                                     float *dbeta, void *stream)
{
    CASPR_REQUIRE(dH, "cnf_act_bwd: dH is NULL");
    return cnf_act_bwd_launch(Z, ldz, b, gate, beta, dH, ldd, nullptr, 0, nullptr, 0, R, n, C, blk, dZ, lddz, dgate, dbeta, stream);
}

// the same for the layer in front of the 3-channel output layer: dH = dZo Wo formed on the fly (dZo (2R, ldo >= 3), Wo (3, ldw >= C))
extern "C" int caspr_cnf_act_bwd_out_f32(const float *Z, int ldz, const float *b, const float *gate, const float *beta,
                                         const float *dZo, int ldo, const float *Wo, int ldw, long R, int n, int C, long blk, float *dZ,
                                         int lddz, float *dgate, float *dbeta, void *stream)
{
    CASPR_REQUIRE(dZo && Wo && ldo >= 3 && ldw >= C, "cnf_act_bwd_out: bad arguments");
    return cnf_act_bwd_launch(Z, ldz, b, gate, beta, nullptr, 0, dZo, ldo, Wo, ldw, R, n, C, blk, dZ, lddz, dgate, dbeta, stream);
}

// ---------------------------------------------------------------------------------------------
// First layer of the ODE function (3 -> C, diffeq_layers.py:83-90) fused with its gate + softplus, on value and
// tangent rows:  zv = W0 y + b0,  zt = W0 e;  H as in cnf_act.  K = 3, so the matrix product is three FMAs per output
// and the layer is a pure streaming write of H (2R x C): no GEMM pass, no stored pre-activation.
// Backward recomputes zv, zt from (y, e) and emits, per (frame, 64-channel chunk) workgroup in a fixed order:
//   dgate, dbeta (frame, C);  dW0 partials (frame, C, 3) = sum_p dzv*y + dzt*e;  dy partials (chunk, R, 3) = sum_c dzv*W0
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
static __device__ __forceinline__ float wave_sum_f(float v)   // uniform result: DPP row sums, then the four rows
{
    v = row_allreduce_add<16>(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}

__global__ __launch_bounds__(256) void cnf_in_fwd_kernel(const float *__restrict__ Yp, const float *__restrict__ E,
                                                         const float *__restrict__ W0, const float *__restrict__ b,
                                                         const float *__restrict__ gate, const float *__restrict__ beta,
                                                         long R, int n, int C, float *__restrict__ H, int shift, long toff)
{
    const int C4 = C >> 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * C4) return;
    const long pt = t / C4, r = cnf_vrow(pt, shift);
    const int c = (int)(t % C4) * 4;
    const long f = pt / n;
    const float y0 = Yp[pt * 3], y1 = Yp[pt * 3 + 1], y2 = Yp[pt * 3 + 2];
    const float e0 = E[pt * 3], e1 = E[pt * 3 + 1], e2 = E[pt * 3 + 2];
    const f32x4 bb = ld4(b + c), g = ld4(gate + f * C + c), be = ld4(beta + f * C + c);
    f32x4 hv, ht;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float *w = W0 + (long)(c + q) * 3;
        const float zv = (w[0] * y0 + w[1] * y1) + w[2] * y2;
        const float zt = (w[0] * e0 + w[1] * e1) + w[2] * e2;
        const float a = (zv + bb[q]) * g[q] + be[q];
        hv[q] = softplus_f(a);
        ht[q] = sigmoid_f(a) * (zt * g[q]);
    }
    st4(H + r * C + c, hv);
    st4(H + (toff + r) * C + c, ht);
}

// The same for C a multiple of 256 (the model's 512): a workgroup owns 32 points of one frame, a thread four channels -- its
// weights, bias, gate and beta in registers for all of them -- and a wave's 64 threads share the point, so y / e are SCALAR
// loads.  The element-per-thread kernel above issues 21 load instructions per two 16-byte stores and ran at 1.7 TB/s of its
// output (193 us per evaluation at cfg-3); this one is bound by the 2R x C write.
__global__ __launch_bounds__(256) void cnf_in_fwd_rows_kernel(const float *__restrict__ Yp, const float *__restrict__ E,
                                                              const float *__restrict__ W0, const float *__restrict__ b,
                                                              const float *__restrict__ gate, const float *__restrict__ beta,
                                                              long R, int n, int C, float *__restrict__ H, int shift, long toff)
{
    const int C4 = C >> 2, RL = 256 / C4;                 // C4 = 64, 128 or 256: whole waves per point
    const int tq = threadIdx.x % C4;
    const int rl = __builtin_amdgcn_readfirstlane(threadIdx.x / C4);
    const long f = blockIdx.y;
    const int c = tq * 4;
    float w[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) w[q][d] = W0[(long)(c + q) * 3 + d];
    const f32x4 bb = ld4(b + c), g = ld4(gate + f * C + c), be = ld4(beta + f * C + c);
    const int p_end = (blockIdx.x * 32 + 32) < n ? (blockIdx.x * 32 + 32) : n;
    for (int pi = blockIdx.x * 32 + rl; pi < p_end; pi += RL) {
        const long pt = f * n + pi, r = cnf_vrow(pt, shift);
        const float *yp = Yp + pt * 3, *ep = E + pt * 3;                 // wave-uniform addresses
        const float y0 = yp[0], y1 = yp[1], y2 = yp[2], e0 = ep[0], e1 = ep[1], e2 = ep[2];
        f32x4 hv, ht;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float zv = (w[q][0] * y0 + w[q][1] * y1) + w[q][2] * y2;
            const float zt = (w[q][0] * e0 + w[q][1] * e1) + w[q][2] * e2;
            const float a = (zv + bb[q]) * g[q] + be[q];
            // softplus and sigmoid from one 2^(-|a| log2 e) on the hardware transcendentals: with the libm forms (~90 instructions per
            // element) this kernel was bound by its arithmetic, not by the 2R x C write
            const float u = __builtin_amdgcn_exp2f(fabsf(a) * -1.44269504088896341f);
            const float rc = __builtin_amdgcn_rcpf(1.0f + u);
            hv[q] = fmaxf(a, 0.0f) + 0.69314718055994531f * __builtin_amdgcn_logf(1.0f + u);
            ht[q] = (a >= 0.0f ? rc : u * rc) * (zt * g[q]);
        }
        st4(H + r * C + c, hv);
        st4(H + (toff + r) * C + c, ht);
    }
}

__global__ __launch_bounds__(256) void cnf_in_bwd_kernel(const float *__restrict__ Yp, const float *__restrict__ E,
                                                         const float *__restrict__ W0, const float *__restrict__ b,
                                                         const float *__restrict__ gate, const float *__restrict__ beta,
                                                         const float *__restrict__ dH, long R, int n, int C,
                                                         float *__restrict__ dgate, float *__restrict__ dbeta,
                                                         float *__restrict__ dW0p, float *__restrict__ dYp, int shift, long toff)
{
    __shared__ float s_red[4][64][8];
    const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int chunk = blockIdx.x, c = chunk * 64 + cl;
    const long f = blockIdx.y;
    const bool ok = c < C;
    float w0 = 0.f, w1 = 0.f, w2 = 0.f, bb = 0.f, g = 0.f, be = 0.f;
    if (ok) { w0 = W0[(long)c * 3]; w1 = W0[(long)c * 3 + 1]; w2 = W0[(long)c * 3 + 2]; bb = b[c]; g = gate[f * C + c]; be = beta[f * C + c]; }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dgate, dbeta, dW0[0..2] (value part), dW0[0..2] (tangent part)
    for (int p = sub; p < n; p += 4) {
        const long r = f * n + p, rv = cnf_vrow(r, shift);
        const float y0 = Yp[r * 3], y1 = Yp[r * 3 + 1], y2 = Yp[r * 3 + 2];
        const float e0 = E[r * 3], e1 = E[r * 3 + 1], e2 = E[r * 3 + 2];
        float dzv = 0.f;
        if (ok) {
            const float zv = (w0 * y0 + w1 * y1) + w2 * y2, zt = (w0 * e0 + w1 * e1) + w2 * e2;
            const float a = (zv + bb) * g + be, ad = zt * g;
            const float s = sigmoid_f(a);
            const float dhv = dH[rv * C + c], dht = dH[(toff + rv) * C + c];
            const float da = dhv * s + dht * (s * (1.0f - s)) * ad;
            const float dad = dht * s;
            dzv = da * g;
            const float dzt = dad * g;
            acc[0] += da * (zv + bb) + dad * zt;
            acc[1] += da;
            acc[2] += dzv * y0; acc[3] += dzv * y1; acc[4] += dzv * y2;
            acc[5] += dzt * e0; acc[6] += dzt * e1; acc[7] += dzt * e2;
        }
        // dy[p][j] = sum over this chunk's 64 channels of dzv * W0[c][j]: wave reduction (the tangent rows carry e, a constant)
        // (DPP row sums + one readlane per 16-lane row: __shfl_xor would be 18 ds_bpermute round trips per point)
        const float d0 = wave_sum_f(dzv * w0), d1 = wave_sum_f(dzv * w1), d2 = wave_sum_f(dzv * w2);
        if (cl == 0) {
            float *o = dYp + ((long)chunk * R + r) * 3;
            o[0] = d0; o[1] = d1; o[2] = d2;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_red[sub][cl][k] = acc[k];
    __syncthreads();
    if (sub == 0 && ok) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (s_red[0][cl][k] + s_red[1][cl][k]) + (s_red[2][cl][k] + s_red[3][cl][k]);
        dgate[f * C + c] = t[0];
        dbeta[f * C + c] = t[1];
        float *o = dW0p + (f * C + c) * 3;
        o[0] = t[2] + t[5]; o[1] = t[3] + t[6]; o[2] = t[4] + t[7];
    }
}

// Backward for C a multiple of 256: a workgroup = (256-channel chunk, frame, point split); a thread owns four channels (16-byte
// reads of dH: the element-per-thread kernel above reads 256 bytes per wave and row), a wave one point at a time (y / e scalar),
// and the channel sums behind dy are three wave reductions per FOUR channels' worth of work.  Partials per point split, summed by
// the caller: dgate / dbeta (frames, nsplit, C), dW0 (frames * nsplit, C, 3), dy (C / 256, R, 3).
__global__ __launch_bounds__(256) void cnf_in_bwd_rows_kernel(const float *__restrict__ Yp, const float *__restrict__ E,
                                                              const float *__restrict__ W0, const float *__restrict__ b,
                                                              const float *__restrict__ gate, const float *__restrict__ beta,
                                                              const float *__restrict__ dH, long R, int n, int C, int nsplit,
                                                              float *__restrict__ dgate, float *__restrict__ dbeta,
                                                              float *__restrict__ dW0p, float *__restrict__ dYp, int shift, long toff)
{
    __shared__ float s_red[3][64][20];
    const int lane = threadIdx.x & 63;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x, c = chunk * 256 + lane * 4;
    const long f = blockIdx.y;
    const int ps = blockIdx.z, per = (n + nsplit - 1) / nsplit;
    const int p_beg = ps * per, p_end = (p_beg + per) < n ? (p_beg + per) : n;
    float w[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) w[q][d] = W0[(long)(c + q) * 3 + d];
    const f32x4 bb = ld4(b + c), g = ld4(gate + f * C + c), be = ld4(beta + f * C + c);
    float acc[20];                 // [q]: dgate, [4 + q]: dbeta, [8 + 3 q + d]: dW0
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = 0.f;
    for (int pi = p_beg + sub; pi < p_end; pi += 4) {
        const long pt = f * n + pi, rv = cnf_vrow(pt, shift);
        const float *yp = Yp + pt * 3, *ep = E + pt * 3;                 // wave-uniform
        const float y0 = yp[0], y1 = yp[1], y2 = yp[2], e0 = ep[0], e1 = ep[1], e2 = ep[2];
        const f32x4 dhv = ld4(dH + rv * C + c), dht = ld4(dH + (toff + rv) * C + c);
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float zv = (w[q][0] * y0 + w[q][1] * y1) + w[q][2] * y2, zt = (w[q][0] * e0 + w[q][1] * e1) + w[q][2] * e2;
            const float a = (zv + bb[q]) * g[q] + be[q], ad = zt * g[q];
            const float sg = sigmoid_fast(a);
            const float da = dhv[q] * sg + dht[q] * (sg * (1.0f - sg)) * ad;
            const float dad = dht[q] * sg;
            const float dzv = da * g[q], dzt = dad * g[q];
            acc[q] += da * (zv + bb[q]) + dad * zt;
            acc[4 + q] += da;
            acc[8 + 3 * q] += dzv * y0 + dzt * e0;
            acc[9 + 3 * q] += dzv * y1 + dzt * e1;
            acc[10 + 3 * q] += dzv * y2 + dzt * e2;
            d0 += dzv * w[q][0];
            d1 += dzv * w[q][1];
            d2 += dzv * w[q][2];
        }
        d0 = wave_sum_f(d0); d1 = wave_sum_f(d1); d2 = wave_sum_f(d2);
        if (lane == 0) {
            float *o = dYp + ((long)chunk * R + pt) * 3;
            o[0] = d0; o[1] = d1; o[2] = d2;
        }
    }
    if (sub > 0) {
#pragma unroll
        for (int k = 0; k < 20; ++k) s_red[sub - 1][lane][k] = acc[k];
    }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = (acc[k] + s_red[0][lane][k]) + (s_red[1][lane][k] + s_red[2][lane][k]);
        const long fs = f * nsplit + ps;
        st4(dgate + fs * C + c, (f32x4){acc[0], acc[1], acc[2], acc[3]});
        st4(dbeta + fs * C + c, (f32x4){acc[4], acc[5], acc[6], acc[7]});
        float *o = dW0p + (fs * C + c) * 3;
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = acc[8 + k];
    }
}

// channels per dy partial / point splits the backward entry uses for this C (the caller sizes dY_part, dgate, dbeta, dW0_part with them)
extern "C" int caspr_cnf_in_bwd_chunk(int C) { return (C > 0 && C % 256 == 0) ? 256 : 64; }
extern "C" int caspr_cnf_in_bwd_splits(int C, int n) { return (C > 0 && C % 256 == 0 && n >= 256) ? 8 : 1; }

extern "C" int caspr_cnf_in_f32(const float *Y, const float *E, const float *W0, const float *b, const float *gate,
                                const float *beta, long R, int n, int C, long blk, float *H, void *stream)
{
    CASPR_REQUIRE(Y && E && W0 && b && gate && beta && H && R > 0 && n > 0 && R % n == 0 && C > 0 && C % 4 == 0, "cnf_in: bad arguments");
    const int shift = cnf_blk_shift(R, blk);
    CASPR_REQUIRE(shift >= -1, "cnf_in: blk=%ld must be R or a power of two that divides R=%ld", blk, R);
    const long frames = R / n;
    if (C % 256 == 0 && C <= 1024 && frames <= 65535)
        cnf_in_fwd_rows_kernel<<<dim3(ceil_div(n, 32), (unsigned)frames), dim3(256), 0, (hipStream_t)stream>>>(Y, E, W0, b, gate, beta, R, n, C, H,
                                                                                                             shift, blk);
    else {
        const long total = R * (C / 4);
        cnf_in_fwd_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(Y, E, W0, b, gate, beta, R, n, C, H, shift, blk);
    }
    CASPR_CHECK_LAUNCH("cnf_in");
    return CASPR_OK;
}

extern "C" int caspr_cnf_in_bwd_f32(const float *Y, const float *E, const float *W0, const float *b, const float *gate,
                                    const float *beta, const float *dH, long R, int n, int C, long blk, float *dgate, float *dbeta,
                                    float *dW0_part, float *dY_part, void *stream)
{
    CASPR_REQUIRE(Y && E && W0 && b && gate && beta && dH && dgate && dbeta && dW0_part && dY_part && R > 0 && n > 0 && R % n == 0 && C > 0,
                  "cnf_in_bwd: bad arguments");
    const int shift = cnf_blk_shift(R, blk);
    CASPR_REQUIRE(shift >= -1, "cnf_in_bwd: blk=%ld must be R or a power of two that divides R=%ld", blk, R);
    const long frames = R / n;
    CASPR_REQUIRE(frames <= 65535, "cnf_in_bwd: %ld frames > 65535", frames);
    if (caspr_cnf_in_bwd_chunk(C) == 256) {
        const int ns = caspr_cnf_in_bwd_splits(C, n);
        cnf_in_bwd_rows_kernel<<<dim3(C / 256, (unsigned)frames, ns), dim3(256), 0, (hipStream_t)stream>>>(Y, E, W0, b, gate, beta, dH, R, n, C, ns,
                                                                                                         dgate, dbeta, dW0_part, dY_part, shift, blk);
    } else
        cnf_in_bwd_kernel<<<dim3(ceil_div(C, 64), (unsigned)frames), dim3(256), 0, (hipStream_t)stream>>>(Y, E, W0, b, gate, beta, dH, R, n, C,
                                                                                                          dgate, dbeta, dW0_part, dY_part, shift, blk);
    CASPR_CHECK_LAUNCH("cnf_in_bwd");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Epilogue of the ODE function's 3-channel output layer on value / tangent rows (odefunc.py:103-105 + the Hutchinson contraction of
// odefunc.py:13-31), forward and backward in one launch each -- as torch element-wise ops these were ~25 launches per evaluation:
//   a[p][j]  = (Zo[v(p)][j] + b[j]) gate[f][j] + beta[f][j]            (dy/dt)
//   nd[p]    = - sum_j Zo[t(p)][j] gate[f][j] e[p][j]                    (- e^T (df/dy) e)
// backward: dZo (2R, 4; column 3 zero), dgate / dbeta (frames, 3) summed over each frame's points in a fixed order.
// gate / beta are rows of a wider (frames, ldg) tensor (the hyper networks' outputs of all layers side by side).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cnf_out_fwd_kernel(const float *__restrict__ Zo, int ldo, const float *__restrict__ b,
                                                          const float *__restrict__ gate, const float *__restrict__ beta, int ldg,
                                                          const float *__restrict__ E, long R, int n, int shift, long toff,
                                                          float *__restrict__ A, float *__restrict__ ND)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= R) return;
    const long f = p / n, rv = cnf_vrow(p, shift), rt = rv + toff;
    float nd = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g = gate[f * ldg + j];
        A[p * 3 + j] = (Zo[rv * ldo + j] + b[j]) * g + beta[f * ldg + j];
        nd += (Zo[rt * ldo + j] * g) * E[p * 3 + j];
    }
    ND[p] = -nd;
}

__global__ __launch_bounds__(256) void cnf_out_bwd_kernel(const float *__restrict__ dA, const float *__restrict__ dND,
                                                          const float *__restrict__ Zo, int ldo, const float *__restrict__ b,
                                                          const float *__restrict__ gate, int ldg, const float *__restrict__ E, long R, int n,
                                                          int shift, long toff, float *__restrict__ dZo, float *__restrict__ dgate,
                                                          float *__restrict__ dbeta)
{
    __shared__ float s_red[4][6];
    const long f = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[3], bb[3], acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) { g[j] = gate[f * ldg + j]; bb[j] = b[j]; }
    for (int pi = threadIdx.x; pi < n; pi += 256) {
        const long p = f * n + pi, rv = cnf_vrow(p, shift), rt = rv + toff;
        const float dnd = dND[p];
        f32x4 dv = (f32x4){0.f, 0.f, 0.f, 0.f}, dt = dv;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float da = dA[p * 3 + j], dad = -dnd * E[p * 3 + j];
            dv[j] = da * g[j];
            dt[j] = dad * g[j];
            acc[j] += da * (Zo[rv * ldo + j] + bb[j]) + dad * Zo[rt * ldo + j];
            acc[3 + j] += da;
        }
        st4(dZo + rv * 4, dv);
        st4(dZo + rt * 4, dt);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = wave_sum_f(acc[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k) s_red[wave][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const float t = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
        if (threadIdx.x < 3) dgate[f * 3 + threadIdx.x] = t;
        else dbeta[f * 3 + threadIdx.x - 3] = t;
    }
}

extern "C" int caspr_cnf_out_f32(const float *Zo, int ldo, const float *b, const float *gate, const float *beta, int ldg, const float *E, long R,
                                 int n, long blk, float *A, float *ND, void *stream)
{
    CASPR_REQUIRE(Zo && b && gate && beta && E && A && ND && R > 0 && n > 0 && R % n == 0 && ldo >= 3 && ldg >= 3, "cnf_out: bad arguments");
    const int shift = cnf_blk_shift(R, blk);
    CASPR_REQUIRE(shift >= -1, "cnf_out: blk=%ld must be R or a power of two that divides R=%ld", blk, R);
    cnf_out_fwd_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(Zo, ldo, b, gate, beta, ldg, E, R, n, shift, blk, A, ND);
    CASPR_CHECK_LAUNCH("cnf_out");
    return CASPR_OK;
}

extern "C" int caspr_cnf_out_bwd_f32(const float *dA, const float *dND, const float *Zo, int ldo, const float *b, const float *gate, int ldg,
                                     const float *E, long R, int n, long blk, float *dZo, float *dgate, float *dbeta, void *stream)
{
    CASPR_REQUIRE(dA && dND && Zo && b && gate && E && dZo && dgate && dbeta && R > 0 && n > 0 && R % n == 0 && ldo >= 3 && ldg >= 3 &&
                      ((uintptr_t)dZo % 16) == 0,
                  "cnf_out_bwd: bad arguments");
    const int shift = cnf_blk_shift(R, blk);
    CASPR_REQUIRE(shift >= -1, "cnf_out_bwd: blk=%ld must be R or a power of two that divides R=%ld", blk, R);
    const long frames = R / n;
    CASPR_REQUIRE(frames <= 2147483647L, "cnf_out_bwd: too many frames");
    cnf_out_bwd_kernel<<<dim3((unsigned)frames), dim3(256), 0, (hipStream_t)stream>>>(dA, dND, Zo, ldo, b, gate, ldg, E, R, n, shift, blk, dZo, dgate,
                                                                                     dbeta);
    CASPR_CHECK_LAUNCH("cnf_out_bwd");
    return CASPR_OK;
}
