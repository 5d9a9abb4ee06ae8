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
#include "common.h"

__global__ __launch_bounds__(256) void cnf_act_fwd_kernel(const float *__restrict__ Z, int ldz, const float *__restrict__ b,
                                                          const float *__restrict__ gate, const float *__restrict__ beta,
                                                          long R, int n, int C, float *__restrict__ H, int ldh)
{
    const int C4 = C >> 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * C4) return;
    const long r = t / C4;
    const int c = (int)(t % C4) * 4;
    const long f = r / n;
    const f32x4 zv = ld4(Z + r * ldz + c), zt = ld4(Z + (R + r) * ldz + c);
    const f32x4 bb = ld4(b + c), g = ld4(gate + f * C + c), be = ld4(beta + f * C + c);
    f32x4 hv, ht;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = (zv[q] + bb[q]) * g[q] + be[q];
        hv[q] = softplus_f(a);
        ht[q] = sigmoid_f(a) * (zt[q] * g[q]);
    }
    st4(H + r * ldh + c, hv);
    st4(H + (R + r) * ldh + c, ht);
}

__global__ __launch_bounds__(256) void cnf_act_bwd_kernel(const float *__restrict__ Z, int ldz, const float *__restrict__ b,
                                                          const float *__restrict__ gate, const float *__restrict__ beta,
                                                          const float *__restrict__ dH, int ldd, long R, int n, int C,
                                                          float *__restrict__ dZ, int lddz, float *__restrict__ dgate,
                                                          float *__restrict__ dbeta)
{
    __shared__ float s_g[4][64], s_b[4][64];
    const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long f = blockIdx.y;
    float acc_g = 0.f, acc_b = 0.f;
    if (c < C) {
        const float bb = b[c], g = gate[f * C + c], be = beta[f * C + c];
        for (int p = sub; p < n; p += 4) {
            const long r = f * n + p;
            const float zv = Z[r * ldz + c], zt = Z[(R + r) * ldz + c];
            const float dhv = dH[r * ldd + c], dht = dH[(R + r) * ldd + c];
            const float a = (zv + bb) * g + be, ad = zt * g;
            const float s = sigmoid_f(a);
            const float da = dhv * s + dht * (s * (1.0f - s)) * ad;
            const float dad = dht * s;
            dZ[r * lddz + c] = da * g;
            dZ[(R + r) * lddz + c] = dad * g;
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
                                 int n, int C, float *H, int ldh, void *stream)
{
    CASPR_REQUIRE(Z && b && gate && beta && H && R > 0 && n > 0 && R % n == 0 && C > 0 && C % 4 == 0 && ldz % 4 == 0 && ldh % 4 == 0 &&
                      ldz >= C && ldh >= C,
                  "cnf_act: bad arguments (C=%d must be a multiple of 4, R a multiple of n)", C);
    const long total = R * (C / 4);
    cnf_act_fwd_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(Z, ldz, b, gate, beta, R, n, C, H, ldh);
    CASPR_CHECK_LAUNCH("cnf_act");
    return CASPR_OK;
}

extern "C" int caspr_cnf_act_bwd_f32(const float *Z, int ldz, const float *b, const float *gate, const float *beta,
                                     const float *dH, int ldd, long R, int n, int C, float *dZ, int lddz, float *dgate,
                                     float *dbeta, void *stream)
{
    CASPR_REQUIRE(Z && b && gate && beta && dH && dZ && dgate && dbeta && R > 0 && n > 0 && R % n == 0 && C > 0 && ldz >= C && ldd >= C &&
                      lddz >= C,
                  "cnf_act_bwd: bad arguments");
    const long frames = R / n;
    CASPR_REQUIRE(frames <= 65535, "cnf_act_bwd: %ld frames > 65535", frames);
    cnf_act_bwd_kernel<<<dim3(ceil_div(C, 64), (unsigned)frames), dim3(256), 0, (hipStream_t)stream>>>(Z, ldz, b, gate, beta, dH, ldd, R, n,
                                                                                                       C, dZ, lddz, dgate, dbeta);
    CASPR_CHECK_LAUNCH("cnf_act_bwd");
    return CASPR_OK;
}
