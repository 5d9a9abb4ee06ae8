// emd.hip -- approximate earth mover's distance (utils/emd.py: emd_cuda.approxmatch_forward + matchcost_forward from
// PyTorchEMD, an un-vendored third-party extension -- its published algorithm, Fan et al. "A Point Set Generation
// Network", is restated here; call site utils/evaluations.py:45).
//
// Ten annealing levels j = 7 .. -2 with kernel exp(level * |p - q|^2), level = -4^j (0 on the last): each level
//   ratioL[k] = remainL[k] / (1e-9 + sum_l K(k,l) remainR[l])
//   sumr[l]   = remainR[l] * sum_k K(k,l) ratioL[k];  ratioR[l] = min(remainR[l]/(sumr+1e-9), 1) * remainR[l];
//   remainR[l] = max(0, remainR[l] - sumr)
//   match[k,l] += K(k,l) ratioL[k] ratioR[l];   remainL[k] = max(0, remainL[k] - sum_l (that increment))
// and cost = sum_{k,l} match[k,l] * |p_k - q_l|.  The match matrix (n*m floats per cloud pair, 2.7 GB at cfg-2) is
// never stored: the cost is linear in it, so every increment is priced as it is produced.
// One workgroup per cloud pair (the levels are a serial chain), the other cloud tiled through LDS.
#include "common.h"

#define EMD_TILE 1024

__global__ __launch_bounds__(256) void emd_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2, int n, int m,
                                                  float *__restrict__ cost, float *__restrict__ ws)
{
    __shared__ float buf[EMD_TILE * 4];
    __shared__ float red[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float *p1 = xyz1 + (long)i * n * 3, *p2 = xyz2 + (long)i * m * 3;
    float *remainL = ws + (long)i * 2 * (n + m), *remainR = remainL + n, *ratioL = remainR + m, *ratioR = ratioL + n;
    const float multiL = n >= m ? 1.0f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.0f;   // integer ratios, as published
    for (int k = tid; k < n; k += 256) remainL[k] = multiL;
    for (int l = tid; l < m; l += 256) remainR[l] = multiR;
    float my_cost = 0.f;
    __syncthreads();
    for (int j = 7; j >= -2; --j) {
        const float level = j == -2 ? 0.0f : -powf(4.0f, (float)j);
        // pass 1: ratioL
        for (int k0 = 0; k0 < n; k0 += 256) {
            const int k = k0 + tid;
            float x1 = 0, y1 = 0, z1 = 0;
            if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
            float suml = 1e-9f;
            for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
                const int lend = (m - l0) < EMD_TILE ? (m - l0) : EMD_TILE;
                for (int l = tid; l < lend; l += 256) {
                    buf[l * 4 + 0] = p2[(l0 + l) * 3]; buf[l * 4 + 1] = p2[(l0 + l) * 3 + 1]; buf[l * 4 + 2] = p2[(l0 + l) * 3 + 2];
                    buf[l * 4 + 3] = remainR[l0 + l];
                }
                __syncthreads();
                for (int l = 0; l < lend; ++l) {
                    const float dx = buf[l * 4] - x1, dy = buf[l * 4 + 1] - y1, dz = buf[l * 4 + 2] - z1;
                    suml += expf(level * (dx * dx + dy * dy + dz * dz)) * buf[l * 4 + 3];
                }
                __syncthreads();
            }
            if (k < n) ratioL[k] = remainL[k] / suml;
        }
        __syncthreads();
        // pass 2: ratioR, remainR
        for (int l0 = 0; l0 < m; l0 += 256) {
            const int l = l0 + tid;
            float x2 = 0, y2 = 0, z2 = 0;
            if (l < m) { x2 = p2[l * 3]; y2 = p2[l * 3 + 1]; z2 = p2[l * 3 + 2]; }
            float sumr = 0.f;
            for (int k0 = 0; k0 < n; k0 += EMD_TILE) {
                const int kend = (n - k0) < EMD_TILE ? (n - k0) : EMD_TILE;
                for (int k = tid; k < kend; k += 256) {
                    buf[k * 4 + 0] = p1[(k0 + k) * 3]; buf[k * 4 + 1] = p1[(k0 + k) * 3 + 1]; buf[k * 4 + 2] = p1[(k0 + k) * 3 + 2];
                    buf[k * 4 + 3] = ratioL[k0 + k];
                }
                __syncthreads();
                for (int k = 0; k < kend; ++k) {
                    const float dx = x2 - buf[k * 4], dy = y2 - buf[k * 4 + 1], dz = z2 - buf[k * 4 + 2];
                    sumr += expf(level * (dx * dx + dy * dy + dz * dz)) * buf[k * 4 + 3];
                }
                __syncthreads();
            }
            if (l < m) {
                const float rr = remainR[l];
                sumr *= rr;
                const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
                ratioR[l] = consumption * rr;
                remainR[l] = fmaxf(0.0f, rr - sumr);
            }
        }
        __syncthreads();
        // pass 3: the match increment, priced immediately; remainL
        for (int k0 = 0; k0 < n; k0 += 256) {
            const int k = k0 + tid;
            float x1 = 0, y1 = 0, z1 = 0, rl = 0;
            if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; rl = ratioL[k]; }
            float suml = 0.f;
            for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
                const int lend = (m - l0) < EMD_TILE ? (m - l0) : EMD_TILE;
                for (int l = tid; l < lend; l += 256) {
                    buf[l * 4 + 0] = p2[(l0 + l) * 3]; buf[l * 4 + 1] = p2[(l0 + l) * 3 + 1]; buf[l * 4 + 2] = p2[(l0 + l) * 3 + 2];
                    buf[l * 4 + 3] = ratioR[l0 + l];
                }
                __syncthreads();
                if (k < n) {
                    for (int l = 0; l < lend; ++l) {
                        const float dx = buf[l * 4] - x1, dy = buf[l * 4 + 1] - y1, dz = buf[l * 4 + 2] - z1;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        const float v = expf(level * d2) * rl * buf[l * 4 + 3];
                        suml += v;
                        my_cost += v * sqrtf(d2);
                    }
                }
                __syncthreads();
            }
            if (k < n) remainL[k] = fmaxf(0.0f, remainL[k] - suml);
        }
        __syncthreads();
    }
    red[tid] = my_cost;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) cost[i] = red[0];
}

extern "C" long caspr_emd_ws_bytes(int B, int n, int m) { return (long)B * 2 * (n + m) * 4 + 256; }

extern "C" int caspr_emd_f32(const float *xyz1, const float *xyz2, int B, int n, int m, float *cost, void *ws, long ws_bytes,
                             void *stream)
{
    CASPR_REQUIRE(xyz1 && xyz2 && cost && ws && B > 0 && n > 0 && m > 0, "emd: bad arguments");
    CASPR_REQUIRE(ws_bytes >= caspr_emd_ws_bytes(B, n, m), "emd: workspace too small");
    emd_kernel<<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(xyz1, xyz2, n, m, cost, (float *)ws);
    CASPR_CHECK_LAUNCH("emd");
    return CASPR_OK;
}
