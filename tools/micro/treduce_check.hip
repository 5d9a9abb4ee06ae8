// treduce_check.hip -- checks xw_treduce16 / xw_rows_add (csrc/x6w_common.h) against a host reduction: lane l, register r holds
// f(l, r); for every half-wave h and r the sum / max / min over its 32 lanes must come out in lane 32 h + 16 p + 4 b + q of w[i]
// with r = 8 i + 4 (b & 1) + 2 (b >> 1) + p.       hipcc --offload-arch=gfx950 -O3 -I../../caspr_amd/csrc -I../../include
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "x6w_common.h"
__global__ void k(float *out)
{
    const int lane = threadIdx.x;
    float x[16], y[16], z[16], m[16], w[2];
    for (int r = 0; r < 16; ++r) {
        x[r] = (float)((lane * 7 + r * 13) % 31) - 11.0f;
        y[r] = x[r];
        z[r] = x[r];
        m[r] = (float)(100 * (lane >> 5) + r);          // equal over the 32 lanes of a half (XwFirst)
    }
    xw_treduce16(x, w, XwAdd{});
    out[lane] = w[0]; out[64 + lane] = w[1];
    xw_treduce16(y, w, XwMax{});
    out[128 + lane] = w[0]; out[192 + lane] = w[1];
    xw_treduce16(z, w, XwMin{});
    out[256 + lane] = w[0]; out[320 + lane] = w[1];
    xw_treduce16(m, w, XwFirst{});
    out[384 + lane] = w[0]; out[448 + lane] = w[1];
    const float s = xw_rows_add(row_allreduce_add<16>((float)((lane * 5) % 17)));
    out[512 + lane] = s;
}
int main()
{
    float *d, h[576];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 2; ++i) {
            const int hh = lane >> 5, p = (lane >> 4) & 1, b = (lane >> 2) & 3;
            const int r = 8 * i + 4 * (b & 1) + 2 * (b >> 1) + p;
            float s = 0, mx = -1e30f, mn = 1e30f;
            for (int l = 32 * hh; l < 32 * hh + 32; ++l) {
                const float v = (float)((l * 7 + r * 13) % 31) - 11.0f;
                s += v; mx = fmaxf(mx, v); mn = fminf(mn, v);
            }
            const float g[4] = {h[64 * i + lane], h[128 + 64 * i + lane], h[256 + 64 * i + lane], h[384 + 64 * i + lane]};
            const float want[4] = {s, mx, mn, (float)(100 * hh + r)};
            for (int q = 0; q < 4; ++q)
                if (g[q] != want[q]) { if (bad < 12) printf("lane %d i %d stat %d: got %g want %g\n", lane, i, q, g[q], want[q]); ++bad; }
        }
    for (int lane = 0; lane < 64; ++lane) {
        float s = 0;
        for (int l = 32 * (lane >> 5); l < 32 * (lane >> 5) + 32; ++l) s += (float)((l * 5) % 17);
        if (h[512 + lane] != s) { if (bad < 16) printf("rows_add lane %d: got %g want %g\n", lane, h[512 + lane], s); ++bad; }
    }
    printf("treduce_check: %d mismatches\n", bad);
    return bad != 0;
}
