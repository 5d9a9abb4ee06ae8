// Are the packed-f32 VALU operations (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) bit-identical to their scalar forms when the wave shares its
// compute unit with a bf16-MFMA kernel of another stream?  Round 6: farthest-point sampling with packed distance arithmetic chose wrong centres
// beside the encoder's stats-only conv; the same kernel with scalar arithmetic never did (profiles/r06_fps_beside_conv.txt).  This kernel repeats
// FPS's distance computation both ways on the same registers and counts disagreements; tools/r06_pk_check.py launches it beside the conv.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pk_check_kernel(const float *__restrict__ pts, int n, int iters, unsigned *__restrict__ bad, unsigned *__restrict__ first, unsigned *__restrict__ detail)
{
    __shared__ float sx[1536 * 3];
    const int tid = threadIdx.x;
    const float *p = pts + (long)blockIdx.x * n * 3;
    for (int i = tid; i < n * 3; i += 256) sx[i] = p[i];
    __syncthreads();
    const int k0 = tid, k1 = tid + 256;
    const f2 X = {sx[k0 * 3 + 0], sx[k1 * 3 + 0]}, Y = {sx[k0 * 3 + 1], sx[k1 * 3 + 1]}, Z = {sx[k0 * 3 + 2], sx[k1 * 3 + 2]};
    unsigned nbad = 0;
    unsigned cnt[6] = {0, 0, 0, 0, 0, 0};
    int old = 0;
    for (int j = 0; j < iters; ++j) {
        const float x1 = sx[old * 3 + 0], y1 = sx[old * 3 + 1], z1 = sx[old * 3 + 2];
        // packed
        const f2 xx = {x1, x1}, yy = {y1, y1}, zz = {z1, z1};
        const f2 dx = X - xx, dy = Y - yy, dz = Z - zz;
        const f2 y2 = dy * dy;
        const f2 t1 = __builtin_elementwise_fma(dx, dx, y2);
        f2 dp = __builtin_elementwise_fma(dz, dz, t1);
        asm volatile("" : "+v"(dp));
        // scalar, kept scalar by opaque copies of the inputs
        float ds[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float ax = X[h], ay = Y[h], az = Z[h];
            asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az));
            const float ex = ax - x1, ey = ay - y1, ez = az - z1;
            float q = ey * ey;
            asm volatile("" : "+v"(q));
            q = __builtin_fmaf(ex, ex, q);
            asm volatile("" : "+v"(q));
            ds[h] = __builtin_fmaf(ez, ez, q);
        }
        // per operation, on the scalar path's own intermediates (opaque copies, so that the compiler keeps both forms)
        {
            f2 X2 = X, Y2 = Y, Z2 = Z;
            asm volatile("" : "+v"(X2), "+v"(Y2), "+v"(Z2));
            f2 sub = Y2 - yy;                                   // v_pk_add_f32 (neg)
            asm volatile("" : "+v"(sub));
            float s0 = Y[0], s1 = Y[1];
            asm volatile("" : "+v"(s0), "+v"(s1));
            s0 = s0 - y1; s1 = s1 - y1;
            asm volatile("" : "+v"(s0), "+v"(s1));
            if (__float_as_uint(sub[0]) != __float_as_uint(s0) || __float_as_uint(sub[1]) != __float_as_uint(s1)) cnt[1] += 1;
            f2 m = sub * sub;                                   // v_pk_mul_f32
            asm volatile("" : "+v"(m));
            float m0 = s0 * s0, m1 = s1 * s1;
            asm volatile("" : "+v"(m0), "+v"(m1));
            if (__float_as_uint(m[0]) != __float_as_uint(m0) || __float_as_uint(m[1]) != __float_as_uint(m1)) cnt[2] += 1;
            f2 dxx = X2 - xx;
            asm volatile("" : "+v"(dxx));
            f2 f = __builtin_elementwise_fma(dxx, dxx, m);      // v_pk_fma_f32
            asm volatile("" : "+v"(f));
            float e0 = dxx[0], e1 = dxx[1];
            asm volatile("" : "+v"(e0), "+v"(e1));
            float f0 = __builtin_fmaf(e0, e0, m0), f1 = __builtin_fmaf(e1, e1, m1);
            asm volatile("" : "+v"(f0), "+v"(f1));
            if (__float_as_uint(f[0]) != __float_as_uint(f0) || __float_as_uint(f[1]) != __float_as_uint(f1)) cnt[3] += 1;
        }
        // the packed form once more: transient or sticky?
        f2 dq;
        {
            f2 X3 = X, Y3 = Y, Z3 = Z;
            asm volatile("" : "+v"(X3), "+v"(Y3), "+v"(Z3));
            const f2 ax = X3 - xx, ay = Y3 - yy, az = Z3 - zz;
            const f2 b2 = ay * ay;
            const f2 c1 = __builtin_elementwise_fma(ax, ax, b2);
            dq = __builtin_elementwise_fma(az, az, c1);
            asm volatile("" : "+v"(dq));
        }
        if (__float_as_uint(dq[0]) != __float_as_uint(ds[0]) || __float_as_uint(dq[1]) != __float_as_uint(ds[1])) cnt[4] += 1;
        if (__float_as_uint(dq[0]) != __float_as_uint(dp[0]) || __float_as_uint(dq[1]) != __float_as_uint(dp[1])) cnt[5] += 1;
        const bool b0 = __float_as_uint(dp[0]) != __float_as_uint(ds[0]), b1 = __float_as_uint(dp[1]) != __float_as_uint(ds[1]);
        if (b0 || b1) {
            if (nbad == 0 && first) { atomicCAS(first, 0u, (unsigned)(j + 1)); }
            if (nbad == 0 && detail) {      // one record: lane, iteration, the three inputs and both results of component 0 / 1
                if (atomicCAS(detail, 0u, 1u) == 0u) {
                    detail[1] = (unsigned)tid; detail[2] = (unsigned)j; detail[3] = blockIdx.x;
                    detail[4] = __float_as_uint(dp[0]); detail[5] = __float_as_uint(ds[0]); detail[6] = __float_as_uint(dp[1]); detail[7] = __float_as_uint(ds[1]);
                    detail[8] = __float_as_uint(X[0]); detail[9] = __float_as_uint(Y[0]); detail[10] = __float_as_uint(Z[0]);
                    detail[11] = __float_as_uint(x1); detail[12] = __float_as_uint(y1); detail[13] = __float_as_uint(z1);
                    detail[14] = __float_as_uint(dq[0]); detail[15] = __float_as_uint(dq[1]);
                }
            }
            nbad += 1;
        }
        // the next reference point: a data-dependent walk, as FPS's
        old = (int)((__float_as_uint(ds[0]) >> 7) % (unsigned)n);
        old = __builtin_amdgcn_readfirstlane(old);
    }
    if (nbad) atomicAdd(bad, nbad);
    for (int c = 1; c < 6; ++c) if (cnt[c]) atomicAdd(bad + c, cnt[c]);
}

extern "C" int pk_check(const float *pts, int frames, int n, int iters, unsigned *bad, unsigned *first, unsigned *detail, void *stream)
{
    if (n > 1536 || n < 512) return -1;
    pk_check_kernel<<<dim3(frames), dim3(256), 0, (hipStream_t)stream>>>(pts, n, iters, bad, first, detail);
    return (int)hipGetLastError();
}
