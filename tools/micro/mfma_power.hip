// mfma_power.hip -- sustained v_mfma_f32_32x32x16_bf16 rate over ~100 ms by operand data: the matrix pipe is clocked by the
// chip's power budget, and the power of an MFMA depends on what it multiplies.  1 wave per SIMD, 4 accumulators, no memory traffic.
//   data 0: zeros   1: one constant   2: random mantissas, same sign / exponent   3: random sign + mantissa + exponent in [2^-8, 1)
//   4: as 3, B operand non-negative   5: as 3, both non-negative   6: as 3, ONE exponent (values in [0.5, 1)), random signs
//   7: the bf16x6 product sequence on three-plane operands as ROUNDING splits them (residual planes: random signs, 2^-8 / 2^-16 down),
//      A = weights (random sign), B = activations (plane 0 positive)      8: the same as TRUNCATION splits them (planes keep the sign)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, int data)
{
    f32x16 c[4] = {};
    bf16x8 a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            unsigned short va = 0, vb = 0;
            if (data == 1) va = vb = 0x3f80;
            if (data == 2) { va = 0x3f80 | ((h >> 8) & 0x7f); vb = 0x3f80 | ((h >> 16) & 0x7f); }
            if (data == 3) { va = (((h >> 3) & 1) << 15) | ((0x77 + ((h >> 5) & 7)) << 7) | ((h >> 8) & 0x7f);
                             vb = (((h >> 4) & 1) << 15) | ((0x77 + ((h >> 20) & 7)) << 7) | ((h >> 16) & 0x7f); }
            if (data >= 4 && data <= 6) {
                const unsigned ea = data == 6 ? 0x7e : 0x77 + ((h >> 5) & 7), eb = data == 6 ? 0x7e : 0x77 + ((h >> 20) & 7);
                const unsigned sa = data == 5 ? 0 : (h >> 3) & 1, sb = data == 6 ? (h >> 4) & 1 : 0;
                va = (sa << 15) | (ea << 7) | ((h >> 8) & 0x7f);
                vb = (sb << 15) | (eb << 7) | ((h >> 16) & 0x7f);
            }
            a[i][j] = (short)va; b[i][j] = (short)vb;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 2) & 3], c[u & 3], 0, 0, 0);
        if (data >= 3 && (it & 63) == 63)      // keep the accumulators bounded
            for (int i = 0; i < 4; ++i) c[i] = c[i] * 0.5f;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the six partial products of the bf16x6 scheme, alternating between two accumulators as the kernels do
__global__ __launch_bounds__(256, 1) void k6(float *out, int iters, int trunc)
{
    f32x16 c[4] = {};
    bf16x8 a[3], b[3];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u;
        const unsigned sa = (h >> 3) & 1, ea = 0x77 + ((h >> 5) & 7), eb = 0x77 + ((h >> 20) & 7);
        unsigned g = h;
        for (int p = 0; p < 3; ++p) {
            g = g * 1664525u + 1013904223u;
            const unsigned drop = p ? ((g >> 28) & 3) : 0;                       // a residual plane starts 0..3 bits below the 8 of its parent
            const unsigned sap = trunc || !p ? sa : (g >> 4) & 1, sbp = trunc || !p ? 0 : (g >> 6) & 1;
            a[p][j] = (short)((sap << 15) | ((ea - 8 * p - drop) << 7) | ((g >> 8) & 0x7f));
            b[p][j] = (short)((sbp << 15) | ((eb - 8 * p - drop) << 7) | ((g >> 16) & 0x7f));
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            c[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c[2 * u], 0, 0, 0);
            c[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c[2 * u + 1], 0, 0, 0);
            c[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c[2 * u], 0, 0, 0);
            c[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c[2 * u + 1], 0, 0, 0);
            c[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c[2 * u], 0, 0, 0);
            c[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c[2 * u + 1], 0, 0, 0);
            c[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c[2 * u], 0, 0, 0);
            c[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c[2 * u + 1], 0, 0, 0);
            c[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c[2 * u], 0, 0, 0);
            c[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c[2 * u + 1], 0, 0, 0);
            c[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c[2 * u], 0, 0, 0);
            c[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c[2 * u + 1], 0, 0, 0);
        }
        if ((it & 63) == 63)
            for (int i = 0; i < 4; ++i) c[i] = c[i] * 0.5f;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    for (int data = 0; data < 9; ++data)
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = data >= 7 ? 266667 : 400000;               // 24 MFMAs per iteration in the plane kernel, 16 in the other
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            if (data >= 7) k6<<<256, 256>>>(out, iters, data == 8);
            else k<<<256, 256>>>(out, iters, data);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("data %d: %8.2f ms  %7.1f TFLOP/s  (= %.3f GHz effective at 32 cycles / MFMA)\n", data, ms,
                   256.0 * 4 * iters * (data >= 7 ? 24 : 16) * 32768.0 / (ms * 1e-3) / 1e12, 256.0 * 4 * iters * (data >= 7 ? 24 : 16) * 32.0 / (ms * 1e-3) / 1e9 / 1024.0);
        }
    return 0;
}
