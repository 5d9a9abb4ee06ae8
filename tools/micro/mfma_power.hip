// mfma_power.hip -- sustained v_mfma_f32_32x32x16_bf16 rate over ~100 ms by operand data: the matrix pipe is clocked by the
// chip's power budget, and the power of an MFMA depends on what it multiplies.  1 wave per SIMD, 4 accumulators, no memory traffic.
//   data 0: zeros   1: one constant   2: random mantissas, same sign / exponent   3: random sign + mantissa + exponent in [2^-8, 1)
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
            a[i][j] = (short)va; b[i][j] = (short)vb;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 2) & 3], c[u & 3], 0, 0, 0);
        if (data == 3 && (it & 63) == 63)      // keep the accumulators bounded
            for (int i = 0; i < 4; ++i) c[i] = c[i] * 0.5f;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    for (int data = 0; data < 4; ++data)
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 400000;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            k<<<256, 256>>>(out, iters, data);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("data %d: %8.2f ms  %7.1f TFLOP/s  (= %.3f GHz effective at 32 cycles / MFMA)\n", data, ms,
                   256.0 * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12, 256.0 * 4 * iters * 16 * 32.0 / (ms * 1e-3) / 1e9 / 1024.0);
        }
    return 0;
}
