// Sustained f32 MFMA rate: a register-only v_mfma_f32_16x16x4_f32 loop (no memory traffic) run for short and long
// durations.  Tells how much of the 157.3 TFLOP/s datasheet peak a long-running kernel can actually hold on this box
// (clock management under sustained matrix load) -- the practical ceiling the roofline fractions should be read against.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_sustained.hip -o /tmp/mfma_sustained ; run: /tmp/mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float *out, long iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 2048 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 2;   // 2 workgroups of 4 waves per CU = 2 waves per SIMD
    for (long iters : {20000L, 20000L, 200000L, 2000000L, 200000L}) {
        hipEventRecord(e0);
        mfma_loop<<<blocks, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 /*waves*/ * iters * 8 * 2048.0;
        printf("iters %8ld  %9.3f ms  %7.1f TFLOP/s\n", iters, ms, flop / ms / 1e9);
    }
    return 0;
}
