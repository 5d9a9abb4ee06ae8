// mfma_fillers.hip -- how many instructions of which kind hide behind v_mfma_f32_32x32x16_bf16 when ONE wave per SIMD issues
// both (the geometry of the 128-point CNF kernel)?  Two alternating accumulators (a dependent MFMA is two slots behind), K
// fillers of one kind after every MFMA, 1 wave per SIMD, 256 workgroups x 256 threads.  Prints cycles per MFMA (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_fillers mfma_fillers.hip && ./mfma_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ f32x4 lds[1024];
    f32x16 c0 = {}, c1 = {};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 3, 4, 5, 6, 7, 8, 9};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    f32x4 r[8];
    for (int i = 0; i < 8; ++i) r[i] = (f32x4){0, 0, 0, 0};
    lds[threadIdx.x] = (f32x4){1, 2, 3, 4};
    lds[threadIdx.x + 256] = (f32x4){1, 2, 3, 4};
    __syncthreads();
    asm volatile("" : "+v"(a), "+v"(b));
    float ka = 1.0001f, kb = 0.5f;
    asm volatile("" : "+v"(ka), "+v"(kb));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < K; ++f) {
                const int j = (u * K + f) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(ka), "v"(kb));      // independent v_fma_f32 (8 chains); asm so that hipcc cannot sink them behind the MFMAs
                if (KIND == 1) v[j] = __builtin_amdgcn_exp2f(v[j]);                              // v_exp_f32
                if (KIND == 2) r[j] = lds[(threadIdx.x + 64 * j) & 1023];                        // ds_read_b128
                if (KIND == 3) asm volatile("s_nop 0");
                if (KIND == 4) v[0] = __builtin_fmaf(v[0], 1.0001f, 0.5f);                       // ONE dependent chain of v_fma_f32
                if (KIND == 5) asm volatile("v_accvgpr_read_b32 %0, a7" : "=v"(v[j]) : : "a7");
                if (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(v[(j + 1) & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) s += v[i] + r[i][0] + r[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int K>
void run(const char *name, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    k<KIND, K><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KIND, K><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s K=%d  %6.2f cycles/MFMA (s_memtime)  %7.3f ms  -> %.0f TFLOP/s\n", name, K, (double)h / (iters * 16.0), ms,
           256.0 * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12);
}
#define ROW(KIND, name) run<KIND, 0>(name, out, cyc); run<KIND, 1>(name, out, cyc); run<KIND, 2>(name, out, cyc); run<KIND, 3>(name, out, cyc); \
    run<KIND, 4>(name, out, cyc); run<KIND, 5>(name, out, cyc); run<KIND, 6>(name, out, cyc); run<KIND, 8>(name, out, cyc);
int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    ROW(0, "v_fma_f32 (independent)")
    ROW(4, "v_fma_f32 (one chain)")
    ROW(1, "v_exp_f32")
    ROW(2, "ds_read_b128")
    ROW(3, "s_nop 0")
    ROW(5, "v_accvgpr_read_b32")
    ROW(6, "v_cvt_pk_bf16_f32")
    return 0;
}
