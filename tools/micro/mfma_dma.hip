// mfma_dma.hip -- what does it cost ONE wave per SIMD to feed LDS while it issues v_mfma_f32_32x32x16_bf16 back to back?
// G memory operations per 16 MFMAs (spread evenly), 1 KB per wave and operation, the source resident in L2:
//   kind 0: global_load_lds_dwordx4 (LDS-DMA; M0 written before each, as the CNF kernel does)
//   kind 1: global_load_dwordx4 into VGPRs
//   kind 2: ds_write_b128
//   kind 3: global_load_dwordx4 + ds_write_b128 of a register loaded earlier (the register-staged form of kind 0)
//   kind 4: global_load_lds_dword (256 B per wave and operation)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_dma mfma_dma.hip && ./mfma_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int G>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, const float *src, int iters)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    f32x16 c0 = {}, c1 = {};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 3, 4, 5, 6, 7, 8, 9};
    asm volatile("" : "+v"(a), "+v"(b));
    f32x4 r[8];
    for (int i = 0; i < 8; ++i) r[i] = (f32x4){0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *gp = src + ((size_t)blockIdx.x * 4 + wave) * 8 * 256;          // 8 KB per wave, cycled
    const unsigned voff = lane * 16;
    const unsigned lbase = (unsigned)(size_t)lds + wave * 8192;
    f32x4 *lw = (f32x4 *)(lds + wave * 8192) + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            if ((u * G) / 16 != ((u + 1) * G) / 16) {
                const int j = ((u * G) / 16) & 7;
                const float *p = gp + j * 256;
                if (KIND == 0)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lbase + j * 1024), "v"(voff), "s"(p) : "memory");
                if (KIND == 1 || KIND == 3) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[j]) : "v"(voff), "s"(p) : "memory");
                if (KIND == 2) lw[64 * j] = r[j];
                if (KIND == 3) lw[64 * j] = r[(j + 4) & 7];
                if (KIND == 4)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(lbase + j * 1024), "v"(lane * 4), "s"(p) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][3];
    s += ((float *)lds)[threadIdx.x];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int G>
void run(const char *name, float *out, unsigned long long *cyc, const float *src)
{
    const int iters = 2000;
    (void)hipFuncSetAttribute((const void *)k<KIND, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    k<KIND, G><<<256, 256, 32768>>>(out, cyc, src, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<KIND, G><<<256, 256, 32768>>>(out, cyc, src, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %2d per 16 MFMAs  %6.2f cycles/MFMA (s_memtime)  %7.3f ms  -> %.0f TFLOP/s\n", name, G, (double)h / (iters * 16.0), ms,
           256.0 * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12);
}
#define ROW(KIND, name) run<KIND, 0>(name, out, cyc, src); run<KIND, 1>(name, out, cyc, src); run<KIND, 2>(name, out, cyc, src); \
    run<KIND, 4>(name, out, cyc, src); run<KIND, 8>(name, out, cyc, src); run<KIND, 16>(name, out, cyc, src);
int main()
{
    float *out, *src; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipMalloc(&src, (size_t)256 * 4 * 8 * 1024); (void)hipMemset(src, 0, (size_t)256 * 4 * 8 * 1024);
    ROW(0, "global_load_lds_dwordx4 (LDS-DMA, 1 KB)")
    ROW(1, "global_load_dwordx4 -> VGPR")
    ROW(2, "ds_write_b128")
    ROW(3, "global_load_dwordx4 + ds_write_b128")
    ROW(4, "global_load_lds_dword (LDS-DMA, 256 B)")
    return 0;
}
