// mfma_pipe.hip -- the memory skeleton of cnf_rk4_x6w_kernel without its arithmetic: 4 waves (one per SIMD), each issuing
// 48 v_mfma_f32_32x32x16_bf16 per 24 KB weight piece; the pieces stream L2 -> LDS by LDS-DMA through a four-deep ring (6 DMA
// instructions per wave and piece, three pieces ahead, counted vmcnt(12) wait + one s_barrier per piece); every MFMA's A
// fragment is a ds_read_b128 issued 6-12 slots earlier (24 per wave and piece), B from registers.  What does each part cost?
//   flags: 1 = LDS-DMA stream   2 = piece barrier + vmcnt wait   4 = fragment reads   8 = the six DMA instructions of a piece spread over
//   the 48 slots behind the barrier (one at a time) instead of back to back;   FILL = dependent v_fma per slot
//   hipcc --offload-arch=gfx950 -O3 -o mfma_pipe mfma_pipe.hip && ./mfma_pipe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PIECE 24576
#define NPIECE 128
#define FENCE __builtin_amdgcn_sched_barrier(0)

template <int FLAGS, int FILL>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, const unsigned char *w, int rounds)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr bool DMA = FLAGS & 1, BAR = FLAGS & 2, RD = FLAGS & 4, SPREAD = FLAGS & 8, NOWAIT = FLAGS & 16, NOBAR = FLAGS & 32, QUIET = FLAGS & 64, PAIRS = FLAGS & 128, AMAJOR = FLAGS & 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 c[4] = {};
    bf16x8 b[3], f[2][2][3];          // fragments: [set of the region's parity][row tile][plane]
    unsigned h = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int p = 0; p < 3; ++p)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            b[p][j] = (short)((p ? ((h >> 4) & 1) << 15 : 0) | ((0x7c - 8 * p + ((h >> 20) & 3)) << 7) | ((h >> 8) & 0x7f));
        }
    for (int t = 0; t < 2; ++t)
        for (int p = 0; p < 3; ++p) f[0][t][p] = f[1][t][p] = b[p];
    float v0 = 1.0f, ka = 1.0001f, kb = 0.5f;
    asm volatile("" : "+v"(ka), "+v"(kb));
    const unsigned lbase = (unsigned)(size_t)lds;
    const unsigned voff = lane * 16;
    // this wave's share of a piece: 6 x 1 KB at wave * 6 KB
    auto dma = [&](int s) __attribute__((always_inline)) {
        if (!DMA) return;
        const unsigned char *src = w + (size_t)(s & (NPIECE - 1)) * PIECE + wave * 6144;
        const unsigned dst = lbase + (s & 3) * PIECE + wave * 6144;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)            // the instruction offset (13 bits) advances the global AND the LDS address
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048"
                         : : "s"(dst + hlf * 3072), "v"(voff), "s"(src + hlf * 3072) : "memory");
    };
    auto dma1 = [&](int s, int j) __attribute__((always_inline)) {       // instruction j of the six
        if (!DMA) return;
        const unsigned char *src = w + (size_t)(s & (NPIECE - 1)) * PIECE + wave * 6144 + j * 1024;
        const unsigned dst = lbase + (s & 3) * PIECE + wave * 6144 + j * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(voff), "s"(src) : "memory");
    };
    // fill the ring: pieces 0, 1, 2; the loop issues piece s + 3 behind the barrier of piece s
    for (int i = tid; i < 4 * PIECE / 16; i += 256) ((uint4 *)lds)[i] = ((const uint4 *)w)[i];
    __syncthreads();
    dma(1); dma(2);
    if (SPREAD || PAIRS) { dma1(3, 0); dma1(3, 1); } else if (!QUIET) dma(3);
    const int total = rounds * NPIECE;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
        const unsigned char *A = lds + (s & 3) * PIECE + lane * 16;
        const unsigned char *An = lds + ((s + 1) & 3) * PIECE + lane * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                if (r == 3 && i == 2) {                  // head of piece s + 1: its share has landed when <= 12 DMAs are outstanding
                    if (BAR) {
                        if (DMA && !NOWAIT) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
                        if (!NOBAR) __builtin_amdgcn_s_barrier();
                    }
                    if (!SPREAD && !QUIET) dma(s + 4);
                    FENCE;
                }
                if (QUIET && PAIRS) {                    // the kernel today: two instructions each in (region 0, slot 7), (region 1, slot 8), (region 3, slot 9)
                    if (r == 0 && i == 7) { dma1(s + 3, 2); dma1(s + 3, 3); }
                    if (r == 1 && i == 8) { dma1(s + 3, 4); dma1(s + 3, 5); }
                    if (r == 3 && i == 9) { dma1(s + 4, 0); dma1(s + 4, 1); }
                } else if (QUIET) {                             // the kernel's slot pattern: producer micro-steps (FILL VALU) in slots 1, 3, 4, 6, 7, 9, 10,
                    if (r < 3 && i == 8) dma1(s + 3, 2 * r);       // reads in slots 0-5: slots 8 and 11 carry nothing but the MFMA
                    if (r < 3 && i == 11) dma1(s + 3, 2 * r + 1);
                } else if (SPREAD) {                            // piece s + 3: instructions 2..5 in regions 0..2, piece s + 4: 0, 1 behind the barrier
                    if (r == 0 && i == 4) dma1(s + 3, 2);
                    if (r == 1 && i == 4) dma1(s + 3, 3);
                    if (r == 2 && i == 4) dma1(s + 3, 4);
                    if (r == 2 && i == 9) dma1(s + 3, 5);
                    if (r == 3 && i == 5) dma1(s + 4, 0);
                    if (r == 3 && i == 9) dma1(s + 4, 1);
                }
                // product order: the kernel's (smallest terms first, the two row tiles alternating) or, flag 256, A-major: the three products
                // of one weight fragment back to back (same A operand in consecutive MFMAs), one row tile after the other
                constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
                constexpr int MA[6] = {2, 1, 1, 0, 0, 0}, MB[6] = {0, 0, 1, 0, 1, 2};
                if (AMAJOR)
                    c[(i / 6) + 2 * (r & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[r & 1][i / 6][MA[i % 6]], b[MB[i % 6]], c[(i / 6) + 2 * (r & 1)], 0, 0, 0);
                else
                    c[(i & 1) + 2 * (r & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[r & 1][i & 1][TA[i >> 1]], b[TB[i >> 1]], c[(i & 1) + 2 * (r & 1)], 0, 0, 0);
                FENCE;
                if (RD) {
                    if (r < 3 && i < 6) f[(r + 1) & 1][i / 3][i % 3] = *(const bf16x8 *)(A + ((r + 1) * 6 + i) * 1024);
                    if (r == 3 && i >= 2 && i < 8) f[0][(i - 2) / 3][(i - 2) % 3] = *(const bf16x8 *)(An + (i - 2) * 1024);
                }
                if (!QUIET || ((0x6da >> i) & 1))        // 0x6da = slots 1, 3, 4, 6, 7, 9, 10
#pragma unroll
                for (int q = 0; q < FILL; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(ka), "v"(kb));
                FENCE;
            }
        }
        if ((s & 15) == 15)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = c[i] * 0.25f;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = v0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) sum += c[i][j];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static int g_rounds = 40;
template <int FLAGS, int FILL>
void run(const char *name, float *out, unsigned long long *cyc, const unsigned char *w)
{
    const int rounds = g_rounds;                // 40: 5120 pieces ~ 5 ms;  ./mfma_pipe 800 for ~100 ms per run (power management settles)
    (void)hipFuncSetAttribute((const void *)k<FLAGS, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * PIECE);
    k<FLAGS, FILL><<<256, 256, 4 * PIECE>>>(out, cyc, w, 2);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<FLAGS, FILL><<<256, 256, 4 * PIECE>>>(out, cyc, w, rounds);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double pieces = rounds * (double)NPIECE;
    printf("%-52s fill %d  %7.1f cycles/piece (s_memtime; 1536 = MFMA only)  %8.3f ms  -> %.0f TFLOP/s  (%.2f s_memtime ticks per ns)\n", name, FILL,
           (double)hc / pieces, ms, 256.0 * 4 * pieces * 48 * 32768.0 / (ms * 1e-3) / 1e12, (double)hc / (ms * 1e6));
}
int main(int argc, char **argv)
{
    if (argc > 1) g_rounds = atoi(argv[1]);
    float *out; unsigned long long *cyc; unsigned char *w;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    const size_t wb = (size_t)NPIECE * PIECE;
    unsigned short *hw = (unsigned short *)malloc(wb);
    unsigned h = 777u;
    for (size_t i = 0; i < wb / 2; ++i) {       // bf16 weights: random sign, exponent over eight binades, random mantissa
        h = h * 1664525u + 1013904223u;
        hw[i] = (unsigned short)((((h >> 3) & 1) << 15) | ((0x77 + ((h >> 5) & 7)) << 7) | ((h >> 8) & 0x7f));
    }
    (void)hipMalloc(&w, wb); (void)hipMemcpy(w, hw, wb, hipMemcpyHostToDevice);
    run<0, 0>("MFMA only (constant fragments)", out, cyc, w);
    run<4, 0>("+ fragment reads", out, cyc, w);
    run<4 | 2, 0>("+ fragment reads + barrier", out, cyc, w);
    run<4 | 1, 0>("+ fragment reads + DMA (no wait: racy, timing only)", out, cyc, w);
    run<4 | 2 | 1, 0>("+ fragment reads + DMA + wait / barrier (the skeleton)", out, cyc, w);
    run<16 | 4 | 2 | 1, 0>("skeleton without the vmcnt wait (racy)", out, cyc, w);
    run<32 | 4 | 2 | 1, 0>("skeleton without the barrier (racy)", out, cyc, w);
    run<32 | 8 | 4 | 2 | 1, 0>("spread, without the barrier (racy)", out, cyc, w);
    run<16 | 8 | 4 | 2 | 1, 0>("spread, without the vmcnt wait (racy)", out, cyc, w);
    run<8 | 4 | 2 | 1, 0>("skeleton, DMA spread over the piece", out, cyc, w);
    run<8 | 4 | 2 | 1, 2>("skeleton, DMA spread over the piece", out, cyc, w);
    run<8 | 4 | 2 | 1, 3>("skeleton, DMA spread over the piece", out, cyc, w);
    run<64 | 4 | 2 | 1, 0>("skeleton, one DMA in slots 8 and 11 of regions 0-2", out, cyc, w);
    run<64 | 4 | 2 | 1, 3>("  ... fill in slots 1,3,4,6,7,9,10 only", out, cyc, w);
    run<64 | 4 | 2 | 1, 4>("  ... fill in slots 1,3,4,6,7,9,10 only", out, cyc, w);
    run<64 | 4 | 2 | 1, 5>("  ... fill in slots 1,3,4,6,7,9,10 only", out, cyc, w);
    run<128 | 64 | 4 | 2 | 1, 0>("skeleton, DMA in three pairs (slots 7 / 8 / 9 of regions 0 / 1 / 3)", out, cyc, w);
    run<128 | 64 | 4 | 2 | 1, 3>("  ... fill in slots 1,3,4,6,7,9,10 only", out, cyc, w);
    run<256 | 64 | 4 | 2 | 1, 0>("quiet-slot skeleton, A-major product order", out, cyc, w);
    run<256 | 64 | 4 | 2 | 1, 3>("  ... fill in slots 1,3,4,6,7,9,10 only", out, cyc, w);
    run<256 | 0, 0>("MFMA only, A-major product order", out, cyc, w);
    run<64 | 4, 4>("no DMA, no barrier, fill in slots 1,3,4,6,7,9,10", out, cyc, w);
    run<64 | 4, 5>("no DMA, no barrier, fill in slots 1,3,4,6,7,9,10", out, cyc, w);
    run<4 | 2 | 1, 1>("skeleton", out, cyc, w);
    run<4 | 2 | 1, 2>("skeleton", out, cyc, w);
    run<4 | 2 | 1, 3>("skeleton", out, cyc, w);
    run<4 | 2 | 1, 4>("skeleton", out, cyc, w);
    run<0, 3>("MFMA only", out, cyc, w);
    return 0;
}
