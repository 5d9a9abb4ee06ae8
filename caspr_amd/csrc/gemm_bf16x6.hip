// Pointwise conv with f32 results on the bf16 matrix pipe ("bf16x6"): the default kernel of the large pointwise convs
// (ops.CONV_BF16X6; the f32 MFMA kernels of gemm.hip take the shapes it does not cover and the "f32" mode).  Same contract
// as caspr_conv1x1_f32.
//
// Measured and dropped (round 2, tools/conv_x6_trace.py; DESIGN.md section 3 has the per-phase cycle counts): (a) one workgroup
// per CU with both operands double-buffered and every transfer inside the product stream (5,700 cycles per 192 MFMAs: a single
// wave per SIMD pays each LDS-DMA issue and each VALU instruction beyond the MFMA shadow in its own stream); (b) the same with
// eight waves / 256 points and one activation buffer (9,000 cycles per 2 x 192 MFMAs per SIMD); (c) a ping-pong of two four-wave
// groups, one multiplying while the other stages: the staging wave's ~250 VALU instructions get about ONE issue slot per MFMA of
// its partner (4,000 cycles, s_setprio makes no difference), so the phases do not shorten.  (d) a double-buffered form -- 8 waves (4 x 2, wave tile 64 x 64), two 72 KB LDS stages, one
// barrier per K chunk with the next chunk's DMA / split-store under the current chunk's MFMAs -- ran 183 f32-equivalent
// TFLOP/s on the 1600 x 1600 head layer against 198 for this kernel (same box, random data): the smaller wave tile reads 30 %
// more fragments per MFMA and eight waves meet one barrier; two independent workgroups per CU already hide each other's
// staging.
//
// Every f32 operand is written as the EXACT sum of three bf16 numbers, x = x1 + x2 + x3 (8 significand bits each, by
// truncation, so every remainder is exact), and a*b is evaluated as a3b1 + a2b2 + a1b3 + a2b1 + a1b2 + a1b1: the six
// partial products are exact inside v_mfma_f32_16x16x32_bf16 and accumulate in f32; the three dropped terms are below
// 2^-23 |a||b|.  Measured against an f64 evaluation the result is as close as a sequential f32 FMA chain (max 3.3e-6 vs
// 4.2e-6 at K=512, tools/micro/bf16x6_gemm.hip), i.e. this is f32 arithmetic on a faster pipe, not a reduced-precision
// mode: six 16-cycle MFMAs replace eight 32-cycle v_mfma_f32_16x16x4_f32 per 16x16x32 block.
//
// Workgroup = 4 waves (2 x 2), tile 256 output channels x 128 points, K chunks of 32; wave tile 128 x 64 (8 x 4 MFMA
// tiles, 128 accumulator registers).  LDS per chunk (72 KB, single buffer, two workgroups per CU): 3 weight planes of
// 256 rows x 64 B, filled by LDS-DMA from a pre-swizzled pack (no staging registers), and 3 activation planes of 128 rows
// x 64 B written by the threads after the split (and the fused GroupNorm/ReLU of the producer, as in gemm.hip).  A
// row's four 16-byte pieces sit at piece ^ swz(row), swz = (row >> 1) & 3: the 16 lanes of every ds_read_b128 lane
// group ({0-3,12-15,20-27}, ...) touch 16 distinct 16-byte slots of the 256-byte bank window, AND the 8 consecutive
// lanes of a ds_write_b128 group (rows r..r+7 of one piece column) touch 8 distinct slots of the 128-byte store
// window (swz = 0,3,2,1 per row QUAD, as in ode_bf16x6.hip, is conflict-free for the reads only: 21 % of this
// kernel's LDS cycles were store conflicts, profiles/r01_bf16x6_optin_pmc_summary.txt).
#include "common.h"
#include <type_traits>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define X6_TM 256
#define X6_TP 128
#define X6_PA (X6_TM * 64)     // bytes of one weight plane of a chunk
#define X6_PB (X6_TP * 64)     // bytes of one activation plane of a chunk
#define X6_CHUNK (3 * X6_PA)   // packed weight bytes per (channel tile, k chunk)
#define X6_LDS (3 * X6_PA + 3 * X6_PB)

__device__ __host__ __forceinline__ int x6_swz(int row) { return (row >> 1) & 3; }
__device__ __forceinline__ int x6_off(int row, int piece) { return row * 64 + ((piece ^ x6_swz(row)) << 4); }

// x = h1 + h2 + h3 exactly; each h keeps the top 8 significand bits of what is left (a bf16 value held in f32)
__device__ __forceinline__ void x6_split(float x, float &h1, float &h2, float &h3)
{
    h1 = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - h1;
    h2 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float r2 = r1 - h2;
    h3 = __uint_as_float(__float_as_uint(r2) & 0xffff0000u);
}
// the same exact split for a pair with the hardware round-to-nearest conversion (v_cvt_pk_bf16_f32): the three packed words
// are the pair's entries of the three planes (remainders after rounding 24 -> 8 bits have <= 15, then <= 7 significant bits)
typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x6_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x6_split_pair(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3)
{
    x6_f32x2 v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));
    x6_f32x2 h = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    v = v - h;
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));
    h = (x6_f32x2){__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xffff0000u)};
    v = v - h;
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));
}
__device__ __forceinline__ unsigned x6_pack(float lo, float hi) { return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u); }

// W (Cout, ldw) f32, columns col0 .. col0+Cin-1 -> [channel tile][k chunk][plane][row 0..255][piece'][8 bf16]; rows past
// Cout are zero.  One thread per (tile, chunk, row, piece).
__global__ void pack_weight_bf16x3_kernel(const float *__restrict__ w, int ldw, int Cout, int col0, int Cin, unsigned char *__restrict__ out,
                                          long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int nk = Cin / 32;
    const int piece = (int)(i & 3), row = (int)((i >> 2) & (X6_TM - 1));
    const long ck = i >> 10;   // tile * nk + chunk
    const int kc = (int)(ck % nk), mt = (int)(ck / nk);
    const int co = mt * X6_TM + row;
    float h[3][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float v = co < Cout ? w[(long)co * ldw + col0 + kc * 32 + piece * 8 + q] : 0.f;
        x6_split(v, h[0][q], h[1][q], h[2][q]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = x6_pack(h[pl][2 * q], h[pl][2 * q + 1]);
        *(u32x4 *)(out + ck * X6_CHUNK + pl * X6_PA + x6_off(row, piece)) = v;
    }
}

// STATS: the GroupNorm that follows the conv (conv -> GroupNorm -> ReLU is the model's building block) gets its statistics
// from the accumulators instead of a second pass over the output: every workgroup tile leaves, per output channel, the mean,
// the sum of squared deviations from it (two passes over the accumulator registers), the max and the min over its 128 points
// (f32; one float4 per (batch entry, point tile, channel) in `part`), conv_gn_finalize_kernel combines them pairwise in f64 in
// a fixed order.  Y may then be NULL (only the statistics are wanted:
// the global PointNet's last layer, whose output is max-pooled).
// act mode 2 (training tier, caspr_conv1x1_cnf_act_bf16x6_f32): the gated softplus layer of the CNF's ODE function applied to value /
// tangent row pairs in the epilogue.  Rows come in blocks of 64 = 32 value rows + the tangent rows of the same 32 points
// (backward_flow.hip, blk = 32): a lane's column tiles 0, 1 are values, 2, 3 their tangents.  The raw product is stored too (the
// backward pass recomputes the activation from it).
// act mode 3 (caspr_conv1x1_cnf_act_bwd_bf16x6_f32): the BACKWARD of that layer in the epilogue of the data-gradient conv that
// produces its dH (this launch multiplies the NEXT layer's dZ by its transposed weight): dZ = act'(Z) dH on the same row pairs, with
// the raw product Z of the layer read back, and the per-frame gate / bias gradients as partial sums per (128-row tile, wave half).
struct X6Act {
    const float *gate;   // (batch entries, cstride); the per-entry bias `bbias` is beta, `bias` the layer bias
    float *z;            // mode 2: raw product out (rows as Y);  mode 3: the layer's raw product in
    int ldz;
    float *red;          // mode 3: [2][batch entries * Pt * 2][cstride] partial sums of dgate | dbeta
};

template <bool FUSED, bool STATS>
__global__ __launch_bounds__(256, 2) void conv1x1_bf16x6_kernel(const unsigned char *__restrict__ wpk, const float *__restrict__ bias,
                                                                const float *__restrict__ bbias, const float *__restrict__ X,
                                                                int ldx, const float *__restrict__ in_scale,
                                                                const float *__restrict__ in_shift, int in_relu, int relu_from,
                                                                float *__restrict__ Y, int ldy, int P, int Cin, int Cout, int act,
                                                                int Mt, int Pt, f32x4 *__restrict__ part, int cstride, unsigned long long *trace,
                                                                X6Act ax)
{
    // debug build (tools/conv_x6_trace.py): s_memtime stamps of workgroup 0, thread 0, five per K chunk
#ifdef CASPR_DEBUG_HOOKS
#define X6_STAMP(i) if (trace && blockIdx.x == 0 && threadIdx.x == 0 && (i) < 160) trace[i] = __builtin_amdgcn_s_memtime();
#else
#define X6_STAMP(i)
#endif
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *sA = lds;
    unsigned char *sB = lds + 3 * X6_PA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware, bijective for any block count: consecutive work items (the channel tiles of one point tile) share an L2
    const int nblk = gridDim.x, lin = blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int mt = work % Mt, pt = (work / Mt) % Pt, b = work / (Mt * Pt);
    const int p0 = pt * X6_TP;
    const int nk = Cin / 32;
    // the last channel tile may be mostly padding (zero rows in the pack): a wave without real channels skips the products
    const int co_w = mt * X6_TM + wm * 128;
    const bool live = co_w < Cout;   // wave-uniform; a dead wave only stages and meets the barriers
    const bool half = co_w + 64 >= Cout;   // at most four of the wave's eight row tiles hold real channels (1600 = 6 x 256 + 64)

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // activation staging: thread = (row xr, 16-float half xh of the chunk); xh is wave-uniform, so the fused transform's
    // scale / shift come through scalar loads
    const int xr = tid & 127, xh = wave >> 1;
    const float *xsrc = X + ((long)b * P + p0 + xr) * ldx + 16 * xh;
    const float *sc = FUSED ? in_scale + (long)b * Cin + 16 * xh : nullptr;
    const float *sh = FUSED ? in_shift + (long)b * Cin + 16 * xh : nullptr;
    f32x4 xreg[4];
    const unsigned char *wsrc = wpk + ((long)mt * nk) * X6_CHUNK + (wave * 12) * 1024;   // wave-uniform; the lane's 16 bytes ride as a
    const unsigned wlane = lane * 16;                                                    // 32-bit offset (no 64-bit add per transfer)
    auto gload = [&](int kc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xreg[q] = ld4(xsrc + kc * 32 + 4 * q);
    };
    auto dma = [&](int kc) {
#pragma unroll
        for (int s = 0; s < 12; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + (long)kc * X6_CHUNK + s * 1024 + wlane),
                                             (__attribute__((address_space(3))) void *)(sA + (wave * 12 + s) * 1024), 16, 0, 0);
    };
    // The weight piece goes through registers between the two barriers of a chunk (12 x 16 B per thread: the fragment registers are
    // free then): an LDS-DMA instruction costs its wave ~125 cycles of issue even on an otherwise idle CU (12 per piece: 1,500 of a
    // lone workgroup's 2,360-cycle stage, tools/conv_x6_trace.py), a plain 16-byte load ~17, and the piece is back (L2) before the
    // activation split is done.  The very first piece still comes by DMA (nothing to overlap it with).
    u32x4 wreg[12];
    auto wload = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 12; ++s) wreg[s] = *(const u32x4 *)(wsrc + (long)kc * X6_CHUNK + s * 1024 + wlane);
    };
    auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 12; ++s) *(u32x4 *)(sA + (wave * 12 + s) * 1024 + wlane) = wreg[s];
    };
    auto lstore = [&](int kc) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            float xv[8];
            // the ReLU switches on at a multiple of 8 channels (checked by the host): one wave-uniform lower bound per piece
            const float lo = (FUSED && in_relu && kc * 32 + 16 * xh + 8 * pc >= relu_from) ? 0.f : -INFINITY;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = xreg[2 * pc + (q >> 2)][q & 3];
                if (FUSED) {
                    const int k = kc * 32 + 8 * pc + q;   // relative to the 16 * xh already folded into sc / sh
                    v = fmaxf(v * sc[k] + sh[k], lo);
                }
                xv[q] = v;
            }
            u32x4 pv[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned p1, p2, p3;
                x6_split_pair(xv[2 * q], xv[2 * q + 1], p1, p2, p3);
                pv[0][q] = p1;
                pv[1][q] = p2;
                pv[2][q] = p3;
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *(u32x4 *)(sB + pl * X6_PB + x6_off(xr, 2 * xh + pc)) = pv[pl];
        }
    };

    gload(0);
    dma(0);
    lstore(0);
    __syncthreads();
    // the K loop, for a wave with NMI live row tiles (two whole copies: a branch inside the loop sends registers to scratch)
    auto kloop = [&](auto nmi) __attribute__((always_inline)) {
    for (int kc = 0; kc < nk; ++kc) {
        const int kn = kc + 1 < nk ? kc + 1 : kc;   // unconditional re-load at the end (a branch here sends registers to scratch)
        X6_STAMP(5 * kc)
        gload(kn);
        if (live) {
            constexpr int NMI = decltype(nmi)::value;
            // B fragments of the chunk, then the row tiles: the three A-fragment reads of tile mi+1 are issued between the first
            // MFMAs of tile mi (two fragment sets; sched_group_barrier pins the pattern, hipcc on its own puts every read right in
            // front of its use and the wave sits out the LDS latency eight times per chunk)
            bf16x8 bf[3][4], af[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bf[pl][ni] = *(const bf16x8 *)(sB + pl * X6_PB + x6_off(wn * 64 + ni * 16 + j, g));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[0][pl] = *(const bf16x8 *)(sA + pl * X6_PA + x6_off(wm * 128 + j, g));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) {
                const bf16x8 (&a)[3] = af[mi & 1];
                if (mi + 1 < NMI) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) af[(mi + 1) & 1][pl] = *(const bf16x8 *)(sA + pl * X6_PA + x6_off(wm * 128 + (mi + 1) * 16 + j, g));
                }
                // smallest terms first; term-major so four independent accumulators sit between dependent MFMAs
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bf[0][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bf[1][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[2][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bf[0][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[1][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[0][ni], acc[mi][ni], 0, 0, 0);
                if (mi + 1 < NMI) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 21, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        X6_STAMP(5 * kc + 1)
        __syncthreads();
        X6_STAMP(5 * kc + 2)
        wload(kn);
        lstore(kn);
        wstore();
        X6_STAMP(5 * kc + 3)
        __syncthreads();
        X6_STAMP(5 * kc + 4)
    }
    };
    if (half) kloop(std::integral_constant<int, 4>());
    else kloop(std::integral_constant<int, 8>());

    // epilogue: lane holds channels co + r (D row = 4g + r) of point p (column j); Cout % 4 == 0
    const float *bb = bbias ? bbias + (long)b * cstride : nullptr;     // cstride: channels of the whole layer (this launch may cover a slice)
    f32x4 *sp = (f32x4 *)lds;   // STATS: [2 point halves][256 channels] {mean, sum of squared deviations, max, min}; the K loop ends on a barrier
    if (STATS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight DMA of the loop's last (repeated) stage targets the same LDS
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int co = co_w + mi * 16 + 4 * g;
        if (co >= Cout) continue;
        if constexpr (!STATS) {
            if ((act & 0xff) == 2) {
                const f32x4 b4 = ld4(bias + co), g4 = ld4(ax.gate + (long)b * cstride + co), be4 = ld4(bb + co);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const long rv = (long)b * P + p0 + wn * 64 + ni * 16 + j, rt = rv + 32;
                    const f32x4 zv = acc[mi][ni], zt = acc[mi][ni + 2];
                    st4(ax.z + rv * ax.ldz + co, zv);
                    st4(ax.z + rt * ax.ldz + co, zt);
                    f32x4 hv, ht;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // softplus and sigmoid from ONE 2^(-|a| log2 e) on the hardware transcendentals (1 ulp each, as the sampling
                        // kernel's softplus_fast): the libm forms of the stand-alone pass (~90 instructions per element, hidden there
                        // behind its memory traffic) made this epilogue half as long as the tile's product loop
                        const float a_ = (zv[r] + b4[r]) * g4[r] + be4[r];
                        const float u = __builtin_amdgcn_exp2f(fabsf(a_) * -1.44269504088896341f);
                        const float rc = __builtin_amdgcn_rcpf(1.0f + u);
                        hv[r] = fmaxf(a_, 0.0f) + 0.69314718055994531f * __builtin_amdgcn_logf(1.0f + u);
                        ht[r] = (a_ >= 0.0f ? rc : u * rc) * (zt[r] * g4[r]);
                    }
                    st4(Y + rv * ldy + co, hv);
                    st4(Y + rt * ldy + co, ht);
                }
                continue;
            }
        }
        if constexpr (!STATS) {
            if ((act & 0xff) == 3) {
                const f32x4 b4 = ld4(bias + co), g4 = ld4(ax.gate + (long)b * cstride + co), be4 = ld4(bb + co);
                f32x4 pg = (f32x4){0.f, 0.f, 0.f, 0.f}, pb = pg;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const long rv = (long)b * P + p0 + wn * 64 + ni * 16 + j, rt = rv + 32;
                    const f32x4 zv = ld4(ax.z + rv * ax.ldz + co), zt = ld4(ax.z + rt * ax.ldz + co);
                    const f32x4 dhv = acc[mi][ni], dht = acc[mi][ni + 2];
                    f32x4 dzv, dzt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float zb = zv[r] + b4[r];
                        const float a_ = zb * g4[r] + be4[r], ad = zt[r] * g4[r];
                        const float u = __builtin_amdgcn_exp2f(fabsf(a_) * -1.44269504088896341f);
                        const float rc = __builtin_amdgcn_rcpf(1.0f + u);
                        const float sg = a_ >= 0.0f ? rc : u * rc;
                        const float da = dhv[r] * sg + dht[r] * (sg * (1.0f - sg)) * ad;
                        const float dad = dht[r] * sg;
                        dzv[r] = da * g4[r];
                        dzt[r] = dad * g4[r];
                        pg[r] += da * zb + dad * zt[r];
                        pb[r] += da;
                    }
                    st4(Y + rv * ldy + co, dzv);
                    st4(Y + rt * ldy + co, dzt);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pg[r] = row_allreduce_add<16>(pg[r]);
                    pb[r] = row_allreduce_add<16>(pb[r]);
                }
                if (j == 0) {
                    const long entry = ((long)b * Pt + pt) * 2 + wn, nent = (long)gridDim.x / Mt * 2;
                    st4(ax.red + entry * cstride + co, pg);
                    st4(ax.red + (nent + entry) * cstride + co, pb);
                }
                continue;
            }
        }
        f32x4 add = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias) add += ld4(bias + co);
        if (bb) add += ld4(bb + co);
        f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f}, q4 = s4;
        f32x4 mx = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mn = (f32x4){INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int p = p0 + wn * 64 + ni * 16 + j;
            f32x4 v = acc[mi][ni] + add;
            if (STATS) {
                s4 += v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mx[r] = fmaxf(mx[r], v[r]);
                    mn[r] = fminf(mn[r], v[r]);
                }
            }
            if ((act & 0xff) == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = sigmoid_f(v[r]);
            }
            if (!STATS || Y) st4(Y + ((long)b * P + p) * ldy + co, v);
        }
        if (STATS) {
            // Moments of the wave's 64 points per channel, TWO passes over the accumulators: the mean first (four DPP steps inside
            // the 16-lane row), then the sum of squared deviations from it -- not sum / sum of squares, whose difference loses
            // (mean / sigma)^2 x 2^-24 of the variance in f32 (a checkpoint with large per-group means; advisor finding, round 2).
#pragma unroll
            for (int r = 0; r < 4; ++r) s4[r] = row_allreduce_add<16>(s4[r]) * (1.0f / 64.0f);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const f32x4 d = acc[mi][ni] + add - s4;
                q4 += d * d;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q4[r] = row_allreduce_add<16>(q4[r]);
                mx[r] = row_allreduce_max<16>(mx[r]);
                mn[r] = -row_allreduce_max<16>(-mn[r]);
            }
            if (j < 4) {
                const float ss = j == 0 ? s4[0] : (j == 1 ? s4[1] : (j == 2 ? s4[2] : s4[3]));
                const float qq = j == 0 ? q4[0] : (j == 1 ? q4[1] : (j == 2 ? q4[2] : q4[3]));
                const float m1 = j == 0 ? mx[0] : (j == 1 ? mx[1] : (j == 2 ? mx[2] : mx[3]));
                const float m0 = j == 0 ? mn[0] : (j == 1 ? mn[1] : (j == 2 ? mn[2] : mn[3]));
                sp[wn * X6_TM + wm * 128 + mi * 16 + 4 * g + j] = (f32x4){ss, qq, m1, m0};      // {mean, M2} of 64 points
            }
        }
    }
    if (STATS) {
        __syncthreads();
        const int co = mt * X6_TM + tid;
        if (co < Cout) {
            // the tile's two 64-point halves (Chan et al.): mean = (m0 + m1) / 2, M2 = M2_0 + M2_1 + (m1 - m0)^2 * 64 * 64 / 128
            const f32x4 a0 = sp[tid], a1 = sp[X6_TM + tid];
            const float dm = a1[0] - a0[0];
            part[((long)b * Pt + pt) * cstride + co] = (f32x4){0.5f * (a0[0] + a1[0]), a0[1] + a1[1] + dm * dm * 32.0f, fmaxf(a0[2], a1[2]), fminf(a0[3], a1[3])};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The REMAINDER of a layer below one 512-channel tile of conv1x1_x6w_kernel (1600 = 3 x 512 + 64): <= 64 output channels over all rows.
// On the kernel above that pass cost 0.79 ms of the 1600 -> 1600 head layer (4 % of the channels, 10 % of the time): half its waves hold
// no channels and only stage, every 32-k chunk goes through LDS behind two barriers.  With <= 64 channels the products are a twentieth of
// a full tile's and the pass is bound by READING the input once, so this kernel has no LDS and no barrier in its K loop: a wave owns 64
// points; lane (j, g) loads the 8 consecutive k (32 bytes) of ITS point straight into B-fragment shape, applies the producer's GroupNorm +
// ReLU, splits into the three planes in registers and multiplies; the weight fragments come from the same pre-swizzled pack (16 rows x
// 64 B = 1 KB per fragment, L1 / L2 hits).  Epilogue and statistics format as above (two 64-point halves per tile, combined through LDS).
// ---------------------------------------------------------------------------------------------------------------------------
template <bool FUSED, bool STATS>
__global__ __launch_bounds__(256) void conv1x1_x6tail_kernel(const unsigned char *__restrict__ wpk, const float *__restrict__ bias,
                                                             const float *__restrict__ bbias, const float *__restrict__ X, int ldx,
                                                             const float *__restrict__ in_scale, const float *__restrict__ in_shift, int in_relu,
                                                             int relu_from, float *__restrict__ Y, int ldy, int P, int Cin, int Cout, int Pt, int ntile,
                                                             f32x4 *__restrict__ part, int cstride)
{
    __shared__ f32x4 sp[2][2][64];                       // [tile of the workgroup][64-point half][channel]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int tl = wave >> 1, hf = wave & 1;
    const int gt0 = blockIdx.x * 2 + tl;
    const bool valid = gt0 < ntile;                      // wave-uniform; an invalid wave computes a copy of the last tile and stores nothing
    const int gt = valid ? gt0 : ntile - 1;
    const int b = gt / Pt, pt = gt - b * Pt;
    const int p0 = pt * X6_TP + hf * 64;
    const int nk = Cin / 32;
    const int nrt = (Cout + 15) >> 4;                    // live row tiles (<= 4)

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *xrow = X + ((long)b * P + p0 + j) * ldx + 8 * g;          // + 16 ni rows, + 32 kc
    const long xstep = 16L * ldx;
    const float *sc = FUSED ? in_scale + (long)b * Cin + 8 * g : nullptr;
    const float *sh = FUSED ? in_shift + (long)b * Cin + 8 * g : nullptr;
    const unsigned char *wl = wpk;                                           // channel tile 0 of the remainder's own pack
    f32x4 ra[4][2], rb[4][2];
    auto gload = [&](f32x4 (&r)[4][2], int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            r[ni][0] = ld4(xrow + ni * xstep + kc * 32);
            r[ni][1] = ld4(xrow + ni * xstep + kc * 32 + 4);
        }
    };
    auto chunk = [&](const f32x4 (&r)[4][2], int kc) __attribute__((always_inline)) {
        bf16x8 af[4][3];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            if (mi < nrt) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[mi][pl] = *(const bf16x8 *)(wl + (long)kc * X6_CHUNK + pl * X6_PA + x6_off(mi * 16 + j, g));
            }
        f32x4 s0, s1, t0, t1;
        float lo = -INFINITY;
        if (FUSED) {
            s0 = ld4(sc + kc * 32); s1 = ld4(sc + kc * 32 + 4);
            t0 = ld4(sh + kc * 32); t1 = ld4(sh + kc * 32 + 4);
            lo = (in_relu && kc * 32 + 8 * g >= relu_from) ? 0.f : -INFINITY;   // the ReLU switches on at a multiple of 8 channels
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            float xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = r[ni][q >> 2][q & 3];
                if (FUSED) v = fmaxf(v * (q < 4 ? s0[q & 3] : s1[q & 3]) + (q < 4 ? t0[q & 3] : t1[q & 3]), lo);
                xv[q] = v;
            }
            u32x4 pv[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned p1, p2, p3;
                x6_split_pair(xv[2 * q], xv[2 * q + 1], p1, p2, p3);
                pv[0][q] = p1;
                pv[1][q] = p2;
                pv[2][q] = p3;
            }
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, pv[0]), b1 = __builtin_bit_cast(bf16x8, pv[1]), b2 = __builtin_bit_cast(bf16x8, pv[2]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                if (mi < nrt) {          // smallest terms first, as in the kernel above
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][2], b0, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][1], b1, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][0], b2, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][1], b0, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][0], b1, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][0], b0, acc[mi][ni], 0, 0, 0);
                }
        }
    };
    gload(ra, 0);
    for (int kc = 0; kc < nk; kc += 2) {
        gload(rb, kc + 1 < nk ? kc + 1 : kc);
        chunk(ra, kc);
        if (kc + 1 < nk) {
            gload(ra, kc + 2 < nk ? kc + 2 : kc + 1);
            chunk(rb, kc + 1);
        }
    }
    // ---- epilogue: bias / per-entry bias, store, statistics of the wave's 64 points (two passes), halves combined below
    const float *bb = bbias ? bbias + (long)b * cstride : nullptr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int co = mi * 16 + 4 * g;
        if (co >= Cout) continue;
        f32x4 add = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias) add += ld4(bias + co);
        if (bb) add += ld4(bb + co);
        f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f}, q4 = s4;
        f32x4 mx = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mn = (f32x4){INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 v = acc[mi][ni] + add;
            if (STATS) {
                s4 += v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mx[r] = fmaxf(mx[r], v[r]);
                    mn[r] = fminf(mn[r], v[r]);
                }
            }
            if (valid && (!STATS || Y)) st4(Y + ((long)b * P + p0 + ni * 16 + j) * ldy + co, v);
        }
        if (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s4[r] = row_allreduce_add<16>(s4[r]) * (1.0f / 64.0f);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const f32x4 d = acc[mi][ni] + add - s4;
                q4 += d * d;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q4[r] = row_allreduce_add<16>(q4[r]);
                mx[r] = row_allreduce_max<16>(mx[r]);
                mn[r] = -row_allreduce_max<16>(-mn[r]);
            }
            if (j < 4) {
                const float ss = j == 0 ? s4[0] : (j == 1 ? s4[1] : (j == 2 ? s4[2] : s4[3]));
                const float qq = j == 0 ? q4[0] : (j == 1 ? q4[1] : (j == 2 ? q4[2] : q4[3]));
                const float m1 = j == 0 ? mx[0] : (j == 1 ? mx[1] : (j == 2 ? mx[2] : mx[3]));
                const float m0 = j == 0 ? mn[0] : (j == 1 ? mn[1] : (j == 2 ? mn[2] : mn[3]));
                sp[tl][hf][co + j] = (f32x4){ss, qq, m1, m0};      // {mean, M2, max, min} of 64 points
            }
        }
    }
    if (STATS) {
        __syncthreads();
        const int t2 = tid >> 7, c = tid & 127;
        const int gtw = blockIdx.x * 2 + t2;
        if (c < Cout && gtw < ntile) {
            // the tile's two 64-point halves (Chan et al.): mean = (m0 + m1) / 2, M2 = M2_0 + M2_1 + (m1 - m0)^2 * 64 * 64 / 128
            const f32x4 a0 = sp[t2][0][c], a1 = sp[t2][1][c];
            const float dm = a1[0] - a0[0];
            part[(long)gtw * cstride + c] = (f32x4){0.5f * (a0[0] + a1[0]), a0[1] + a1[1] + dm * dm * 32.0f, fmaxf(a0[2], a1[2]), fminf(a0[3], a1[3])};
        }
    }
}

// Folds the per-tile statistics of a STATS conv: one workgroup per (group, batch entry) sums the tiles' channel sums in f64
// (a thread's elements and the tree over threads are fixed by (P, C, G) only: a batch entry's result does not depend on the
// batch it sits in), then emits what caspr_gn_stats_f32 does: scale / shift per (b, c), optionally the max over points of the
// normalised output (from the channel's raw max or min, by the sign of its scale) and the moments.
__global__ __launch_bounds__(256) void conv_gn_finalize_kernel(const f32x4 *__restrict__ part, int PT, int P, int C, int G,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                               float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ pmax,
                                                               float *__restrict__ mean_out, float *__restrict__ rstd_out, int g0 = 0)
{
    __shared__ double s_sum[256], s_sq[256], s_m2[256];
    __shared__ float s_mx[256], s_mn[256];
    const int g = g0 + blockIdx.x, b = blockIdx.y, cpg = C / G, tid = threadIdx.x;      // g0: a launch may finalize a RANGE of groups
    const f32x4 *base = part + (long)b * PT * C + g * cpg;
    // every element of `part` = {mean, M2, max, min} of one channel over one 128-point tile.  ONE pass: a thread walks the tiles of
    // ITS channel (cpg <= 256: 256 / cpg tile lanes per channel), summing the tile means, their squares and the tiles' M2 in f64 --
    // group M2 = sum M2_t + 128 (sum m_t^2 - T mean^2), exact enough in f64 for any |mean| / sigma a float can hold -- and keeping
    // the channel's max / min.  (Two passes plus a per-channel serial walk for the max took 70-105 us on the 1600-channel layers.)
    const int lanes = cpg <= 256 ? 256 / cpg : 1;
    double s1 = 0.0, s2 = 0.0, sm = 0.0;
    for (int c0 = 0; c0 < cpg; c0 += 256) {                    // one round unless a group is wider than 256 channels
        const int cc = c0 + (cpg <= 256 ? tid % cpg : tid), pl = cpg <= 256 ? tid / cpg : 0;
        float m1 = -INFINITY, m0 = INFINITY;
        if (cc < cpg && pl < lanes) {
#pragma unroll 4
            for (int pt = pl; pt < PT; pt += lanes) {
                const f32x4 v = base[(long)pt * C + cc];
                s1 += (double)v[0];
                s2 += (double)v[0] * (double)v[0];
                sm += (double)v[1];
                m1 = fmaxf(m1, v[2]);
                m0 = fminf(m0, v[3]);
            }
        }
        if (pmax) {
            __syncthreads();
            s_mx[tid] = m1;
            s_mn[tid] = m0;
            __syncthreads();
            if (cc < cpg && pl == 0) {
                for (int l = 1; l < lanes; ++l) {
                    m1 = fmaxf(m1, s_mx[l * cpg + cc]);
                    m0 = fminf(m0, s_mn[l * cpg + cc]);
                }
                s_mx[tid] = m1;          // kept for the scale's sign below (this thread writes channel cc)
                s_mn[tid] = m0;
            }
        }
    }
    s_sum[tid] = s1;
    s_sq[tid] = s2;
    s_m2[tid] = sm;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) {
            s_sum[tid] += s_sum[tid + off];
            s_sq[tid] += s_sq[tid + off];
            s_m2[tid] += s_m2[tid + off];
        }
        __syncthreads();
    }
    const double T = (double)PT * cpg;
    const double mean = s_sum[0] / T;
    double m2 = s_m2[0] + 128.0 * (s_sq[0] - T * mean * mean);
    const double cnt = (double)P * cpg;
    double var = m2 / cnt;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    if (mean_out && tid == 0) {
        mean_out[b * G + g] = (float)mean;
        rstd_out[b * G + g] = (float)rstd;
    }
    if (cpg <= 256) {
        if (tid < cpg) {
            const int c = g * cpg + tid;
            const float sc = (float)((double)gamma[c] * rstd);
            const float sf = (float)((double)beta[c] - mean * (double)gamma[c] * rstd);
            scale[(long)b * C + c] = sc;
            shift[(long)b * C + c] = sf;
            if (pmax) pmax[(long)b * C + c] = (sc >= 0.f ? s_mx[tid] : s_mn[tid]) * sc + sf;
        }
    } else {
        for (int cc = tid; cc < cpg; cc += 256) {
            const int c = g * cpg + cc;
            const float sc = (float)((double)gamma[c] * rstd);
            const float sf = (float)((double)beta[c] - mean * (double)gamma[c] * rstd);
            scale[(long)b * C + c] = sc;
            shift[(long)b * C + c] = sf;
            if (pmax) {
                float m1 = -INFINITY, m0 = INFINITY;
                for (int pt = 0; pt < PT; ++pt) {
                    const f32x4 v = base[(long)pt * C + cc];
                    m1 = fmaxf(m1, v[2]);
                    m0 = fminf(m0, v[3]);
                }
                pmax[(long)b * C + c] = (sc >= 0.f ? m1 : m0) * sc + sf;
            }
        }
    }
}

extern "C" long caspr_bf16x3_packed_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || Cin % 32) return 0;
    return (long)ceil_div(Cout, X6_TM) * (Cin / 32) * X6_CHUNK;
}

extern "C" int caspr_pack_weight_bf16x3(const float *w, int ldw, int Cout, int col0, int ncols, void *packed, void *stream)
{
    CASPR_REQUIRE(w && packed && Cout > 0 && ncols > 0 && col0 >= 0 && ldw >= col0 + ncols, "pack_weight_bf16x3: bad arguments");
    CASPR_REQUIRE(ncols % 32 == 0, "pack_weight_bf16x3: the input width %d must be a multiple of 32", ncols);
    CASPR_REQUIRE(((uintptr_t)packed % 16) == 0, "pack_weight_bf16x3: packed must be 16-byte aligned");
    const long total = (long)ceil_div(Cout, X6_TM) * (ncols / 32) * 1024;
    pack_weight_bf16x3_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, ldw, Cout, col0, ncols, (unsigned char *)packed, total);
    CASPR_CHECK_LAUNCH("pack_weight_bf16x3");
    return CASPR_OK;
}

#ifdef CASPR_DEBUG_HOOKS
static unsigned long long *g_conv_x6_trace = nullptr;
extern "C" void caspr_debug_set_conv_x6_trace(unsigned long long *dev_buf) { g_conv_x6_trace = dev_buf; }   // debug build only: >= 160 u64
#endif

static int conv_x6_launch(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx, const float *in_scale,
                          const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B, int P, int Cin, int Cout,
                          int act, f32x4 *part, void *stream, int cstride = 0, X6Act ax = X6Act{nullptr, nullptr, 0, nullptr})
{
    if (cstride == 0) cstride = Cout;
    CASPR_REQUIRE(wpk && X && (Y || part) && B > 0 && P > 0 && Cin > 0 && Cout > 0, "conv1x1_bf16x6: bad arguments");
    CASPR_REQUIRE(Cin % 32 == 0 && Cout % 4 == 0 && P % X6_TP == 0,
                  "conv1x1_bf16x6: needs Cin %% 32 == 0, Cout %% 4 == 0 and P %% 128 == 0 (Cin=%d Cout=%d P=%d); use caspr_conv1x1_f32", Cin, Cout, P);
    CASPR_REQUIRE(ldx % 4 == 0 && ldx >= Cin, "conv1x1_bf16x6: ldx=%d must be a multiple of 4 and >= Cin=%d", ldx, Cin);
    CASPR_REQUIRE(!Y || (ldy % 4 == 0 && ldy >= Cout), "conv1x1_bf16x6: ldy=%d must be a multiple of 4 and >= Cout=%d", ldy, Cout);
    CASPR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv1x1_bf16x6: in_scale/in_shift must be given together");
    CASPR_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)wpk % 16) == 0, "conv1x1_bf16x6: pointers must be 16-byte aligned");
    CASPR_REQUIRE((bias == nullptr || ((uintptr_t)bias % 16) == 0) && (bbias == nullptr || ((uintptr_t)bbias % 16) == 0),
                  "conv1x1_bf16x6: bias pointers must be 16-byte aligned");
    CASPR_REQUIRE(in_relu_from >= 0 && in_relu_from % 8 == 0, "conv1x1_bf16x6: in_relu_from=%d must be a non-negative multiple of 8", in_relu_from);
    const int Mt = ceil_div(Cout, X6_TM), Pt = P / X6_TP;
    const long nblk = (long)Mt * Pt * B;
    CASPR_REQUIRE(nblk < (1L << 31), "conv1x1_bf16x6: too many tiles (%ld)", nblk);
    unsigned long long *trace = nullptr;
    CASPR_IF_DEBUG(trace = g_conv_x6_trace;)
    static CasprLdsOptIn optin[4];
    const void *kern[4] = {(const void *)conv1x1_bf16x6_kernel<false, false>, (const void *)conv1x1_bf16x6_kernel<true, false>,
                           (const void *)conv1x1_bf16x6_kernel<false, true>, (const void *)conv1x1_bf16x6_kernel<true, true>};
    const int which = (in_scale ? 1 : 0) + (part ? 2 : 0);
    const hipError_t e1 = caspr_lds_opt_in(optin[which], kern[which], X6_LDS);
    if (e1 != hipSuccess) {
        caspr_set_error("conv1x1_bf16x6: hipFuncSetAttribute failed: %s", hipGetErrorString(e1));
        return CASPR_ELAUNCH;
    }
#define X6_LAUNCH(F, S)                                                                                                          \
    conv1x1_bf16x6_kernel<F, S><<<dim3((unsigned)nblk), dim3(256), X6_LDS, (hipStream_t)stream>>>(                                \
        (const unsigned char *)wpk, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, P, Cin, Cout, act, Mt, Pt, part, cstride, trace, ax)
    if (which == 0) X6_LAUNCH(false, false);
    else if (which == 1) X6_LAUNCH(true, false);
    else if (which == 2) X6_LAUNCH(false, true);
    else X6_LAUNCH(true, true);
#undef X6_LAUNCH
    CASPR_CHECK_LAUNCH("conv1x1_bf16x6");
    return CASPR_OK;
}

extern "C" int caspr_conv1x1_cnf_act_bf16x6_f32(const void *wpk, const float *b, const float *gate, const float *beta, const float *X, int ldx,
                                                float *Z, int ldz, float *H, int ldh, int frames, int n, int Cin, int Cout, void *stream)
{
    CASPR_REQUIRE(b && gate && beta && Z && H && frames > 0 && n > 0 && n % 64 == 0, "conv1x1_cnf_act: bad arguments (n=%d must be a multiple of 64)", n);
    CASPR_REQUIRE(ldz % 4 == 0 && ldz >= Cout && ((uintptr_t)Z % 16) == 0 && ((uintptr_t)gate % 16) == 0 && ((uintptr_t)beta % 16) == 0 &&
                      ((uintptr_t)b % 16) == 0,
                  "conv1x1_cnf_act: Z / gate / beta / b must be 16-byte aligned, ldz a multiple of 4 and >= Cout");
    return conv_x6_launch(wpk, b, beta, X, ldx, nullptr, nullptr, 0, 0, H, ldh, frames, 2 * n, Cin, Cout, 2, nullptr, stream, 0, X6Act{gate, Z, ldz, nullptr});
}

// sums the per-(tile, wave half) partials of act mode 3 in index order: one thread per (frame, channel)
__global__ void cnf_act_bwd_reduce_kernel(const float *__restrict__ red, int frames, int per, int C, float *__restrict__ dgate,
                                          float *__restrict__ dbeta)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)frames * C) return;
    const long f = i / C;
    const int c = (int)(i - f * C);
    const long nent = (long)frames * per;
    float sg = 0.f, sb = 0.f;
    for (int e = 0; e < per; ++e) {
        sg += red[(f * per + e) * C + c];
        sb += red[(nent + f * per + e) * C + c];
    }
    dgate[i] = sg;
    dbeta[i] = sb;
}

extern "C" long caspr_conv1x1_cnf_act_bwd_ws_bytes(int frames, int n, int Cout) { return (long)frames * (2 * n / X6_TP) * 2 * 2 * Cout * 4 + 256; }

extern "C" int caspr_conv1x1_cnf_act_bwd_bf16x6_f32(const void *wpk, const float *X, int ldx, const float *Z, int ldz, const float *b,
                                                    const float *gate, const float *beta, float *dZ, int lddz, float *dgate, float *dbeta,
                                                    void *ws, long ws_bytes, int frames, int n, int Cin, int Cout, void *stream)
{
    CASPR_REQUIRE(Z && b && gate && beta && dZ && dgate && dbeta && ws && frames > 0 && n > 0 && n % 64 == 0,
                  "conv1x1_cnf_act_bwd: bad arguments (n=%d must be a multiple of 64)", n);
    CASPR_REQUIRE(ldz % 4 == 0 && ldz >= Cout && ((uintptr_t)Z % 16) == 0 && ((uintptr_t)gate % 16) == 0 && ((uintptr_t)beta % 16) == 0 &&
                      ((uintptr_t)b % 16) == 0 && ((uintptr_t)ws % 16) == 0,
                  "conv1x1_cnf_act_bwd: Z / gate / beta / b / ws must be 16-byte aligned, ldz a multiple of 4 and >= Cout");
    CASPR_REQUIRE(ws_bytes >= caspr_conv1x1_cnf_act_bwd_ws_bytes(frames, n, Cout), "conv1x1_cnf_act_bwd: workspace too small");
    const int rc = conv_x6_launch(wpk, b, beta, X, ldx, nullptr, nullptr, 0, 0, dZ, lddz, frames, 2 * n, Cin, Cout, 3, nullptr, stream, 0,
                                  X6Act{gate, const_cast<float *>(Z), ldz, (float *)ws});
    if (rc != CASPR_OK) return rc;
    const long tot = (long)frames * Cout;
    cnf_act_bwd_reduce_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>((const float *)ws, frames, 2 * n / X6_TP * 2, Cout,
                                                                                                         dgate, dbeta);
    CASPR_CHECK_LAUNCH("conv1x1_cnf_act_bwd");
    return CASPR_OK;
}

extern "C" int caspr_conv1x1_bf16x6_f32(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx,
                                        const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y,
                                        int ldy, int B, int P, int Cin, int Cout, int act, void *stream)
{
    CASPR_REQUIRE(Y, "conv1x1_bf16x6: Y is NULL");
    return conv_x6_launch(wpk, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, Cout, act, nullptr, stream);
}

extern "C" long caspr_conv_gn_ws_bytes(int B, int P, int Cout)
{
    if (B <= 0 || P <= 0 || Cout <= 0 || P % X6_TP) return 0;
    return (long)B * (P / X6_TP) * Cout * 16;
}

// pool: consecutive batch entries whose GroupNorm statistics are taken TOGETHER (1 = per entry).  The conv sees B entries -- each
// with its own in_scale / in_shift / bbias row -- the statistics B / pool groups of pool x P points: the head's first layer reads
// per-FRAME normalised PointNet++ features but normalises its own output per SEQUENCE (tpointnet2.py:96-99).  The per-tile
// partials of consecutive entries are contiguous, so the pooled finalize is the plain one over (B / pool, pool x P / 128 tiles).
static int conv_gn_x6_impl(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx, const float *in_scale,
                           const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B, int P, int Cin, int Cout, int G,
                           int pool, const float *gamma, const float *beta, float eps, float *scale, float *shift, float *pmax, float *mean,
                           float *rstd, void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(gamma && beta && scale && shift && ws && G > 0, "conv1x1_gn_bf16x6: bad arguments");
    CASPR_REQUIRE(Cout % G == 0, "conv1x1_gn_bf16x6: Cout=%d is not a multiple of the %d groups", Cout, G);
    CASPR_REQUIRE((mean == nullptr) == (rstd == nullptr), "conv1x1_gn_bf16x6: mean / rstd must be given together");
    CASPR_REQUIRE(pool >= 1 && B % pool == 0, "conv1x1_gn_bf16x6: pool=%d must divide B=%d", pool, B);
    CASPR_REQUIRE(B <= 65535 * (long)pool, "conv1x1_gn_bf16x6: B too large");
    CASPR_REQUIRE(P % X6_TP == 0 && ws_bytes >= caspr_conv_gn_ws_bytes(B, P, Cout) && ((uintptr_t)ws % 16) == 0,
                  "conv1x1_gn_bf16x6: workspace too small or misaligned (%ld < %ld)", ws_bytes, caspr_conv_gn_ws_bytes(B, P, Cout));
    const int rc = conv_x6_launch(wpk, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, Cout, 0,
                                  (f32x4 *)ws, stream);
    if (rc != CASPR_OK) return rc;
    conv_gn_finalize_kernel<<<dim3(G, B / pool), dim3(256), 0, (hipStream_t)stream>>>((const f32x4 *)ws, pool * (P / X6_TP), pool * P, Cout, G, gamma,
                                                                                      beta, eps, scale, shift, pmax, mean, rstd);
    CASPR_CHECK_LAUNCH("conv1x1_gn_bf16x6");
    return CASPR_OK;
}

extern "C" int caspr_conv1x1_gn_bf16x6_f32(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx,
                                           const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y,
                                           int ldy, int B, int P, int Cin, int Cout, int G, const float *gamma, const float *beta,
                                           float eps, float *scale, float *shift, float *pmax, float *mean, float *rstd, void *ws,
                                           long ws_bytes, void *stream)
{
    return conv_gn_x6_impl(wpk, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, Cout, G, 1, gamma, beta, eps, scale,
                           shift, pmax, mean, rstd, ws, ws_bytes, stream);
}

// ... with the statistics pooled over `pool` consecutive batch entries: scale / shift / pmax (B / pool, Cout), mean / rstd (B / pool, G)
extern "C" int caspr_conv1x1_gn_pooled_bf16x6_f32(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx,
                                                  const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y,
                                                  int ldy, int B, int P, int Cin, int Cout, int G, int pool, const float *gamma,
                                                  const float *beta, float eps, float *scale, float *shift, float *pmax, float *mean,
                                                  float *rstd, void *ws, long ws_bytes, void *stream)
{
    return conv_gn_x6_impl(wpk, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, Cout, G, pool, gamma, beta, eps,
                           scale, shift, pmax, mean, rstd, ws, ws_bytes, stream);
}

// remainder (<= 64 channels) of a layer whose first channels ran on the 512-channel kernel: conv1x1_x6tail_kernel; `part` / `bbias` are
// already offset to the remainder's first channel, cstride = the layer's full channel count
static int conv_x6tail_launch(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx, const float *in_scale,
                              const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B, int P, int Cin, int Cout, f32x4 *part,
                              void *stream, int cstride)
{
    CASPR_REQUIRE(wpk && X && (Y || part) && Cout > 0 && Cout <= 64 && Cout % 4 == 0 && Cin % 32 == 0 && P % X6_TP == 0, "conv1x1_x6tail: bad arguments");
    const int Pt = P / X6_TP;
    const long ntile = (long)B * Pt;
    const unsigned grid = (unsigned)((ntile + 1) / 2);
#define X6T_GO(F, S)                                                                                                                          \
    conv1x1_x6tail_kernel<F, S><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>((const unsigned char *)wpk, bias, bbias, X, ldx, in_scale, in_shift, \
                                                                                  in_relu, in_relu_from, Y, ldy, P, Cin, Cout, Pt, (int)ntile, part, cstride)
    if (in_scale && part) X6T_GO(true, true);
    else if (in_scale) X6T_GO(true, false);
    else if (part) X6T_GO(false, true);
    else X6T_GO(false, false);
#undef X6T_GO
    CASPR_CHECK_LAUNCH("conv1x1_x6tail");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The large layers: gemm_bf16x6w.hip's 128-point x 512-channel kernel on the first Cout - Cout % 512 channels, the kernel above
// on the remainder (1600 = 3 x 512 + 64), both writing into one output / one statistics array; G > 0 adds the GroupNorm
// statistics of the output (as caspr_conv1x1_gn_bf16x6_f32).
// ---------------------------------------------------------------------------------------------------------------------------
int caspr_conv_x6w_launch(const void *wpk, const float *bias, const float *bbias, int bb_stride, const float *X, int ldx, const float *in_scale,
                          const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B, int P, int Cin, int mt_begin, int mt_end,
                          void *part, int part_stride, int reserve_cus, hipStream_t stream) __attribute__((visibility("hidden")));

static int conv_x6w_impl(const void *wpk_main, const void *wpk_tail, const float *bias, const float *bbias, const float *X, int ldx,
                         const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B,
                         int P, int Cin, int Cout, int G, int pool, const float *gamma, const float *beta, float eps, float *scale,
                         float *shift, float *pmax, float *mean, float *rstd, void *ws, long ws_bytes, void *stream)
{
    const int Cmain = Cout - Cout % 512, Ctail = Cout - Cmain;
    CASPR_REQUIRE(wpk_main && X && B > 0 && P > 0 && Cmain >= 512 && (Ctail == 0 || wpk_tail), "conv1x1_x6w: bad arguments (Cout=%d needs >= 512 channels%s)", Cout,
                  Ctail ? " and the bf16x3 pack of the remainder" : "");
    CASPR_REQUIRE(Cin % 32 == 0 && Cin >= 64 && Cout % 4 == 0 && P % 128 == 0, "conv1x1_x6w: needs Cin %% 32 == 0, Cout %% 4 == 0 and P %% 128 == 0 (Cin=%d Cout=%d P=%d)", Cin, Cout, P);
    CASPR_REQUIRE(ldx % 4 == 0 && ldx >= Cin && (!Y || (ldy % 4 == 0 && ldy >= Cout)), "conv1x1_x6w: row strides must be multiples of 4 and cover the channels");
    CASPR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv1x1_x6w: in_scale/in_shift must be given together");
    CASPR_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)wpk_main % 16) == 0 && ((uintptr_t)wpk_tail % 16) == 0 &&
                  ((uintptr_t)bias % 16) == 0 && ((uintptr_t)bbias % 16) == 0, "conv1x1_x6w: pointers must be 16-byte aligned");
    CASPR_REQUIRE(in_relu_from >= 0 && in_relu_from % 8 == 0, "conv1x1_x6w: in_relu_from=%d must be a non-negative multiple of 8", in_relu_from);
    CASPR_REQUIRE((long)B * (P / 128) * (Cmain / 512) < (1L << 31) && B <= 65535 * (long)(pool > 0 ? pool : 1), "conv1x1_x6w: too many tiles");
    f32x4 *part = nullptr;
    if (G > 0) {
        CASPR_REQUIRE(gamma && beta && scale && shift && ws && Cout % G == 0 && (mean == nullptr) == (rstd == nullptr), "conv1x1_x6w: bad GroupNorm arguments");
        CASPR_REQUIRE(pool >= 1 && B % pool == 0, "conv1x1_x6w: pool=%d must divide B=%d", pool, B);
        CASPR_REQUIRE(ws_bytes >= caspr_conv_gn_ws_bytes(B, P, Cout) && ((uintptr_t)ws % 16) == 0, "conv1x1_x6w: workspace too small or misaligned");
        part = (f32x4 *)ws;
    } else {
        CASPR_REQUIRE(Y, "conv1x1_x6w: Y is NULL");
    }
    int rc = caspr_conv_x6w_launch(wpk_main, bias, bbias, Cout, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, 0, Cmain / 512, part,
                                   Cout, 0, (hipStream_t)stream);
    if (rc != CASPR_OK) return rc;
    CASPR_CHECK_LAUNCH("conv1x1_x6w");
    if (Ctail && Ctail <= 64) {
        rc = conv_x6tail_launch(wpk_tail, bias ? bias + Cmain : nullptr, bbias ? bbias + Cmain : nullptr, X, ldx, in_scale, in_shift, in_relu, in_relu_from,
                                Y ? Y + Cmain : nullptr, ldy, B, P, Cin, Ctail, part ? part + Cmain : nullptr, stream, Cout);
        if (rc != CASPR_OK) return rc;
    } else if (Ctail) {
        rc = conv_x6_launch(wpk_tail, bias ? bias + Cmain : nullptr, bbias ? bbias + Cmain : nullptr, X, ldx, in_scale, in_shift, in_relu, in_relu_from,
                            Y ? Y + Cmain : nullptr, ldy, B, P, Cin, Ctail, 0, part ? part + Cmain : nullptr, stream, Cout);
        if (rc != CASPR_OK) return rc;
    }
    if (G > 0) {
        // pooled statistics (pool consecutive batch entries together): the per-tile partials of consecutive entries are contiguous
        conv_gn_finalize_kernel<<<dim3(G, B / pool), dim3(256), 0, (hipStream_t)stream>>>((const f32x4 *)ws, pool * (P / X6_TP), pool * P, Cout, G, gamma, beta,
                                                                                          eps, scale, shift, pmax, mean, rstd);
        CASPR_CHECK_LAUNCH("conv1x1_x6w (GroupNorm statistics)");
    }
    return CASPR_OK;
}

extern "C" int caspr_conv1x1_x6w_f32(const void *wpk_main, const void *wpk_tail, const float *bias, const float *bbias, const float *X, int ldx,
                                     const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B,
                                     int P, int Cin, int Cout, int G, const float *gamma, const float *beta, float eps, float *scale,
                                     float *shift, float *pmax, float *mean, float *rstd, void *ws, long ws_bytes, void *stream)
{
    return conv_x6w_impl(wpk_main, wpk_tail, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, Cout, G, 1, gamma, beta,
                         eps, scale, shift, pmax, mean, rstd, ws, ws_bytes, stream);
}

// ... with the GroupNorm statistics pooled over `pool` consecutive batch entries (as caspr_conv1x1_gn_pooled_bf16x6_f32; G > 0)
extern "C" int caspr_conv1x1_x6w_pooled_f32(const void *wpk_main, const void *wpk_tail, const float *bias, const float *bbias, const float *X, int ldx,
                                            const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B,
                                            int P, int Cin, int Cout, int G, int pool, const float *gamma, const float *beta, float eps,
                                            float *scale, float *shift, float *pmax, float *mean, float *rstd, void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(G > 0, "conv1x1_x6w_pooled: needs the GroupNorm statistics (G > 0)");
    return conv_x6w_impl(wpk_main, wpk_tail, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, Cout, G, pool, gamma,
                         beta, eps, scale, shift, pmax, mean, rstd, ws, ws_bytes, stream);
}

// The same layer in PIECES, for a caller that needs some of the output's GroupNorm statistics before the whole layer is done (the
// encoder's head: the latent ODE starts from the max over points of the first 64 normalised channels of the 1600 -> 1600 layer,
// tpointnet2.py:100,111 + caspr.py:169 -- group 0 of 16 -- which are complete after the FIRST channel tile):
//   caspr_conv1x1_x6w_part_f32      channel tiles mt_begin .. mt_end - 1 of the 512-channel kernel (+ the < 512-channel remainder when
//                                   with_tail), output and per-tile statistics partials into ws; no finalize.  reserve_cus compute
//                                   units are left to a kernel of another stream that runs beside it;
//   caspr_conv_gn_finalize_f32      scale / shift / pmax / mean / rstd of groups g_begin .. g_end - 1 from the partials in ws.
// Pieces + finalize over all groups == caspr_conv1x1_x6w_f32 / _pooled_f32, bit for bit (tiles and groups are independent).
extern "C" int caspr_conv1x1_x6w_part_f32(const void *wpk_main, const void *wpk_tail, const float *bias, const float *bbias, const float *X, int ldx,
                                          const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B,
                                          int P, int Cin, int Cout, int mt_begin, int mt_end, int with_tail, int reserve_cus, void *ws, long ws_bytes,
                                          void *stream)
{
    const int Cmain = Cout - Cout % 512, Ctail = Cout - Cmain;
    CASPR_REQUIRE(wpk_main && X && B > 0 && P > 0 && Cmain >= 512 && (Ctail == 0 || !with_tail || wpk_tail), "conv1x1_x6w_part: bad arguments");
    CASPR_REQUIRE(mt_begin >= 0 && mt_begin <= mt_end && mt_end <= Cmain / 512, "conv1x1_x6w_part: channel tiles %d..%d of %d", mt_begin, mt_end, Cmain / 512);
    CASPR_REQUIRE(Cin % 32 == 0 && Cin >= 64 && Cout % 4 == 0 && P % 128 == 0, "conv1x1_x6w_part: needs Cin %% 32 == 0, Cout %% 4 == 0 and P %% 128 == 0");
    CASPR_REQUIRE(ldx % 4 == 0 && ldx >= Cin && (!Y || (ldy % 4 == 0 && ldy >= Cout)), "conv1x1_x6w_part: row strides must be multiples of 4 and cover the channels");
    CASPR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv1x1_x6w_part: in_scale/in_shift must be given together");
    CASPR_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)wpk_main % 16) == 0 && ((uintptr_t)wpk_tail % 16) == 0 &&
                  ((uintptr_t)bias % 16) == 0 && ((uintptr_t)bbias % 16) == 0 && ((uintptr_t)ws % 16) == 0, "conv1x1_x6w_part: pointers must be 16-byte aligned");
    CASPR_REQUIRE(in_relu_from >= 0 && in_relu_from % 8 == 0, "conv1x1_x6w_part: in_relu_from must be a non-negative multiple of 8");
    CASPR_REQUIRE((long)B * (P / 128) * (Cmain / 512) < (1L << 31), "conv1x1_x6w_part: too many tiles");
    CASPR_REQUIRE(ws && ws_bytes >= caspr_conv_gn_ws_bytes(B, P, Cout), "conv1x1_x6w_part: workspace too small");
    f32x4 *part = (f32x4 *)ws;
    int rc = CASPR_OK;
    if (mt_end > mt_begin) {
        rc = caspr_conv_x6w_launch(wpk_main, bias, bbias, Cout, X, ldx, in_scale, in_shift, in_relu, in_relu_from, Y, ldy, B, P, Cin, mt_begin, mt_end, part,
                                   Cout, reserve_cus, (hipStream_t)stream);
        if (rc != CASPR_OK) return rc;
        CASPR_CHECK_LAUNCH("conv1x1_x6w_part");
    }
    if (Ctail && with_tail && Ctail <= 64)
        rc = conv_x6tail_launch(wpk_tail, bias ? bias + Cmain : nullptr, bbias ? bbias + Cmain : nullptr, X, ldx, in_scale, in_shift, in_relu, in_relu_from,
                                Y ? Y + Cmain : nullptr, ldy, B, P, Cin, Ctail, part + Cmain, stream, Cout);
    else if (Ctail && with_tail)
        rc = conv_x6_launch(wpk_tail, bias ? bias + Cmain : nullptr, bbias ? bbias + Cmain : nullptr, X, ldx, in_scale, in_shift, in_relu, in_relu_from,
                            Y ? Y + Cmain : nullptr, ldy, B, P, Cin, Ctail, 0, part + Cmain, stream, Cout);
    return rc;
}

extern "C" int caspr_conv_gn_finalize_f32(const void *ws, long ws_bytes, int B, int P, int Cout, int G, int g_begin, int g_end, int pool,
                                          const float *gamma, const float *beta, float eps, float *scale, float *shift, float *pmax, float *mean,
                                          float *rstd, void *stream)
{
    CASPR_REQUIRE(ws && gamma && beta && scale && shift && G > 0 && Cout % G == 0 && (mean == nullptr) == (rstd == nullptr), "conv_gn_finalize: bad arguments");
    CASPR_REQUIRE(0 <= g_begin && g_begin < g_end && g_end <= G, "conv_gn_finalize: groups %d..%d of %d", g_begin, g_end, G);
    CASPR_REQUIRE(pool >= 1 && B % pool == 0 && B / pool <= 65535 && P % X6_TP == 0, "conv_gn_finalize: pool=%d must divide B=%d; P %% 128 == 0", pool, B);
    CASPR_REQUIRE(ws_bytes >= caspr_conv_gn_ws_bytes(B, P, Cout) && ((uintptr_t)ws % 16) == 0, "conv_gn_finalize: workspace too small or misaligned");
    conv_gn_finalize_kernel<<<dim3(g_end - g_begin, B / pool), dim3(256), 0, (hipStream_t)stream>>>((const f32x4 *)ws, pool * (P / X6_TP), pool * P, Cout, G, gamma,
                                                                                                     beta, eps, scale, shift, pmax, mean, rstd, g_begin);
    CASPR_CHECK_LAUNCH("conv_gn_finalize");
    return CASPR_OK;
}
