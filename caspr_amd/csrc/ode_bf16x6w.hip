// ode_bf16x6w.hip -- the point-CNF SAMPLING solve (cnf.py:70-128 with logpx = None; odefunc.py:98-105,
// diffeq_layers.py:83-90) on the bf16 matrix pipe, 128 points per workgroup: the default kernel of CaSPR.reconstruct's
// dominant stage since round 3 (the 64-point kernel of ode_bf16x6.hip keeps the divergence variant).
//
// Why another geometry.  The 64-point kernel re-streams the 3 MB of split hidden-layer weights L2 -> LDS once per 64 points
// and stage (491 GB of L2 -> LDS traffic per cfg-2 launch, 12 LDS-DMA instructions and 48 fragment reads per wave per 1536
// matrix-pipe cycles) and sits at 0.53 of the bf16x6 ceiling with the pipe busy 64 % of the cycles.  Here a wave owns 32
// points and ALL 512 hidden units: v_mfma_f32_32x32x16_bf16 (32 matrix-pipe cycles each, twice the work per instruction and
// per operand fetch of the 16x16x32 form) on 32-row x 32-point tiles, so every weight fragment fetched from LDS feeds twice
// the products, the weight stream per point is halved, and a wave issues half as many instructions per product.
//
//  * workgroup = 4 waves = 128 points of one frame; lane (j = lane & 31, h = lane >> 5) is point 32 wave + j;
//  * layer 1 (512 x 512): all 16 row tiles accumulate at once, k-chunk-major; the input-layer values (3 -> 512, gate folded
//    into the weights per stage) of chunk kc + 1 are produced inside chunk kc's MFMA shadow;
//  * layer 2: four passes over 128 output rows (acc2, 64 registers).  Pass 0 applies layer 1's gate / bias / softplus IN
//    PLACE while it splits k-step t + 1 into bf16 planes; passes 1-3 only split the stored activations again.  Each pass
//    ends with its rows' epilogue and their share of the 512 -> 3 output layer;
//  * a D fragment of the 32x32 form holds rows (r & 3) + 8 (r >> 2) + 4 h of its tile in register r: registers 8u .. 8u + 7
//    ARE the B fragment of k-step 2 T + u of the next layer once the weight pack lists k in that order -- the hidden
//    activation never leaves the registers of its lane;
//  * THE ACCUMULATOR FILE IS MANAGED BY HAND: layer 1's 256 accumulators / activations live at fixed addresses a[16 T + r]
//    (row tile T, register r) and are touched only by this file's inline-asm statements (v_mfma with a literal a[..] range,
//    v_accvgpr_read / _write).  hipcc keeps the VGPR half: layer 2's accumulators, the fragments, the producers.  (Left to
//    hipcc, an element update of a 16-register accumulator tuple copies the whole tuple between the two halves and the
//    activations spill to scratch -- whose reloads drain the LDS-DMA queue with vmcnt(0).)  Every statement that writes the
//    accumulator file names all of a0..a255 as clobbered, so hipcc parks nothing there; tests/test_host_cpu.py checks the
//    disassembly for compiler-generated accumulator moves and for scratch;
//  * every MFMA has its own scheduling slot (sched_barrier after each): the MFMA, at most one fragment read, and a hand-picked
//    micro-step of the producers -- two row tiles alternate, so a dependent MFMA is two issue slots behind its predecessor;
//  * weights: 24 KB pieces [k-step 2][row tile 4][plane 3][fragment image], a fragment = 1 KB in lane order (conflict-free
//    ds_read_b128 by construction), through a four-deep LDS ring by LDS-DMA, three to four pieces ahead behind COUNTED vmcnt
//    waits, one raw s_barrier per piece of 48 MFMAs (1536 matrix-pipe cycles); the stream never stops: the last region of a
//    stage already reads the first fragments of the next stage;
//  * layer 1 is a runtime loop over chunk pairs, passes 1-3 a runtime loop over the pass.
#include "x6w_common.h"

#define XW_RING 4
#define XW_TAB (XW_RING * XW_PIECE)
// tables (floats): hb0[512] w0g[3][512] g1[512] hb1[512] g2[512] hb2[512] w3[3][512] w0[512][3] g3[8]
#define XW_TAB_FLOATS (14 * XC_H + 8)
#define XW_LDS (XW_TAB + XW_TAB_FLOATS * 4)

// XW_EXP (debug flavours only, build.py CASPR_XW_EXP): timing experiments, results are WRONG with any bit set.
//   1 no piece barriers / waits   2 no weight DMA   4 no producers at all   8 no fragment reads   16 no stamps
//   32 no layer-1 producers   64 no layer-2 producers   128 producers without transcendentals   256 producers without
//   accumulator-file moves   512 no layer-2 epilogues   1024 no table build / accumulator zeroing
#ifndef XW_EXP
#define XW_EXP 0
#endif
#if defined(CASPR_DEBUG_HOOKS) && !(XW_EXP & 16)
#define XW_STAMP(i) if (a.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && step == 0 && stage == 1) a.trace[i] = __builtin_amdgcn_s_memtime();
#else
#define XW_STAMP(i)
#endif
#if defined(CASPR_DEBUG_HOOKS) && !(XW_EXP & 16)
#define XW_STAMPC(c, i) if (a.trace && (c) && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && step == 0 && stage == 1) a.trace[i] = __builtin_amdgcn_s_memtime();
#else
#define XW_STAMPC(c, i)
#endif
__global__ __launch_bounds__(256, 1) void cnf_rk4_x6w_kernel(CnfX6Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *wbuf = lds;                      // [XW_RING][XW_PIECE]
    float *s_hb0 = (float *)(lds + XW_TAB);         // [512]     layer 0: bias * gate + hyper bias
    float *s_w0g = s_hb0 + XC_H;                    // [3][512]  layer 0: weight column d * gate
    float *s_g1 = s_w0g + 3 * XC_H;                 // [512]     sigmoid gate of hidden layer 1
    float *s_hb1 = s_g1 + XC_H;
    float *s_g2 = s_hb1 + XC_H;
    float *s_hb2 = s_g2 + XC_H;
    float *s_w3 = s_hb2 + XC_H;                     // [3][512]  output layer
    float *s_w0 = s_w3 + 3 * XC_H;                  // [512][3]  input layer (raw)
    float *s_g3 = s_w0 + 3 * XC_H;                  // [8]: gate3[3], pad, hb3[3]

    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = blockIdx.y;
    const float *hy = a.hyper + (long)bt * a.ldh;
    constexpr int BOFF = 3 * XC_H + 3;

    for (int i = tid; i < 3 * XC_H; i += 256) {
        s_w0[i] = a.w0[i];
        s_w3[i] = a.w3[i];
    }

    // state of the lane's point, all three components (both halves h of a column hold the same copy)
    const int col = blockIdx.x * XW_PTS + 32 * wave + (lane0 & 31);
    const bool cvalid = col < a.n;
    const int ccol = cvalid ? col : a.n - 1;
    float y[3], kacc[3] = {0.f, 0.f, 0.f}, kprev[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = a.y_in[((long)bt * a.n + ccol) * 3 + d];
        if (a.mbn_in) {
            const float w = a.mbn_in[d], bb = a.mbn_in[3 + d], mean = a.mbn_in[6 + d], var = a.mbn_in[9 + d];
            if (a.reverse) v = (v - bb) * expf(-w) * expf(0.5f * logf(var + 1e-4f)) + mean;   // normalization.py:92-94
            else v = (v - mean) * expf(-0.5f * logf(var + 1e-4f)) * expf(w) + bb;             // normalization.py:70-74
        }
        y[d] = v;
    }

    // per-thread constants of the stage tables: units tid and tid + 256 of the three gated layers (context part of the hyper
    // networks' gate / bias, their time columns, the layer biases), and -- threads 0..2 -- the output layer's
    float kc_g[2][3], kc_hb[2][3], kc_tg[2][3], kc_tb[2][3], kc_b[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int i = l * XC_H + tid + 256 * u;
            kc_g[u][l] = hy[i];
            kc_hb[u][l] = hy[BOFF + i];
            kc_tg[u][l] = a.tcol[i];
            kc_tb[u][l] = a.tcol[BOFF + i];
            kc_b[u][l] = (l == 0 ? a.b0 : (l == 1 ? a.b1 : a.b2))[tid + 256 * u];
        }
    const int t3 = tid < 3 ? tid : 0;
    const float k3_g = hy[3 * XC_H + t3], k3_hb = hy[BOFF + 3 * XC_H + t3], k3_tg = a.tcol[3 * XC_H + t3], k3_tb = a.tcol[BOFF + 3 * XC_H + t3], k3_b = a.b3[t3];

    // The weight stream of one stage is a fixed sequence of 128 pieces: layer 1 chunk-major (s = 4 kc + rq), then layer 2
    // pass-major (s = 64 + 16 q + kc); piece (rq, kc) of a layer's pack sits at (rq * 16 + kc) * XW_PIECE.  Ring slot s & 3.
    auto piece_src = [&](int s_) -> const unsigned char * {
        s_ &= 127;
        const int l2 = s_ >> 6, t_ = s_ & 63;
        const int rq = l2 ? (t_ >> 4) : (t_ & 3), kc = l2 ? (t_ & 15) : (t_ >> 2);
        return (l2 ? a.w2x : a.w1x) + (long)(rq * 16 + kc) * XW_PIECE;
    };
    // LDS-DMA of a third of this wave's share (6 KB = 6 wave-instructions) of sequence piece s_: two global_load_lds_dwordx4 in
    // the SADDR form -- the piece's address in SGPRs (SALU arithmetic only), the lane's 16 bytes as ONE 32-bit VGPR offset that
    // never changes, the second kilobyte through the instruction offset, which advances the global AND the LDS address (M0
    // base + offset + 16 lane).  As inline asm: hipcc's own selection adds a 64-bit VALU add, a 64-bit literal move and an M0
    // write per instruction (~20 issue cycles each in a single-wave MFMA stream; measured 9 ms of a 46 ms launch for the DMA).
    // M0 has no other user in this kernel (checked on the disassembly by tests/test_host_cpu.py).
    auto dma = [&](int s_, int lane16, int i0) XW_INL {
        if constexpr (XW_EXP & 2) return;
        const unsigned char *src = piece_src(s_) + (wave * 6 + i0) * 1024;
        const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(wbuf + (s_ & 3) * XW_PIECE + (wave * 6 + i0) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                     : : "s"(dst), "v"(lane16), "s"(src) : "memory");
    };
    // ONE instruction per scheduling slot inside the product stream: a lone LDS-DMA hides behind the slot's MFMA, the second of a
    // back-to-back pair waits for the first (tools/micro/mfma_pipe.hip: 1764 cycles per 48-MFMA piece with three pairs, 1663
    // with six singles in slots that carry no producer step, 1632 without any DMA)
    auto dma1 = [&](int s_, int lane16, int i0) XW_INL {
        if constexpr (XW_EXP & 2) return;
        if constexpr (XW_EXP & 16384) {          // debug flavour: the previous placement, pairs in the slot of the even instruction
            if (!(i0 & 1)) dma(s_, lane16, i0);
            return;
        }
        const unsigned char *src = piece_src(s_) + (wave * 6 + i0) * 1024;
        const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(wbuf + (s_ & 3) * XW_PIECE + (wave * 6 + i0) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(lane16), "s"(src) : "memory");
    };
    const double t0 = a.reverse ? (double)a.t_end : 0.0, t1 = a.reverse ? 0.0 : (double)a.t_end;
    const double h = (t1 - t0) / (double)a.steps;
    const float hh = (float)h, h2 = (float)(0.5 * h), h6 = (float)(h / 6.0);

    // pieces 0, 1, 2 and the first third of piece 3 in flight before the first one is consumed
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
        dma(s_, lane0 * 16, 0);
        dma(s_, lane0 * 16, 2);
        dma(s_, lane0 * 16, 4);
    }
    dma(3, lane0 * 16, 0);

    f32x16 acc2[4];               // layer 2: the 128 rows of the running pass
    bf16x8 fX[2][3], fY[2][3];    // A fragments: two sets of two row tiles x three planes
    u32x4 b1w[2][2][3];           // layer 1 B planes [chunk parity][k-step of the chunk][plane]
    u32x4 b2w[2][3];              // layer 2 B planes [k-step parity][plane]

#if XW_EXP & (4 | 8 | 32 | 64)
    // timing experiments without producers / fragment reads: defined (opaque) operands
    for (int i_ = 0; i_ < 2; ++i_)
        for (int j_ = 0; j_ < 3; ++j_) {
            fX[i_][j_] = fY[i_][j_] = (bf16x8){1, 2, 3, 4, 5, 6, 7, 8};
            b2w[i_][j_] = (u32x4){1u, 2u, 3u, 4u};
            b1w[i_][0][j_] = b1w[i_][1][j_] = (u32x4){1u, 2u, 3u, 4u};
            asm volatile("" : "+v"(fX[i_][j_]), "+v"(fY[i_][j_]), "+v"(b2w[i_][j_]), "+v"(b1w[i_][0][j_]), "+v"(b1w[i_][1][j_]));
        }
#endif
    // fragments of region 0 of piece 0 (k-step 0, row tiles 0, 1): the only exposed fragment read of the kernel
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const unsigned char *A0 = wbuf + lane0 * 16;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fX[u][pl] = *(const bf16x8 *)(A0 + (u * 3 + pl) * XW_FRAG);
    }

    for (int step = 0; step < a.steps; ++step) {
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
            const double tc = (stage == 0) ? 0.0 : (stage == 3 ? 1.0 : 0.5);
            const float t = (float)(t0 + (double)step * h + tc * h);
            const float aw = (stage == 0) ? 0.f : (stage == 3 ? hh : h2);
            int lane = lane0;
            asm volatile("" : "+v"(lane));     // opaque: nothing derived from the lane id is hoisted out of the stage loop
            const int hq = (lane >> 5) * 4, lane16 = lane * 16;
            XW_STAMP(300)
#ifdef CASPR_DEBUG_HOOKS
            if (a.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && step == 0 && stage == 2) a.trace[306] = __builtin_amdgcn_s_memtime();
#endif
            // layer 1 accumulates from zero (issued before the barrier: overlaps the other waves' arrival)
            if constexpr (!(XW_EXP & 1024)) xw_for<0, 256>([&](auto N) XW_INL { xw_acc_zero<decltype(N)::value>(); });
            __syncthreads();   // the previous stage's epilogues are done with the tables
            // the stage's tables from the per-thread constants loaded once (kc_*): no global memory round trip per stage
#pragma unroll
            for (int u = 0; u < ((XW_EXP & 1024) ? 0 : 2); ++u) {
                const int i = tid + 256 * u;
                const float g0 = sigmoid_fast(fmaf(t, kc_tg[u][0], kc_g[u][0]));
                s_hb0[i] = fmaf(kc_b[u][0], g0, fmaf(t, kc_tb[u][0], kc_hb[u][0]));
                s_w0g[i] = s_w0[3 * i] * g0;
                s_w0g[XC_H + i] = s_w0[3 * i + 1] * g0;
                s_w0g[2 * XC_H + i] = s_w0[3 * i + 2] * g0;
                const float g1 = sigmoid_fast(fmaf(t, kc_tg[u][1], kc_g[u][1]));
                s_g1[i] = g1;
                s_hb1[i] = fmaf(kc_b[u][1], g1, fmaf(t, kc_tb[u][1], kc_hb[u][1]));
                const float g2 = sigmoid_fast(fmaf(t, kc_tg[u][2], kc_g[u][2]));
                s_g2[i] = g2;
                s_hb2[i] = fmaf(kc_b[u][2], g2, fmaf(t, kc_tb[u][2], kc_hb[u][2]));
            }
            if (tid < 3) {
                const float gt = sigmoid_fast(fmaf(t, k3_tg, k3_g));
                s_g3[tid] = gt;
                s_g3[4 + tid] = fmaf(k3_b, gt, fmaf(t, k3_tb, k3_hb));
            }
            __syncthreads();
            XW_STAMP(301)

            float ys[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) ys[d] = (stage == 0) ? y[d] : y[d] + aw * kprev[d];

            // barrier in front of the next sequence piece (placed in the last region of a piece): this wave's share of it has
            // landed once at most the 12 DMA instructions of the two younger pieces are outstanding; lgkmcnt: this wave's
            // reads of the ring slot that the DMA issued right after refills
            auto piece_head = [&](int sn) XW_INL {
                XW_STAMP(2 * (sn & 127))
                if constexpr (XW_EXP & 1) return;
                asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                XW_STAMP(2 * (sn & 127) + 1)
                asm volatile("" ::: "memory");
            };
            // one scheduling slot per MFMA: MFMA i of a region multiplies term i >> 1 (smallest first) into row tile i & 1 of the
            // pair; slots 0-5 also read the six fragments of the NEXT region, `fill` adds the slot's producer micro-step.  The
            // MFMA goes first, fenced: the lgkmcnt wait hipcc puts in front of it must not cover a read issued in the same slot
            // (measured: one exposed LDS round trip per region, +500 cycles per piece)
            constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
            // layer 1: row tiles T0, T0 + 1 in the hand-managed accumulator file
            auto region_a = [&](auto T0C, const bf16x8 (&fc)[2][3], const u32x4 (&b)[3], bf16x8 (&fn)[2][3], const unsigned char *An_, auto &&fill) XW_INL {
                constexpr int T0 = decltype(T0C)::value;
                const bf16x8 bb[3] = {__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[1]), __builtin_bit_cast(bf16x8, b[2])};
                xw_for<0, 12>([&](auto I) XW_INL {
                    constexpr int i = decltype(I)::value;
                    xw_mfma_a<T0 + (i & 1), (i < 2)>(fc[i & 1][TA[i >> 1]], bb[TB[i >> 1]]);
                    XW_FENCE;
                    if constexpr (i < 6 && !(XW_EXP & 8)) fn[i / 3][i % 3] = *(const bf16x8 *)(An_ + i * XW_FRAG);
                    fill(I);
                    XW_FENCE;
                });
            };
            // ... the last region of a piece: the next piece's barrier after the first two MFMAs, its first fragments after it
            auto region_a_last = [&](auto T0C, const bf16x8 (&fc)[2][3], const u32x4 (&b)[3], bf16x8 (&fn)[2][3], const unsigned char *An_, int sn, auto &&fill) XW_INL {
                constexpr int T0 = decltype(T0C)::value;
                const bf16x8 bb[3] = {__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[1]), __builtin_bit_cast(bf16x8, b[2])};
                xw_for<0, 12>([&](auto I) XW_INL {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i == 2) {
                        piece_head(sn);
                        XW_FENCE;
                    }
                    xw_mfma_a<T0 + (i & 1), (i < 2)>(fc[i & 1][TA[i >> 1]], bb[TB[i >> 1]]);
                    XW_FENCE;
                    if constexpr (i >= 2 && i < 8 && !(XW_EXP & 8)) fn[(i - 2) / 3][(i - 2) % 3] = *(const bf16x8 *)(An_ + (i - 2) * XW_FRAG);
                    fill(I);
                    XW_FENCE;
                });
            };
            // layer 2: row tiles c0, c1 in hipcc's registers
            auto region_v = [&](f32x16 &c0, f32x16 &c1, const bf16x8 (&fc)[2][3], const u32x4 (&b)[3], bf16x8 (&fn)[2][3], const unsigned char *An_, auto &&fill) XW_INL {
                const bf16x8 bb[3] = {__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[1]), __builtin_bit_cast(bf16x8, b[2])};
                xw_for<0, 12>([&](auto I) XW_INL {
                    constexpr int i = decltype(I)::value;
                    if constexpr ((i & 1) == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc[0][TA[i >> 1]], bb[TB[i >> 1]], c0, 0, 0, 0);
                    else c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc[1][TA[i >> 1]], bb[TB[i >> 1]], c1, 0, 0, 0);
                    XW_FENCE;
                    if constexpr (i < 6 && !(XW_EXP & 8)) fn[i / 3][i % 3] = *(const bf16x8 *)(An_ + i * XW_FRAG);
                    fill(I);
                    XW_FENCE;
                });
            };
            auto region_v_last = [&](f32x16 &c0, f32x16 &c1, const bf16x8 (&fc)[2][3], const u32x4 (&b)[3], bf16x8 (&fn)[2][3], const unsigned char *An_, int sn, auto &&fill) XW_INL {
                const bf16x8 bb[3] = {__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[1]), __builtin_bit_cast(bf16x8, b[2])};
                xw_for<0, 12>([&](auto I) XW_INL {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i == 2) {
                        piece_head(sn);
                        XW_FENCE;
                    }
                    if constexpr ((i & 1) == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc[0][TA[i >> 1]], bb[TB[i >> 1]], c0, 0, 0, 0);
                    else c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc[1][TA[i >> 1]], bb[TB[i >> 1]], c1, 0, 0, 0);
                    XW_FENCE;
                    if constexpr (i >= 2 && i < 8 && !(XW_EXP & 8)) fn[(i - 2) / 3][(i - 2) % 3] = *(const bf16x8 *)(An_ + (i - 2) * XW_FRAG);
                    fill(I);
                    XW_FENCE;
                });
            };

            // ================= layer 1: 16 chunks x 4 row quarters, sequence pieces 0..63 =================
            // B fragment of k-step t (T = t >> 1, u = t & 1), lane (j, h): slot s <-> unit 32 T + 16 u + (s & 3) + 8 (s >> 2) + 4 h:
            // group 0 (slots 0-3, words 0, 1) = four consecutive units from ub = 16 t + 4 h, group 1 (words 2, 3) from ub + 8
            f32x4 tin[4];         // input-layer tables of the group being produced: hb0, w0g x / y / z
            XwPair pa, pb;
            auto l1_tab = [&](int ub) XW_INL {
                if constexpr (XW_EXP & (4 | 32)) return;
                tin[0] = ld4(s_hb0 + ub);
                tin[1] = ld4(s_w0g + ub);
                tin[2] = ld4(s_w0g + XC_H + ub);
                tin[3] = ld4(s_w0g + 2 * XC_H + ub);
            };
            auto l1_pre = [&](XwPair &p, int pr) XW_INL {      // units 2 pr, 2 pr + 1 of the group
                p.x0 = fmaf(tin[1][2 * pr], ys[0], fmaf(tin[2][2 * pr], ys[1], fmaf(tin[3][2 * pr], ys[2], tin[0][2 * pr])));
                p.x1 = fmaf(tin[1][2 * pr + 1], ys[0], fmaf(tin[2][2 * pr + 1], ys[1], fmaf(tin[3][2 * pr + 1], ys[2], tin[0][2 * pr + 1])));
            };
            // the seven micro-steps of pair pr of a group, by slot (1, 3, 4, 6, 7, 9, 10)
            auto l1_pair_step = [&](auto I, u32x4 (&bw)[3], int grp, int pr) XW_INL {
                constexpr int i = decltype(I)::value;
                if constexpr (XW_EXP & (4 | 32)) return;
                if constexpr (i == 1) l1_pre(pa, pr);
                if constexpr (i == 3) xw_sp1(pa);
                if constexpr (i == 4) xw_sp2(pa);
                if constexpr (i == 6) xw_sp3(pa);
                if constexpr (i == 7) xw_split1(pa);
                if constexpr (i == 9) xw_split2(pa);
                if constexpr (i == 10) xw_split3(pa, bw, 2 * grp + pr);
            };
            // chunk 0 of the input layer up front (exposed: 1/16 of the input layer)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    l1_tab(16 * ks + 8 * grp + hq);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        l1_pre(pa, pr);
                        xw_sp1(pa);
                        xw_sp2(pa);
                        xw_sp3(pa);
                        xw_split1(pa);
                        xw_split2(pa);
                        xw_split3(pa, b1w[0][ks], 2 * grp + pr);
                    }
                }
            XW_FENCE;
            XW_STAMP(302)
#pragma unroll 1
            for (int it = 0; it < 8; ++it) {
                xw_for<0, 8>([&](auto PC) XW_INL {
                    constexpr int pc = decltype(PC)::value, par = pc >> 2, rq = pc & 3;   // chunk kc = 2 it + par, ring slot = rq
                    const int s1 = 8 * it + pc;                                            // sequence piece
                    const unsigned char *A = wbuf + rq * XW_PIECE + lane16;
                    const unsigned char *An = wbuf + ((rq + 1) & 3) * XW_PIECE + lane16;
                    // producers of chunk kc + 1 (B set par ^ 1): piece rq makes group (ks = rq >> 1, grp = rq & 1): tables in
                    // region 0, one pair in regions 1 and 2 each.  (Chunk 16 does not exist: the last pass reads table
                    // entries past the layer's 512 -- still inside the table block -- into a B set nobody multiplies.)
                    const int ubn = 32 * (2 * it + par + 1) + 16 * (rq >> 1) + 8 * (rq & 1) + hq;
                    u32x4 (&bn)[3] = b1w[par ^ 1][rq >> 1];
                    // region 0: k-step 0, row tiles 0, 1 (fX) | reads k-step 0, row tiles 2, 3 -> fY
                    XW_STAMPC(it == 1 || it == 2, 320 + 32 * (it - 1) + 4 * pc)
                    region_a(std::integral_constant<int, 4 * rq>{}, fX, b1w[par][0], fY, A + 6 * XW_FRAG, [&](auto I) XW_INL {
                        constexpr int i = decltype(I)::value;
                        if constexpr (i == 0) l1_tab(ubn);
                        if constexpr (i == 7) dma1(s1 + 3, lane16, 2);
                        if constexpr (i == 10) dma1(s1 + 3, lane16, 3);
                    });
                    // region 1: k-step 0, row tiles 2, 3 (fY) | reads k-step 1, row tiles 0, 1 -> fX
                    XW_STAMPC(it == 1 || it == 2, 320 + 32 * (it - 1) + 4 * pc + 1)
                    region_a(std::integral_constant<int, 4 * rq + 2>{}, fY, b1w[par][0], fX, A + 12 * XW_FRAG, [&](auto I) XW_INL {
                        constexpr int i = decltype(I)::value;
                        l1_pair_step(I, bn, rq & 1, 0);
                        if constexpr (i == 8) dma1(s1 + 3, lane16, 4);       // slots 8 and 11 carry no producer step
                        if constexpr (i == 11) dma1(s1 + 3, lane16, 5);
                    });
                    // region 2: k-step 1, row tiles 0, 1 (fX) | reads k-step 1, row tiles 2, 3 -> fY
                    XW_STAMPC(it == 1 || it == 2, 320 + 32 * (it - 1) + 4 * pc + 2)
                    region_a(std::integral_constant<int, 4 * rq>{}, fX, b1w[par][1], fY, A + 18 * XW_FRAG, [&](auto I) XW_INL {
                        l1_pair_step(I, bn, rq & 1, 1);
                    });
                    // region 3: barrier of the next piece | k-step 1, row tiles 2, 3 (fY) | reads the next piece's k-step 0, row
                    // tiles 0, 1 -> fX | first third of the DMA of piece s1 + 4 into the slot just released
                    XW_STAMPC(it == 1 || it == 2, 320 + 32 * (it - 1) + 4 * pc + 3)
                    region_a_last(std::integral_constant<int, 4 * rq + 2>{}, fY, b1w[par][1], fX, An, s1 + 1, [&](auto I) XW_INL {
                        constexpr int i = decltype(I)::value;
                        if constexpr (i == 8) dma1(s1 + 4, lane16, 0);
                        if constexpr (i == 11) dma1(s1 + 4, lane16, 1);
                    });
                });
            }
            XW_STAMP(303)

            // ================= layer 2: four passes of 16 pieces, sequence pieces 64 + 16 q + kc =================
            float part[3] = {0.f, 0.f, 0.f};
            f32x4 tt[2][2];       // gate1 / hb1 of a group, [region parity][gate | bias], read one region ahead
            float qv[4];
            auto l2_tab = [&](int set, int t_, int grp) XW_INL {
                if constexpr (XW_EXP & (4 | 64)) return;
                const int c = 16 * t_ + 8 * grp + hq;
                tt[set][0] = ld4(s_g1 + c);
                tt[set][1] = ld4(s_hb1 + c);
            };
            // Producer of group GRP of k-step T_ (registers a[16 (T_ >> 1) + 8 (T_ & 1) + 4 GRP + r], r = 0..3), by slot.  FIRST
            // pass: gate / bias / softplus applied in place; later passes: the stored activation is only split again.
            auto l2_step = [&](auto I, auto TC, auto GC, auto FC, u32x4 (&bw)[3], int set) XW_INL {
                constexpr int i = decltype(I)::value, t_ = decltype(TC)::value, grp = decltype(GC)::value;
                constexpr bool first = decltype(FC)::value;
                constexpr int base = 16 * (t_ >> 1) + 8 * (t_ & 1) + 4 * grp;
                if constexpr ((XW_EXP & (4 | 64)) != 0 && i >= 0) return;
                if constexpr (first) {
                    if constexpr (i == 0) {
                        qv[0] = xw_acc_rd<base>();
                        qv[1] = xw_acc_rd<base + 1>();
                        qv[2] = xw_acc_rd<base + 2>();
                        qv[3] = xw_acc_rd<base + 3>();
                    }
                    if constexpr (i == 1) {
                        pa.x0 = fmaf(qv[0], tt[set][0][0], tt[set][1][0]);
                        pa.x1 = fmaf(qv[1], tt[set][0][1], tt[set][1][1]);
                        pb.x0 = fmaf(qv[2], tt[set][0][2], tt[set][1][2]);
                        pb.x1 = fmaf(qv[3], tt[set][0][3], tt[set][1][3]);
                    }
                    if constexpr (i == 2) xw_sp1(pa);
                    if constexpr (i == 3) xw_sp1(pb);
                    if constexpr (i == 4) xw_sp2(pa);
                    if constexpr (i == 5) xw_sp2(pb);
                    if constexpr (i == 6) {
                        xw_sp3(pa);
                        xw_acc_wr<base>(pa.x0);
                        xw_acc_wr<base + 1>(pa.x1);
                    }
                    if constexpr (i == 7) {
                        xw_sp3(pb);
                        xw_acc_wr<base + 2>(pb.x0);
                        xw_acc_wr<base + 3>(pb.x1);
                    }
                    if constexpr (i == 8) xw_split1(pa);
                    if constexpr (i == 9) xw_split2(pa);
                    if constexpr (i == 10) {
                        xw_split3(pa, bw, 2 * grp);
                        xw_split1(pb);
                    }
                    if constexpr (i == 11) {
                        xw_split2(pb);
                        xw_split3(pb, bw, 2 * grp + 1);
                    }
                } else {
                    if constexpr (i == 0) {
                        pa.x0 = xw_acc_rd<base>();
                        pa.x1 = xw_acc_rd<base + 1>();
                    }
                    if constexpr (i == 1) xw_split1(pa);
                    if constexpr (i == 2) xw_split2(pa);
                    if constexpr (i == 3) {
                        xw_split3(pa, bw, 2 * grp);
                        pb.x0 = xw_acc_rd<base + 2>();
                        pb.x1 = xw_acc_rd<base + 3>();
                    }
                    if constexpr (i == 4) xw_split1(pb);
                    if constexpr (i == 5) xw_split2(pb);
                    if constexpr (i == 6) xw_split3(pb, bw, 2 * grp + 1);
                }
            };
            auto pass = [&](int q, auto FC) XW_INL {
                constexpr bool first = decltype(FC)::value;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[mi][r] = 0.f;
                // k-step 0 of this pass up front (exposed); the tables of (k-step 1, group 0) for region 0
                if constexpr (first) {
                    l2_tab(0, 0, 0);
                    l2_tab(1, 0, 1);
                }
                xw_for<0, 12>([&](auto I) XW_INL { l2_step(I, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, FC, b2w[0], 0); });
                xw_for<0, 12>([&](auto I) XW_INL { l2_step(I, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, FC, b2w[0], 1); });
                if constexpr (first) l2_tab(0, 1, 0);
                XW_FENCE;
                xw_for<0, 16>([&](auto KC) XW_INL {
                    constexpr int kc = decltype(KC)::value;
                    const int sq = 64 + 16 * q + kc;
                    const unsigned char *A = wbuf + (kc & 3) * XW_PIECE + lane16;            // sq & 3 == kc & 3
                    const unsigned char *An = wbuf + ((kc + 1) & 3) * XW_PIECE + lane16;
                    constexpr int tb = 2 * kc + 1, tn = (kc < 15 ? 2 * kc + 2 : 0);
                    // k-step t + 1 is produced during k-step t: group 0 in the region of row tiles 0, 1 (tables in tt[0]), group 1 in
                    // the region of row tiles 2, 3 (tt[1]); slot 6 of a region reads the tables of the next region's group
                    // (behind the fragment reads of slots 0-5, well ahead of the lgkmcnt wait that opens the next region)
                    // region 0: k-step 2 kc, row tiles 0, 1 (fX) | reads row tiles 2, 3 -> fY | group 0 of k-step tb
                    region_v(acc2[0], acc2[1], fX, b2w[0], fY, A + 6 * XW_FRAG, [&](auto I) XW_INL {
                        constexpr int i = decltype(I)::value;
                        l2_step(I, std::integral_constant<int, tb>{}, std::integral_constant<int, 0>{}, FC, b2w[1], 0);
                        if constexpr (first && i == 6) l2_tab(1, tb, 1);
                        if constexpr (i == 7) dma1(sq + 3, lane16, 2);
                        if constexpr (i == 10) dma1(sq + 3, lane16, 3);
                    });
                    // region 1: k-step 2 kc, row tiles 2, 3 (fY) | reads tb, row tiles 0, 1 -> fX | group 1 of tb
                    region_v(acc2[2], acc2[3], fY, b2w[0], fX, A + 12 * XW_FRAG, [&](auto I) XW_INL {
                        constexpr int i = decltype(I)::value;
                        l2_step(I, std::integral_constant<int, tb>{}, std::integral_constant<int, 1>{}, FC, b2w[1], 1);
                        if constexpr (first && kc < 15 && i == 6) l2_tab(0, tn, 0);
                        if constexpr (i == 8) dma1(sq + 3, lane16, 4);
                        if constexpr (i == 11) dma1(sq + 3, lane16, 5);
                    });
                    // region 2: k-step tb, row tiles 0, 1 (fX) | reads tb, row tiles 2, 3 -> fY | group 0 of k-step tb + 1
                    region_v(acc2[0], acc2[1], fX, b2w[1], fY, A + 18 * XW_FRAG, [&](auto I) XW_INL {
                        constexpr int i = decltype(I)::value;
                        if constexpr (kc < 15) l2_step(I, std::integral_constant<int, tn>{}, std::integral_constant<int, 0>{}, FC, b2w[0], 0);
                        if constexpr (first && kc < 15 && i == 6) l2_tab(1, tn, 1);
                    });
                    // region 3: barrier of the next piece | k-step tb, row tiles 2, 3 (fY) | reads the next piece's first
                    // fragments -> fX | DMA of piece sq + 4 | group 1 of k-step tb + 1
                    region_v_last(acc2[2], acc2[3], fY, b2w[1], fX, An, sq + 1, [&](auto I) XW_INL {
                        constexpr int i = decltype(I)::value;
                        if constexpr (kc < 15) l2_step(I, std::integral_constant<int, tn>{}, std::integral_constant<int, 1>{}, FC, b2w[0], 1);
                        if constexpr (first && kc < 15 && i == 6) l2_tab(0, tn + 1, 0);
                        if constexpr (i == 8) dma1(sq + 4, lane16, 0);
                        if constexpr (i == 11) dma1(sq + 4, lane16, 1);
                    });
                });
                // ---- epilogue of hidden layer 2 for rows 128 q .. 128 q + 127 + their share of the 512 -> 3 output layer:
                // acc2[rt] register r <-> unit 128 q + 32 rt + 8 (r >> 2) + 4 h + (r & 3)
                XW_STAMP(310 + 2 * q)
                int le = lane;   // opaque again: the table addresses must not be hoisted above the product loop
                asm volatile("" : "+v"(le));
                const int cq = 128 * q + (le >> 5) * 4;
#pragma unroll
                for (int rt = 0; rt < ((XW_EXP & 512) ? 1 : 4); ++rt)
#pragma unroll
                    for (int rr = 0; rr < ((XW_EXP & 512) ? 1 : 4); ++rr) {
                        const int c = cq + 32 * rt + 8 * rr;
                        const f32x4 gt = ld4(s_g2 + c), hb = ld4(s_hb2 + c);
                        const f32x4 wx3 = ld4(s_w3 + c), wy3 = ld4(s_w3 + XC_H + c), wz3 = ld4(s_w3 + 2 * XC_H + c);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float hv = softplus_fast(fmaf(acc2[rt][4 * rr + r], gt[r], hb[r]));
                            part[0] = fmaf(wx3[r], hv, part[0]);
                            part[1] = fmaf(wy3[r], hv, part[1]);
                            part[2] = fmaf(wz3[r], hv, part[2]);
                        }
                    }
                XW_FENCE;
                XW_STAMP(311 + 2 * q)
            };
            int q0 = 0;
            asm volatile("" : "+s"(q0));     // opaque: the piece addresses of pass 0 are computed like those of passes 1-3 (SALU), not
                                             // precomputed for all 16 pieces and spilled
            pass(q0, std::true_type{});
            XW_STAMP(304)
#pragma unroll 1
            for (int q = 1; q < 4; ++q) pass(q, std::false_type{});
            XW_STAMP(305)

            // ---- output ConcatSquash (no softplus: odefunc.py:103): the two halves of a column hold disjoint rows
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v = part[d];
                v += __shfl_xor(v, 32);
                const float od = fmaf(v, s_g3[d], s_g3[4 + d]);
                kprev[d] = od;
                kacc[d] = (stage == 0) ? od : ((stage == 3) ? kacc[d] + od : kacc[d] + 2.0f * od);
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) y[d] = y[d] + h6 * kacc[d];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pieces left in flight by the last stage

    if (cvalid && lane0 < 32) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float v = y[d];
            if (a.mbn_out) {
                const float w = a.mbn_out[d], bb = a.mbn_out[3 + d], mean = a.mbn_out[6 + d], var = a.mbn_out[9 + d];
                if (a.reverse) v = (v - bb) * expf(-w) * expf(0.5f * logf(var + 1e-4f)) + mean;
                else v = (v - mean) * expf(-0.5f * logf(var + 1e-4f)) * expf(w) + bb;
            }
            a.y_out[((long)bt * a.n + col) * 3 + d] = v;
        }
    }
}

// wide pack: (512, ldw) f32 -> [row quarter 4][k chunk 16][k-step 2][row tile 4][plane 3][lane 64][8 bf16]: lane (i = l & 31,
// h = l >> 5) of fragment (rq, kc, ks, rt) holds row 128 rq + 32 rt + i, k slots s = 0..7 <-> k = 32 kc + 16 ks + (s & 3) +
// 8 (s >> 2) + 4 h (the D-fragment order of the producing layer, see the header)
__global__ void pack_weight_cnf_x6w_kernel(const float *__restrict__ w, int ldw, unsigned char *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // ((rq * 16 + kc) * 8 + ks * 4 + rt) * 64 + lane
    if (i >= 4 * 16 * 8 * 64) return;
    const int l = i & 63, fr = (i >> 6) & 7, ck = i >> 9;
    const int rt = fr & 3, ks = fr >> 2, kc = ck & 15, rq = ck >> 4;
    const int row = 128 * rq + 32 * rt + (l & 31), hh = l >> 5;
    float hs[3][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = 32 * kc + 16 * ks + (s & 3) + 8 * (s >> 2) + 4 * hh;
        xc_split(w[(long)row * ldw + k], hs[0][s], hs[1][s], hs[2][s]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = xc_pack(hs[pl][2 * q], hs[pl][2 * q + 1]);
        *(u32x4 *)(out + (long)ck * XW_PIECE + ((long)(fr * 3 + pl) * 64 + l) * 16) = v;
    }
}

int caspr_cnf_x6w_pack(const float *w, int ldw, unsigned char *out, hipStream_t stream)
{
    pack_weight_cnf_x6w_kernel<<<4 * 16 * 8 * 64 / 256, 256, 0, stream>>>(w, ldw, out);
    return CASPR_OK;
}

int caspr_cnf_x6w_launch(const CnfX6Args &a, int BT, hipStream_t stream)
{
    static CasprLdsOptIn optin;
    const hipError_t err = caspr_lds_opt_in(optin, (const void *)cnf_rk4_x6w_kernel, XW_LDS);
    if (err != hipSuccess) {
        caspr_set_error("cnf_rk4_x6: hipFuncSetAttribute failed: %s", hipGetErrorString(err));
        return CASPR_ELAUNCH;
    }
    cnf_rk4_x6w_kernel<<<dim3(ceil_div(a.n, XW_PTS), BT), dim3(256), XW_LDS, stream>>>(a);
    return CASPR_OK;
}
