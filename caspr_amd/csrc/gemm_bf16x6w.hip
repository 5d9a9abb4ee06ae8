// gemm_bf16x6w.hip -- pointwise conv (nn.Conv1d k = 1; pointnet2.py:525,247, tpointnet2.py:96-105) on the bf16 matrix pipe in the
// exact three-way split, for the LARGE layers of the encoder (the 1600-wide head, the 512-wide feature-propagation / final
// layers): workgroup tile 128 points x 512 output channels on v_mfma_f32_32x32x16_bf16, the technique of ode_bf16x6w.hip.
//
// Why a second conv kernel.  conv1x1_bf16x6_kernel (gemm_bf16x6.hip: 128 points x 256 channels, 16x16x32 MFMAs, two workgroups
// per CU, both operands through LDS) re-reads and re-splits the activation once per 256 output channels -- 7 passes over the
// 2.1 GB input of a 1600 -> 1600 layer, FETCH_SIZE 23 GB raw -- and sits at 0.42-0.45 of the bf16x6 ceiling.  Here
//  * a wave owns 128 channels x ALL 128 points of the tile: 4 x 4 tiles of 32 x 32 = 256 accumulators, kept in the accumulator
//    file BY HAND (x6w_common.h), so the activation is read and split once per 512 channels (4 passes at 1600: 3 full tiles
//    here, the 64-channel remainder on the other kernel) and a fragment read feeds four times the products (0.25 reads / MFMA);
//  * the WEIGHT fragments never touch LDS: a wave's 128 channels are its own, so it loads its 12 fragments of a k-step straight
//    from global memory (L2-resident pack in fragment order, 1 KB coalesced per instruction) one k-step (3072 matrix-pipe
//    cycles) ahead into a second register set -- no LDS-DMA (60+ issue cycles each), no fragment reads for that operand;
//  * LDS holds only the split ACTIVATION of the running 32-k chunk (24 KB, double-buffered): every thread loads 16 values of a
//    row, applies the producer's GroupNorm + ReLU (scale / shift through scalar loads: the k range of a thread is wave-uniform),
//    splits them into the three planes in the MFMA shadow and writes them in fragment order; one barrier per chunk of 192 MFMAs;
//  * one scheduling slot per MFMA, two accumulator tiles alternating, as in the CNF kernel;
//  * epilogue: bias / per-batch bias, optional GroupNorm statistics of the output per 128-point tile (mean, squared deviations,
//    max, min: the format of conv1x1_bf16x6_kernel<., true>, finalised by the same conv_gn_finalize_kernel).
// Contract: Cin % 32 == 0, P % 128 == 0, Cout % 512 == 0 (the host wrapper sends a remainder of the channels to the other kernel).
#include "x6w_common.h"

// XW_EXP (debug flavours, build.py CASPR_XW_EXP): timing experiments, WRONG results: 2048 no weight loads inside the K loop,
// 4096 no activation staging inside the K loop, 8192 no epilogue
#ifndef XW_EXP
#define XW_EXP 0
#endif
#define CW_TP 128
#define CW_TM 512
#define CW_FRAG 1024
#define CW_BCHUNK (2 * 4 * 3 * CW_FRAG)      // split activation of one 32-k chunk: [k-step 2][column tile 4][plane 3][fragment]

#define CW_SPART (2 * CW_BCHUNK)            // STATS: [512 channels] {mean, M2, max, min} of the tile being finished: a region of its own
#define CW_LDS (2 * CW_BCHUNK + CW_TM * 16)

struct ConvWArgs {
    const unsigned char *wpk;
    const float *bias, *bbias, *X;
    float *Y;
    f32x4 *part;
    int ldx, ldy, P, Cin, Cout, in_relu, relu_from, Mt, Pt, part_stride, bb_stride, ntiles, mt0;
};

// One chunk of the flattened (tile, k-chunk) sequence a workgroup walks through: everything here is wave-uniform (SGPRs).
struct CwChunk {
    int lin;                      // tile id (channel-tile-major over the whole problem)
    int kc;                       // 32-k chunk inside the tile
    int mt, b, pt;                // its channel tile, batch entry, point tile
};

// PERSISTENT since round 4.  The round-3 form launched one workgroup per tile, one workgroup per CU (512 registers): a tile's
// prologue (first weights, first two activation chunks from HBM, chunk 0 split and staged: ~2-3 us) and its epilogue (256 KB of
// output per tile) were exposed, and -- all resident workgroups running the same K in loose lockstep -- the layer's whole OUTPUT
// (2.1 GB at 1600 channels) left in bursts between K loops instead of underneath them: timed over K, a layer cost
// 0.73 ms + 4.5 us per input channel at cfg-2's 327,680 rows, the constant being a fifth of the 576 -> 1600 layer and a sixth of every
// 512-wide one.  Now a workgroup walks through its tiles (blockIdx.x + i * gridDim.x, still channel-tile-major) as ONE stream of
// k-chunks: the last chunks of tile i already load / split / stage the first chunks of tile i + 1 and fetch its first weights, the
// stores of tile i drain under tile i + 1's products, and only the accumulator read-out itself (256 values per lane) sits between.
// in_scale / in_shift come as kernel arguments of their own (const __restrict__: hipcc then reads them through the scalar cache;
// as members of the argument struct they became per-lane global loads followed by vmcnt(0), eight times per chunk)
template <bool FUSED, bool STATS>
__global__ __launch_bounds__(256, 1) void conv1x1_x6w_kernel(ConvWArgs a, const float *__restrict__ in_scale, const float *__restrict__ in_shift)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tile order: CHANNEL-TILE-MAJOR over the whole problem.  The weight slice of one channel tile (512 x Cin x 6 B: 4.9 MB at Cin = 1600)
    // is streamed by every workgroup that works on that tile, 16 B / clk / CU: it has to come out of the XCDs' L2s, not the fabric.
    // The gridDim.x resident workgroups take tiles w, w + G, w + 2 G, ...: at any time they all work on the same channel tile (except
    // around the Mt - 1 switches) and stream the same slice in loose lockstep; the activation tile is then read Mt times, far apart
    // in time (3 x 2.1 GB per 1600-wide layer, HBM / MALL).  No assumption about placement: any order is correct.
    const int G = gridDim.x;
    const int npt = a.ntiles / a.Mt;                  // point tiles over all batch entries
    const int nk = a.Cin / 32;
    const int nmine = (a.ntiles - (int)blockIdx.x + G - 1) / G;
    const int total = nmine * nk;                     // chunks of this workgroup
    const int lin_last = blockIdx.x + (nmine - 1) * G;

    auto decode = [&](CwChunk &c) XW_INL {
        const int mtl = c.lin / npt;
        const int gpt = c.lin - mtl * npt;
        c.mt = a.mt0 + mtl;                 // a launch may cover a RANGE of channel tiles (mt0 .. mt0 + Mt - 1)
        c.b = gpt / a.Pt;
        c.pt = gpt - c.b * a.Pt;
    };
    // the chunk after c in this workgroup's stream; past the end it stays on the last chunk (whatever is loaded / staged for it
    // again is harmless and never consumed)
    auto advance = [&](CwChunk &c) XW_INL {
        if (c.kc + 1 < nk) {
            c.kc += 1;
        } else if (c.lin != lin_last) {
            c.kc = 0;
            c.lin += G;
            decode(c);
        }
    };

    // activation staging: thread = (row xr of the tile, k-step xh of the chunk: 16 consecutive k); xh is wave-uniform
    const int xr = tid & 127, xh = wave >> 1;
    const unsigned xoff = (unsigned)xr * (unsigned)a.ldx + 16u * xh;                  // floats, inside the tile's 128 rows
    float scv[16], shv[16];       // the scale / shift of the thread's 16 channels of the chunk being staged (SGPRs: wave-uniform)
    const unsigned wdst = ((xh * 4 + (xr >> 5)) * 3) * CW_FRAG + (xr & 31) * 16;     // + plane * CW_FRAG + half * 512
    // weights: [channel tile][k-step][wave][row tile 4][plane 3][fragment]: 12 KB per wave and k-step
    const long wstep = 4L * 12 * CW_FRAG;

    bf16x8 afr[2][4][3];          // weight fragments [k-step parity][row tile][plane]
    bf16x8 bfr[2][2][3];          // activation fragments [column-tile pair][tile of the pair][plane]
    f32x4 xra[4], xrb[4];         // 16 raw values of the thread's row for the next chunk and the one after (loaded a whole chunk ahead: HBM latency)
    u32x4 pv[2][3];               // their three planes [half of the k-step][plane]
    XwPair sp;

    auto aload = [&](const CwChunk &c, int ks, auto SETC, auto RTC) XW_INL {        // the three planes of row tile RT of k-step ks of chunk c -> set SET
        constexpr int set = decltype(SETC)::value, rt = decltype(RTC)::value;
        const unsigned char *p = a.wpk + ((long)c.mt * (2 * nk) * 4 + wave) * (12 * CW_FRAG) + (long)(2 * c.kc + ks) * wstep + rt * 3 * CW_FRAG + lane0 * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) afr[set][rt][pl] = *(const bf16x8 *)(p + pl * CW_FRAG);
    };
    auto gload = [&](f32x4 (&xv)[4], const CwChunk &c) XW_INL {
        const float *base = a.X + ((long)c.b * a.P + c.pt * CW_TP) * a.ldx + c.kc * 32;      // wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = ld4(base + xoff + 4 * q);
    };
    auto sload = [&](const CwChunk &c) XW_INL {
        if constexpr (FUSED) {
            const float *sc = in_scale + (long)c.b * a.Cin + 16 * xh + c.kc * 32, *sh = in_shift + (long)c.b * a.Cin + 16 * xh + c.kc * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                scv[q] = sc[q];
                shv[q] = sh[q];
            }
        }
    };
    // pair m (values 2m, 2m + 1 of the thread's 16) of chunk kc: producer transform, then the split micro-steps
    auto st_pre = [&](const f32x4 (&xv)[4], int kc, auto MC) XW_INL {
        constexpr int m = decltype(MC)::value;
        float v0 = xv[m >> 1][(2 * m) & 3], v1 = xv[m >> 1][(2 * m + 1) & 3];
        if constexpr (FUSED) {
            // the ReLU switches on at a multiple of 8 channels (checked by the host)
            const float lo = (a.in_relu && kc * 32 + 16 * xh + 2 * m >= a.relu_from) ? 0.f : -INFINITY;
            v0 = fmaxf(fmaf(v0, scv[2 * m], shv[2 * m]), lo);
            v1 = fmaxf(fmaf(v1, scv[2 * m + 1], shv[2 * m + 1]), lo);
        }
        sp.x0 = v0;
        sp.x1 = v1;
    };
    auto st_put = [&](auto MC) XW_INL {
        constexpr int m = decltype(MC)::value;
        xw_split3(sp, pv[m >> 2], m & 3);
    };
    auto st_write = [&](int buf, auto HC, auto PC) XW_INL {
        constexpr int hf = decltype(HC)::value, pl = decltype(PC)::value;
        *(u32x4 *)(lds + buf * CW_BCHUNK + wdst + pl * CW_FRAG + hf * 512) = pv[hf][pl];
    };
    // fragment read: tile c of the pair CTP of k-step ks of the chunk in buffer buf, plane pl
    auto bread = [&](int buf, int ks, auto CTPC, auto IC) XW_INL {
        constexpr int ctp = decltype(CTPC)::value, i = decltype(IC)::value;     // i = tile * 3 + plane
        bfr[ctp][i / 3][i % 3] = *(const bf16x8 *)(lds + buf * CW_BCHUNK + ((ks * 4 + 2 * ctp + i / 3) * 3 + i % 3) * CW_FRAG + lane0 * 16);
    };

    // ---- the stream's three live positions: the chunk being multiplied, the next one (staged under it), the one after (loaded under it)
    CwChunk c0, c1, c2;
    c0.lin = blockIdx.x;
    c0.kc = 0;
    decode(c0);
    c1 = c0;
    advance(c1);
    c2 = c1;
    advance(c2);

    xw_for<0, 256>([&](auto N) XW_INL { xw_acc_zero<decltype(N)::value>(); });
    // ---- prologue (once per workgroup): weights of k-step 0, chunk 0 staged (exposed), its first fragments
    xw_for<0, 4>([&](auto RT) XW_INL { aload(c0, 0, std::integral_constant<int, 0>{}, RT); });
    gload(xra, c0);
    sload(c0);
    gload(xrb, c1);
    xw_for<0, 8>([&](auto M) XW_INL {
        st_pre(xra, 0, M);
        xw_split1(sp);
        xw_split2(sp);
        st_put(M);
    });
    xw_for<0, 2>([&](auto H) XW_INL { xw_for<0, 3>([&](auto PL) XW_INL { st_write(0, H, PL); }); });
    __syncthreads();
    xw_for<0, 6>([&](auto I) XW_INL { bread(0, 0, std::integral_constant<int, 0>{}, I); });
    XW_FENCE;

    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
    // one region = 12 MFMAs on accumulator tiles (RT, 2 CTP), (RT, 2 CTP + 1), one scheduling slot each
    auto region = [&](auto KSC, auto CTPC, auto RTC, auto &&fill) XW_INL {
        constexpr int ks = decltype(KSC)::value, ctp = decltype(CTPC)::value, rt = decltype(RTC)::value;
        xw_for<0, 12>([&](auto I) XW_INL {
            constexpr int i = decltype(I)::value;
            xw_mfma_a<4 * rt + 2 * ctp + (i & 1), (i < 2)>(afr[ks][rt][TA[i >> 1]], bfr[ctp][i & 1][TB[i >> 1]]);
            XW_FENCE;
            fill(I);
            XW_FENCE;
        });
    };

    // one 32-k chunk (c0) in LDS buffer cur: 16 regions; stages chunk c1 from xs (loaded during the previous chunk) into the other
    // buffer, loads chunk c2 into xl.  c1 / c2 may belong to the NEXT tile of this workgroup.
    auto chunk = [&](int cur, const f32x4 (&xs)[4], f32x4 (&xl)[4]) XW_INL {
        const int nxt = cur ^ 1;
        // Staging of chunk c1: loads in region (0,0,0); pair g in region g of the eight regions (0,1,*) and (1,0,*): transform,
        // split x 2, put at slots 1, 3, 5, 7; the three planes of a half written at slots 8-10 of regions 3 and 7
        auto stage = [&](auto GC, auto I) XW_INL {
            constexpr int g = decltype(GC)::value, i = decltype(I)::value;
            if constexpr ((XW_EXP & 4096) != 0 && i >= 0) return;
            if constexpr (i == 1) st_pre(xs, c1.kc, GC);
            if constexpr (i == 3) xw_split1(sp);
            if constexpr (i == 5) xw_split2(sp);
            if constexpr (i == 7) st_put(GC);
            if constexpr ((g & 3) == 3 && i >= 8 && i < 11) st_write(nxt, std::integral_constant<int, (g >> 2)>{}, std::integral_constant<int, i - 8>{});
        };
        // ---- k-step 0 of the chunk (weights: set 0); set 1 <- k-step 1 of this chunk (3 loads in slots 2 of regions (0,0,*))
        xw_for<0, 4>([&](auto RT) XW_INL {
            constexpr int rt = decltype(RT)::value;
            region(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, RT, [&](auto I) XW_INL {
                constexpr int i = decltype(I)::value;
                if constexpr (rt == 0 && i < 6) bread(cur, 0, std::integral_constant<int, 1>{}, I);          // column tiles 2, 3 of this k-step
                if constexpr (i == 6 && !(XW_EXP & 2048)) aload(c0, 1, std::integral_constant<int, 1>{}, RT);
                if constexpr (rt == 0 && i == 7 && !(XW_EXP & 4096)) sload(c1);
            });
        });
        xw_for<0, 4>([&](auto RT) XW_INL {
            constexpr int rt = decltype(RT)::value;
            region(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, RT, [&](auto I) XW_INL {
                constexpr int i = decltype(I)::value;
                if constexpr (rt == 2 && i < 6) bread(cur, 1, std::integral_constant<int, 0>{}, I);          // column tiles 0, 1 of k-step 1
                stage(RT, I);
                // the raw activations of the chunk after next: issued BEHIND this k-step's weight loads, so that the counted waits for
                // those do not cover them, a whole chunk (6144 matrix-pipe cycles) before the staging that consumes them
                if constexpr (rt == 0 && i == 8 && !(XW_EXP & 4096)) gload(xl, c2);
            });
        });
        // ---- k-step 1 (weights: set 1); set 0 <- k-step 0 of the next chunk
        xw_for<0, 4>([&](auto RT) XW_INL {
            constexpr int rt = decltype(RT)::value;
            region(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, RT, [&](auto I) XW_INL {
                constexpr int i = decltype(I)::value;
                if constexpr (rt == 0 && i < 6) bread(cur, 1, std::integral_constant<int, 1>{}, I);          // column tiles 2, 3 of k-step 1
                stage(std::integral_constant<int, 4 + rt>{}, I);
                if constexpr (i == 11 && !(XW_EXP & 2048)) aload(c1, 0, std::integral_constant<int, 0>{}, RT);
            });
        });
        xw_for<0, 4>([&](auto RT) XW_INL {
            constexpr int rt = decltype(RT)::value;
            if constexpr (rt == 0) {
                // every wave's planes of chunk c1 are written (and this wave is done reading the buffer the chunk after that
                // will overwrite): one barrier per chunk
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                XW_FENCE;
            }
            region(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, RT, [&](auto I) XW_INL {
                constexpr int i = decltype(I)::value;
                if constexpr (rt == 0 && i < 6) bread(nxt, 0, std::integral_constant<int, 0>{}, I);          // column tiles 0, 1 of the next chunk
            });
        });
    };

    // ---- a tile's read-out: bias / per-batch bias, store (the stores drain under the next tile's products), statistics, accumulators
    // back to zero.  Everything the stream keeps in flight (weight set 0 and the first fragments of the next chunk, the raw rows of the
    // two chunks after it) stays live across it.
    auto finish = [&](const CwChunk &c) XW_INL {
        if constexpr (XW_EXP & 8192) return;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results, before the accumulator file is read
        const int lane = lane0, j = lane & 31, hq = (lane >> 5) * 4;
        const int p0 = c.pt * CW_TP;
        const int cw = c.mt * CW_TM + wave * 128;               // first channel of this wave
        f32x4 *spart = (f32x4 *)(lds + CW_SPART);
        xw_for<0, 4>([&](auto RT) XW_INL {
            constexpr int rt = decltype(RT)::value;
            float v[4][16];
            xw_for<0, 4>([&](auto CT) XW_INL {
                constexpr int ct = decltype(CT)::value;
                xw_for<0, 16>([&](auto R) XW_INL {
                    constexpr int r = decltype(R)::value;
                    v[ct][r] = xw_acc_rd<16 * (4 * rt + ct) + r>();
                });
            });
            xw_for<0, 64>([&](auto N) XW_INL { xw_acc_zero<64 * rt + decltype(N)::value>(); });
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int co = cw + 32 * rt + 8 * rr + hq;
                f32x4 add = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a.bias) add += ld4(a.bias + co);
                if (a.bbias) add += ld4(a.bbias + (long)c.b * a.bb_stride + co);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[ct][4 * rr + r] += add[r];
                    if (!STATS || a.Y)
                        st4(a.Y + ((long)c.b * a.P + p0 + 32 * ct + j) * a.ldy + co, (f32x4){v[ct][4 * rr], v[ct][4 * rr + 1], v[ct][4 * rr + 2], v[ct][4 * rr + 3]});
                }
            }
            if (STATS) {
                // per channel over the tile's 128 points = 4 column tiles x 32 lanes: mean first, then the squared deviations from it
                // (two passes over the registers), max and min.  The mean is all-reduced (every lane needs it for the second pass);
                // deviations / max / min go through the transposing reduction of x6w_common.h, the mean follows them by selection only
                float mean[16], q[16], mx[16], mn[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = (v[0][r] + v[1][r]) + (v[2][r] + v[3][r]);
                    mx[r] = fmaxf(fmaxf(v[0][r], v[1][r]), fmaxf(v[2][r], v[3][r]));
                    mn[r] = fminf(fminf(v[0][r], v[1][r]), fminf(v[2][r], v[3][r]));
                    s = xw_rows_add(row_allreduce_add<16>(s));
                    mean[r] = s * (1.0f / 128.0f);
                    float qq = 0.f;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const float d = v[ct][r] - mean[r];
                        qq = fmaf(d, d, qq);
                    }
                    q[r] = qq;
                }
                float mean2[2], q2[2], mx2[2], mn2[2];
                xw_treduce16(q, q2, XwAdd{});
                xw_treduce16(mx, mx2, XwMax{});
                xw_treduce16(mn, mn2, XwMin{});
                xw_treduce16(mean, mean2, XwFirst{});
                if ((lane & 3) == 0) {
                    const int bk = (lane >> 2) & 3, rho = (lane >> 4) & 1;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int r = 8 * i + 4 * (bk & 1) + 2 * (bk >> 1) + rho;
                        spart[wave * 128 + 32 * rt + 8 * (r >> 2) + hq + (r & 3)] = (f32x4){mean2[i], q2[i], mx2[i], mn2[i]};
                    }
                }
            }
        });
        if (STATS) {
            // spart is rewritten one whole tile (>= 2 chunk barriers) later: the copy-out below is long done by then
            __syncthreads();
            for (int ch = tid; ch < CW_TM; ch += 256) a.part[((long)c.b * a.Pt + c.pt) * a.part_stride + c.mt * CW_TM + ch] = spart[ch];
        }
    };
    auto step = [&]() XW_INL {        // c0 is done: read its tile out if that was the tile's last chunk, move the three positions on
        if (c0.kc == nk - 1) finish(c0);
        c0 = c1;
        c1 = c2;
        advance(c2);
    };
#pragma unroll 1
    for (int g = 0; g < total; g += 2) {
        chunk(0, xrb, xra);
        step();
        if (g + 1 < total) {
            chunk(1, xra, xrb);
            step();
        }
    }
}

// (Cout, ldw) f32 [+ column offset / count] -> [channel tile of 512][k-step][wave 4][row tile 4][plane 3][lane 64][8 bf16]: lane
// (i = l & 31, h = l >> 5) of fragment (mt, t, w, rt) holds row 512 mt + 128 w + 32 rt + i (zero beyond Cout), k = 16 t + 8 h + s
__global__ void pack_weight_x6w_kernel(const float *__restrict__ w, int ldw, int Cout, int col0, int Cin, unsigned char *__restrict__ out, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // ((mt * nks + t) * 16 + w * 4 + rt) * 64 + lane
    if (i >= total) return;
    const int l = (int)(i & 63), wr = (int)((i >> 6) & 15);
    const long mk = i >> 10;
    const int nks = Cin / 16;
    const int t = (int)(mk % nks), mt = (int)(mk / nks);
    const int row = mt * CW_TM + (wr >> 2) * 128 + (wr & 3) * 32 + (l & 31), hh = l >> 5;
    float hs[3][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float x = row < Cout ? w[(long)row * ldw + col0 + 16 * t + 8 * hh + s] : 0.f;
        xc_split(x, hs[0][s], hs[1][s], hs[2][s]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = xc_pack(hs[pl][2 * q], hs[pl][2 * q + 1]);
        *(u32x4 *)(out + ((mk * 16 + wr) * 3 + pl) * CW_FRAG + l * 16) = v;
    }
}

extern "C" long caspr_x6w_packed_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || Cin % 32) return 0;
    return (long)ceil_div(Cout, CW_TM) * (Cin / 16) * 16 * 3 * CW_FRAG;
}

extern "C" int caspr_pack_weight_x6w(const float *w, int ldw, int Cout, int col0, int ncols, void *packed, void *stream)
{
    CASPR_REQUIRE(w && packed && Cout > 0 && ncols > 0 && ncols % 32 == 0 && col0 >= 0 && ldw >= col0 + ncols, "pack_weight_x6w: bad arguments");
    CASPR_REQUIRE(((uintptr_t)packed % 16) == 0, "pack_weight_x6w: packed must be 16-byte aligned");
    const long total = (long)ceil_div(Cout, CW_TM) * (ncols / 16) * 16 * 64;
    pack_weight_x6w_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, ldw, Cout, col0, ncols, (unsigned char *)packed, total);
    CASPR_CHECK_LAUNCH("pack_weight_x6w");
    return CASPR_OK;
}

int caspr_conv_x6w_launch(const void *wpk, const float *bias, const float *bbias, int bb_stride, const float *X, int ldx, const float *in_scale,
                          const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B, int P, int Cin, int mt_begin, int mt_end,
                          void *part, int part_stride, int reserve_cus, hipStream_t stream) __attribute__((visibility("hidden")));

// channel tiles mt_begin .. mt_end - 1 (512 channels each) of the layer; reserve_cus: compute units left free for a kernel of another
// stream that is to run BESIDE this one (the persistent grid is static: a workgroup that has to wait for a unit delays its whole share)
int caspr_conv_x6w_launch(const void *wpk, const float *bias, const float *bbias, int bb_stride, const float *X, int ldx, const float *in_scale,
                          const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B, int P, int Cin, int mt_begin, int mt_end,
                          void *part, int part_stride, int reserve_cus, hipStream_t stream)
{
    ConvWArgs a;
    a.wpk = (const unsigned char *)wpk; a.bias = bias; a.bbias = bbias; a.X = X; a.Y = Y;
    a.part = (f32x4 *)part; a.ldx = ldx; a.ldy = ldy; a.P = P; a.Cin = Cin; a.Cout = mt_end * CW_TM; a.in_relu = in_relu; a.relu_from = in_relu_from;
    a.Mt = mt_end - mt_begin; a.mt0 = mt_begin; a.Pt = P / CW_TP; a.part_stride = part_stride; a.bb_stride = bb_stride;
    const long ntiles = (long)B * a.Mt * a.Pt;
    a.ntiles = (int)ntiles;
    // persistent: one workgroup per CU (512 registers per lane: one wave per SIMD), each walking through its share of the tiles
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, v = 0;
        (void)hipGetDevice(&dev);
        n_cu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    long grid = n_cu - (reserve_cus > 0 ? reserve_cus : 0);
    if (grid < 1) grid = 1;
    const long nblk = ntiles < grid ? ntiles : grid;
    const bool fused = in_scale != nullptr, stats = part != nullptr;
#define CW_GO(F, S) conv1x1_x6w_kernel<F, S><<<dim3((unsigned)nblk), dim3(256), CW_LDS, stream>>>(a, in_scale, in_shift)     /* 56 KB: below the 64 KB opt-in limit */
    if (fused && stats) CW_GO(true, true);
    else if (fused) CW_GO(true, false);
    else if (stats) CW_GO(false, true);
    else CW_GO(false, false);
#undef CW_GO
    return CASPR_OK;
}
