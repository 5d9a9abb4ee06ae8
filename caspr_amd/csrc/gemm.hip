// gemm.hip -- pointwise convolution (nn.Conv1d k=1 / nn.Linear) on the f32 MFMA pipe of gfx950,
// weight packing, and GroupNorm statistics.  Replaces the cuDNN/cuBLAS calls behind
// models/pointnet.py:37-41, models/pointnet2.py:525,247 and models/tpointnet2.py:99-105.
//
// conv1x1: Y[b,p,co] = act(sum_k W[co,k] * in(X[b,p,k]) + bias[co] + bbias[b,co]).
//   Block tile 128 (co) x 128 (points), K tile 32, 256 threads = 4x1 waves of 32x128 (2x8 MFMA
//   16x16x4 tiles, 64 accumulator VGPRs).  The point operand is staged through a double-buffered,
//   XOR-swizzled LDS B-tile (register staging, so the previous layer's GroupNorm+ReLU is applied
//   on the fly: one pass over the activations instead of three).  Weights are read straight from
//   the packed A-fragment stream (one coalesced 1-KiB dwordx4 load per wave per 16x16 tile per
//   16 k), prefetched one chunk ahead: they never touch LDS.
#include <stdlib.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------
extern "C" long caspr_packed_size(int Cout, int Cin)
{
    const long mt = (Cout + 15) / 16, kc = 2L * ((Cin + 31) / 32);
    return mt * kc * 256;
}

__global__ void pack_weight_kernel(const float *__restrict__ w, int ldw, int Cout, int col0, int ncols, int KC,
                                   float *__restrict__ packed, long total)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int q = (int)(t & 3), l = (int)((t >> 2) & 63);
    const long tile = t >> 8;
    const int kc = (int)(tile % KC), mt = (int)(tile / KC);
    const int row = mt * 16 + (l & 15), k = kc * 16 + 4 * (l >> 4) + q;
    packed[t] = (row < Cout && k < ncols) ? w[(long)row * ldw + col0 + k] : 0.0f;
}

extern "C" int caspr_pack_weight_f32(const float *w, int ldw, int Cout, int col0, int ncols, float *packed,
                                     void *stream)
{
    CASPR_REQUIRE(w && packed && Cout > 0 && ncols > 0 && col0 >= 0 && ldw >= col0 + ncols,
                  "pack_weight: bad arguments");
    const int KC = 2 * ((ncols + 31) / 32);
    const long total = caspr_packed_size(Cout, ncols);
    pack_weight_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        w, ldw, Cout, col0, ncols, KC, packed, total);
    CASPR_CHECK_LAUNCH("pack_weight");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// conv1x1
// ---------------------------------------------------------------------------------------------
#define GEMM_MT 128
#define GEMM_NT 128
#define GEMM_KT 32
#define GEMM_MAXC 2048
#ifndef GEMM_TOUCH_B
#define GEMM_TOUCH_B 1
#endif

// NT = points per block (128 or 64): wave tile 32 (co) x NT.  NT = 64 halves the accumulators (4 resident waves per
// SIMD instead of 3) and wastes nothing on the 64-point coarse levels; measured +1.4 % on the 1600x1600 layer, -0.9 ms
// on the cfg-2 step, -8 ms on the training step, so it is the default (CASPR_GEMM_KERNEL=1 selects NT = 128).
template <int NT>
__global__ __launch_bounds__(256) void conv1x1_kernel(const float *__restrict__ wp, const float *__restrict__ bias,
                                                      const float *__restrict__ bbias, const float *__restrict__ X,
                                                      int ldx, const float *__restrict__ in_scale,
                                                      const float *__restrict__ in_shift, int in_relu,
                                                      int relu_from, float *__restrict__ Y, int ldy, int P, int Cin,
                                                      int Cout, int act, unsigned long long *trace)
{
    __shared__ __attribute__((aligned(16))) float sB[2][8 * NT * 4];  // 2 x (NT/8) KiB (K tile 32 = 2 chunks; K tile 64 measured slower: fewer blocks per CU)
    // per-batch GroupNorm scale / shift of the input channels, staged once per block: reading them from global
    // memory inside store_stage exposed an L2 round trip per K tile (tools/gemm_phase_trace.py: 3.6-5.9k of 12.5k cycles)
    __shared__ __attribute__((aligned(16))) float sSS[2][GEMM_MAXC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 4 x 1 wave layout: each wave owns 32 output channels x all 128 points.  The weight fragments (streamed from
    // L2, the scarce per-CU resource at ~10-12 B/clk) are then fetched exactly once per block; the activation
    // fragments are re-read by all four waves, but from LDS.  (2 x 2 waves fetched every weight fragment twice.)
    const int wm = wave;
    const int g = lane >> 4, j = lane & 15;
    // XCD-aware work mapping.  The dispatcher places linear block L on XCD L % 8 (MI355X_MICROARCH.md); each XCD has
    // a private 4 MiB L2.  Work items are renumbered so that XCD k owns a contiguous range of point tiles and walks
    // all output-channel tiles of one point tile back to back: the activation tile (the big operand: 128 x Cin x 4 B)
    // is fetched into ONE L2 and re-used by its Mt consumers there, instead of being fetched by up to 8 L2s.
    // Placement only changes speed, never results.
    const int Mt = gridDim.x, Pt = gridDim.y;
    const int nblk = Mt * Pt * gridDim.z;
    const int lin = blockIdx.x + Mt * (blockIdx.y + Pt * blockIdx.z);
    const int xcd = lin & 7, slot = lin >> 3;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int wmt = work % Mt, wpt = (work / Mt) % Pt;
    const int b = work / (Mt * Pt);
    const int p0 = wpt * NT;
    const int co0 = wmt * GEMM_MT;
    const int KC = 2 * ((Cin + 31) / 32);
    const int MT16 = (Cout + 15) / 16;
    const int ntiles = KC / 2;

    const float *Xb = X + (long)b * P * ldx;
    const float *sc = in_scale ? in_scale + (long)b * Cin : nullptr;
    const float *sh = in_scale ? in_shift + (long)b * Cin : nullptr;
    const bool ss_lds = sc && Cin <= GEMM_MAXC;
    if (ss_lds) {
        for (int c = tid * 4; c < Cin; c += 1024) {   // Cin % 4 == 0 when the transform is fused
            st4(&sSS[0][c], ld4(sc + c));
            st4(&sSS[1][c], ld4(sh + c));
        }
        __syncthreads();   // the first store_stage below reads sSS
    }

    // which 16-row tiles of the packed stream this wave owns (wave-uniform validity)
    // Weight fragments: buffer loads (SGPR resource + SGPR tile/chunk offset + one lane-offset VGPR), no VALU
    // address arithmetic in the MFMA shadow.  K tiles are walked in an order rotated by the point-tile index so
    // that the blocks sharing a weight slab do not hit the same L2 channel in lockstep (fp32 sums reassociate
    // per point tile, independent of the batch index).
    const int mt0 = (co0 >> 4) + wm * 2;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, MT16 * KC * 1024, 0x00020000);
    const int wvoff = lane * 16;
    int wsoff[2];
    bool mvalid[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        mvalid[mi] = (mt0 + mi) < MT16;
        wsoff[mi] = (mvalid[mi] ? mt0 + mi : 0) * KC * 1024;
    }
    // CASPR_CONV_ROW_INVARIANT: every row tile walks K in the same order (a row's result must not depend on which tile it is in)
    const int rot = (act & CASPR_CONV_ROW_INVARIANT) ? 0 : wpt % ntiles;

    f32x4 acc[2][NT / 16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT / 16; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging assignment: float4 f = tid + 256*i  ->  kq = f & 7, col = f >> 3
    f32x4 stage[NT / 32];
    // The global loads only ISSUE here (raw data stays in flight across the MFMA block); masking and the fused
    // GroupNorm+ReLU transform run in store_stage, when the data has long arrived -- applying them at load time
    // would put an s_waitcnt vmcnt(0) in front of the MFMAs.
    auto load_stage = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NT / 32; ++i) {
            const int f = tid + 256 * i;
            const int kq = f & 7, col = f >> 3;
            const int k = kt * GEMM_KT + kq * 4;
            const int p = p0 + col;
            const bool ok = p < P && k < Cin;
            stage[i] = ld4(Xb + (ok ? ((long)p * ldx + k) : 0));   // ldx >= roundup4(Cin): the tail quad is readable
        }
    };
    auto store_stage = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < NT / 32; ++i) {
            const int f = tid + 256 * i;
            const int kq = f & 7, col = f >> 3;
            const int k = kt * GEMM_KT + kq * 4;
            const int p = p0 + col;
            f32x4 v = stage[i];
            if (p < P && k < Cin) {
                if (sc) {
                    const f32x4 s4 = ss_lds ? ld4(&sSS[0][k]) : ld4(sc + k), t4 = ss_lds ? ld4(&sSS[1][k]) : ld4(sh + k);
                    v = v * s4 + t4;
                    if (in_relu && k >= relu_from) {  // relu_from % 4 == 0
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k + q >= Cin) v[q] = 0.f;
            } else {
                v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            st4(&sB[buf][btile_off(kq, col, NT)], v);
        }
    };

    f32x4 a0[2], a1[2], b0[NT / 16];   // one activation-fragment set: a second one costs 32 VGPRs = the third resident block per CU
    auto load_a = [&](f32x4(&a)[2], int kc) {   // kc = chunk index in the packed stream (already rotated)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            a[mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff[mi] + kc * 1024, 0));
    };
    auto tile_of = [&](int it) { const int t = it + rot; return t >= ntiles ? t - ntiles : t; };
    auto load_b = [&](f32x4(&bf)[NT / 16], int buf, int c) {
#pragma unroll
        for (int ni = 0; ni < NT / 16; ++ni) bf[ni] = ld4(&sB[buf][btile_off(c * 4 + g, ni * 16 + j, NT)]);
    };
    float lds_one = 1.0f;
    asm volatile("" : "+v"(lds_one));
    auto touch_b = [&](f32x4(&bf)[NT / 16]) {   // experiment: a VALU op between the LDS reads and the MFMAs
        if (GEMM_TOUCH_B) {
#pragma unroll
            for (int ni = 0; ni < NT / 16; ++ni) bf[ni] = bf[ni] * lds_one;
        }
    };
    auto mma = [&](const f32x4(&a)[2], const f32x4(&bf)[NT / 16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT / 16; ++ni) acc[mi][ni] = mfma16(a[mi][q], bf[ni][q], acc[mi][ni]);
    };

    load_stage(tile_of(0));
    load_a(a0, tile_of(0) * 2);
    store_stage(0, tile_of(0));
    __syncthreads();

    // Two register sets per operand: chunk c+1's weight (L2) and activation (LDS) fragments are in flight while
    // chunk c's 64 MFMAs run.  sched_barrier pins the issue point: hipcc otherwise sinks loads to their first use.
    for (int it = 0; it < ntiles; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < ntiles;
        const int kt = tile_of(it), ktn = tile_of(more ? it + 1 : it);
#ifdef CASPR_DEBUG_HOOKS
#define GEMM_STAMP(i) if (trace && wmt == 3 && wpt == 5 && b == 0 && tid == 0 && it >= 8 && it < 12) trace[(it - 8) * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define GEMM_STAMP(i)
#endif
        GEMM_STAMP(0)
        if (more) load_stage(ktn);             // next tile's global loads stay in flight during this tile's MFMAs
        load_b(b0, buf, 0);
        load_a(a1, kt * 2 + 1);
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STAMP(1)
        touch_b(b0);
        mma(a0, b0);
        GEMM_STAMP(2)
        load_b(b0, buf, 1);
        load_a(a0, ktn * 2);
        __builtin_amdgcn_sched_barrier(0);
        touch_b(b0);
        mma(a1, b0);
        GEMM_STAMP(3)
        if (more) {
            store_stage(buf ^ 1, ktn);
            GEMM_STAMP(4)
            __syncthreads();
            GEMM_STAMP(5)
        }
    }

    // epilogue: lane holds co = co0 + wm*32 + mi*16 + 4g + r (r = 0..3) for point p0 + ni*16 + j
    const float *bb = bbias ? bbias + (long)b * Cout : nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        if (!mvalid[mi]) continue;
        const int co = co0 + wm * 32 + mi * 16 + 4 * g;
        float add[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if (co + r < Cout) {
                if (bias) v += bias[co + r];
                if (bb) v += bb[co + r];
            }
            add[r] = v;
        }
#pragma unroll
        for (int ni = 0; ni < NT / 16; ++ni) {
            const int p = p0 + ni * 16 + j;
            if (p >= P) continue;
            f32x4 v = acc[mi][ni];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] += add[r];
                if ((act & 0xff) == 1) v[r] = sigmoid_f(v[r]);
            }
            float *dst = Y + ((long)b * P + p) * ldy + co;
            if (co + 3 < Cout) {
                st4(dst, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < Cout) dst[r] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv1x1, streaming variant: no LDS tiles, no barrier in the K loop.
// Diagnostics on conv1x1_kernel (skipping parts of its K loop) showed the activation staging path -- global load ->
// transform -> LDS store -> workgroup barrier -- costing more than anything else (139 TFLOP/s without it and without the
// weight loads, 117 with staging only, 127 with weight loads only, 114 with both).  In the point-major layout the B
// fragment of lane (g, j) for a 16-k chunk IS a contiguous 16-byte piece of row p = p0 + ni*16 + j (channels 16kc+4g..+3),
// so each wave can fetch its own activation fragments straight from global memory (the second chunk of a 128-byte line
// hits L1) and nothing is shared between waves: 1 x 4 waves, each 128 (co) x 32 (points) = 8 x 2 MFMA tiles.  Every wave
// streams the whole 128-row weight slab (4x the fragment traffic of the 4 x 1 layout -- measured harmless above).
// ---------------------------------------------------------------------------------------------
#ifndef ST_TOUCH_A
#define ST_TOUCH_A 0   // the same VALU touch on the weight fragments: within noise (+3 % unfused, -1.5 % fused)
#endif
#ifndef ST_DB
#define ST_DB 6   // activation fragment sets in flight (chunks)
#endif
#ifndef ST_DA
#define ST_DA 3   // weight fragment sets in flight (chunks); ST_DB must be a multiple of ST_DA and even
#endif

// NMI = row tiles of 16 output channels per wave: 8, or 1 for convs with up to 16 outputs (the T-NOCS regression 1600 -> 4,
// tpointnet2.py:105: with the 128-row tile 7/8 of its MFMAs multiplied padding and the layer ran at 1.9 TB/s of its 2.1 GB input)
template <bool FUSED, int NMI>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const float *__restrict__ wp, const float *__restrict__ bias,
                                                                const float *__restrict__ bbias, const float *__restrict__ X,
                                                                int ldx, const float *__restrict__ in_scale,
                                                                const float *__restrict__ in_shift, int in_relu, int relu_from,
                                                                float *__restrict__ Y, int ldy, int P, int Cin, int Cout, int act)
{
    __shared__ __attribute__((aligned(16))) float sSS[2][GEMM_MAXC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int Mt = gridDim.x, Pt = gridDim.y;
    const int nblk = Mt * Pt * gridDim.z;
    const int lin = blockIdx.x + Mt * (blockIdx.y + Pt * blockIdx.z);
    const int xcd = lin & 7, slot = lin >> 3;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int wmt = work % Mt, wpt = (work / Mt) % Pt;
    const int b = work / (Mt * Pt);
    const int p0 = wpt * 128 + wave * 32;
    const int co0 = wmt * GEMM_MT;
    const int KC = 2 * ((Cin + 31) / 32);
    const int MT16 = (Cout + 15) / 16;
    const int Cin4 = (Cin + 3) & ~3;

    const float *sc = in_scale ? in_scale + (long)b * Cin : nullptr;
    const float *sh = in_scale ? in_shift + (long)b * Cin : nullptr;
    if (sc) {
        for (int c = tid * 4; c < Cin; c += 1024) {
            st4(&sSS[0][c], ld4(sc + c));
            st4(&sSS[1][c], ld4(sh + c));
        }
        __syncthreads();
    }

    const int mt0 = co0 >> 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, MT16 * KC * 1024, 0x00020000);
    const int wvoff = lane * 16;
    int wsoff[NMI];
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) wsoff[mi] = ((mt0 + mi) < MT16 ? mt0 + mi : 0) * KC * 1024;
    // activation rows of this lane (two column tiles); rows past P read row P-1 (finite data, columns never stored)
    const float *xrow[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int p = p0 + ni * 16 + j;
        xrow[ni] = X + ((long)b * P + (p < P ? p : P - 1)) * ldx + 4 * g;
    }

    f32x4 acc[NMI][2];
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 af[ST_DA][NMI], bfr[ST_DB][2];
    auto load_a = [&](int set, int kc) {
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi)
            af[set][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff[mi] + kc * 1024, 0));
    };
    // Whole chunks (all 16 k below Cin) run in an unconditional, ST_DB-times unrolled loop; the ragged end (a partial
    // chunk and the zero padding up to an even chunk count) goes through the masked path.
    auto load_b = [&](int set, int kc) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bfr[set][ni] = ld4(xrow[ni] + 16 * kc);
    };
    auto load_b_edge = [&](int set, int kc) {
        const bool in = 16 * kc + 4 * g < Cin4;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bfr[set][ni] = in ? ld4(xrow[ni] + 16 * kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    float opaque_one = 1.0f;
    asm volatile("" : "+v"(opaque_one));   // the compiler must not fold the multiply below away
    auto fix_b = [&](int set, int kc) {
        if (!FUSED) {
            // Without the fused transform the MFMAs would read the registers the global loads returned into directly;
            // measured 103-108 TFLOP/s that way versus 116-126 when a VALU op sits in between (as in the fused path).
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bfr[set][ni] = bfr[set][ni] * opaque_one;
        }
        if (FUSED) {
            const int k = 16 * kc + 4 * g;
            const f32x4 s4 = ld4(&sSS[0][k]), t4 = ld4(&sSS[1][k]);
            const bool relu = in_relu && k >= relu_from;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                f32x4 v = bfr[set][ni] * s4 + t4;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (relu && !(v[q] > 0.f)) ? 0.f : v[q];
                bfr[set][ni] = v;
            }
        }
    };
    auto fix_b_edge = [&](int set, int kc) {
        const int k = 16 * kc + 4 * g;
        if (k < Cin) {
            if (FUSED) fix_b(set, kc);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k + q >= Cin) bfr[set][ni][q] = 0.f;
        } else {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bfr[set][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto mma = [&](int aset, int bset) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma16(af[aset][mi][q], bfr[bset][ni][q], acc[mi][ni]);
    };

    const int nfull = (Cin / 16) / ST_DB * ST_DB;   // chunks handled by the unconditional loop (multiple of ST_DB, even)
    if (nfull > 0) {
#pragma unroll
        for (int d = 0; d < ST_DB - 1; ++d) load_b(d, d);   // nfull >= ST_DB
#pragma unroll
        for (int d = 0; d < ST_DA - 1; ++d) load_a(d, d);
        for (int kc0 = 0; kc0 < nfull; kc0 += ST_DB) {
#pragma unroll
            for (int d = 0; d < ST_DB; ++d) {
                const int kc = kc0 + d;
                // prefetch: activations ST_DB-1 chunks ahead, weights ST_DA-1 chunks ahead (clamped re-loads at the very
                // end keep the loop free of branches; their results are overwritten or unused)
                const int kb = kc + ST_DB - 1 < nfull ? kc + ST_DB - 1 : nfull - 1;
                const int ka = kc + ST_DA - 1 < KC ? kc + ST_DA - 1 : KC - 1;
                load_b((d + ST_DB - 1) % ST_DB, kb);
                load_a((d + ST_DA - 1) % ST_DA, ka);
                __builtin_amdgcn_sched_barrier(0);
                fix_b(d, kc);
                if (ST_TOUCH_A) {
#pragma unroll
                    for (int mi = 0; mi < NMI; ++mi) af[d % ST_DA][mi] = af[d % ST_DA][mi] * opaque_one;
                }
                mma(d % ST_DA, d);
            }
        }
    }
    // ragged end (KC and nfull are even): masked, shallow prefetch, static register sets.  Weight chunks nfull .. nfull+ST_DA-2
    // may already sit in sets 0 .. ST_DA-2 from the loop above; they are simply fetched again.
    for (int kc = nfull; kc < KC; kc += 2) {
        load_a(0, kc);
        load_a(1, kc + 1);
        load_b_edge(0, kc);
        load_b_edge(1, kc + 1);
        fix_b_edge(0, kc);
        mma(0, 0);
        fix_b_edge(1, kc + 1);
        mma(1, 1);
    }

    // epilogue: lane holds co = co0 + mi*16 + 4g + r for point p0 + ni*16 + j
    const float *bb = bbias ? bbias + (long)b * Cout : nullptr;
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {
        if (mt0 + mi >= MT16) continue;
        const int co = co0 + mi * 16 + 4 * g;
        float add[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if (co + r < Cout) {
                if (bias) v += bias[co + r];
                if (bb) v += bb[co + r];
            }
            add[r] = v;
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int p = p0 + ni * 16 + j;
            if (p >= P) continue;
            f32x4 v = acc[mi][ni];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] += add[r];
                if ((act & 0xff) == 1) v[r] = sigmoid_f(v[r]);
            }
            float *dst = Y + ((long)b * P + p) * ldy + co;
            if (co + 3 < Cout) {
                st4(dst, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < Cout) dst[r] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv1x1, narrow inputs (Cin < 32 * ST_DB = 192) over many rows: the set-abstraction MLPs of the TRAINING encoder run as plain convs
// over the grouped rows (1.3 - 2.6 M rows of 9 - 131 channels into 16 - 128, forward and data gradient; the inference path has them
// in sa_mlp.hip).  Such a call is a pure stream -- 2.6 M x (32 + 32) floats in and out against 5 GFLOP -- and on the LDS-tiled kernel
// (128 output channels x 64 points per workgroup, every K tile staged through LDS behind a barrier) it moved ~2 TB/s.  Here a wave
// owns 32 rows and ALL the output channels of its 128-channel slab (NMI = 1 / 2 / 4 / 8 row tiles, chosen from Cout on the host: no
// MFMAs on padding tiles), issues every activation load of its rows up front (at most 12 chunks x 2 column tiles of 16 bytes: the whole
// K extent is in flight before the first MFMA) and reads the weight fragments from the pack as it goes (a few KB, L1 resident).
// Products are accumulated in ascending k like the other f32 kernels.
// ---------------------------------------------------------------------------------------------
#define NW_MAXKC 12
template <int NMI>
__global__ __launch_bounds__(256, 2) void conv1x1_narrow_kernel(const float *__restrict__ wp, const float *__restrict__ bias,
                                                                const float *__restrict__ bbias, const float *__restrict__ X,
                                                                int ldx, float *__restrict__ Y, int ldy, int P, int Cin, int Cout,
                                                                int act)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.z;
    const int p0 = blockIdx.y * 128 + wave * 32;
    const int co0 = blockIdx.x * (NMI * 16);
    const int KC = 2 * ((Cin + 31) / 32);
    const int MT16 = (Cout + 15) / 16;
    const int Cin4 = (Cin + 3) & ~3;
    const int mt0 = co0 >> 4;
    if (p0 >= P) return;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, MT16 * KC * 1024, 0x00020000);
    const int wvoff = lane * 16;
    int wsoff[NMI];
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) wsoff[mi] = ((mt0 + mi) < MT16 ? mt0 + mi : 0) * KC * 1024;
    const float *xrow[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int p = p0 + ni * 16 + j;
        xrow[ni] = X + ((long)b * P + (p < P ? p : P - 1)) * ldx + 4 * g;
    }

    f32x4 bfr[NW_MAXKC][2];
#pragma unroll
    for (int kc = 0; kc < NW_MAXKC; ++kc) {
        if (kc < KC) {
            const int k = 16 * kc + 4 * g;
            const bool in = k < Cin4;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bfr[kc][ni] = in ? ld4(xrow[ni] + 16 * kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 acc[NMI][2];
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kc = 0; kc < NW_MAXKC; ++kc) {
        if (kc < KC) {
            f32x4 af[NMI];
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
                af[mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff[mi] + kc * 1024, 0));
            const int k = 16 * kc + 4 * g;
            if (k + 4 > Cin) {   // the padding columns of the last float4 hold whatever the producer left there
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (k + q >= Cin) bfr[kc][ni][q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma16(af[mi][q], bfr[kc][ni][q], acc[mi][ni]);
        }
    }

    const float *bb = bbias ? bbias + (long)b * Cout : nullptr;
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {
        if (mt0 + mi >= MT16) continue;
        const int co = co0 + mi * 16 + 4 * g;
        float add[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if (co + r < Cout) {
                if (bias) v += bias[co + r];
                if (bb) v += bb[co + r];
            }
            add[r] = v;
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int p = p0 + ni * 16 + j;
            if (p >= P) continue;
            f32x4 v = acc[mi][ni];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] += add[r];
                if ((act & 0xff) == 1) v[r] = sigmoid_f(v[r]);
            }
            float *dst = Y + ((long)b * P + p) * ldy + co;
            if (co + 3 < Cout) {
                st4(dst, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < Cout) dst[r] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv1x1 over at most 16 rows per batch entry (the latent ODE's layers in training: 8 sequences per GPU, 576 products per step of
// 64-512 channels; the per-sequence bias of the head conv): one workgroup per (16 output channels, batch entry) -- the choice of
// this kernel depends on P only, never on B, so a batch entry's result does not depend on the batch around it --, the K chunks dealt round-robin to its four waves (A fragment straight from
// the packed weight, B fragment = 16 bytes of the lane's row: columns past P read row P-1 and are not stored), partial tiles
// summed in wave order through LDS.  The tiled kernels above spend ~36 us on such a call (four workgroups walking K = 512 through
// their LDS stages); this one is a few microseconds of weight streaming.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1x1_skinny_kernel(const float *__restrict__ wp, const float *__restrict__ bias,
                                                             const float *__restrict__ bbias, const float *__restrict__ X, int ldx,
                                                             float *__restrict__ Y, int ldy, int P, int Cin, int Cout, int act)
{
    __shared__ __attribute__((aligned(16))) float s_part[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int mt = blockIdx.x, b = blockIdx.y;
    const int KC = 2 * ((Cin + 31) / 32);
    const int Cin4 = (Cin + 3) & ~3;
    const float *xrow = X + ((long)b * P + (j < P ? j : P - 1)) * ldx + 4 * g;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc = wave; kc < KC; kc += 4) {
        const f32x4 af = ld4(wp + (((long)mt * KC + kc) * 64 + lane) * 4);     // zero beyond Cout / Cin (pack_weight_kernel)
        const int k = 16 * kc + 4 * g;
        f32x4 bf = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (k < Cin4) bf = ld4(xrow + 16 * kc);                               // ldx >= roundup4(Cin); the pad columns meet zero weights
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k + q >= Cin) bf[q] = 0.f;                                    // (they may hold anything)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = mfma16(af[q], bf[q], acc);
    }
    st4(&s_part[wave][lane * 4], acc);
    __syncthreads();
    if (wave == 0 && j < P) {
        f32x4 v = (ld4(&s_part[0][lane * 4]) + ld4(&s_part[1][lane * 4])) + (ld4(&s_part[2][lane * 4]) + ld4(&s_part[3][lane * 4]));
        const int co = mt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (co + r >= Cout) continue;
            float o = v[r] + (bias ? bias[co + r] : 0.f);
            if (bbias) o += bbias[(long)b * Cout + co + r];
            if ((act & 0xff) == 1) o = sigmoid_f(o);
            Y[((long)b * P + j) * ldy + co + r] = o;
        }
    }
}

#ifdef CASPR_DEBUG_HOOKS
extern "C" int caspr_debug_gemm_occupancy(void)   // debug build only: resident conv1x1_kernel blocks per CU
{
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)conv1x1_kernel<64>, 256, 0);
    return n;
}
static unsigned long long *g_gemm_trace = nullptr;
extern "C" void caspr_debug_set_gemm_trace(unsigned long long *dev_buf) { g_gemm_trace = dev_buf; }   // debug build only
#define GEMM_TRACE_PTR g_gemm_trace
#else
#define GEMM_TRACE_PTR nullptr
#endif

extern "C" int caspr_conv1x1_f32(const float *wp, const float *bias, const float *bbias, const float *X, int ldx,
                                 const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y,
                                 int ldy, int B, int P, int Cin, int Cout, int act, void *stream)
{
    CASPR_REQUIRE(wp && X && Y && B > 0 && P > 0 && Cin > 0 && Cout > 0, "conv1x1: bad arguments");
    CASPR_REQUIRE(ldx % 4 == 0 && ldx >= ((Cin + 3) & ~3), "conv1x1: ldx=%d must be a multiple of 4 and >= roundup4(Cin=%d)", ldx, Cin);
    CASPR_REQUIRE(ldy % 4 == 0 && ldy >= Cout, "conv1x1: ldy=%d must be a multiple of 4 and >= Cout=%d", ldy, Cout);
    CASPR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv1x1: in_scale/in_shift must be given together");
    CASPR_REQUIRE(in_scale == nullptr || Cin % 4 == 0, "conv1x1: fused input GroupNorm needs Cin %% 4 == 0 (Cin=%d)", Cin);
    CASPR_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)wp % 16) == 0, "conv1x1: pointers must be 16-byte aligned");
    CASPR_REQUIRE(B <= 65535, "conv1x1: B=%d > 65535", B);
    CASPR_REQUIRE(in_relu_from >= 0 && in_relu_from % 4 == 0, "conv1x1: in_relu_from=%d must be a non-negative multiple of 4", in_relu_from);
    const int force = CASPR_DEBUG_ENV_INT("CASPR_GEMM_KERNEL");   // debug build only: 1 = 128-point tiles, 7 = streaming, 8 = narrow where its contract holds (experiments)
    const size_t lds_pad = (size_t)CASPR_DEBUG_ENV_INT("CASPR_GEMM_LDS_PAD") * 1024;   // occupancy experiments, debug build only
    CASPR_REQUIRE(ceil_div(P, 128) <= 65535, "conv1x1: P=%d rows per batch entry exceed the grid (split the call)", P);
    if (force == 0 && P <= 16 && !in_scale && !(act & CASPR_CONV_ROW_INVARIANT)) {
        conv1x1_skinny_kernel<<<dim3(ceil_div(Cout, 16), B), dim3(256), 0, (hipStream_t)stream>>>(wp, bias, bbias, X, ldx, Y, ldy, P, Cin, Cout,
                                                                                                  act);
        CASPR_CHECK_LAUNCH("conv1x1(skinny)");
        return CASPR_OK;
    }
    // default: the streaming kernel wherever a wave's 32 points and the unrolled K loop are filled; the LDS-tiled kernel
    // for short rows-per-batch (coarse levels) and narrow inputs (set-abstraction MLPs).  CASPR_GEMM_KERNEL: 1 / 4 force
    // the LDS kernel with 128 / 64-point tiles, 7 forces streaming.
    // row-invariant calls always take the LDS kernel (the kernel choice itself must not depend on P)
    // narrow inputs over many rows (the training encoder's set-abstraction convs): the whole K extent of a wave's rows in flight at once
    if ((force == 0 || force == 8) && P >= 128 && Cin < 32 * ST_DB && !in_scale && !(act & CASPR_CONV_ROW_INVARIANT)) {
        const int mt16 = ceil_div(Cout, 16);
        const int nmi = mt16 >= 5 ? 8 : mt16 >= 3 ? 4 : mt16;
        dim3 grid(ceil_div(mt16, nmi), ceil_div(P, 128), B);
#define NW_LAUNCH(N) conv1x1_narrow_kernel<N><<<grid, dim3(256), 0, (hipStream_t)stream>>>(wp, bias, bbias, X, ldx, Y, ldy, P, Cin, Cout, act)
        if (nmi == 8) NW_LAUNCH(8);
        else if (nmi == 4) NW_LAUNCH(4);
        else if (nmi == 2) NW_LAUNCH(2);
        else NW_LAUNCH(1);
#undef NW_LAUNCH
        CASPR_CHECK_LAUNCH("conv1x1(narrow)");
        return CASPR_OK;
    }
    if (force == 7 || (force == 0 && P >= 128 && Cin >= 32 * ST_DB && !(act & CASPR_CONV_ROW_INVARIANT))) {
        dim3 grid(ceil_div(Cout, GEMM_MT), ceil_div(P, 128), B);
        // The narrow variant (<= 16 outputs: the T-NOCS regression, a pure stream over 2.1 GB) is launched POLITE: 16 KB of unused dynamic LDS on
        // top of its 16 KB, so that at most five of its workgroups share a compute unit (it needs ~8 MB in flight to saturate HBM and has
        // 37 MB with three) instead of ten that fill every wave slot -- the small kernels of another stream (the flow's context cat and
        // hyper-network conv, which the flow waits for) then find slots at once instead of queueing behind 2,560 workgroups -- and exactly one
        // still fits beside a workgroup of the flow (127 KB).
        const size_t st_pad = Cout <= 16 ? 16 * 1024 : 0;
#define ST_LAUNCH(F, N) conv1x1_stream_kernel<F, N><<<grid, dim3(256), st_pad, (hipStream_t)stream>>>(wp, bias, bbias, X, ldx, in_scale, in_shift, in_relu, \
                                                                                    in_relu_from, Y, ldy, P, Cin, Cout, act)
        if (Cout <= 16) {
            if (in_scale) ST_LAUNCH(true, 1);
            else ST_LAUNCH(false, 1);
        } else {
            if (in_scale) ST_LAUNCH(true, 8);
            else ST_LAUNCH(false, 8);
        }
#undef ST_LAUNCH
        CASPR_CHECK_LAUNCH("conv1x1(stream)");
        return CASPR_OK;
    }
    if (force == 1 || ceil_div(P, 64) > 65535) {
        dim3 grid(ceil_div(Cout, GEMM_MT), ceil_div(P, 128), B);
        conv1x1_kernel<128><<<grid, dim3(256), lds_pad, (hipStream_t)stream>>>(wp, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from,
                                                                               Y, ldy, P, Cin, Cout, act, GEMM_TRACE_PTR);
    } else {
        dim3 grid(ceil_div(Cout, GEMM_MT), ceil_div(P, 64), B);
        conv1x1_kernel<64><<<grid, dim3(256), lds_pad, (hipStream_t)stream>>>(wp, bias, bbias, X, ldx, in_scale, in_shift, in_relu, in_relu_from,
                                                                              Y, ldy, P, Cin, Cout, act, GEMM_TRACE_PTR);
    }
    CASPR_CHECK_LAUNCH("conv1x1");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics (+ optional max over points of the normalised output).
// Pass 1: grid (G, S, B): one workgroup per (group, point split, batch) accumulates sum / sum of
// squares in f64 and per-channel max / min.  Pass 2: one thread per (b, c) combines the S partials in
// a fixed order (deterministic) and emits scale / shift / pmax.
// ---------------------------------------------------------------------------------------------
#define GN_SPLIT 1024

extern "C" long caspr_gn_ws_bytes(int B, int P, int C, int G)
{
    const long S = (P + GN_SPLIT - 1) / GN_SPLIT;
    return (long)B * G * S * 16 + (long)B * C * S * 8 + 64;
}

__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ Y, int ldy, int P, int C, int G,
                                                         double *__restrict__ psum, float *__restrict__ pmm)
{
    __shared__ double s_sum[256], s_sq[256];
    __shared__ float s_mx[256 * 4], s_mn[256 * 4];
    const int g = blockIdx.x, s = blockIdx.y, b = blockIdx.z, S = gridDim.y;
    const int cpg = C / G, Q4 = cpg >> 2;  // host guarantees cpg % 4 == 0 and Q4 <= 256
    const int TP = 256 / Q4;
    const int tq = threadIdx.x % Q4, tp = threadIdx.x / Q4;
    const int pbeg = s * GN_SPLIT, pend = (pbeg + GN_SPLIT) < P ? (pbeg + GN_SPLIT) : P;
    double sum = 0.0, sq = 0.0;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    if (tp < TP) {
        const float *base = Y + (long)b * P * ldy + g * cpg + tq * 4;
#pragma unroll 4
        for (int p = pbeg + tp; p < pend; p += TP) {
            const f32x4 v = ld4(base + (long)p * ldy);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sum += (double)v[q];
                sq += (double)v[q] * (double)v[q];
                mx[q] = v[q] > mx[q] ? v[q] : mx[q];
                mn[q] = v[q] < mn[q] ? v[q] : mn[q];
            }
        }
    }
    s_sum[threadIdx.x] = sum;
    s_sq[threadIdx.x] = sq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s_mx[threadIdx.x * 4 + q] = mx[q];
        s_mn[threadIdx.x * 4 + q] = mn[q];
    }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
            s_sq[threadIdx.x] += s_sq[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double *o = psum + (((long)b * G + g) * S + s) * 2;
        o[0] = s_sum[0];
        o[1] = s_sq[0];
    }
    if (threadIdx.x < cpg) {  // one thread per channel of the group folds the TP row partials
        const int c = threadIdx.x, q4 = c >> 2, q = c & 3;
        float m1 = -INFINITY, m0 = INFINITY;
        for (int r = 0; r < TP; ++r) {
            const float a = s_mx[(r * Q4 + q4) * 4 + q], bb = s_mn[(r * Q4 + q4) * 4 + q];
            m1 = a > m1 ? a : m1;
            m0 = bb < m0 ? bb : m0;
        }
        float *o = pmm + (((long)b * C + g * cpg + c) * S + s) * 2;
        o[0] = m1;
        o[1] = m0;
    }
}

// The same partials with ROW-contiguous reads, for C <= 1024: one workgroup per (point split, batch entry) covers all channels --
// thread (tp, tq) walks rows tp, tp + TP, ... of the split with the 16 bytes of channel quad tq, so a wave reads whole rows
// (gn_partial_kernel above gives every group its own workgroup: at 4 channels per group each reads 16 bytes per row, a quarter
// of every 64-byte sector it touches, and the 16 workgroups of a split fetch the same rows 16 times: 0.8 TB/s on the
// 327,680 x 64 layer of the global PointNet, 103 us; this form 1/4 of that).  Same sums in the same f64 arithmetic; the order of
// the additions differs (fixed, so still deterministic).
__global__ __launch_bounds__(256) void gn_partial_rows_kernel(const float *__restrict__ Y, int ldy, int P, int C, int G,
                                                              double *__restrict__ psum, float *__restrict__ pmm)
{
    __shared__ double s_sum[256], s_sq[256];
    __shared__ float s_mx[256 * 4], s_mn[256 * 4];
    const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
    const int CQ = C >> 2, cpg = C / G, qpg = cpg >> 2;     // host: C % 4 == 0, cpg % 4 == 0, CQ <= 256
    const int TP = 256 / CQ;
    const int tq = threadIdx.x % CQ, tp = threadIdx.x / CQ;
    const int pbeg = s * GN_SPLIT, pend = (pbeg + GN_SPLIT) < P ? (pbeg + GN_SPLIT) : P;
    double sum = 0.0, sq = 0.0;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    if (tp < TP) {
        const float *base = Y + (long)b * P * ldy + tq * 4;
#pragma unroll 8
        for (int p = pbeg + tp; p < pend; p += TP) {
            const f32x4 v = ld4(base + (long)p * ldy);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sum += (double)v[q];
                sq += (double)v[q] * (double)v[q];
                mx[q] = v[q] > mx[q] ? v[q] : mx[q];
                mn[q] = v[q] < mn[q] ? v[q] : mn[q];
            }
        }
    }
    s_sum[threadIdx.x] = sum;
    s_sq[threadIdx.x] = sq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s_mx[threadIdx.x * 4 + q] = mx[q];
        s_mn[threadIdx.x * 4 + q] = mn[q];
    }
    __syncthreads();
    if (threadIdx.x < G) {        // one thread per group: its quads x the TP row lanes, in index order
        const int g = threadIdx.x;
        double a0 = 0.0, a1 = 0.0;
        for (int r = 0; r < TP; ++r)
            for (int k = 0; k < qpg; ++k) {
                a0 += s_sum[r * CQ + g * qpg + k];
                a1 += s_sq[r * CQ + g * qpg + k];
            }
        double *o = psum + (((long)b * G + g) * S + s) * 2;
        o[0] = a0;
        o[1] = a1;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        const int q4 = c >> 2, q = c & 3;
        float m1 = -INFINITY, m0 = INFINITY;
        for (int r = 0; r < TP; ++r) {
            const float a = s_mx[(r * CQ + q4) * 4 + q], bb = s_mn[(r * CQ + q4) * 4 + q];
            m1 = a > m1 ? a : m1;
            m0 = bb < m0 ? bb : m0;
        }
        float *o = pmm + (((long)b * C + c) * S + s) * 2;
        o[0] = m1;
        o[1] = m0;
    }
}

__global__ void gn_finalize_kernel(const double *__restrict__ psum, const float *__restrict__ pmm, int B, int P, int C,
                                   int G, int S, const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float eps, float *__restrict__ scale, float *__restrict__ shift,
                                   float *__restrict__ pmax, float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const int b = t / C, c = t % C, cpg = C / G, g = c / cpg;
    double sum = 0.0, sq = 0.0;
    for (int s = 0; s < S; ++s) {
        sum += psum[(((long)b * G + g) * S + s) * 2 + 0];
        sq += psum[(((long)b * G + g) * S + s) * 2 + 1];
    }
    const double cnt = (double)P * cpg;
    const double mean = sum / cnt;
    double var = sq / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * rstd);
    const float sf = (float)((double)beta[c] - mean * (double)gamma[c] * rstd);
    scale[t] = sc;
    shift[t] = sf;
    if (mean_out && c % cpg == 0) {   // training tier: the moments themselves, one per (b, group)
        mean_out[b * G + g] = (float)mean;
        rstd_out[b * G + g] = (float)rstd;
    }
    if (pmax) {
        float m1 = -INFINITY, m0 = INFINITY;
        for (int s = 0; s < S; ++s) {
            const float a = pmm[((long)t * S + s) * 2 + 0], bb = pmm[((long)t * S + s) * 2 + 1];
            m1 = a > m1 ? a : m1;
            m0 = bb < m0 ? bb : m0;
        }
        pmax[t] = (sc >= 0.f ? m1 : m0) * sc + sf;
    }
}

static int gn_stats_impl(const float *Y, int ldy, int B, int P, int C, int G, const float *gamma, const float *beta,
                         float eps, float *scale, float *shift, float *pmax, float *mean, float *rstd, void *ws,
                         long ws_bytes, void *stream)
{
    CASPR_REQUIRE(Y && gamma && beta && scale && shift && ws && B > 0 && P > 0 && C > 0 && G > 0, "gn_stats: bad arguments");
    CASPR_REQUIRE(C % G == 0 && (C / G) % 4 == 0 && (C / G) <= 256, "gn_stats: C/G=%d must be a multiple of 4 and <= 256", C / G);
    CASPR_REQUIRE(ldy % 4 == 0 && ldy >= C && ((uintptr_t)Y % 16) == 0, "gn_stats: ldy=%d must be a multiple of 4 and >= C", ldy);
    CASPR_REQUIRE(ws_bytes >= caspr_gn_ws_bytes(B, P, C, G), "gn_stats: workspace too small (%ld < %ld)", ws_bytes, caspr_gn_ws_bytes(B, P, C, G));
    CASPR_REQUIRE(B <= 65535, "gn_stats: B too large");
    const int S = ceil_div(P, GN_SPLIT);
    double *psum = (double *)ws;
    float *pmm = (float *)((char *)ws + (((long)B * G * S * 16 + 63) & ~63L));
    hipStream_t st = (hipStream_t)stream;
    if (C <= 1024 && G <= 256 && 256 % (C >> 2) == 0)
        gn_partial_rows_kernel<<<dim3(S, B), dim3(256), 0, st>>>(Y, ldy, P, C, G, psum, pmm);
    else
        gn_partial_kernel<<<dim3(G, S, B), dim3(256), 0, st>>>(Y, ldy, P, C, G, psum, pmm);
    gn_finalize_kernel<<<dim3(ceil_div(B * C, 256)), dim3(256), 0, st>>>(psum, pmm, B, P, C, G, S, gamma, beta, eps, scale,
                                                                         shift, pmax, mean, rstd);
    CASPR_CHECK_LAUNCH("gn_stats");
    return CASPR_OK;
}

extern "C" int caspr_gn_stats_f32(const float *Y, int ldy, int B, int P, int C, int G, const float *gamma,
                                  const float *beta, float eps, float *scale, float *shift, float *pmax, void *ws,
                                  long ws_bytes, void *stream)
{
    return gn_stats_impl(Y, ldy, B, P, C, G, gamma, beta, eps, scale, shift, pmax, nullptr, nullptr, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// Feature propagation with its first conv on the COARSE level (round 5, second part; pointnet2.py:514-525).  The reference interpolates the
// coarser level's features to the fine points, concatenates the skip features and applies conv -> GroupNorm -> ReLU; interpolation and a
// pointwise conv commute -- W [sum_k w_k h_k ; s] + b = sum_k w_k (W_p h_k) + W_s s + b -- so the conv's large part runs over the m coarse
// rows (caspr_conv1x1_*: u = W_p h, no bias) and this kernel produces the layer's raw output on the n fine rows: the three-neighbour
// combination of u, plus the skip part on the vector pipe (C2 <= 8 channels: the finest level's 6 augmented coordinates), plus the bias,
// with the GroupNorm statistics of the result in f64 per 128-row block (measured 0.448 / 0.420 / 0.396 / 0.409 ms at 32 / 64 / 128 / 256 rows) (the layout gn_finalize_kernel reads).  At cfg-2's finest level:
// a 544 -> 512 conv over 327,680 rows becomes a 512 -> 512 conv over 163,840.
// ---------------------------------------------------------------------------------------------
#define TIA_ROWS 128
__global__ __launch_bounds__(256) void three_interp_add_gn_kernel(const float *__restrict__ u, int ldu, const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ weight, const float *__restrict__ skip, int lds,
                                                                  int C2, const float *__restrict__ wsk, const float *__restrict__ bias, int m,
                                                                  int n, int C, int G, float *__restrict__ y, int ldy, double *__restrict__ psum)
{
    __shared__ double s_sum[256], s_sq[256];
    const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
    const int CQ = C >> 2, cpg = C / G, qpg = cpg >> 2;     // host: C % 4 == 0, cpg % 4 == 0, CQ <= 256
    const int TP = 256 / CQ;
    const int tq = threadIdx.x % CQ, tp = threadIdx.x / CQ;
    const int pbeg = s * TIA_ROWS, pend = (pbeg + TIA_ROWS) < n ? (pbeg + TIA_ROWS) : n;
    double sum = 0.0, sq = 0.0;
    if (tp < TP) {
        float wk[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 8; ++c) wk[q][c] = c < C2 ? wsk[(long)(4 * tq + q) * C2 + c] : 0.f;
        const f32x4 b4 = bias ? ld4(bias + 4 * tq) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *ub = u + (long)b * m * ldu + 4 * tq;
#pragma unroll 8
        for (int p = pbeg + tp; p < pend; p += TP) {
            const long row = (long)b * n + p;
            const int i0 = idx[row * 3 + 0], i1 = idx[row * 3 + 1], i2 = idx[row * 3 + 2];
            const float w0 = weight[row * 3 + 0], w1 = weight[row * 3 + 1], w2 = weight[row * 3 + 2];
            const f32x4 a = ld4(ub + (long)i0 * ldu), bb = ld4(ub + (long)i1 * ldu), cc = ld4(ub + (long)i2 * ldu);
            float sk[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) sk[c] = c < C2 ? skip[row * lds + c] : 0.f;
            f32x4 r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = b4[q];
#pragma unroll
                for (int c = 0; c < 8; ++c) t = fmaf(wk[q][c], sk[c], t);
                r[q] = fmaf(w2, cc[q], fmaf(w1, bb[q], w0 * a[q])) + t;
                sum += (double)r[q];
                sq += (double)r[q] * (double)r[q];
            }
            st4(y + row * ldy + 4 * tq, r);
        }
    }
    s_sum[threadIdx.x] = sum;
    s_sq[threadIdx.x] = sq;
    __syncthreads();
    if (threadIdx.x < G) {        // one thread per group: its quads x the TP row lanes, in index order (deterministic)
        const int g = threadIdx.x;
        double a0 = 0.0, a1 = 0.0;
        for (int r = 0; r < TP; ++r)
            for (int k = 0; k < qpg; ++k) {
                a0 += s_sum[r * CQ + g * qpg + k];
                a1 += s_sq[r * CQ + g * qpg + k];
            }
        double *o = psum + (((long)b * G + g) * S + s) * 2;
        o[0] = a0;
        o[1] = a1;
    }
}

extern "C" long caspr_three_interp_add_gn_ws_bytes(int B, int n, int G) { return (long)B * G * ((n + TIA_ROWS - 1) / TIA_ROWS) * 2 * (long)sizeof(double); }

extern "C" int caspr_three_interp_add_gn_f32(const float *u, int ldu, const int32_t *idx, const float *weight, const float *skip, int lds,
                                             int C2, const float *wskip, const float *bias, int B, int m, int n, int C, float *y, int ldy,
                                             int G, const float *gamma, const float *beta, float eps, float *scale, float *shift, void *ws,
                                             long ws_bytes, void *stream)
{
    CASPR_REQUIRE(u && idx && weight && y && gamma && beta && scale && shift && ws && B > 0 && B <= 65535 && m > 0 && n > 0 && C > 0 && G > 0,
                  "three_interp_add_gn: bad arguments");
    CASPR_REQUIRE(C % G == 0 && (C / G) % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, "three_interp_add_gn: C=%d / G=%d: need C/G %% 4 == 0 and C/4 a divisor of 256", C, G);
    CASPR_REQUIRE(ldu % 4 == 0 && ldu >= C && ldy % 4 == 0 && ldy >= C, "three_interp_add_gn: ldu=%d ldy=%d must be multiples of 4 and >= C", ldu, ldy);
    CASPR_REQUIRE(C2 >= 0 && C2 <= 8 && (C2 == 0 || (skip && wskip && lds >= C2)), "three_interp_add_gn: C2=%d skip channels (0..8) need skip / wskip", C2);
    const int S = ceil_div(n, TIA_ROWS);
    CASPR_REQUIRE(ws_bytes >= caspr_three_interp_add_gn_ws_bytes(B, n, G), "three_interp_add_gn: workspace of %ld bytes is too small", ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    double *psum = reinterpret_cast<double *>(ws);
    three_interp_add_gn_kernel<<<dim3(S, B), dim3(256), 0, st>>>(u, ldu, idx, weight, skip, lds, C2, wskip, bias, m, n, C, G, y, ldy, psum);
    gn_finalize_kernel<<<dim3(ceil_div(B * C, 256)), dim3(256), 0, st>>>(psum, nullptr, B, n, C, G, S, gamma, beta, eps, scale, shift, nullptr, nullptr, nullptr);
    CASPR_CHECK_LAUNCH("three_interp_add_gn");
    return CASPR_OK;
}

// training tier: same statistics, and the per-(batch, group) mean / rstd the backward pass needs
extern "C" int caspr_gn_stats_train_f32(const float *Y, int ldy, int B, int P, int C, int G, const float *gamma,
                                        const float *beta, float eps, float *scale, float *shift, float *pmax,
                                        float *mean, float *rstd, void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(mean && rstd, "gn_stats_train: mean / rstd outputs are required");
    return gn_stats_impl(Y, ldy, B, P, C, G, gamma, beta, eps, scale, shift, pmax, mean, rstd, ws, ws_bytes, stream);
}
