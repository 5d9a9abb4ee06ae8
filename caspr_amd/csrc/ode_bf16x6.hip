// ode_bf16x6.hip -- the point-CNF solve (cnf.py:70-128) with the two 512x512 hidden layers on the bf16 matrix pipe in the
// exact three-way split of gemm_bf16x6.hip: the DEFAULT kernel of the CNF (ops.CNF_BF16X6; cnf_rk4_kernel of ode.hip is
// the f32-MFMA alternative).  Two instantiations: DIV = false integrates the state only (sampling, cnf.py:71-74 with
// logpx = None); DIV = true also integrates the Hutchinson divergence estimate (odefunc.py:119-142) for forward() / NLL:
// forward-mode tangents J e ride as the UPPER 8 columns of every wave's 16-column tile next to their 8 points (e^T J^T e ==
// e^T J e), the value column's pre-activation reaches its tangent column through one DPP row rotate per layer.
//
// Different geometry from the f32 kernel, forced by the 2.67x faster products (weights can no longer be streamed from L2
// into every wave's fragments: 4x the bytes per MFMA cycle):
//  * a 256-thread workgroup owns 64 points of one frame; WAVE w owns ALL 512 hidden units of its 16 points
//    (32 row tiles x 1 column tile = 128 accumulator registers), one wave per SIMD, one workgroup per CU;
//  * the hidden activation never touches LDS: an accumulator (D) fragment holds rows 4g..4g+3 of a 16-row tile for
//    column j, a bf16 B fragment holds 8 k-slots for column j -- two row tiles ARE one B fragment of the next layer
//    once the weight pack lists its k in the same order (slot s of lane group g <-> unit 32kc + 4g + s for s < 4,
//    32kc + 16 + 4g + s - 4 above);
//  * B fragments are produced WHILE the layer runs (pieces go k-chunk-major): chunk kc+1's input-layer values (layer 1)
//    or its slice of layer 1's epilogue (layer 2) are computed in quarters inside chunk kc's MFMA runs, hidden behind
//    the matrix pipe; two accumulator sets instead of 192 fragment registers;
//  * LDS is the weight stage shared by the four waves: 48 KB pieces (256 rows x 32 k x 3 planes, pre-swizzled image)
//    arrive by LDS-DMA, double-buffered, one raw s_barrier per piece of 96 MFMAs = 1536 matrix-pipe cycles per wave;
//  * output layer 512 -> 3, the RK4 state and its update are wave-local (state component d of column j lives in lane
//    16 d + j).
//
// Measured at cfg-2 (160 frames x 2048 points, 8 RK4 steps): 51.3 ms per launch against 80.4 ms for the f32 kernel
// (DESIGN.md section 3 lists the steps from 64.8 ms and what is left: MFMA issue floor 26 ms, exposed VALU 3.6 ms,
// piece barriers ~6 ms, weight DMA 8-9.5 ms -- the 3 MB of split weights go L2 -> LDS once per 64 points and stage).
#include "ode_x6.h"

#define XC_COLS 64
#define XC_PA (256 * 64)          // one plane of a piece
#define XC_PIECE (3 * XC_PA)      // 48 KB: 256 rows x 32 k x 3 planes
#define XC_NPIECE 32              // per layer: 16 k chunks x 2 row halves
#define XC_RING 2                 // LDS double buffer of pieces
#define XC_LDS (XC_RING * XC_PIECE + (6 * XC_H + 3 * XC_H + 3 * XC_H + 8) * 4)

// value of lane (l ^ 8) inside its 16-lane row: the partner column (value <-> tangent) of the same hidden-unit rows
__device__ __forceinline__ float xc_partner(float v) { return dpp_mov<0x128>(v); }   // row_ror:8
// tangent lanes (j >= 8 = DPP banks 2, 3 of every row) take the partner's value, value lanes keep their own: ONE
// v_mov_b32_dpp row_ror:8 bank_mask:0xC in place -- no select, no second register
__device__ __forceinline__ float xc_value_pre(float v)
{
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0x128, 0xf, 0xC, false));
}

template <bool DIV>
__global__ __launch_bounds__(256, 1) void cnf_rk4_x6_kernel(CnfX6Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *wbuf = lds;                                  // [XC_RING][XC_PIECE]
    float *s_gate = (float *)(lds + XC_RING * XC_PIECE);              // [3][512] sigmoid gates of layers 0,1,2
    float *s_hb = s_gate + 3 * XC_H;                            // [3][512] layer bias * gate + hyper bias
    float *s_w0 = s_hb + 3 * XC_H;                              // [512][3]
    float *s_w3 = s_w0 + 3 * XC_H;                              // [3][512] output layer
    float *s_g3 = s_w3 + 3 * XC_H;                              // [8]: gate3[3], hb3[3]

    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = lane0 >> 4;
    const int bt = blockIdx.y;
    // DIV: a wave's 16 columns are 8 points (j < 8) and their 8 tangent columns (j >= 8); the workgroup owns 32 points
    const int col = DIV ? blockIdx.x * (XC_COLS / 2) + 8 * wave + (lane0 & 7) : blockIdx.x * XC_COLS + 16 * wave + (lane0 & 15);
    const bool tg0 = DIV && (lane0 & 8);
    const bool cvalid = col < a.n;
    const int ccol = cvalid ? col : a.n - 1;
    const float *hy = a.hyper + (long)bt * a.ldh;
    constexpr int GOFF = 0, BOFF = 3 * XC_H + 3;
    const int sd = g0 < 3 ? g0 : 0;   // state component of this lane (lanes g == 3 carry a copy of component 0, never stored)

    for (int i = tid; i < 3 * XC_H; i += 256) {
        s_w0[i] = a.w0[i];
        s_w3[i] = a.w3[i];
    }

    float y, kacc = 0.f, kprev = 0.f;
    float lp = 0.f, lacc = 0.f;   // DIV: log-density state of point j, owned by lane (g = 3, j < 8)
    {
        float v = a.y_in[((long)bt * a.n + ccol) * 3 + sd];
        if (a.mbn_in) {
            const float w = a.mbn_in[sd], bb = a.mbn_in[3 + sd], mean = a.mbn_in[6 + sd], var = a.mbn_in[9 + sd];
            if (a.reverse) v = (v - bb) * expf(-w) * expf(0.5f * logf(var + 1e-4f)) + mean;   // normalization.py:92-94
            else v = (v - mean) * expf(-0.5f * logf(var + 1e-4f)) * expf(w) + bb;             // normalization.py:70-74
        }
        y = v;
        if (DIV) {
            if (tg0) y = a.e[((long)bt * a.n + ccol) * 3 + sd];     // tangent lanes carry e_d (constant over the solve) in the state register
            lp = a.logp_in ? a.logp_in[(long)bt * a.n + ccol] : 0.f;
            if (a.mbn_in) {
                float ld = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) ld += -0.5f * logf(a.mbn_in[9 + d] + 1e-4f) + a.mbn_in[d];   // normalization.py:103-108
                lp = a.reverse ? lp + ld : lp - ld;
            }
        }
    }

    // LDS-DMA of piece p (k chunk p >> 1, row half p & 1) of a layer's pack [row half][k chunk][48 KB image] into
    // buffer p & 1: scalar base + one 32-bit lane offset (anything lane-dependent that is hoisted out of the stage
    // loop ends up in scratch).  12 wave-instructions of 1 KB per wave, issued in three parts.
    auto dma = [&](const unsigned char *wx, int p, int lane16, int s0 = 0, int s1 = 12) {
        const unsigned char *src = wx + (long)((p & 1) * 16 + (p >> 1)) * XC_PIECE + (wave * 12) * 1024;
#pragma unroll
        for (int s = s0; s < s1; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + s * 1024 + lane16),
                                             (__attribute__((address_space(3))) void *)(wbuf + (p & 1) * XC_PIECE + (wave * 12 + s) * 1024), 16, 0, 0);
    };
    const double t0 = a.reverse ? (double)a.t_end : 0.0, t1 = a.reverse ? 0.0 : (double)a.t_end;
    const double h = (t1 - t0) / (double)a.steps;
    const float hh = (float)h, h2 = (float)(0.5 * h), h6 = (float)(h / 6.0);

    // the first piece of layer 1; every layer pass leaves the NEXT pass's first piece in flight
    dma(a.w1x, 0, lane0 * 16);

    // The B fragments of a layer input are PRODUCED WHILE THE LAYER RUNS: the pieces go k-chunk-major, so chunk kc+1's three
    // planes (12 registers) are only needed when chunk kc's two pieces are done -- its input-layer values (layer 1) or its
    // slice of layer 1's epilogue (layer 2: gate * acc + bias, softplus, split) are computed in quarters inside the
    // MFMA runs of chunk kc, where the VALU work hides behind the matrix pipe.  Two accumulator sets (layer 1's is
    // consumed chunk by chunk while layer 2 fills its own) instead of 192 fragment registers.
    f32x4 acc1[32], acc2[32];
    u32x4 bkw[2][3];              // B-fragment planes of the current / next k chunk, by chunk parity
    f32x4 tg_, tb, tw[3];         // table values of the half chunk being produced (gate, bias, input-layer weights)

    for (int step = 0; step < a.steps; ++step) {
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
            const double tc = (stage == 0) ? 0.0 : (stage == 3 ? 1.0 : 0.5);
            const float t = (float)(t0 + (double)step * h + tc * h);
            const float aw = (stage == 0) ? 0.f : (stage == 3 ? hh : h2);
            // Opaque copy of the lane id: everything derived from it (LDS offsets, DMA offsets, table addresses) is
            // recomputed per stage instead of being hoisted out of the 32-stage loop and spilled (as in ode.hip).
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            const int g = lane >> 4, j = lane & 15, lane16 = lane * 16;
            // A-fragment read offset inside a plane: row (lane & 15) of a 16-row tile, piece g, swizzled as the pack
            const int aoff = j * 64 + ((g ^ ((0 - (j >> 2)) & 3)) << 4);
            __syncthreads();   // the previous stage's epilogues are done with the gate tables
            for (int i = tid; i < 3 * XC_H; i += 256) {
                const float gt = sigmoid_fast(hy[GOFF + i] + t * a.tcol[GOFF + i]);
                const float hb = hy[BOFF + i] + t * a.tcol[BOFF + i];
                const float bl = i < XC_H ? a.b0[i] : (i < 2 * XC_H ? a.b1[i - XC_H] : a.b2[i - 2 * XC_H]);
                s_gate[i] = gt;
                s_hb[i] = bl * gt + hb;
            }
            if (tid < 3) {
                const float gt = sigmoid_fast(hy[GOFF + 3 * XC_H + tid] + t * a.tcol[GOFF + 3 * XC_H + tid]);
                const float hb = hy[BOFF + 3 * XC_H + tid] + t * a.tcol[BOFF + 3 * XC_H + tid];
                s_g3[tid] = gt;
                s_g3[4 + tid] = a.b3[tid] * gt + hb;
            }
            __syncthreads();

            // ---- stage input of this lane's column, all three components
            const bool tg = DIV && (lane & 8);               // this lane's column is a tangent column
            const float ystage = (stage == 0 || tg) ? y : y + aw * kprev;   // tangent lanes: e (kprev stays 0 there)
            const int jp = DIV ? (j & 7) : j;                // the point's value column
            const float y0 = __shfl(ystage, jp), y1 = __shfl(ystage, 16 + jp), y2 = __shfl(ystage, 32 + jp);
            float e0 = 0.f, e1 = 0.f, e2 = 0.f;              // DIV: the point's noise vector
            if (DIV) {
                e0 = __shfl(ystage, 8 + jp);
                e1 = __shfl(ystage, 24 + jp);
                e2 = __shfl(ystage, 40 + jp);
            }
            // gated softplus layer on a value / tangent column pair (odefunc.py:98-105 and its forward-mode derivative):
            //   value   : softplus(pre)                 pre = gate * (W h) + bias  (the VALUE column's, `pre`)
            //   tangent : gate * (W h_t) * sigmoid(pre)                            (`lin_t` = gate * (W h_t))
            // with u = e^-|pre| shared: softplus = max(pre, 0) + ln(1 + u), sigmoid = (pre >= 0 ? 1 : u) / (1 + u)
            auto act_pair = [&](float pre, float lin_t) __attribute__((always_inline)) -> float {
                if (!DIV) return softplus_fast(pre);
                const float u = __builtin_amdgcn_exp2f(fabsf(pre) * -1.44269504088896341f);
                const float w1 = 1.0f + u;
                const float sp = fmaxf(pre, 0.0f) + 0.69314718055994531f * __builtin_amdgcn_logf(w1);
                const float sg = (pre >= 0.0f ? 1.0f : u) * __builtin_amdgcn_rcpf(w1);
                return tg ? lin_t * sg : sp;
            };

            // ---- producers of B fragments, a quarter (two k-slots) at a time; the tables of a half chunk one region earlier
            auto put_pair = [&](int kc, int q, float v0, float v1) __attribute__((always_inline)) {
                unsigned p1, p2, p3;
                xc_split_pair(v0, v1, p1, p2, p3);
                bkw[kc & 1][0][q] = p1;
                bkw[kc & 1][1][q] = p2;
                bkw[kc & 1][2][q] = p3;
            };
            // input layer 3 -> 512 (diffeq_layers.py:83-90 + softplus): slots 2q, 2q+1 of chunk kc = units 32kc + 16h + 4g + r
            auto tab_in = [&](int kc, int hf) __attribute__((always_inline)) {
                const int c = 32 * kc + 16 * hf + 4 * g;
                tg_ = ld4(s_gate + c);
                tb = ld4(s_hb + c);
                tw[0] = ld4(s_w0 + 3 * c);
                tw[1] = ld4(s_w0 + 3 * c + 4);
                tw[2] = ld4(s_w0 + 3 * c + 8);
            };
            auto quad_in = [&](int kc, int q) __attribute__((always_inline)) {
                const float w[12] = {tw[0][0], tw[0][1], tw[0][2], tw[0][3], tw[1][0], tw[1][1], tw[1][2], tw[1][3], tw[2][0], tw[2][1], tw[2][2], tw[2][3]};
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * (q & 1) + e;
                    const float pre = (w[3 * r] * y0 + w[3 * r + 1] * y1 + w[3 * r + 2] * y2) * tg_[r] + tb[r];
                    const float lin_t = DIV ? (w[3 * r] * e0 + w[3 * r + 1] * e1 + w[3 * r + 2] * e2) * tg_[r] : 0.f;
                    v[e] = act_pair(pre, lin_t);
                }
                put_pair(kc, q, v[0], v[1]);
            };
            // epilogue of hidden layer 1 for chunk kc of layer 2: units 32kc + 16h + 4g + r = rows of acc1[2kc + h]
            auto tab_e1 = [&](int kc, int hf) __attribute__((always_inline)) {
                const int c = 32 * kc + 16 * hf + 4 * g;
                tg_ = ld4(s_gate + XC_H + c);
                tb = ld4(s_hb + XC_H + c);
            };
            auto quad_e1 = [&](int kc, int q) __attribute__((always_inline)) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * (q & 1) + e;
                    const float lin = acc1[2 * kc + (q >> 1)][r] * tg_[r];
                    float pre = lin + tb[r];
                    if (DIV) pre = xc_value_pre(pre);                // the value column's pre-activation
                    v[e] = act_pair(pre, lin);
                }
                put_pair(kc, q, v[0], v[1]);
            };

            // 4 / 20 MFMAs of four row tiles: smallest terms first; term-major, i.e. four independent accumulators between
            // dependent MFMAs
            auto mma_head = [&](f32x4 (&acc)[32], const bf16x8 (&af)[4][3], const u32x4 (&b)[3], int m0) __attribute__((always_inline)) {
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[0]);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][2], b0, acc[m0 + u], 0, 0, 0);
            };
            auto mma_tail = [&](f32x4 (&acc)[32], const bf16x8 (&af)[4][3], const u32x4 (&b)[3], int m0) __attribute__((always_inline)) {
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[0]), b1 = __builtin_bit_cast(bf16x8, b[1]), b2 = __builtin_bit_cast(bf16x8, b[2]);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][1], b1, acc[m0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][0], b2, acc[m0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][1], b0, acc[m0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][0], b1, acc[m0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][0], b0, acc[m0 + u], 0, 0, 0);
            };

            // One layer pass.  A piece (48 KB, k chunk p >> 1, row half p & 1) is four groups of four row tiles (24 MFMAs each),
            // read into two fragment sets alternately and skewed by one group: a scheduling region = the 12 reads of group
            // G+1, interleaved two per MFMA with the first MFMAs of the 20 that remain of group G ("tail"), then the first
            // four MFMAs of group G+1 ("head") -- hipcc waits with lgkmcnt(0), never a counted wait, before the first use of
            // a set, and at the head that wait is free.  Six of the eight regions of a k chunk also carry a quarter of the
            // next chunk's B fragments (VALU) or the table reads for it.  sched_group_barrier builds the patterns,
            // sched_barrier(0) closes a region (hipcc otherwise sinks every read to just before its use).  The last group of
            // piece p-1 finishes after the barrier of piece p.
#define XC_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define XC_RM6 XC_SGB(0x100, 2) XC_SGB(0x008, 1) XC_SGB(0x100, 2) XC_SGB(0x008, 1) XC_SGB(0x100, 2) XC_SGB(0x008, 1) XC_SGB(0x100, 2) \
    XC_SGB(0x008, 1) XC_SGB(0x100, 2) XC_SGB(0x008, 1) XC_SGB(0x100, 2) XC_SGB(0x008, 1)
#define XC_VM4 XC_SGB(0x002, 2) XC_SGB(0x008, 1) XC_SGB(0x002, 2) XC_SGB(0x008, 1) XC_SGB(0x002, 2) XC_SGB(0x008, 1) XC_SGB(0x002, 2) XC_SGB(0x008, 1)
#define XC_REGION_PLAIN XC_RM6 XC_SGB(0x008, 18) __builtin_amdgcn_sched_barrier(0);
#define XC_REGION_TAB XC_RM6 XC_SGB(0x100, 5) XC_SGB(0x008, 18) __builtin_amdgcn_sched_barrier(0);
#define XC_REGION_VALU_S XC_RM6 XC_VM4 XC_VM4 XC_VM4 XC_VM4 XC_SGB(0x002, 2) XC_SGB(0x008, 2) __builtin_amdgcn_sched_barrier(0);
// DIV carries ~1.7x the producer VALU (value + tangent forms of the gated softplus): the reads are pinned as in the sampling
// variant, the VALU is left to the scheduler between the remaining MFMAs (a fixed 2-per-MFMA pattern strands the rest
// behind the region's last MFMA and sends ~450 registers to scratch)
#define XC_VM3 XC_SGB(0x002, 3) XC_SGB(0x008, 1) XC_SGB(0x002, 3) XC_SGB(0x008, 1) XC_SGB(0x002, 3) XC_SGB(0x008, 1) XC_SGB(0x002, 3) XC_SGB(0x008, 1)
#define XC_REGION_VALU_D XC_RM6 XC_VM3 XC_VM3 XC_VM3 XC_VM3 XC_SGB(0x002, 8) XC_SGB(0x008, 2) __builtin_amdgcn_sched_barrier(0);
#define XC_REGION_VALU if constexpr (DIV) { XC_REGION_VALU_D } else { XC_REGION_VALU_S }
            auto layer = [&](const unsigned char *wx, const unsigned char *wnext, f32x4 (&acc)[32], auto tab, auto quad) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < 32; ++mi) acc[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
                bf16x8 af0[4][3], af1[4][3];
                auto rd = [&](bf16x8 (&af)[4][3], const unsigned char *A, int G) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) af[u][pl] = *(const bf16x8 *)(A + pl * XC_PA + (4 * G + u) * 1024);
                };
#pragma unroll
                for (int p = 0; p < XC_NPIECE; ++p) {
                    // piece p (its DMA was issued one piece ago) has landed once nothing is outstanding; lgkmcnt: this wave's
                    // reads of the buffer about to be refilled.  Raw barrier: no compiler-added waits.
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();   // piece p is there for every wave; everybody is done with buffer (p + 1) & 1
                    asm volatile("" ::: "memory");
                    const int kc = p >> 1, mt = p & 1;
                    const int kcp = (p - 1) >> 1, mtp = (p - 1) & 1;
                    const bool more = kc + 1 < 16;   // a next chunk to produce
                    const unsigned char *wq = p + 1 < XC_NPIECE ? wx : wnext;
                    const int pn = p + 1 < XC_NPIECE ? p + 1 : 0;
                    const unsigned char *A = wbuf + (p & 1) * XC_PIECE + aoff;
                    // region 0: read group 0 | the rest of the previous piece's group 3 | first MFMAs of group 0
                    __builtin_amdgcn_sched_barrier(0);
                    rd(af0, A, 0);
                    dma(wq, pn, lane16, 0, 4);
                    if (p > 0) mma_tail(acc, af1, bkw[kcp & 1], 16 * mtp + 12);
                    mma_head(acc, af0, bkw[kc & 1], 16 * mt);
                    if (more && mt == 0) {
                        tab(kc + 1, 0);
                        XC_REGION_TAB
                    } else if (more) {
                        quad(kc + 1, 2);
                        XC_REGION_VALU
                    } else {
                        XC_REGION_PLAIN
                    }
                    // region 1
                    rd(af1, A, 1);
                    dma(wq, pn, lane16, 4, 8);
                    mma_tail(acc, af0, bkw[kc & 1], 16 * mt);
                    mma_head(acc, af1, bkw[kc & 1], 16 * mt + 4);
                    if (more) {
                        quad(kc + 1, mt == 0 ? 0 : 3);
                        XC_REGION_VALU
                    } else {
                        XC_REGION_PLAIN
                    }
                    // region 2
                    rd(af0, A, 2);
                    dma(wq, pn, lane16, 8, 12);
                    mma_tail(acc, af1, bkw[kc & 1], 16 * mt + 4);
                    mma_head(acc, af0, bkw[kc & 1], 16 * mt + 8);
                    if (more && mt == 0) {
                        quad(kc + 1, 1);
                        XC_REGION_VALU
                    } else {
                        XC_REGION_PLAIN
                    }
                    // region 3
                    rd(af1, A, 3);
                    mma_tail(acc, af0, bkw[kc & 1], 16 * mt + 8);
                    mma_head(acc, af1, bkw[kc & 1], 16 * mt + 12);
                    if (more && mt == 0) {
                        tab(kc + 1, 1);
                        XC_REGION_TAB
                    } else {
                        XC_REGION_PLAIN
                    }
                }
                mma_tail(acc, af1, bkw[((XC_NPIECE - 1) >> 1) & 1], 16 * ((XC_NPIECE - 1) & 1) + 12);
            };

            float part[3] = {0.f, 0.f, 0.f};
            if (!(a.diag & 2)) {
                // chunk 0 of layer 1 up front (exposed: 1/16 of the input layer)
                tab_in(0, 0);
                quad_in(0, 0);
                quad_in(0, 1);
                tab_in(0, 1);
                quad_in(0, 2);
                quad_in(0, 3);
                layer(a.w1x, a.w2x, acc1, tab_in, quad_in);
                // chunk 0 of layer 2
                tab_e1(0, 0);
                quad_e1(0, 0);
                quad_e1(0, 1);
                tab_e1(0, 1);
                quad_e1(0, 2);
                quad_e1(0, 3);
                layer(a.w2x, a.w1x, acc2, tab_e1, quad_e1);
            }
            {
                // ---- epilogue of hidden layer 2 + the 512 -> 3 output layer as a per-lane partial dot product (tables one
                // row tile ahead)
                int le = lane;   // opaque again: the table addresses must not be hoisted above the product loop
                asm volatile("" : "+v"(le));
                const int ge = le >> 4;
                f32x4 tq[2][5];
                auto ld_e2 = [&](int set, int mi) __attribute__((always_inline)) {
                    const int c = 16 * mi + 4 * ge;
                    tq[set][0] = ld4(s_gate + 2 * XC_H + c);
                    tq[set][1] = ld4(s_hb + 2 * XC_H + c);
                    tq[set][2] = ld4(s_w3 + c);
                    tq[set][3] = ld4(s_w3 + XC_H + c);
                    tq[set][4] = ld4(s_w3 + 2 * XC_H + c);
                };
                ld_e2(0, 0);
#pragma unroll
                for (int mi = 0; mi < 32; ++mi) {
                    if (mi + 1 < 32) ld_e2((mi + 1) & 1, mi + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 gt = tq[mi & 1][0], hb = tq[mi & 1][1], wx3 = tq[mi & 1][2], wy3 = tq[mi & 1][3], wz3 = tq[mi & 1][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float lin = acc2[mi][r] * gt[r];
                        float pre = lin + hb[r];
                        if (DIV) pre = xc_value_pre(pre);
                        const float hv = act_pair(pre, lin);
                        part[0] += wx3[r] * hv;
                        part[1] += wy3[r] * hv;
                        part[2] += wz3[r] * hv;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- output ConcatSquash (no softplus: odefunc.py:103): sum the four lane groups, every lane gets all three
            float o[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v = part[d];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                o[d] = (DIV && tg) ? v * s_g3[d] : v * s_g3[d] + s_g3[4 + d];    // tangent columns: J e has no bias term
            }
            const float od = sd == 0 ? o[0] : (sd == 1 ? o[1] : o[2]);
            if (!DIV || !tg) {
                kprev = od;
                kacc = (stage == 0) ? od : ((stage == 3) ? kacc + od : kacc + 2.0f * od);
            }
            if (DIV) {
                // -divergence estimate = -(e . J e) (odefunc.py:26,136): tangent lanes (g < 3) hold e_g and all of J e
                float dv = (tg && g < 3) ? ystage * od : 0.f;
                dv += __shfl_xor(dv, 16);
                dv += __shfl_xor(dv, 32);
                const float nd = -xc_partner(dv);            // lands in the point's value column (every g)
                lacc = (stage == 0) ? nd : ((stage == 3) ? lacc + nd : lacc + 2.0f * nd);
            }
        }
        if (!DIV || !tg0) y = y + h6 * kacc;
        if (DIV) lp = lp + h6 * lacc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetch left in flight by the last layer pass

    if (DIV && cvalid && g0 == 3 && !tg0) {
        if (a.mbn_out) {
            float ld = 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) ld += -0.5f * logf(a.mbn_out[9 + d] + 1e-4f) + a.mbn_out[d];
            lp = a.reverse ? lp + ld : lp - ld;
        }
        a.logp_out[(long)bt * a.n + col] = lp;
    }
    if (cvalid && g0 < 3 && !tg0) {
        float v = y;
        if (a.mbn_out) {
            const float w = a.mbn_out[sd], bb = a.mbn_out[3 + sd], mean = a.mbn_out[6 + sd], var = a.mbn_out[9 + sd];
            if (a.reverse) v = (v - bb) * expf(-w) * expf(0.5f * logf(var + 1e-4f)) + mean;
            else v = (v - mean) * expf(-0.5f * logf(var + 1e-4f)) * expf(w) + bb;
        }
        a.y_out[((long)bt * a.n + col) * 3 + sd] = v;
    }
}

// (512, ldw) f32 hidden-layer weight -> [row half 2][k chunk 16][plane 3][row 0..255][piece'][8 bf16], k listed in the
// D-fragment order of the producing layer (see the header): piece g, element q <-> k = 32 kc + (q < 4 ? 4g + q : 16 + 4g + q - 4)
__global__ void pack_weight_cnf_x6_kernel(const float *__restrict__ w, int ldw, unsigned char *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (row half * 16 + chunk) * 1024 + row * 4 + piece
    if (i >= 2 * 16 * 1024) return;
    const int piece = i & 3, row = (i >> 2) & 255;
    const int ck = i >> 10, kc = ck & 15, mt = ck >> 4;
    const int co = mt * 256 + row;
    float hs[3][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = 32 * kc + (q < 4 ? 4 * piece + q : 16 + 4 * piece + q - 4);
        xc_split(w[(long)co * ldw + k], hs[0][q], hs[1][q], hs[2][q]);
    }
    const int sw = (0 - (row >> 2)) & 3;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = xc_pack(hs[pl][2 * q], hs[pl][2 * q + 1]);
        *(u32x4 *)(out + (long)ck * XC_PIECE + pl * XC_PA + row * 64 + ((piece ^ sw) << 4)) = v;
    }
}

#ifdef CASPR_DEBUG_HOOKS
static unsigned long long *g_x6_trace = nullptr;
extern "C" void caspr_debug_set_x6_trace(unsigned long long *dev_buf) { g_x6_trace = dev_buf; }   // debug build only: >= 512 u64
#endif
// one buffer per hidden layer holds BOTH images: [0, XC_PACK) the 48 KB-piece layout of cnf_rk4_x6_kernel (divergence
// variant), [XC_PACK, XC_PACK + XW_PACK) the 24 KB-piece fragment-order layout of cnf_rk4_x6w_kernel (sampling)
#define XC_PACK (2L * 16 * XC_PIECE)
extern "C" long caspr_cnf_x6_packed_bytes(void) { return XC_PACK + XW_PACK; }

extern "C" int caspr_pack_weight_cnf_x6(const float *w, int ldw, void *packed, void *stream)
{
    CASPR_REQUIRE(w && packed && ldw >= XC_H, "pack_weight_cnf_x6: bad arguments");
    CASPR_REQUIRE(((uintptr_t)packed % 16) == 0, "pack_weight_cnf_x6: packed must be 16-byte aligned");
    pack_weight_cnf_x6_kernel<<<2 * 16 * 1024 / 256, 256, 0, (hipStream_t)stream>>>(w, ldw, (unsigned char *)packed);
    caspr_cnf_x6w_pack(w, ldw, (unsigned char *)packed + XC_PACK, (hipStream_t)stream);
    CASPR_CHECK_LAUNCH("pack_weight_cnf_x6");
    return CASPR_OK;
}

extern "C" int caspr_cnf_rk4_x6_f32(const float *y_in, const float *hyper, int ldh, const float *tcol, const float *w0,
                                    const float *b0, const void *w1x, const float *b1, const void *w2x, const float *b2,
                                    const float *w3, const float *b3, int H, float t_end, int steps, int reverse,
                                    const float *mbn_in, const float *mbn_out, const float *e, const float *logp_in,
                                    float *logp_out, float *y_out, int BT, int n, void *stream)
{
    CASPR_REQUIRE(y_in && hyper && tcol && w0 && b0 && w1x && b1 && w2x && b2 && w3 && b3 && y_out, "cnf_rk4_x6: null pointer");
    CASPR_REQUIRE(H == XC_H, "cnf_rk4_x6: hidden width %d unsupported (kernel is built for 512-512-512, flow.py:89)", H);
    CASPR_REQUIRE(BT > 0 && BT <= 65535 && n > 0 && steps > 0 && ldh >= 2 * (3 * H + 3), "cnf_rk4_x6: bad sizes");
    CASPR_REQUIRE((e == nullptr) == (logp_out == nullptr), "cnf_rk4_x6: e and logp_out must be given together");
    CASPR_REQUIRE(((uintptr_t)w1x % 16) == 0 && ((uintptr_t)w2x % 16) == 0 && ((uintptr_t)w0 % 16) == 0 && ((uintptr_t)w3 % 16) == 0,
                  "cnf_rk4_x6: weights must be 16-byte aligned");
    CnfX6Args a;
    a.y_in = y_in; a.hyper = hyper; a.tcol = tcol; a.w0 = w0; a.b0 = b0; a.b1 = b1; a.b2 = b2; a.w3 = w3; a.b3 = b3;
    a.mbn_in = mbn_in; a.mbn_out = mbn_out; a.w1x = (const unsigned char *)w1x; a.w2x = (const unsigned char *)w2x;
    a.e = e; a.logp_in = logp_in; a.logp_out = logp_out;
    a.trace = nullptr;
    CASPR_IF_DEBUG(a.trace = g_x6_trace;)
    a.diag = CASPR_DEBUG_ENV_INT("CASPR_X6_DIAG");   // timing experiments, debug build only
    const bool narrow = (reverse & CASPR_CNF_NARROW) != 0;      // the caller asks for the 64-point sampling kernel (include/caspr_hip.h)
    reverse &= 1;
    a.y_out = y_out; a.ldh = ldh; a.n = n; a.steps = steps; a.reverse = reverse; a.t_end = t_end;
    // Kernel choice by the presence of e only, never by BT or n: a frame's result does not depend on the batch around it.
    // Sampling (no divergence): the 128-point kernel of ode_bf16x6w.hip; the debug build can force the 64-point one
    // (CASPR_X6_NARROW=1) for A/B timing (tools/cnf_x6w_trace.py).
    if (!e && !narrow && CASPR_DEBUG_ENV_INT("CASPR_X6_NARROW") == 0) {
        a.w1x += XC_PACK;
        a.w2x += XC_PACK;
        const int rc = caspr_cnf_x6w_launch(a, BT, (hipStream_t)stream);
        if (rc != CASPR_OK) return rc;
        CASPR_CHECK_LAUNCH("cnf_rk4_x6 (128-point kernel)");
        return CASPR_OK;
    }
    static CasprLdsOptIn optin_s, optin_d;
    const hipError_t err = e ? caspr_lds_opt_in(optin_d, (const void *)cnf_rk4_x6_kernel<true>, XC_LDS)
                             : caspr_lds_opt_in(optin_s, (const void *)cnf_rk4_x6_kernel<false>, XC_LDS);
    if (err != hipSuccess) {
        caspr_set_error("cnf_rk4_x6: hipFuncSetAttribute failed: %s", hipGetErrorString(err));
        return CASPR_ELAUNCH;
    }
    if (e) cnf_rk4_x6_kernel<true><<<dim3(ceil_div(n, XC_COLS / 2), BT), dim3(256), XC_LDS, (hipStream_t)stream>>>(a);
    else cnf_rk4_x6_kernel<false><<<dim3(ceil_div(n, XC_COLS), BT), dim3(256), XC_LDS, (hipStream_t)stream>>>(a);
    CASPR_CHECK_LAUNCH("cnf_rk4_x6");
    return CASPR_OK;
}
