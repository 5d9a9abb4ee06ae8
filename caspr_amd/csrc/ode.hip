// ode.hip -- the "advect" and "sample" stages for gfx950: fixed-step RK4 of the latent dynamics net
// (models/latent_ode_model.py:45-70,139-147) and of the point CNF's gated ODE function
// (models/cnf.py:70-128, odefunc.py:98-142, diffeq_layers.py:83-90, normalization.py:59-108).
//
// cnf_rk4_kernel: ONE launch integrates the whole flow.  A 256-thread workgroup owns 32 columns of one
// frame (32 points when sampling; 16 points + their 16 Hutchinson tangents when the divergence is
// integrated); two workgroups share a CU.  The 512 x 32 hidden activation lives in LDS as an XOR-swizzled
// MFMA B-tile (64 KiB); each of the 4 waves owns 128 hidden units x 32 columns (64 accumulator VGPRs) and
// streams its slice of the packed 512x512 weights straight from L2 into A fragments with buffer loads (two
// register sets, prefetched one 16-k chunk ahead, pinned with sched_barrier because hipcc sinks loads to
// their first use).  ConcatSquash gate/bias, softplus, the 3->512 input layer and the
// 512->3 output layer (fused into the last hidden layer's epilogue as a register-level partial dot
// product) never leave the CU.  The divergence uses the forward-mode identity
// e^T (df/dy)^T e == e^T (df/dy) e: tangents ride along as 32 extra columns of the same GEMMs.
#include <stdlib.h>

#include "common.h"

#define CNF_H 512
#define CNF_NCOL 32
#define CNF_KC (CNF_H / 16)  // 32 chunks of 16 k

// ---------------------------------------------------------------------------------------------
// latent ODE.  The dynamics net is a GEMV per sequence -- latency-bound if run one sequence per
// workgroup (each thread a 512-long dependent FMA chain on L2 loads).  Instead ONE 512-thread workgroup
// advects up to 16 sequences at once: the batch is the N=16 column dimension of the f32 MFMA tile, the
// hidden state is an LDS B-tile, each wave owns 64 of the 512 hidden units and streams its packed
// weights from L2 (read once per evaluation for all 16 sequences).  RK4 state lives in LDS.
// ---------------------------------------------------------------------------------------------
#define LAT_NCOL 16

// out rows [16*rt0, 16*(rt0+nrt)) of W (packed, KC chunks) times the B-tile `in`; epilogue(rt, acc)
template <int NRT, typename Epi>
__device__ __forceinline__ void lat_layer(const float *__restrict__ wp, int KC, int rt0, const float *in, int lane, Epi epi)
{
    // One CU streams the whole weight set from L2 every evaluation: keep 4 chunks (4 x NRT KiB per wave,
    // 128 KiB per workgroup) of A fragments in flight in a register ring to cover the L2 latency.
    const int g = lane >> 4, j = lane & 15;
    f32x4 acc[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *wb = wp + ((long)rt0 * KC) * 256 + lane * 4;
    f32x4 ring[4][NRT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NRT; ++i) ring[s][i] = ld4(wb + ((long)i * KC + (s < KC ? s : 0)) * 256);
    for (int kc0 = 0; kc0 < KC; kc0 += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kc = kc0 + s;
            if (kc < KC) {
                const f32x4 bf = ld4(in + btile_off(kc * 4 + g, j, LAT_NCOL));
#pragma unroll
                for (int i = 0; i < NRT; ++i) {
                    const f32x4 af = ring[s][i];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i] = mfma16(af[q], bf[q], acc[i]);
                }
                const int kn = (kc + 4 < KC) ? kc + 4 : kc;
#pragma unroll
                for (int i = 0; i < NRT; ++i) ring[s][i] = ld4(wb + ((long)i * KC + kn) * 256);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NRT; ++i) epi(rt0 + i, acc[i]);
}

__global__ __launch_bounds__(512) void latent_rk4_kernel(const float *__restrict__ z0, int ldz,
                                                         const float *__restrict__ times, int B, int Tu, int D, int H,
                                                         int steps, const float *__restrict__ w0p,
                                                         const float *__restrict__ b0, const float *__restrict__ w1p,
                                                         const float *__restrict__ b1, const float *__restrict__ w2p,
                                                         const float *__restrict__ b2, const float *__restrict__ w3p,
                                                         const float *__restrict__ b3, float *__restrict__ out)
{
    // B-tiles: stage input (64 x 16, padded to 32 k-rows of 4), two hidden buffers (512 x 16)
    __shared__ __attribute__((aligned(16))) float s_in[16 * LAT_NCOL * 4];
    __shared__ __attribute__((aligned(16))) float s_h1[128 * LAT_NCOL * 4], s_h2[128 * LAT_NCOL * 4];
    __shared__ float s_z[64 * LAT_NCOL], s_acc[64 * LAT_NCOL], s_k[64 * LAT_NCOL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int b0i = blockIdx.x * LAT_NCOL;
    const int KC0 = 2 * ((D + 31) / 32), KCH = 2 * ((H + 31) / 32);
    const int RTH = (H + 15) / 16, RTD = (D + 15) / 16;

    for (int i = tid; i < 64 * LAT_NCOL; i += 512) {
        const int d = i / LAT_NCOL, c = i % LAT_NCOL;
        const float v = (d < D && b0i + c < B) ? z0[(long)(b0i + c) * ldz + d] : 0.f;
        s_z[i] = v;
        if (d < D && b0i + c < B) out[((long)(b0i + c) * Tu) * D + d] = v;
    }
    for (int i = tid; i < 16 * LAT_NCOL * 4; i += 512) s_in[i] = 0.f;   // K padding rows stay zero
    for (int i = tid; i < 128 * LAT_NCOL * 4; i += 512) { s_h1[i] = 0.f; s_h2[i] = 0.f; }
    __syncthreads();

    auto write_in = [&](float a, const float *kv) {  // s_in = z + a*k as a B-tile (row d, col c)
        for (int i = tid; i < 64 * LAT_NCOL; i += 512) {
            const int d = i / LAT_NCOL, c = i % LAT_NCOL;
            if (d < D) s_in[btile_off(d >> 2, c, LAT_NCOL) + (d & 3)] = kv ? s_z[i] + a * kv[i] : s_z[i];
        }
    };
    auto dyn = [&]() {  // s_in -> s_k ; latent_ode_model.py:139-147 (Linear-Tanh x3, Linear)
        __syncthreads();
        for (int rt = wave * 4; rt < RTH; rt += 32)
            lat_layer<4>(w0p, KC0, rt, s_in, lane, [&](int r, f32x4 a) {
                const f32x4 bb = ld4(b0 + r * 16 + 4 * g);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = tanhf(a[q] + bb[q]);
                st4(s_h1 + btile_off(r * 4 + g, j, LAT_NCOL), v);
            });
        __syncthreads();
        for (int rt = wave * 4; rt < RTH; rt += 32)
            lat_layer<4>(w1p, KCH, rt, s_h1, lane, [&](int r, f32x4 a) {
                const f32x4 bb = ld4(b1 + r * 16 + 4 * g);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = tanhf(a[q] + bb[q]);
                st4(s_h2 + btile_off(r * 4 + g, j, LAT_NCOL), v);
            });
        __syncthreads();
        for (int rt = wave * 4; rt < RTH; rt += 32)
            lat_layer<4>(w2p, KCH, rt, s_h2, lane, [&](int r, f32x4 a) {
                const f32x4 bb = ld4(b2 + r * 16 + 4 * g);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = tanhf(a[q] + bb[q]);
                st4(s_h1 + btile_off(r * 4 + g, j, LAT_NCOL), v);
            });
        __syncthreads();
        if (wave < RTD)
            lat_layer<1>(w3p, KCH, wave, s_h1, lane, [&](int r, f32x4 a) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int d = r * 16 + 4 * g + q;
                    if (d < D) s_k[d * LAT_NCOL + j] = a[q] + b3[d];
                }
            });
        __syncthreads();
    };

    const float t_first = times[0];
    for (int ti = 1; ti < Tu; ++ti) {
        const float r0 = times[ti - 1] - t_first, r1 = times[ti] - t_first;  // latent_ode_model.py:58
        const double h = ((double)r1 - (double)r0) / (double)steps;
        const float hh = (float)h, h2 = (float)(0.5 * h), h6 = (float)(h / 6.0);
        // a repeated time stamp (the host may pass the sorted, NOT de-duplicated times of a whole batch so that it never
        // has to synchronise on torch.unique's data-dependent size) is a zero-length interval: the state is unchanged
        for (int s = 0; s < (r1 != r0 ? steps : 0); ++s) {
            write_in(0.f, nullptr);
            dyn();
            for (int i = tid; i < 64 * LAT_NCOL; i += 512) s_acc[i] = s_k[i];
            write_in(h2, s_k);
            dyn();
            for (int i = tid; i < 64 * LAT_NCOL; i += 512) s_acc[i] = s_acc[i] + 2.0f * s_k[i];
            write_in(h2, s_k);
            dyn();
            for (int i = tid; i < 64 * LAT_NCOL; i += 512) s_acc[i] = s_acc[i] + 2.0f * s_k[i];
            write_in(hh, s_k);
            dyn();
            for (int i = tid; i < 64 * LAT_NCOL; i += 512) {
                const float a = s_acc[i] + s_k[i];
                s_z[i] = s_z[i] + h6 * a;
            }
            __syncthreads();
        }
        for (int i = tid; i < 64 * LAT_NCOL; i += 512) {
            const int d = i / LAT_NCOL, c = i % LAT_NCOL;
            if (d < D && b0i + c < B) out[((long)(b0i + c) * Tu + ti) * D + d] = s_z[i];
        }
    }
}

extern "C" int caspr_latent_rk4_f32(const float *z0, int ldz, const float *times, int B, int Tu, int D, int H,
                                    int steps, const float *w0p, const float *b0, const float *w1p, const float *b1,
                                    const float *w2p, const float *b2, const float *w3p, const float *b3, float *out,
                                    void *stream)
{
    CASPR_REQUIRE(z0 && times && out && w0p && w1p && w2p && w3p && b0 && b1 && b2 && b3, "latent_rk4: null pointer");
    CASPR_REQUIRE(B > 0 && Tu > 0 && steps > 0 && D > 0 && D <= 64 && H > 0 && H <= 512 && H % 64 == 0 && ldz >= D,
                  "latent_rk4: unsupported sizes D=%d H=%d (need D<=64, H<=512, H %% 64 == 0)", D, H);
    latent_rk4_kernel<<<dim3(ceil_div(B, LAT_NCOL)), dim3(512), 0, (hipStream_t)stream>>>(z0, ldz, times, B, Tu, D, H, steps, w0p,
                                                                                          b0, w1p, b1, w2p, b2, w3p, b3, out);
    CASPR_CHECK_LAUNCH("latent_rk4");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// latent ODE across 32 compute units.  The single-workgroup kernel above streams all 2.2 MB of weights from L2 through
// ONE CU at every evaluation (51 us each, 72 evaluations at cfg-2 = 3.7 ms, a serial chain that nothing overlaps).
// Here workgroup w of a 32-workgroup team keeps rows [16w, 16w+16) of the three 512-row layers -- and the matching 16
// columns of the 64 x 512 output layer -- RESIDENT IN LDS (72 KB) for the whole solve.  Per evaluation:
//   layer 0 / 1: each workgroup computes its 16 x 16 output tile (K split over its 4 waves), writes it to a global
//                exchange buffer in B-tile layout, team barrier, everybody reads the full 512 x 16 activation back;
//   layer 2 -> 3: the workgroup's own 16 outputs of layer 2 ARE the B fragment of its K-slice of the output layer (the
//                D-fragment / B-fragment identity): it writes a 64 x 16 partial, team barrier, and every workgroup adds
//                the 32 partials in index order (fixed order: results do not depend on timing).
// Three team barriers per evaluation (monotonic counter, release / acquire at agent scope, bounded spin).  All
// workgroups carry the RK4 state redundantly (64 x 16 floats).  32 x ceil(B/16) workgroups must be co-resident: they are
// tiny (256 threads) and 256 CUs are available; a team member that is not yet scheduled only delays the others.
// ---------------------------------------------------------------------------------------------
#define LM_TEAM 32
#define LM_SPIN_LIMIT (1u << 24)

struct LatTeam {
    unsigned *counter;     // arrivals, monotonic
    unsigned *error;       // set if a barrier spin gave up
    float *hbuf;           // [2][128 * 16 * 4] exchange B-tiles
    float *pbuf;           // [LM_TEAM][4][256] output-layer partials
};

__device__ __forceinline__ void team_barrier(const LatTeam &t, unsigned &gen, bool &dead)
{
    // fence-release + relaxed arrive, relaxed poll + fence-acquire: ONE cache write-back (by the only wave that stored
    // to the exchange buffers) and one invalidate per wave, instead of a write-back per atomic and per wave
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    ++gen;
    if (threadIdx.x == 0 && !dead) {
        __hip_atomic_fetch_add(t.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = gen * LM_TEAM;
        unsigned spins = 0;
        while (__hip_atomic_load(t.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > LM_SPIN_LIMIT) {   // never observed; keeps a broken launch from hanging the device
                __hip_atomic_store(t.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    if (__hip_atomic_load(t.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave: later loads see the other workgroups' stores
}

// Training tier (MODE 1 / 2; caspr_latent_rk4_team_tape_f32 / _adjoint_f32).  The discrete RK4 map is differentiated by hand
// (caspr_amd/train/flow_grad.py: LatentSolve): its forward pass needs, per evaluation e, the layer inputs x_e, h1_e, h2_e, h3_e (the "tape":
// rows [e][sequence][channel]), its reverse sweep is the SAME chain of four products with the transposed weights in reverse order --
// W3^T (64 -> 512) takes the place of layer 0, W2^T, W1^T of the 512 x 512 layers, W0^T (512 -> 64) of the output layer -- with the tanh
// replaced by a multiplication with 1 - h^2 from the tape, the RK4 combinations by their adjoints, and every product's input kept as the
// layer's delta (rows as the tape: the weight gradients are then ONE product per layer over all evaluations, conv1x1_wgrad on the host).
// One launch each way instead of ~650 (8-row products, tanh, addcmul: 72 evaluations at cfg-3).
struct LatTape {
    float *x, *h1, *h2, *h3;          // MODE 1: written; MODE 2: h1..h3 read.  x rows are xw floats wide, the others 512
    float *d0, *d1, *d2, *d3;         // MODE 2: deltas of the four layers (d3 rows xw wide)
    const float *gout;                // MODE 2: (B, Tu, D) gradient of the solve's output
    float *gz;                        // MODE 2: (B, D) gradient of z0
    int xw;
};

template <int MODE>
__global__ __launch_bounds__(256) void latent_rk4_team_kernel(const float *__restrict__ z0, int ldz,
                                                              const float *__restrict__ times, int B, int Tu, int D,
                                                              int steps, const float *__restrict__ w0p,
                                                              const float *__restrict__ b0, const float *__restrict__ w1p,
                                                              const float *__restrict__ b1, const float *__restrict__ w2p,
                                                              const float *__restrict__ b2, const float *__restrict__ w3p,
                                                              const float *__restrict__ b3, float *__restrict__ out,
                                                              char *ws, long ws_stride, LatTape tp)
{
    constexpr int KCH = 32;                       // 512 / 16
    __shared__ __attribute__((aligned(16))) float sW0[4 * 256], sW1[KCH * 256], sW2[KCH * 256], sW3[4 * 256];
    __shared__ __attribute__((aligned(16))) float s_in[16 * LAT_NCOL * 4], s_h[128 * LAT_NCOL * 4], s_part[4][256];
    __shared__ float s_z[64 * LAT_NCOL], s_acc[64 * LAT_NCOL], s_k[64 * LAT_NCOL], s_acc2[MODE == 2 ? 64 * LAT_NCOL : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int w = blockIdx.x, grp = blockIdx.y;
    const int b0i = grp * LAT_NCOL;
    const int KC0 = 2 * ((D + 31) / 32);
    char *base = ws + (long)grp * ws_stride;
    LatTeam team{(unsigned *)base, (unsigned *)(base + 64), (float *)(base + 256), (float *)(base + 256) + 2 * 128 * LAT_NCOL * 4};
    unsigned gen = 0;
    bool dead = false;

    // resident weights: this workgroup's row tile of layers 0-2 (packed streams are row-tile-major: contiguous), and
    // chunk kc = w of each of the 4 row tiles of the output layer
    for (int i = tid; i < KC0 * 64; i += 256) st4(&sW0[i * 4], ld4(w0p + ((long)w * KC0) * 256 + i * 4));
    for (int i = tid; i < KCH * 64; i += 256) {
        st4(&sW1[i * 4], ld4(w1p + ((long)w * KCH) * 256 + i * 4));
        st4(&sW2[i * 4], ld4(w2p + ((long)w * KCH) * 256 + i * 4));
    }
    {
        const int mt = tid >> 6;   // 4 row tiles x 64 lanes
        st4(&sW3[tid * 4], ld4(w3p + ((long)mt * KCH + w) * 256 + lane * 4));
    }
    for (int i = tid; i < 64 * LAT_NCOL; i += 256) {
        const int d = i / LAT_NCOL, c = i % LAT_NCOL;
        float v;
        if (MODE == 2) v = (d < D && b0i + c < B) ? tp.gout[((long)(b0i + c) * Tu + (Tu - 1)) * D + d] : 0.f;     // adjoint state: dL/dz(t_last)
        else v = (d < D && b0i + c < B) ? z0[(long)(b0i + c) * ldz + d] : 0.f;
        s_z[i] = v;
        if (MODE != 2 && w == 0 && d < D && b0i + c < B) out[((long)(b0i + c) * Tu) * D + d] = v;
    }
    for (int i = tid; i < 16 * LAT_NCOL * 4; i += 256) s_in[i] = 0.f;
    __syncthreads();

    // 16 x 16 output tile = sW (KC chunks of this row tile) x in ; K split over the 4 waves, combined by wave 0 in order
    auto tile = [&](const float *sW, int KC, const float *in) -> f32x4 {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kc = wave; kc < KC; kc += 4) {
            const f32x4 af = ld4(sW + (kc * 64 + lane) * 4);
            const f32x4 bf = ld4(in + btile_off(kc * 4 + g, j, LAT_NCOL));
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = mfma16(af[q], bf[q], acc);
        }
        st4(&s_part[wave][lane * 4], acc);
        __syncthreads();
        f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wave == 0) r = (ld4(&s_part[0][lane * 4]) + ld4(&s_part[1][lane * 4])) + (ld4(&s_part[2][lane * 4]) + ld4(&s_part[3][lane * 4]));
        __syncthreads();
        return r;
    };
    // the hidden layers' epilogue on this workgroup's 16 units x 16 sequences (lane: units 16 w + 4 g .., sequence column j).
    // MODE 0 / 1: tanh(a + bias) [-> tape];  MODE 2: a (1 - h^2) with h from the tape, kept as the layer's delta
    const bool col_ok = b0i + j < B;
    auto act = [&](f32x4 a, const float *bias, float *tape_h, float *delta, int e) {
        f32x4 v;
        const long off = ((long)e * B + b0i + j) * 512 + w * 16 + 4 * g;
        if (MODE == 2) {
            const f32x4 t = col_ok ? ld4(tape_h + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a[q] * (1.0f - t[q] * t[q]);
            if (wave == 0 && col_ok) st4(delta + off, v);
        } else {
            const f32x4 bb = ld4(bias + w * 16 + 4 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = tanhf(a[q] + bb[q]);
            if (MODE == 1 && wave == 0 && col_ok) st4(tape_h + off, v);
        }
        return v;
    };
    auto fetch_h = [&](const float *src) {   // full 512 x 16 activation of the team -> LDS
        for (int i = tid; i < 128 * LAT_NCOL; i += 256) st4(&s_h[i * 4], ld4(src + i * 4));
        __syncthreads();
    };
    // s_in = ca * A + cb * Bv as a B-tile (row d, column c); Bv may be null
    auto write_in = [&](float ca, const float *A, float cb, const float *Bv) {
        for (int i = tid; i < 64 * LAT_NCOL; i += 256) {
            const int d = i / LAT_NCOL, c = i % LAT_NCOL;
            // the forward forms stay z + a k / z exactly as written since round 1 (ca == 1 there)
            if (d < D) s_in[btile_off(d >> 2, c, LAT_NCOL) + (d & 3)] = Bv ? (MODE == 2 ? ca * A[i] + cb * Bv[i] : A[i] + cb * Bv[i]) : (MODE == 2 ? ca * A[i] : A[i]);
        }
        __syncthreads();
    };
    float *hb0 = team.hbuf, *hb1 = team.hbuf + 128 * LAT_NCOL * 4;
    auto dyn = [&](int e) {   // s_in -> s_k   (latent_ode_model.py:139-147; MODE 2: its transpose)
        if (MODE != 0 && w == 0) {   // the evaluation's input: x_e of the tape, or the output layer's delta
            float *dst = MODE == 1 ? tp.x : tp.d3;
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) {
                const int d = i / LAT_NCOL, c = i % LAT_NCOL;
                if (d < D && b0i + c < B) dst[((long)e * B + b0i + c) * tp.xw + d] = s_in[btile_off(d >> 2, c, LAT_NCOL) + (d & 3)];
            }
        }
        f32x4 v = act(tile(sW0, KC0, s_in), b0, MODE == 2 ? tp.h3 : tp.h1, tp.d2, e);
        if (wave == 0) st4(hb0 + btile_off(w * 4 + g, j, LAT_NCOL), v);
        team_barrier(team, gen, dead);
        fetch_h(hb0);
        v = act(tile(sW1, KCH, s_h), b1, tp.h2, tp.d1, e);
        if (wave == 0) st4(hb1 + btile_off(w * 4 + g, j, LAT_NCOL), v);
        team_barrier(team, gen, dead);
        fetch_h(hb1);
        v = act(tile(sW2, KCH, s_h), b2, MODE == 2 ? tp.h1 : tp.h3, tp.d0, e);
        if (wave == 0) {   // own 16 hidden units x 16 columns = the B fragment of K-slice w of the output layer
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 af = ld4(&sW3[(mt * 64 + lane) * 4]);
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = mfma16(af[q], v[q], acc);
                st4(team.pbuf + ((long)w * 4 + mt) * 256 + lane * 4, acc);
            }
        }
        team_barrier(team, gen, dead);
        {
            const int mt = tid >> 6;
            f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < LM_TEAM; ++k) sum = sum + ld4(team.pbuf + ((long)k * 4 + mt) * 256 + lane * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = mt * 16 + 4 * g + q;
                if (d < D) s_k[d * LAT_NCOL + j] = MODE == 2 ? sum[q] : sum[q] + b3[d];
            }
        }
        __syncthreads();
    };

    const float t_first = times[0];
    if (MODE == 2) {
        // reverse sweep: the evaluations in the opposite order, e counts down from the forward pass's total
        int e = 0;
        for (int ti = 1; ti < Tu; ++ti) e += (times[ti] - t_first != times[ti - 1] - t_first) ? 4 * steps : 0;     // the forward pass's own test
        for (int ti = Tu - 1; ti >= 1; --ti) {
            const float r0 = times[ti - 1] - t_first, r1 = times[ti] - t_first;
            const double h = ((double)r1 - (double)r0) / (double)steps;
            const float hh = (float)h, h2 = (float)(0.5 * h), h3 = (float)(h / 3.0), h6 = (float)(h / 6.0);
            for (int s = 0; s < (r1 != r0 ? steps : 0); ++s) {
                e -= 4;
                write_in(h6, s_z, 0.f, nullptr);          // dL/dk4
                dyn(e + 3);
                for (int i = tid; i < 64 * LAT_NCOL; i += 256) s_acc[i] = s_k[i];
                __syncthreads();
                write_in(h3, s_z, hh, s_k);               // dL/dk3 = h/3 gz + h g4
                dyn(e + 2);
                for (int i = tid; i < 64 * LAT_NCOL; i += 256) s_acc[i] = s_acc[i] + s_k[i];
                __syncthreads();
                write_in(h3, s_z, h2, s_k);               // dL/dk2 = h/3 gz + h/2 g3
                dyn(e + 1);
                for (int i = tid; i < 64 * LAT_NCOL; i += 256) s_acc2[i] = s_k[i];
                __syncthreads();
                write_in(h6, s_z, h2, s_k);               // dL/dk1 = h/6 gz + h/2 g2
                dyn(e);
                for (int i = tid; i < 64 * LAT_NCOL; i += 256) s_z[i] = (s_z[i] + s_acc[i]) + (s_acc2[i] + s_k[i]);
                __syncthreads();
            }
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) {
                const int d = i / LAT_NCOL, c = i % LAT_NCOL;
                if (d < D && b0i + c < B) s_z[i] += tp.gout[((long)(b0i + c) * Tu + (ti - 1)) * D + d];
            }
            __syncthreads();
        }
        if (w == 0)
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) {
                const int d = i / LAT_NCOL, c = i % LAT_NCOL;
                if (d < D && b0i + c < B) tp.gz[(long)(b0i + c) * D + d] = dead ? __builtin_nanf("") : s_z[i];
            }
        return;
    }
    int e = 0;
    for (int ti = 1; ti < Tu; ++ti) {
        const float r0 = times[ti - 1] - t_first, r1 = times[ti] - t_first;
        const double h = ((double)r1 - (double)r0) / (double)steps;
        const float hh = (float)h, h2 = (float)(0.5 * h), h6 = (float)(h / 6.0);
        for (int s = 0; s < (r1 != r0 ? steps : 0); ++s) {
            write_in(1.f, s_z, 0.f, nullptr);
            dyn(e);
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) s_acc[i] = s_k[i];
            __syncthreads();
            write_in(1.f, s_z, h2, s_k);
            dyn(e + 1);
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) s_acc[i] = s_acc[i] + 2.0f * s_k[i];
            __syncthreads();
            write_in(1.f, s_z, h2, s_k);
            dyn(e + 2);
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) s_acc[i] = s_acc[i] + 2.0f * s_k[i];
            __syncthreads();
            write_in(1.f, s_z, hh, s_k);
            dyn(e + 3);
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) {
                const float a = s_acc[i] + s_k[i];
                s_z[i] = s_z[i] + h6 * a;
            }
            __syncthreads();
            e += 4;
        }
        if (w == 0)
            for (int i = tid; i < 64 * LAT_NCOL; i += 256) {
                const int d = i / LAT_NCOL, c = i % LAT_NCOL;
                // a team barrier that gave up (never observed) must not pass for a result: poison the output
                if (d < D && b0i + c < B) out[((long)(b0i + c) * Tu + ti) * D + d] = dead ? __builtin_nanf("") : s_z[i];
            }
    }
}

#define LM_WS_STRIDE (256 + (2 * 128 * LAT_NCOL * 4 + LM_TEAM * 4 * 256) * 4)

__global__ void latent_team_zero_kernel(char *ws, long stride)
{
    reinterpret_cast<unsigned *>(ws + (long)blockIdx.x * stride)[threadIdx.x] = 0u;      // 64 lanes x 4 B = the 256-byte header
}

extern "C" long caspr_latent_team_ws_bytes(int B) { return (long)ceil_div(B, LAT_NCOL) * LM_WS_STRIDE + 256; }

extern "C" int caspr_latent_rk4_team_f32(const float *z0, int ldz, const float *times, int B, int Tu, int D, int H,
                                         int steps, const float *w0p, const float *b0, const float *w1p, const float *b1,
                                         const float *w2p, const float *b2, const float *w3p, const float *b3, float *out,
                                         void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(z0 && times && out && w0p && w1p && w2p && w3p && b0 && b1 && b2 && b3 && ws, "latent_rk4_team: null pointer");
    CASPR_REQUIRE(B > 0 && Tu > 0 && steps > 0 && D > 0 && D <= 64 && H == 512 && ldz >= D,
                  "latent_rk4_team: needs D<=64, H==512 (got D=%d H=%d); use caspr_latent_rk4_f32 otherwise", D, H);
    CASPR_REQUIRE(ws_bytes >= caspr_latent_team_ws_bytes(B) && ((uintptr_t)ws % 256) == 0, "latent_rk4_team: workspace too small or misaligned");
    const int groups = ceil_div(B, LAT_NCOL);
    CASPR_REQUIRE(groups * LM_TEAM <= 128, "latent_rk4_team: %d sequences need %d co-resident workgroups (> 128); use caspr_latent_rk4_f32", B, groups * LM_TEAM);
    hipStream_t st = (hipStream_t)stream;
    // the barrier words of every group, zeroed by a KERNEL: a hipMemsetAsync here is a runtime blit, and behind a long kernel of
    // the same stream it started ~215 us after that kernel's end (step timelines of round 3) -- on the critical path of the step
    latent_team_zero_kernel<<<dim3(groups), dim3(64), 0, st>>>((char *)ws, (long)LM_WS_STRIDE);
    latent_rk4_team_kernel<0><<<dim3(LM_TEAM, groups), dim3(256), 0, st>>>(z0, ldz, times, B, Tu, D, steps, w0p, b0, w1p, b1, w2p, b2, w3p, b3,
                                                                           out, (char *)ws, (long)LM_WS_STRIDE, LatTape{});
    CASPR_CHECK_LAUNCH("latent_rk4_team");
    return CASPR_OK;
}

// Training tier: the same solve leaving its tape -- x (E, B, xw), h1 / h2 / h3 (E, B, 512), E = 4 * steps * (Tu - 1) evaluations in the order
// they are made (rows of evaluations a zero-length interval skips are not written: the caller zero-fills) ...
extern "C" int caspr_latent_rk4_team_tape_f32(const float *z0, int ldz, const float *times, int B, int Tu, int D, int H, int steps,
                                              const float *w0p, const float *b0, const float *w1p, const float *b1, const float *w2p,
                                              const float *b2, const float *w3p, const float *b3, float *out, float *tape_x, int xw,
                                              float *tape_h1, float *tape_h2, float *tape_h3, void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(z0 && times && out && w0p && w1p && w2p && w3p && b0 && b1 && b2 && b3 && ws && tape_x && tape_h1 && tape_h2 && tape_h3,
                  "latent_rk4_team_tape: null pointer");
    CASPR_REQUIRE(B > 0 && Tu > 0 && steps > 0 && D > 0 && D <= 64 && H == 512 && ldz >= D && xw >= D && xw % 4 == 0,
                  "latent_rk4_team_tape: needs D<=64, H==512, xw >= D a multiple of 4 (got D=%d H=%d xw=%d)", D, H, xw);
    CASPR_REQUIRE(ws_bytes >= caspr_latent_team_ws_bytes(B) && ((uintptr_t)ws % 256) == 0, "latent_rk4_team_tape: workspace too small or misaligned");
    CASPR_REQUIRE(((uintptr_t)tape_h1 % 16) == 0 && ((uintptr_t)tape_h2 % 16) == 0 && ((uintptr_t)tape_h3 % 16) == 0, "latent_rk4_team_tape: tape must be 16-byte aligned");
    const int groups = ceil_div(B, LAT_NCOL);
    CASPR_REQUIRE(groups * LM_TEAM <= 128, "latent_rk4_team_tape: %d sequences need %d co-resident workgroups (> 128)", B, groups * LM_TEAM);
    hipStream_t st = (hipStream_t)stream;
    LatTape tp{};
    tp.x = tape_x; tp.h1 = tape_h1; tp.h2 = tape_h2; tp.h3 = tape_h3; tp.xw = xw;
    latent_team_zero_kernel<<<dim3(groups), dim3(64), 0, st>>>((char *)ws, (long)LM_WS_STRIDE);
    latent_rk4_team_kernel<1><<<dim3(LM_TEAM, groups), dim3(256), 0, st>>>(z0, ldz, times, B, Tu, D, steps, w0p, b0, w1p, b1, w2p, b2, w3p, b3,
                                                                           out, (char *)ws, (long)LM_WS_STRIDE, tp);
    CASPR_CHECK_LAUNCH("latent_rk4_team_tape");
    return CASPR_OK;
}

// ... and the reverse sweep of that discrete map: gout (B, Tu, D) -> gz (B, D) = dL/dz0 and the four layers' deltas (rows as the tape:
// d0 / d1 / d2 (E, B, 512), d3 (E, B, xw)), with w3tp .. w0tp the packs of the TRANSPOSED weights (W3^T: 512 x D first, W0^T: D x 512 last).
// dW_l = delta_l^T tape_l (conv1x1_wgrad over the E B rows), db_l = column sums of delta_l.  Deterministic (no atomics on data).
extern "C" int caspr_latent_rk4_team_adjoint_f32(const float *gout, const float *times, int B, int Tu, int D, int H, int steps,
                                                 const float *w3tp, const float *w2tp, const float *w1tp, const float *w0tp,
                                                 const float *tape_h1, const float *tape_h2, const float *tape_h3, float *d0, float *d1,
                                                 float *d2, float *d3, int xw, float *gz, void *ws, long ws_bytes, void *stream)
{
    CASPR_REQUIRE(gout && times && gz && w0tp && w1tp && w2tp && w3tp && ws && tape_h1 && tape_h2 && tape_h3 && d0 && d1 && d2 && d3,
                  "latent_rk4_team_adjoint: null pointer");
    CASPR_REQUIRE(B > 0 && Tu > 0 && steps > 0 && D > 0 && D <= 64 && H == 512 && xw >= D && xw % 4 == 0,
                  "latent_rk4_team_adjoint: needs D<=64, H==512, xw >= D a multiple of 4 (got D=%d H=%d xw=%d)", D, H, xw);
    CASPR_REQUIRE(ws_bytes >= caspr_latent_team_ws_bytes(B) && ((uintptr_t)ws % 256) == 0, "latent_rk4_team_adjoint: workspace too small or misaligned");
    CASPR_REQUIRE(((uintptr_t)tape_h1 % 16) == 0 && ((uintptr_t)tape_h2 % 16) == 0 && ((uintptr_t)tape_h3 % 16) == 0 && ((uintptr_t)d0 % 16) == 0 &&
                      ((uintptr_t)d1 % 16) == 0 && ((uintptr_t)d2 % 16) == 0, "latent_rk4_team_adjoint: tape / deltas must be 16-byte aligned");
    const int groups = ceil_div(B, LAT_NCOL);
    CASPR_REQUIRE(groups * LM_TEAM <= 128, "latent_rk4_team_adjoint: %d sequences need %d co-resident workgroups (> 128)", B, groups * LM_TEAM);
    hipStream_t st = (hipStream_t)stream;
    LatTape tp{};
    tp.h1 = const_cast<float *>(tape_h1); tp.h2 = const_cast<float *>(tape_h2); tp.h3 = const_cast<float *>(tape_h3);
    tp.d0 = d0; tp.d1 = d1; tp.d2 = d2; tp.d3 = d3; tp.gout = gout; tp.gz = gz; tp.xw = xw;
    latent_team_zero_kernel<<<dim3(groups), dim3(64), 0, st>>>((char *)ws, (long)LM_WS_STRIDE);
    latent_rk4_team_kernel<2><<<dim3(LM_TEAM, groups), dim3(256), 0, st>>>(nullptr, 0, times, B, Tu, D, steps, w3tp, nullptr, w2tp, nullptr, w1tp, nullptr,
                                                                           w0tp, nullptr, nullptr, (char *)ws, (long)LM_WS_STRIDE, tp);
    CASPR_CHECK_LAUNCH("latent_rk4_team_adjoint");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// point CNF
// ---------------------------------------------------------------------------------------------
struct CnfArgs {
    const float *y_in, *hyper, *tcol, *w0, *b0, *w1p, *b1, *w2p, *b2, *w3, *b3, *mbn_in, *mbn_out, *e, *logp_in;
    float *logp_out, *y_out;
    int ldh, n, steps, reverse;
    float t_end;
    unsigned long long *trace;   // debug: per-phase s_memtime stamps of workgroup (0,0), wave 0 (NULL in production)
};

// Geometry (every step measured on MI355X, profiles/r01_*): a 256-thread workgroup = 4 waves owns CNF_NCOL = 32
// columns of one frame and 76 KB of LDS, so TWO workgroups share a CU and run unsynchronised: while one is in a
// VALU epilogue (gate * acc + bias, softplus, output-layer dot: ~10 % of a stage) the other keeps the matrix pipe busy.
//   8 waves x 64 columns, one workgroup per CU (both waves of a SIMD barrier-locked)          105 ms  0.67 of peak
//   4 waves x 64 columns, one workgroup per CU (one wave per SIMD, epilogues exposed)           86 ms  0.82
//   4 waves x 32 columns, two workgroups per CU, 64-bit VALU address arithmetic per load        98 ms  0.71
//   4 waves x 32 columns, two workgroups per CU, buffer loads + per-tile k rotation (this)      81 ms  0.87
// Co-resident MFMA streams only pay off once the loop carries no VALU address arithmetic (buffer loads with
// SGPR offsets) and no register spills (the opaque thread id below); s_setprio around the MFMA loops: -1 %.
#define CNF_WAVES 4
#define CNF_NT (CNF_WAVES * 64)
#define CNF_MI 8   // 16-row tiles per wave: 128 hidden units
#define CNF_CT 2   // 16-column tiles: 32 columns

// one hidden layer: acc[mi][ct] = sum_k W[128*wave + 16*mi + row][k] * Hbuf[k][16*ct + col]
// Software pipeline, two register sets: while the 128 MFMAs of chunk kc run on (a0,b0), the A fragments
// (L2 -> VGPR) and B fragments (LDS -> VGPR) of chunk kc+1 are already in flight into (a1,b1), and vice versa.
//  * A fragments come through buffer loads (SGPR resource + SGPR chunk offset + one lane-offset VGPR): no
//    per-load 64-bit VALU address arithmetic in the MFMA shadow.
//  * `rot` (a multiple of 4 chunks, taken from the tile index) rotates the k order per workgroup: all 1024
//    waves of the chip otherwise walk the same 1 MiB weight matrix in lockstep and pile onto one L2 channel
//    at a time.  fp32 sums are reassociated per tile position, deterministically (independent of the batch).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// The prefetch of the next chunk (8 weight buffer loads, 2 LDS reads) is spread through the 128 MFMAs of the current one
// by sched_group_barrier -- one load per 8 MFMAs -- instead of being issued as a burst in front of them (hipcc on its
// own sinks the loads to just before their use: no prefetch at all); sched_barrier(0) closes the region.
#define CNF_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define CNF_PREFETCH_PATTERN                                                                                                  \
    CNF_SGB(0x100, 2) CNF_SGB(0x008, 4) CNF_SGB(0x020, 1) CNF_SGB(0x008, 8) CNF_SGB(0x020, 1) CNF_SGB(0x008, 8) CNF_SGB(0x020, 1) \
    CNF_SGB(0x008, 8) CNF_SGB(0x020, 1) CNF_SGB(0x008, 8) CNF_SGB(0x020, 1) CNF_SGB(0x008, 8) CNF_SGB(0x020, 1) CNF_SGB(0x008, 8) \
    CNF_SGB(0x020, 1) CNF_SGB(0x008, 8) CNF_SGB(0x020, 1) CNF_SGB(0x008, 68) __builtin_amdgcn_sched_barrier(0);
__device__ __forceinline__ void cnf_mfma_layer(const float *__restrict__ wp, const float *Hbuf, int wave, int lane, int rot,
                                               f32x4 (&acc)[CNF_MI][CNF_CT])
{
    const int g = lane >> 4, j = lane & 15;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, CNF_H * CNF_H * 4, 0x00020000);
    const int voff = lane * 16;
    const int sbase = (wave * CNF_MI) * CNF_KC * 1024;   // bytes; row tile mi adds mi*CNF_KC*1024, chunk kc adds kc*1024
    // B-tile read offsets: kq = 4*kc + g  ->  (kq & 15) = (4*kc + g) & 15 alternates with kc & 3
    int boff[4][CNF_CT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < CNF_CT; ++ct) boff[r][ct] = btile_off(r * 4 + g, ct * 16 + j, CNF_NCOL);
#pragma unroll
    for (int mi = 0; mi < CNF_MI; ++mi)
#pragma unroll
        for (int ct = 0; ct < CNF_CT; ++ct) acc[mi][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a0[CNF_MI], a1[CNF_MI], b0[CNF_CT], b1[CNF_CT];
#pragma unroll
    for (int mi = 0; mi < CNF_MI; ++mi) a0[mi] = buf_ld4(rs, voff, sbase + (mi * CNF_KC + rot) * 1024);
#pragma unroll
    for (int ct = 0; ct < CNF_CT; ++ct) b0[ct] = ld4(Hbuf + rot * 4 * CNF_NCOL * 4 + boff[0][ct]);
#pragma unroll 1
    for (int kc = 0; kc < CNF_KC; kc += 4) {
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            const int k1 = (kc + u + 1 + rot) & (CNF_KC - 1);   // chunk prefetched into set 1
#pragma unroll
            for (int mi = 0; mi < CNF_MI; ++mi) a1[mi] = buf_ld4(rs, voff, sbase + (mi * CNF_KC + k1) * 1024);
#pragma unroll
            for (int ct = 0; ct < CNF_CT; ++ct) b1[ct] = ld4(Hbuf + (k1 - (u + 1)) * 4 * CNF_NCOL * 4 + boff[u + 1][ct]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < CNF_MI; ++mi)
#pragma unroll
                    for (int ct = 0; ct < CNF_CT; ++ct) acc[mi][ct] = mfma16(a0[mi][q], b0[ct][q], acc[mi][ct]);
            CNF_PREFETCH_PATTERN
            // prefetch the chunk after that into set 0 (wraps harmlessly on the last pair)
            const int un = (u + 2) & 3;
            const int k2 = (kc + u + 2 + rot) & (CNF_KC - 1);
#pragma unroll
            for (int mi = 0; mi < CNF_MI; ++mi) a0[mi] = buf_ld4(rs, voff, sbase + (mi * CNF_KC + k2) * 1024);
#pragma unroll
            for (int ct = 0; ct < CNF_CT; ++ct) b0[ct] = ld4(Hbuf + (k2 - un) * 4 * CNF_NCOL * 4 + boff[un][ct]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < CNF_MI; ++mi)
#pragma unroll
                    for (int ct = 0; ct < CNF_CT; ++ct) acc[mi][ct] = mfma16(a1[mi][q], b1[ct][q], acc[mi][ct]);
            CNF_PREFETCH_PATTERN
        }
    }
}

template <bool WITH_DIV>
__global__ __launch_bounds__(CNF_NT, 2) void cnf_rk4_kernel(CnfArgs a)
{
    constexpr int PT = WITH_DIV ? CNF_NCOL / 2 : CNF_NCOL;  // points per workgroup (the other half are tangents)
    constexpr int NSTATE = 3 * CNF_NCOL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Hbuf = smem;                                   // [128 kq][32 col][4]   64 KiB
    float *s_gate = Hbuf + (CNF_H / 4) * CNF_NCOL * 4;    // [2][512] sigmoid gate of hidden layers 1,2
    float *s_hb = s_gate + 2 * CNF_H;                     // [2][512] layer bias*gate + hyper bias
    float *s_red = s_hb + 2 * CNF_H;                      // [4][3][32] per-wave partial outputs
    float *s_ys = s_red + CNF_WAVES * 3 * CNF_NCOL;       // [32][4] stage input (value cols) / e (tangent cols)
    float *s_out = s_ys + CNF_NCOL * 4;                   // [3][32] stage output dy / J e
    float *s_g3 = s_out + 3 * CNF_NCOL;                   // [8]: gate3[3], hb3[3]

    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // wave-uniform (SGPR)
    const int bt = blockIdx.y;
    const int p0 = blockIdx.x * PT;
    const float *hy = a.hyper + (long)bt * a.ldh;
    constexpr int GOFF = 0, BOFF = 3 * CNF_H + 3;  // column offsets of the gate / bias blocks
    int tid = tid0;
    const int rot = (blockIdx.x & 7) * 4;   // k-order rotation of this tile position (see cnf_mfma_layer)

    // ---- state: thread (d = tid / 32, col = tid % 32), tid < 96
    const int sd = tid0 / CNF_NCOL, scol = tid0 % CNF_NCOL;
    const bool is_state = tid0 < NSTATE && scol < PT;
    const int spt = p0 + scol;
    const bool pvalid = is_state && spt < a.n;
    float y = 0.f, kacc = 0.f, ev = 0.f;
    float lp = 0.f, lacc = 0.f;  // log-density state, owned by tid < PT (sd == 0)
    if (pvalid) {
        float v = a.y_in[((long)bt * a.n + spt) * 3 + sd];
        if (a.mbn_in) {
            const float w = a.mbn_in[sd], bb = a.mbn_in[3 + sd], mean = a.mbn_in[6 + sd], var = a.mbn_in[9 + sd];
            if (a.reverse) v = (v - bb) * expf(-w) * expf(0.5f * logf(var + 1e-4f)) + mean;   // normalization.py:92-94
            else v = (v - mean) * expf(-0.5f * logf(var + 1e-4f)) * expf(w) + bb;             // normalization.py:70-74
        }
        y = v;
        if (WITH_DIV) ev = a.e[((long)bt * a.n + spt) * 3 + sd];
    }
    if (WITH_DIV) {
        if (tid < PT && p0 + tid < a.n) {
            lp = a.logp_in ? a.logp_in[(long)bt * a.n + p0 + tid] : 0.f;
            if (a.mbn_in) {
                float ld = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) ld += -0.5f * logf(a.mbn_in[9 + d] + 1e-4f) + a.mbn_in[d];  // :103-108
                lp = a.reverse ? lp + ld : lp - ld;
            }
        }
        // tangent seed e sits in the tangent columns of s_ys for the whole solve
        if (is_state) s_ys[(PT + scol) * 4 + sd] = ev;
    }

    const double t0 = a.reverse ? (double)a.t_end : 0.0, t1 = a.reverse ? 0.0 : (double)a.t_end;
    const double h = (t1 - t0) / (double)a.steps;
    const float hh = (float)h, h2 = (float)(0.5 * h), h6 = (float)(h / 6.0);
    float kprev = 0.f;

    for (int step = 0; step < a.steps; ++step) {
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
            // Opaque copy of the thread id: everything derived from it below (LDS / global addresses, XOR swizzles,
            // input-layer weights) is recomputed per stage instead of being hoisted out of the 32-stage loop, where
            // ~150 loop-invariant VGPRs were spilled and reloaded with exposed scratch latency in every epilogue.
            asm volatile("" : "+v"(tid));
#ifdef CASPR_DEBUG_HOOKS
#define CNF_STAMP(i)                                                                                                   \
    if (a.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid0 == 0 && step == 0) a.trace[stage * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define CNF_STAMP(i)
#endif
            CNF_STAMP(0)
            const int lane = tid & 63, g = lane >> 4, j = lane & 15;
            const int kq0 = tid & 127, cg0 = tid >> 7;   // input layer: rows 4*kq0 .. +3, column group
            const double tc = (stage == 0) ? 0.0 : (stage == 3 ? 1.0 : 0.5);
            const float t = (float)(t0 + (double)step * h + tc * h);
            const float aw = (stage == 0) ? 0.f : (stage == 3 ? hh : h2);
            // ---- stage input + gates
            if (is_state) s_ys[scol * 4 + sd] = (stage == 0) ? y : y + aw * kprev;
            for (int i = tid; i < 2 * CNF_H; i += CNF_NT) {   // hidden layers 1,2 (layer 0's gates live in registers)
                const int c = CNF_H + i;
                const float gt = sigmoid_fast(hy[GOFF + c] + t * a.tcol[GOFF + c]);
                const float hb = hy[BOFF + c] + t * a.tcol[BOFF + c];
                const float bl = (i < CNF_H) ? a.b1[i] : a.b2[i - CNF_H];
                s_gate[i] = gt;
                s_hb[i] = bl * gt + hb;
            }
            if (tid < 3) {
                const float gt = sigmoid_fast(hy[GOFF + 3 * CNF_H + tid] + t * a.tcol[GOFF + 3 * CNF_H + tid]);
                const float hb = hy[BOFF + 3 * CNF_H + tid] + t * a.tcol[BOFF + 3 * CNF_H + tid];
                s_g3[tid] = gt;
                s_g3[4 + tid] = a.b3[tid] * gt + hb;
            }
            CNF_STAMP(1)
            __syncthreads();
            CNF_STAMP(2)
            // ---- input layer 3 -> 512 straight into the B-tile (diffeq_layers.py:83-90 + softplus)
            {
                float gt[4], hb[4], w0r[4][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = kq0 * 4 + q;
                    w0r[q][0] = a.w0[c * 3 + 0];
                    w0r[q][1] = a.w0[c * 3 + 1];
                    w0r[q][2] = a.w0[c * 3 + 2];
                    gt[q] = sigmoid_fast(hy[GOFF + c] + t * a.tcol[GOFF + c]);
                    hb[q] = a.b0[c] * gt[q] + (hy[BOFF + c] + t * a.tcol[BOFF + c]);
                }
#pragma unroll 4
                for (int c = 0; c < CNF_NCOL / 2; ++c) {
                    const int col = cg0 * (CNF_NCOL / 2) + c;
                    const f32x4 in = ld4(s_ys + col * 4);
                    f32x4 v;
                    if (!WITH_DIV || col < PT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float pre = (w0r[q][0] * in[0] + w0r[q][1] * in[1] + w0r[q][2] * in[2]) * gt[q] + hb[q];
                            v[q] = softplus_fast(pre);
                        }
                    } else {
                        const f32x4 yv = ld4(s_ys + (col - PT) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float pre = (w0r[q][0] * yv[0] + w0r[q][1] * yv[1] + w0r[q][2] * yv[2]) * gt[q] + hb[q];
                            const float tg = (w0r[q][0] * in[0] + w0r[q][1] * in[1] + w0r[q][2] * in[2]) * gt[q];
                            v[q] = tg * sigmoid_fast(pre);
                        }
                    }
                    st4(Hbuf + btile_off(kq0, col, CNF_NCOL), v);
                }
            }
            CNF_STAMP(3)
            __syncthreads();
            CNF_STAMP(4)

            f32x4 acc[CNF_MI][CNF_CT];
            // ---- hidden layer 1
            cnf_mfma_layer(a.w1p, Hbuf, wave, lane, rot, acc);
            CNF_STAMP(5)
#pragma unroll
            for (int mi = 0; mi < CNF_MI; ++mi) {
                const int co = (wave * CNF_MI + mi) * 16 + 4 * g;
                const f32x4 gt = ld4(s_gate + co), hb = ld4(s_hb + co);
#pragma unroll
                for (int ct = 0; ct < (WITH_DIV ? CNF_CT / 2 : CNF_CT); ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = acc[mi][ct][r] * gt[r] + hb[r];
                        acc[mi][ct][r] = softplus_fast(pre);
                        if (WITH_DIV) acc[mi][ct + CNF_CT / 2][r] = acc[mi][ct + CNF_CT / 2][r] * gt[r] * sigmoid_fast(pre);
                    }
                __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting all 8 tiles' gate loads (VGPR blow-up)
            }
            CNF_STAMP(6)
            __syncthreads();  // every wave has finished reading Hbuf
            CNF_STAMP(7)
#pragma unroll
            for (int mi = 0; mi < CNF_MI; ++mi)
#pragma unroll
                for (int ct = 0; ct < CNF_CT; ++ct)
                    st4(Hbuf + btile_off((wave * CNF_MI + mi) * 4 + g, ct * 16 + j, CNF_NCOL), acc[mi][ct]);
            CNF_STAMP(8)
            __syncthreads();
            CNF_STAMP(9)
            // ---- hidden layer 2 + fused output layer 512 -> 3
            cnf_mfma_layer(a.w2p, Hbuf, wave, lane, rot, acc);
            CNF_STAMP(10)
            float part[3][CNF_CT];
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int ct = 0; ct < CNF_CT; ++ct) part[d][ct] = 0.f;
#pragma unroll
            for (int mi = 0; mi < CNF_MI; ++mi) {
                const int co = (wave * CNF_MI + mi) * 16 + 4 * g;
                const f32x4 gt = ld4(s_gate + CNF_H + co), hb = ld4(s_hb + CNF_H + co);
                const f32x4 wx = ld4(a.w3 + co), wy = ld4(a.w3 + CNF_H + co), wz = ld4(a.w3 + 2 * CNF_H + co);
#pragma unroll
                for (int ct = 0; ct < (WITH_DIV ? CNF_CT / 2 : CNF_CT); ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = acc[mi][ct][r] * gt[r] + hb[r];
                        const float hv = softplus_fast(pre);
                        part[0][ct] += wx[r] * hv;
                        part[1][ct] += wy[r] * hv;
                        part[2][ct] += wz[r] * hv;
                        if (WITH_DIV) {
                            const float tv = acc[mi][ct + CNF_CT / 2][r] * gt[r] * sigmoid_fast(pre);
                            part[0][ct + CNF_CT / 2] += wx[r] * tv;
                            part[1][ct + CNF_CT / 2] += wy[r] * tv;
                            part[2][ct + CNF_CT / 2] += wz[r] * tv;
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int ct = 0; ct < CNF_CT; ++ct) {
                    float v = part[d][ct];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if (g == 0) s_red[(wave * 3 + d) * CNF_NCOL + ct * 16 + j] = v;
                }
            CNF_STAMP(11)
            __syncthreads();
            CNF_STAMP(12)
            // ---- combine the wave partials, apply the output ConcatSquash (no softplus: odefunc.py:103)
            if (tid < NSTATE) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < CNF_WAVES; ++w) s += s_red[(w * 3 + sd) * CNF_NCOL + scol];
                float o;
                if (!WITH_DIV || scol < PT) o = s * s_g3[sd] + s_g3[4 + sd];
                else o = s * s_g3[sd];
                if (WITH_DIV) s_out[sd * CNF_NCOL + scol] = o;
                if (is_state) {
                    kprev = o;
                    kacc = (stage == 0) ? o : ((stage == 3) ? kacc + o : kacc + 2.0f * o);
                }
            }
            if (WITH_DIV) {
                __syncthreads();
                if (tid < PT) {
                    // -divergence = -(e . J e)   (odefunc.py:26,136)
                    float dv = 0.f;
#pragma unroll
                    for (int d = 0; d < 3; ++d) dv += s_ys[(PT + tid) * 4 + d] * s_out[d * CNF_NCOL + PT + tid];
                    const float o = -dv;
                    lacc = (stage == 0) ? o : ((stage == 3) ? lacc + o : lacc + 2.0f * o);
                }
            }
            CNF_STAMP(13)
            __syncthreads();
            CNF_STAMP(14)
        }
        if (is_state) y = y + h6 * kacc;
        if (WITH_DIV && tid < PT) lp = lp + h6 * lacc;
    }

    if (pvalid) {
        float v = y;
        if (a.mbn_out) {
            const float w = a.mbn_out[sd], bb = a.mbn_out[3 + sd], mean = a.mbn_out[6 + sd], var = a.mbn_out[9 + sd];
            if (a.reverse) v = (v - bb) * expf(-w) * expf(0.5f * logf(var + 1e-4f)) + mean;
            else v = (v - mean) * expf(-0.5f * logf(var + 1e-4f)) * expf(w) + bb;
        }
        a.y_out[((long)bt * a.n + spt) * 3 + sd] = v;
    }
    if (WITH_DIV && tid < PT && p0 + tid < a.n) {
        if (a.mbn_out) {
            float ld = 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) ld += -0.5f * logf(a.mbn_out[9 + d] + 1e-4f) + a.mbn_out[d];
            lp = a.reverse ? lp + ld : lp - ld;
        }
        a.logp_out[(long)bt * a.n + p0 + tid] = lp;
    }
}

#ifdef CASPR_DEBUG_HOOKS
static unsigned long long *g_cnf_trace = nullptr;
// debug build only (not part of include/caspr_hip.h): device buffer of >= 64 u64 receiving per-phase cycle stamps
extern "C" void caspr_debug_set_cnf_trace(unsigned long long *dev_buf) { g_cnf_trace = dev_buf; }
#endif

extern "C" int caspr_cnf_rk4_f32(const float *y_in, const float *hyper, int ldh, const float *tcol, const float *w0,
                                 const float *b0, const float *w1p, const float *b1, const float *w2p, const float *b2,
                                 const float *w3, const float *b3, int H, float t_end, int steps, int reverse,
                                 const float *mbn_in, const float *mbn_out, const float *e, const float *logp_in,
                                 float *logp_out, float *y_out, int BT, int n, void *stream)
{
    CASPR_REQUIRE(y_in && hyper && tcol && w0 && b0 && w1p && b1 && w2p && b2 && w3 && b3 && y_out, "cnf_rk4: null pointer");
    CASPR_REQUIRE(H == CNF_H, "cnf_rk4: hidden width %d unsupported (kernel is built for 512-512-512, flow.py:89)", H);
    CASPR_REQUIRE(BT > 0 && BT <= 65535 && n > 0 && steps > 0 && ldh >= 2 * (3 * H + 3), "cnf_rk4: bad sizes");
    CASPR_REQUIRE((e == nullptr) == (logp_out == nullptr), "cnf_rk4: e and logp_out must be given together");
    CASPR_REQUIRE(((uintptr_t)w1p % 16) == 0 && ((uintptr_t)w2p % 16) == 0, "cnf_rk4: packed weights must be 16-byte aligned");
    CnfArgs a;
    a.y_in = y_in; a.hyper = hyper; a.tcol = tcol; a.w0 = w0; a.b0 = b0; a.w1p = w1p; a.b1 = b1; a.w2p = w2p; a.b2 = b2;
    a.w3 = w3; a.b3 = b3; a.mbn_in = mbn_in; a.mbn_out = mbn_out; a.e = e; a.logp_in = logp_in; a.logp_out = logp_out;
    a.y_out = y_out; a.ldh = ldh; a.n = n; a.steps = steps; a.reverse = reverse; a.t_end = t_end;
    a.trace = nullptr;
    CASPR_IF_DEBUG(a.trace = g_cnf_trace;)
    const size_t shmem = ((size_t)(CNF_H / 4) * CNF_NCOL * 4 + 4 * CNF_H + CNF_WAVES * 3 * CNF_NCOL + CNF_NCOL * 4 + 3 * CNF_NCOL + 8) * 4
                         + (size_t)CASPR_DEBUG_ENV_INT("CASPR_CNF_LDS_PAD") * 1024;   // pad: occupancy experiments, debug build only
    hipStream_t st = (hipStream_t)stream;
    hipError_t err;
    static CasprLdsOptIn optin_div, optin_nodiv;
    if (e) {
        auto kern = cnf_rk4_kernel<true>;
        err = caspr_lds_opt_in(optin_div, (const void *)kern, shmem);
        if (err == hipSuccess) kern<<<dim3(ceil_div(n, CNF_NCOL / 2), BT), dim3(CNF_NT), shmem, st>>>(a);
    } else {
        auto kern = cnf_rk4_kernel<false>;
        err = caspr_lds_opt_in(optin_nodiv, (const void *)kern, shmem);
        if (err == hipSuccess) kern<<<dim3(ceil_div(n, CNF_NCOL), BT), dim3(CNF_NT), shmem, st>>>(a);
    }
    if (err != hipSuccess) {
        caspr_set_error("cnf_rk4: hipFuncSetAttribute(%zu) failed: %s", shmem, hipGetErrorString(err));
        return CASPR_ELAUNCH;
    }
    CASPR_CHECK_LAUNCH("cnf_rk4");
    return CASPR_OK;
}
