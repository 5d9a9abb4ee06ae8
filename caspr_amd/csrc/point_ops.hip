// point_ops.hip -- index / gather kernels of the TPointNet++ set-abstraction and feature-propagation
// stack for gfx950: farthest point sampling, ball query, grouping, three-NN, three-interpolate,
// Chamfer, input preparation.  HBM/LDS-bound integer + f32-compare work: wave64 ballot / shuffle
// reductions, clouds resident in registers / LDS.  Compiled with -ffp-contract=off: the squared
// distances must round exactly like the upstream CUDA binaries' (restated in oracle/point_ops.c)
// because the integer outputs are compared bit-exactly.  The only fused multiply-adds are the two
// explicit ones of sqsum3(): nvcc's default -fmad=true turns  a*a + b*b + c*c  into
// mul(b,b); fma(a,a,.); fma(c,c,.).
#include <stdarg.h>
#include <string.h>

#include "common.h"

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void caspr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *caspr_last_error_string(void) { return g_err; }
extern "C" int caspr_abi_version(void) { return 1; }

__device__ __forceinline__ float sqsum3(float a, float b, float c)
{
    const float yy = b * b;
    return __builtin_fmaf(c, c, __builtin_fmaf(a, a, yy));
}
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    return sqsum3(ax - bx, ay - by, az - bz);
}

// ---------------------------------------------------------------------------------------------
// input preparation (tpointnet2.py:79-90)
// ---------------------------------------------------------------------------------------------
__global__ void prep_input_kernel(const float *__restrict__ x, long total, int quad, int pairs,
                                  float *__restrict__ xyz, float *__restrict__ feat)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const f32x4 v = ld4(x + i * 4);
    xyz[i * 3 + 0] = v[0];
    xyz[i * 3 + 1] = v[1];
    xyz[i * 3 + 2] = v[2];
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int c = 0;
    if (quad) {
        f[0] = v[0] * v[0];
        f[1] = v[1] * v[1];
        f[2] = v[2] * v[2];
        c = 3;
    }
    if (pairs) {
        f[c + 0] = v[0] * v[2];  // xz
        f[c + 1] = v[0] * v[1];  // xy
        f[c + 2] = v[2] * v[1];  // yz
    }
    f32x4 a = {f[0], f[1], f[2], f[3]}, b = {f[4], f[5], f[6], f[7]};
    st4(feat + i * 8, a);
    st4(feat + i * 8 + 4, b);
}

extern "C" int caspr_prep_input_f32(const float *x, int BT, int N, int quad, int pairs, float *xyz,
                                    float *feat, void *stream)
{
    CASPR_REQUIRE(x && xyz && feat && BT > 0 && N > 0, "prep_input: bad arguments");
    const long total = (long)BT * N;
    prep_input_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        x, total, quad, pairs, xyz, feat);
    CASPR_CHECK_LAUNCH("prep_input");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Copy of a frame's cloud (count floats) into LDS by NT threads.  The plain loop `for (i = tid; i < count; i += NT) dst[i] = src[i]` compiles to one
// 4-byte load per thread and trip with s_waitcnt vmcnt(0) behind it: 24-48 dependent global round trips in front of every index kernel (a fifth of the
// thread-per-centre ball query's time).  Here: 16-byte loads, four in flight per thread, when the frame is 16-byte aligned (n % 4 == 0 for every cloud of
// the model); scalar loads, eight in flight, otherwise.  Same bytes in the same places.
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void stage_cloud(float *__restrict__ dst, const float *__restrict__ src, int count, int tid)
{
    if ((count & 3) == 0 && (reinterpret_cast<unsigned long long>(src) & 15ull) == 0ull) {
        const int n4 = count >> 2;
        for (int i = tid; i < n4; i += 4 * NT) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = i + u * NT;
                v[u] = j < n4 ? ld4(src + 4 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = i + u * NT;
                if (j < n4) st4(dst + 4 * j, v[u]);
            }
        }
    } else {
        for (int i = tid; i < count; i += 8 * NT) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = i + u * NT;
                v[u] = j < count ? src[j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = i + u * NT;
                if (j < count) dst[j] = v[u];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// farthest point sampling (pointnet2.py:384).  One 256-thread workgroup per cloud; the cloud and
// the running min-distance live in registers (PPT points per thread, point k = tid + 256*i), a
// copy of xyz in LDS serves the "last selected point" broadcast.  Arg-max key = 64-bit
// (f32 bits of the distance + 1) << 32 | ~(bitrev(k mod bs) << 16 | k): max over keys reproduces the
// upstream block reduction's total order (value desc, bit-reversed k mod bs asc -- the shared-memory
// tree keeps the lower slot on equality, its last level deciding on bit 0 --, k asc); 0 is the identity
// (best=-1, besti=0).
// Wave reduction by DPP row steps + v_readlane (wave_max_u64), 4 wave results through a double-buffered LDS slot:
// one barrier per round.
// ---------------------------------------------------------------------------------------------
// Round 6: the per-round instruction stream (the rounds are a dependent chain: 1,023 of them per frame at the first level, ~1,900 cycles each in
// round 5) was cut without touching a single rounding: 810 -> 575 us at the first level, 470 -> 230 us at the second.
//  * the per-thread arg-max on 32-bit operands: the thread's points are HELD IN PRIORITY ORDER of the tie key (smaller (bitrev(k mod bs), k) first:
//    for bs = 512 the points with even i before the odd ones, ascending otherwise); distances travel as their bit patterns (a non-negative float
//    orders like its pattern read as a signed integer; points outside the cloud / under the origin guard carry -1.0f, below every distance, the
//    identity): v_min_i32 / v_max3_i32 per point instead of a 64-bit key build, a 64-bit compare and two selects;
//  * the wave arg-max as two 32-bit DPP reductions (max distance, then max tie key among the lanes that hold it -- each lane's candidate is its FIRST
//    slot equal to the wave's maximum) instead of one 64-bit one; the four wave results still meet as 64-bit keys in the double-buffered LDS slot,
//    one barrier per round.
// NO PACKED f32 ARITHMETIC.  The first form of this kernel computed two distances per v_pk_add / v_pk_mul / v_pk_fma_f32 -- bit-exact on an idle chip,
// and WRONG beside the global PointNet's conv on another stream: in a few frames, from some round on, a different centre.  tools/micro/pk_check.hip
// reduced it to this: a packed-f32 operation that consumes a register an LDS read has just returned (here: the selected point's y, behind
// s_waitcnt lgkmcnt) now and then sees the register's OLD content in one 16-lane pass when ANOTHER kernel that executes packed-f32 instructions
// (conv1x1_bf16x6_kernel as it was compiled then) shares the compute unit; the same arithmetic in scalar form on the same registers, and the packed
// form repeated an instruction later, are right (profiles/r06_pk_check.txt: the failing subtraction used y_ref = 0; beside the same conv compiled
// WITHOUT packed instructions the test kernel is clean: it takes two).  The whole library is therefore compiled WITHOUT packed-f32 instructions
// (csrc/build.py: -target-feature -packed-fp32-ops; csrc/audit.py refuses an object that holds one): the compiler had put 9,000 of them into the
// product kernels on its own (SLP vectorisation), and taking them out cost nothing (the headline step: 69.0 -> 68.85 ms).

// max over the wave of a signed 32-bit value / an unsigned one: four DPP row steps with the operation on the DPP operand itself, the four
// row results through v_readlane.  (Distances travel as their BIT PATTERNS: a non-negative float orders like its pattern read as a
// signed integer, and the "never a candidate" value -1.0f, 0xBF800000, is below all of them -- integer min / max have no NaN
// canonicalisation step in front of them, v_min_f32 / v_max_f32 under IEEE mode do.)
__device__ __forceinline__ int fps_wave_max_i32(int x)
{
    int o;
    o = dpp_mov_i32<0xB1>(x);  x = o > x ? o : x;
    o = dpp_mov_i32<0x4E>(x);  x = o > x ? o : x;
    o = dpp_mov_i32<0x141>(x); x = o > x ? o : x;
    o = dpp_mov_i32<0x140>(x); x = o > x ? o : x;
    int best = __builtin_amdgcn_readlane(x, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const int r = __builtin_amdgcn_readlane(x, row * 16);
        best = r > best ? r : best;
    }
    return best;
}
__device__ __forceinline__ unsigned fps_wave_max_u32(unsigned x)
{
    unsigned o;
    o = (unsigned)dpp_mov_i32<0xB1>((int)x);  x = o > x ? o : x;
    o = (unsigned)dpp_mov_i32<0x4E>((int)x);  x = o > x ? o : x;
    o = (unsigned)dpp_mov_i32<0x141>((int)x); x = o > x ? o : x;
    o = (unsigned)dpp_mov_i32<0x140>((int)x); x = o > x ? o : x;
    unsigned best = (unsigned)__builtin_amdgcn_readlane((int)x, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const unsigned r = (unsigned)__builtin_amdgcn_readlane((int)x, row * 16);
        best = r > best ? r : best;
    }
    return best;
}

template <int PPT>
__global__ __launch_bounds__(256) void fps_kernel(const float *__restrict__ xyz, int n, int M, int bs_bits,
                                                  int guard, int32_t *__restrict__ idx,
                                                  float *__restrict__ new_xyz)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sx = smem;  // n*3
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(smem + ((n * 3 + 3) & ~3));  // [2][4]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *p = xyz + (long)b * n * 3;
    stage_cloud<256>(sx, p, n * 3, tid);
    __syncthreads();

    constexpr int NP = (PPT + 1) / 2;          // slots come in pairs (PPT = 1: the second one never holds a point)
    float px[2 * NP], py[2 * NP], pz[2 * NP];
    int tmp[2 * NP];                           // running minimum distance, as its bit pattern
    unsigned kkey[2 * NP];
#pragma unroll
    for (int s = 0; s < 2 * NP; ++s) {
        // slot s holds the thread's point of priority s: its tie key (bitrev(k mod bs) << 16 | k, smaller wins) ascends with s.  k = tid + 256 i;
        // for bs = 512 the bit reversal turns bit 8 of k mod 512 -- the parity of i -- into the lowest bit of the reversed field, below the
        // bits tid fixes: even i first, then odd i, ascending inside each; for bs <= 256 the reversed part is the same for all i: ascending i.
        int i = s;
        if (PPT >= 2 && bs_bits == 9) i = (s < PPT / 2) ? 2 * s : 2 * (s - PPT / 2) + 1;
        const int k = tid + 256 * i;
        const bool in = s < PPT && k < n;
        const float x_ = in ? sx[k * 3 + 0] : 0.f, y_ = in ? sx[k * 3 + 1] : 0.f, z_ = in ? sx[k * 3 + 2] : 0.f;
        px[s] = x_;
        py[s] = y_;
        pz[s] = z_;
        const float mag = sqsum3(x_, y_, z_);
        const bool ok = in && !(guard && mag <= 1e-3f);
        tmp[s] = __builtin_bit_cast(int, ok ? 1e10f : -1.0f);      // -1: never a candidate (every real distance is >= +0: min(d, -1) stays -1)
        // bit reversal of (k mod bs) over bs_bits bits (bs_bits = 0: a one-thread block, no tie key)
        const unsigned rev = bs_bits ? (__brev((unsigned)k) >> (32 - bs_bits)) : 0u;
        kkey[s] = ~((rev << 16) | (unsigned)k);
    }
    int32_t *out = idx + (long)b * M;
    float *oxyz = new_xyz ? new_xyz + (long)b * M * 3 : nullptr;
    int old = 0;
    if (tid == 0) {
        out[0] = 0;
        if (oxyz) { oxyz[0] = sx[0]; oxyz[1] = sx[1]; oxyz[2] = sx[2]; }
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int never = __builtin_bit_cast(int, -1.0f);
    for (int j = 1; j < M; ++j) {
        const float x1 = sx[old * 3 + 0], y1 = sx[old * 3 + 1], z1 = sx[old * 3 + 2];
        int bd = never;
#pragma unroll
        for (int s_ = 0; s_ < 2 * NP; ++s_) {
            const float d = sqdist3(px[s_], py[s_], pz[s_], x1, y1, z1);
            const int di = __builtin_bit_cast(int, d), t = tmp[s_];
            const int d2 = di < t ? di : t;               // = (d < tmp ? d : tmp) on the patterns: d >= +0, tmp >= +0 or -1.0f
            tmp[s_] = d2;
            bd = d2 > bd ? d2 : bd;
        }
        const int wm = fps_wave_max_i32(bd);
        // the tie key of this thread's candidate: its FIRST slot (best tie key) that holds the wave's maximum, 0 if none does
        unsigned bk = 0u;
#pragma unroll
        for (int s = 2 * NP - 1; s >= 0; --s) bk = tmp[s] == wm ? kkey[s] : bk;
        const unsigned wk = fps_wave_max_u32(bk);
        unsigned long long *s = slot + (j & 1) * 4;
        if (lane == 0) s[wave] = wm == never ? 0ull : (((unsigned long long)((unsigned)wm + 1u) << 32) | (unsigned long long)wk);
        __syncthreads();
        unsigned long long m = s[0];
        m = s[1] > m ? s[1] : m;
        m = s[2] > m ? s[2] : m;
        m = s[3] > m ? s[3] : m;
        old = (m == 0ull) ? 0 : (int)((~(unsigned)m) & 0xffffu);
        if (tid == 0) {
            out[j] = old;
            if (oxyz) {
                oxyz[j * 3 + 0] = sx[old * 3 + 0];
                oxyz[j * 3 + 1] = sx[old * 3 + 1];
                oxyz[j * 3 + 2] = sx[old * 3 + 2];
            }
        }
    }
}

// n > 4096 (no config of BASELINE.json; Kaolin's kernel takes any n): 1024 threads, the running minimum distance in LDS
// (4 bytes per point: up to 36,864 points in 144 KB), the cloud re-read from global memory every round (L1 / L2 resident:
// 12 n bytes), the same keys with the point index in 22 bits.  Same selections as fps_kernel, bit for bit.
#define FPS_BIG_MAX_N 36864
__global__ __launch_bounds__(1024) void fps_big_kernel(const float *__restrict__ xyz, int n, int M, int bs_bits, int guard,
                                                       int32_t *__restrict__ idx, float *__restrict__ new_xyz)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tmp = smem;                                                                              // [n]
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(smem + ((n + 3) & ~3));       // [2][16]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *p = xyz + (long)b * n * 3;
    for (int k = tid; k < n; k += 1024) tmp[k] = 1e10f;
    int32_t *out = idx + (long)b * M;
    float *oxyz = new_xyz ? new_xyz + (long)b * M * 3 : nullptr;
    int old = 0;
    if (tid == 0) {
        out[0] = 0;
        if (oxyz) { oxyz[0] = p[0]; oxyz[1] = p[1]; oxyz[2] = p[2]; }
    }
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    for (int j = 1; j < M; ++j) {
        const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
        unsigned long long best = 0ull;
        for (int k = tid; k < n; k += 1024) {
            const float px = p[k * 3 + 0], py = p[k * 3 + 1], pz = p[k * 3 + 2];
            if (guard && sqsum3(px, py, pz) <= 1e-3f) continue;
            const float d = sqdist3(px, py, pz, x1, y1, z1);
            const float t = tmp[k];
            const float d2 = d < t ? d : t;
            tmp[k] = d2;
            const unsigned rev = bs_bits ? (__brev((unsigned)k) >> (32 - bs_bits)) : 0u;
            const unsigned long long key = ((unsigned long long)(__float_as_uint(d2) + 1u) << 32) | (unsigned long long)(~((rev << 22) | (unsigned)k));
            best = key > best ? key : best;
        }
        best = wave_max_u64(best);
        unsigned long long *s = slot + (j & 1) * 16;
        if (lane == 0) s[wave] = best;
        __syncthreads();
        unsigned long long m = s[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) m = s[w] > m ? s[w] : m;
        old = (m == 0ull) ? 0 : (int)((~(unsigned)m) & 0x3fffffu);
        if (tid == 0) {
            out[j] = old;
            if (oxyz) {
                oxyz[j * 3 + 0] = p[old * 3 + 0];
                oxyz[j * 3 + 1] = p[old * 3 + 1];
                oxyz[j * 3 + 2] = p[old * 3 + 2];
            }
        }
    }
}

static int fps_block_size(int n)
{
    int bs = 1;
    while (bs * 2 <= n && bs * 2 <= 512) bs *= 2;
    return bs;
}

extern "C" int caspr_fps_f32(const float *xyz, int B, int n, int M, int guard, int32_t *idx,
                             float *new_xyz, void *stream)
{
    CASPR_REQUIRE(xyz && idx && B > 0 && n > 0 && M > 0, "fps: bad arguments");
    CASPR_REQUIRE(n <= FPS_BIG_MAX_N, "fps: n=%d > %d unsupported (the running minimum of a cloud is kept in LDS)", n, FPS_BIG_MAX_N);
    const int bs = fps_block_size(n);
    int bs_bits = 0;
    while ((1 << bs_bits) < bs) ++bs_bits;
    hipStream_t st = (hipStream_t)stream;
    if (n > 4096) {
        const size_t shb = (size_t)((n + 3) & ~3) * 4 + 2 * 16 * 8;
        static CasprLdsOptIn optin;
        if (caspr_lds_opt_in(optin, (const void *)fps_big_kernel, FPS_BIG_MAX_N * 4 + 2 * 16 * 8) != hipSuccess) {
            caspr_set_error("fps: hipFuncSetAttribute failed");
            return CASPR_ELAUNCH;
        }
        fps_big_kernel<<<dim3(B), dim3(1024), shb, st>>>(xyz, n, M, bs_bits, guard, idx, new_xyz);
        CASPR_CHECK_LAUNCH("fps");
        return CASPR_OK;
    }
    const size_t sh = (size_t)((n * 3 + 3) & ~3) * 4 + 64;
    const int ppt = ceil_div(n, 256);
#define FPS_LAUNCH(P) fps_kernel<P><<<dim3(B), dim3(256), sh, st>>>(xyz, n, M, bs_bits, guard, idx, new_xyz)
    if (ppt <= 1) FPS_LAUNCH(1);
    else if (ppt <= 2) FPS_LAUNCH(2);
    else if (ppt <= 4) FPS_LAUNCH(4);
    else if (ppt <= 8) FPS_LAUNCH(8);
    else FPS_LAUNCH(16);
#undef FPS_LAUNCH
    CASPR_CHECK_LAUNCH("fps");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// gather rows of a point-major feature tensor (pointnet2.py:385)
// ---------------------------------------------------------------------------------------------
__global__ void gather_points_kernel(const float *__restrict__ feat, int ldf, const int32_t *__restrict__ idx,
                                     int n, int M, int C, float *__restrict__ out, int ldo, long total)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % C);
    const long bm = t / C;
    const int b = (int)(bm / M);
    const int k = idx[bm];
    out[bm * ldo + c] = feat[((long)b * n + k) * ldf + c];
}

extern "C" int caspr_gather_points_f32(const float *feat, int ldf, const int32_t *idx, int B, int n, int M,
                                       int C, float *out, int ldo, void *stream)
{
    CASPR_REQUIRE(feat && idx && out && B > 0 && n > 0 && M > 0 && C > 0 && ldf >= C && ldo >= C,
                  "gather_points: bad arguments");
    const long total = (long)B * M * C;
    gather_points_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        feat, ldf, idx, n, M, C, out, ldo, total);
    CASPR_CHECK_LAUNCH("gather_points");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// ball query (pointnet2.py:391).  One wave per centre scans the cloud 64 points at a time in index
// order; ballot + prefix popcount assign output slots in ascending-k order; the remaining slots
// keep the first hit (upstream fills all ns slots on the first hit).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ xyz,
                                                         const float *__restrict__ new_xyz, int n, int M,
                                                         float r2, int ns, int32_t *__restrict__ idx,
                                                         long centres)
{
    const int lane = threadIdx.x & 63;
    const long c = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= centres) return;
    const int b = (int)(c / M);
    const float *p = xyz + (long)b * n * 3;
    const float cx = new_xyz[c * 3 + 0], cy = new_xyz[c * 3 + 1], cz = new_xyz[c * 3 + 2];
    int32_t *o = idx + c * ns;
    int cnt = 0, first = 0;
    for (int base = 0; base < n && cnt < ns; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) {
            const float d2 = sqdist3(cx, cy, cz, p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]);
            hit = d2 < r2;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            if (cnt == 0) first = base + (__ffsll((long long)mask) - 1);
            const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && slot < ns) o[slot] = k;
            cnt += __popcll(mask);
        }
    }
    cnt = cnt < ns ? cnt : ns;
    for (int s = cnt + lane; s < ns; s += 64) o[s] = first;  // first == 0 when there was no hit
}

// The same scan with the cloud in LDS (round 5, second part).  The kernel above reads the frame's 12 n bytes from L1 / L2 once per CENTRE:
// 1024 centres x 24 KB x 160 frames = 3.9 GB per call at cfg-2's first level -- 12 TB/s of cache traffic in its 330 us, the bound.  Here a
// workgroup copies the cloud into LDS once and its four waves walk BQ_CPW centres each against that copy: same comparisons in the same order
// (bit-identical index rows by construction), the traffic on the LDS array.  For clouds that fit (n <= BQ_LDS_MAX_N) with enough centres
// per frame to pay for the copy.
#define BQ_CPW 16
#define BQ_LDS_MAX_N 12288
__global__ __launch_bounds__(256) void ball_query_lds_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int n, int M,
                                                             float r2, int ns, int32_t *__restrict__ idx)
{
    extern __shared__ __attribute__((aligned(16))) float sp[];     // [n * 3]: stride-3 word reads across a wave are bank-conflict free
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float *p = xyz + (long)b * n * 3;
    stage_cloud<256>(sp, p, n * 3, threadIdx.x);
    __syncthreads();
    const int m0 = (blockIdx.x * 4 + wave) * BQ_CPW;
    for (int cc = 0; cc < BQ_CPW; ++cc) {
        const int m = m0 + cc;
        if (m >= M) break;                                         // wave-uniform
        const long c = (long)b * M + m;
        const float cx = new_xyz[c * 3 + 0], cy = new_xyz[c * 3 + 1], cz = new_xyz[c * 3 + 2];
        int32_t *o = idx + c * ns;
        int cnt = 0, first = 0;
        for (int base = 0; base < n && cnt < ns; base += 64) {
            const int k = base + lane;
            bool hit = false;
            if (k < n) {
                const float d2 = sqdist3(cx, cy, cz, sp[k * 3 + 0], sp[k * 3 + 1], sp[k * 3 + 2]);
                hit = d2 < r2;
            }
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                if (cnt == 0) first = base + (__ffsll((long long)mask) - 1);
                const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                if (hit && slot < ns) o[slot] = k;
                cnt += __popcll(mask);
            }
        }
        cnt = cnt < ns ? cnt : ns;
        for (int s = cnt + lane; s < ns; s += 64) o[s] = first;  // first == 0 when there was no hit
    }
}

extern "C" int caspr_ball_query_f32(const float *xyz, const float *new_xyz, int B, int n, int M, float radius, int ns, int32_t *idx, void *stream);

// The two ball queries of a set-abstraction level (two scales: pointnet2.py:340-342 builds one grouper per radius over the SAME xyz / new_xyz) in
// ONE pass: the cloud staged in LDS once, one distance per (centre, point) compared against both radii, two ballot / prefix-popcount
// streams.  Each scale's row is what ball_query_lds_kernel writes for it, bit for bit (the same comparisons in the same order; a scale
// that has its ns hits stops recording, the walk goes on until both have theirs or the cloud ends).  At the first level the two queries
// were 0.5 + 0.38 ms in front of the first set-abstraction kernel with the chip otherwise idle (profiles/r06c_step_timeline.txt).
__global__ __launch_bounds__(256) void ball_query2_lds_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int n, int M,
                                                              float r2a, int nsa, int32_t *__restrict__ idxa, float r2b, int nsb,
                                                              int32_t *__restrict__ idxb)
{
    extern __shared__ __attribute__((aligned(16))) float sp[];     // [n * 3]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float *p = xyz + (long)b * n * 3;
    stage_cloud<256>(sp, p, n * 3, threadIdx.x);
    __syncthreads();
    const int m0 = (blockIdx.x * 4 + wave) * BQ_CPW;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int cc = 0; cc < BQ_CPW; ++cc) {
        const int m = m0 + cc;
        if (m >= M) break;                                         // wave-uniform
        const long c = (long)b * M + m;
        const float cx = new_xyz[c * 3 + 0], cy = new_xyz[c * 3 + 1], cz = new_xyz[c * 3 + 2];
        int32_t *oa = idxa + c * nsa, *ob = idxb + c * nsb;
        int cnta = 0, firsta = 0, cntb = 0, firstb = 0;
        for (int base = 0; base < n && (cnta < nsa || cntb < nsb); base += 64) {
            const int k = base + lane;
            bool hita = false, hitb = false;
            if (k < n) {
                const float d2 = sqdist3(cx, cy, cz, sp[k * 3 + 0], sp[k * 3 + 1], sp[k * 3 + 2]);
                hita = d2 < r2a;
                hitb = d2 < r2b;
            }
            const unsigned long long ma = cnta < nsa ? __ballot(hita) : 0ull;     // a scale that is full records nothing more (as its own kernel: it has left the loop)
            const unsigned long long mb = cntb < nsb ? __ballot(hitb) : 0ull;
            if (ma) {
                if (cnta == 0) firsta = base + (__ffsll((long long)ma) - 1);
                const int slot = cnta + __popcll(ma & below);
                if (hita && slot < nsa) oa[slot] = k;
                cnta += __popcll(ma);
            }
            if (mb) {
                if (cntb == 0) firstb = base + (__ffsll((long long)mb) - 1);
                const int slot = cntb + __popcll(mb & below);
                if (hitb && slot < nsb) ob[slot] = k;
                cntb += __popcll(mb);
            }
        }
        cnta = cnta < nsa ? cnta : nsa;
        cntb = cntb < nsb ? cntb : nsb;
        for (int s = cnta + lane; s < nsa; s += 64) oa[s] = firsta;  // first == 0 when there was no hit
        for (int s = cntb + lane; s < nsb; s += 64) ob[s] = firstb;
    }
}

// (Round 6 also built a ONE-CENTRE-PER-LANE form -- the wave walks the cloud point by point through LDS broadcast reads, every lane appends hits to rows
// of its own, no ballots: ~17 instructions per 64 (centre, point) pairs against ~40 here -- bit-identical rows, and SLOWER: 393 us for the first level's
// pair of queries against 322 us for the kernel above, 260 / 291 against 164 / 193 us for single queries, 0.1-0.2 ms on the whole step
// (gpurun r06aj).  The instruction count is not what bounds this kernel; the form was dropped.  What DID help every index kernel is stage_cloud.)
extern "C" int caspr_ball_query2_f32(const float *xyz, const float *new_xyz, int B, int n, int M, float radius_a, int ns_a, int32_t *idx_a,
                                     float radius_b, int ns_b, int32_t *idx_b, void *stream)
{
    CASPR_REQUIRE(xyz && new_xyz && idx_a && idx_b && B > 0 && n > 0 && M > 0 && ns_a > 0 && ns_b > 0, "ball_query2: bad arguments");
    if (n <= BQ_LDS_MAX_N && M >= 4 * BQ_CPW && n >= 256 && B <= 65535) {
        // (the same rule on the frame's shape as caspr_ball_query_f32: whichever path runs, the rows are the same bits)
        volatile float r2a = radius_a * radius_a, r2b = radius_b * radius_b;
        const size_t sh = (size_t)n * 3 * sizeof(float);
        if (sh > 64 * 1024) {
            static CasprLdsOptIn optin;
            if (caspr_lds_opt_in(optin, (const void *)ball_query2_lds_kernel, BQ_LDS_MAX_N * 3 * sizeof(float)) != hipSuccess) {
                caspr_set_error("ball_query2: hipFuncSetAttribute failed");
                return CASPR_ELAUNCH;
            }
        }
        ball_query2_lds_kernel<<<dim3(ceil_div(M, 4 * BQ_CPW), B), dim3(256), sh, (hipStream_t)stream>>>(xyz, new_xyz, n, M, r2a, ns_a, idx_a, r2b, ns_b, idx_b);
        CASPR_CHECK_LAUNCH("ball_query2");
        return CASPR_OK;
    }
    const int ra = caspr_ball_query_f32(xyz, new_xyz, B, n, M, radius_a, ns_a, idx_a, stream);
    return ra != CASPR_OK ? ra : caspr_ball_query_f32(xyz, new_xyz, B, n, M, radius_b, ns_b, idx_b, stream);
}

extern "C" int caspr_ball_query_f32(const float *xyz, const float *new_xyz, int B, int n, int M, float radius,
                                    int ns, int32_t *idx, void *stream)
{
    CASPR_REQUIRE(xyz && new_xyz && idx && B > 0 && n > 0 && M > 0 && ns > 0, "ball_query: bad arguments");
    const long centres = (long)B * M;
    volatile float r2 = radius * radius;
    if (n <= BQ_LDS_MAX_N && M >= 4 * BQ_CPW && n >= 256 && B <= 65535) {
        // a rule on the frame's shape only (never on the number of frames); the two kernels write the same bits anyway
        const size_t sh = (size_t)n * 3 * sizeof(float);
        if (sh > 64 * 1024) {
            static CasprLdsOptIn optin;
            if (caspr_lds_opt_in(optin, (const void *)ball_query_lds_kernel, BQ_LDS_MAX_N * 3 * sizeof(float)) != hipSuccess) {
                caspr_set_error("ball_query: hipFuncSetAttribute failed");
                return CASPR_ELAUNCH;
            }
        }
        ball_query_lds_kernel<<<dim3(ceil_div(M, 4 * BQ_CPW), B), dim3(256), sh, (hipStream_t)stream>>>(xyz, new_xyz, n, M, r2, ns, idx);
        CASPR_CHECK_LAUNCH("ball_query");
        return CASPR_OK;
    }
    ball_query_kernel<<<dim3((unsigned)((centres + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        xyz, new_xyz, n, M, r2, ns, idx, centres);
    CASPR_CHECK_LAUNCH("ball_query");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// standalone grouping (pointnet2.py:391-398) -- reference-shaped (B,M,3+C,ns) output; the
// production path uses the fused caspr_sa_mlp_max_f32 instead.
// ---------------------------------------------------------------------------------------------
__global__ void group_points_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                    const float *__restrict__ feat, int ldf, const int32_t *__restrict__ idx,
                                    int n, int M, int C, int ns, float *__restrict__ out, long total)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int s = (int)(t % ns);
    const int ch = (int)((t / ns) % (C + 3));
    const long bm = t / ((long)ns * (C + 3));
    const int b = (int)(bm / M);
    const int k = idx[bm * ns + s];
    float v;
    if (ch < 3) v = xyz[((long)b * n + k) * 3 + ch] - new_xyz[bm * 3 + ch];
    else v = feat[((long)b * n + k) * ldf + (ch - 3)];
    out[t] = v;
}

extern "C" int caspr_group_points_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                                      const int32_t *idx, int B, int n, int M, int C, int ns, float *out,
                                      void *stream)
{
    CASPR_REQUIRE(xyz && new_xyz && idx && out && (C == 0 || feat), "group_points: bad arguments");
    const long total = (long)B * M * (C + 3) * ns;
    group_points_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        xyz, new_xyz, feat, ldf, idx, n, M, C, ns, out, total);
    CASPR_CHECK_LAUNCH("group_points");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// three_nn (pointnet2.py:514-518): one thread per unknown point, the known cloud staged in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ unknown,
                                                       const float *__restrict__ known, int n, int m,
                                                       float *__restrict__ dist, int32_t *__restrict__ idx,
                                                       float *__restrict__ weight)
{
    extern __shared__ __attribute__((aligned(16))) float sk[];
    const int b = blockIdx.y;
    const float *kp = known + (long)b * m * 3;
    stage_cloud<256>(sk, kp, m * 3, threadIdx.x);
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long row = (long)b * n + i;
    const float ux = unknown[row * 3 + 0], uy = unknown[row * 3 + 1], uz = unknown[row * 3 + 2];
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;  // upstream starts at 1e40 (double) == +inf as f32
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < m; ++k) {
        const float d = sqdist3(ux, uy, uz, sk[k * 3 + 0], sk[k * 3 + 1], sk[k * 3 + 2]);
        if (d < b1) {
            b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
        } else if (d < b2) {
            b3 = b2; i3 = i2; b2 = d; i2 = k;
        } else if (d < b3) {
            b3 = d; i3 = k;
        }
    }
    const float d1 = sqrtf(b1), d2 = sqrtf(b2), d3 = sqrtf(b3);  // correctly rounded (hipcc default)
    dist[row * 3 + 0] = d1; dist[row * 3 + 1] = d2; dist[row * 3 + 2] = d3;
    idx[row * 3 + 0] = i1; idx[row * 3 + 1] = i2; idx[row * 3 + 2] = i3;
    if (weight) {
        const float v1 = 1.0f / (d1 + 1e-8f), v2 = 1.0f / (d2 + 1e-8f), v3 = 1.0f / (d3 + 1e-8f);
        const float t0 = v1 + v2;
        const float tot = t0 + v3;
        weight[row * 3 + 0] = v1 / tot;
        weight[row * 3 + 1] = v2 / tot;
        weight[row * 3 + 2] = v3 / tot;
    }
}

extern "C" int caspr_three_nn_f32(const float *unknown, const float *known, int B, int n, int m, float *dist,
                                  int32_t *idx, float *weight, void *stream)
{
    CASPR_REQUIRE(unknown && known && dist && idx && B > 0 && n > 0 && m > 0, "three_nn: bad arguments");
    CASPR_REQUIRE(m <= 8192, "three_nn: m=%d > 8192 unsupported", m);
    three_nn_kernel<<<dim3(ceil_div(n, 256), B), dim3(256), (size_t)m * 12, (hipStream_t)stream>>>(
        unknown, known, n, m, dist, idx, weight);
    CASPR_CHECK_LAUNCH("three_nn");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// three_interpolate + concat with the skip features (pointnet2.py:519-523), point-major rows:
// one 64-lane wave per output row, 16-byte loads of the three neighbour rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_interp_kernel(const float *__restrict__ feat, int ldf,
                                                           const int32_t *__restrict__ idx,
                                                           const float *__restrict__ weight,
                                                           const float *__restrict__ in_scale,
                                                           const float *__restrict__ in_shift, int in_relu,
                                                           const float *__restrict__ skip, int lds, int m,
                                                           int n, int C, int C2, float *__restrict__ out,
                                                           int ldo, long rows)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row / n);
    const int i0 = idx[row * 3 + 0], i1 = idx[row * 3 + 1], i2 = idx[row * 3 + 2];
    const float w0 = weight[row * 3 + 0], w1 = weight[row * 3 + 1], w2 = weight[row * 3 + 2];
    const float *f0 = feat + ((long)b * m + i0) * ldf, *f1 = feat + ((long)b * m + i1) * ldf,
                *f2 = feat + ((long)b * m + i2) * ldf;
    float *o = out + row * ldo;
    for (int c = lane * 4; c < C; c += 256) {  // C % 4 == 0 checked by the host
        f32x4 a = ld4(f0 + c), bb = ld4(f1 + c), cc = ld4(f2 + c);
        if (in_scale) {  // previous layer's GroupNorm(+ReLU) folded into the load
            const f32x4 s4 = ld4(in_scale + (long)b * C + c), t4 = ld4(in_shift + (long)b * C + c);
            a = a * s4 + t4;
            bb = bb * s4 + t4;
            cc = cc * s4 + t4;
            if (in_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = a[q] > 0.f ? a[q] : 0.f;
                    bb[q] = bb[q] > 0.f ? bb[q] : 0.f;
                    cc[q] = cc[q] > 0.f ? cc[q] : 0.f;
                }
            }
        }
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t0 = w0 * a[q], t1 = w1 * bb[q], t2 = w2 * cc[q];
            const float s = t0 + t1;
            r[q] = s + t2;
        }
        st4(o + c, r);
    }
    const float *sk = skip ? skip + row * lds : nullptr;
    for (int c = C + lane; c < ldo; c += 64) o[c] = (c - C < C2) ? sk[c - C] : 0.0f;
}

extern "C" int caspr_three_interp_f32(const float *feat, int ldf, const int32_t *idx, const float *weight,
                                      const float *in_scale, const float *in_shift, int in_relu, const float *skip,
                                      int lds, int B, int m, int n, int C, int C2, float *out, int ldo, void *stream)
{
    CASPR_REQUIRE(feat && idx && weight && out && B > 0 && m > 0 && n > 0, "three_interp: bad arguments");
    CASPR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "three_interp: in_scale/in_shift must be given together");
    CASPR_REQUIRE(C % 4 == 0 && ldf % 4 == 0 && ldo % 4 == 0 && ldo >= C + C2 && (C2 == 0 || skip),
                  "three_interp: C=%d ldf=%d ldo=%d C2=%d: need C,ldf,ldo %% 4 == 0 and ldo >= C+C2", C, ldf, ldo, C2);
    const long rows = (long)B * n;
    three_interp_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        feat, ldf, idx, weight, in_scale, in_shift, in_relu, skip, lds, m, n, C, C2, out, ldo, rows);
    CASPR_CHECK_LAUNCH("three_interp");
    return CASPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Chamfer (evaluations.py:40): thread per query point, the other cloud tiled through LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chamfer_kernel(const float *__restrict__ p, const float *__restrict__ q,
                                                      int n, int m, float *__restrict__ dist)
{
    __shared__ float sq[1024 * 3];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float *pb = p + (long)b * n * 3, *qb = q + (long)b * m * 3;
    float px = 0, py = 0, pz = 0;
    if (i < n) { px = pb[i * 3]; py = pb[i * 3 + 1]; pz = pb[i * 3 + 2]; }
    float best = INFINITY;
    for (int base = 0; base < m; base += 1024) {
        const int cnt = (m - base) < 1024 ? (m - base) : 1024;
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += 256) sq[t] = qb[base * 3 + t];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float d = sqdist3(px, py, pz, sq[j * 3], sq[j * 3 + 1], sq[j * 3 + 2]);
            best = d < best ? d : best;
        }
    }
    if (i < n) dist[(long)b * n + i] = best;
}

extern "C" int caspr_chamfer_f32(const float *p, const float *q, int B, int n, int m, float *dist1, float *dist2,
                                 void *stream)
{
    CASPR_REQUIRE(p && q && dist1 && dist2 && B > 0 && n > 0 && m > 0, "chamfer: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    chamfer_kernel<<<dim3(ceil_div(n, 256), B), dim3(256), 0, st>>>(p, q, n, m, dist1);
    chamfer_kernel<<<dim3(ceil_div(m, 256), B), dim3(256), 0, st>>>(q, p, m, n, dist2);
    CASPR_CHECK_LAUNCH("chamfer");
    return CASPR_OK;
}
