// common.h -- shared device/host helpers for libcaspr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/caspr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void caspr_set_error(const char *fmt, ...);

#define CASPR_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            caspr_set_error(__VA_ARGS__);   \
            return CASPR_EINVAL;            \
        }                                   \
    } while (0)

#define CASPR_CHECK_LAUNCH(name)                                                        \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            caspr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return CASPR_ELAUNCH;                                                       \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// MFMA f32 16x16x4 conventions (see DESIGN.md "fragment conventions")
//
//   lane l : g = l >> 4 (k sub-index / row quad), j = l & 15 (A row / B,D column)
//   A operand  : A[i = j][kk = g]      B operand : B[kk = g][col = j]
//   D[reg r]   : row = 4*g + r, col = j
//
// Packed weights ("A-pack"), produced by caspr_pack_weight_f32:
//   packed[((mt*KC + kc)*64 + l)*4 + q] = W[mt*16 + (l&15)][kc*16 + 4*(l>>4) + q]
// so one float4 per lane feeds four consecutive MFMA k-steps q = 0..3 of chunk kc, where k-step q
// contracts k in {kc*16 + 4*kk + q : kk = 0..3}.
//
// LDS operand tiles ("B-tile") are stored as [kq = k/4][col][4] with the 16-byte slot XOR-swizzled
// by kq so that both the 16-B fragment reads (lane (g,j) reads kq = 4*kc+g, col = 16*ct+j) and the
// 16-B epilogue writes are bank-conflict free:
//   float offset = (kq*NCOL + (col ^ (kq & 15))) * 4
// The D fragment of row tile mt is exactly the float4 at kq = 4*mt + g, col = 16*ct + j of the next
// layer's B-tile, so chained layers never shuffle data between lanes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int btile_off(int kq, int col, int ncol)
{
    return (kq * ncol + (col ^ (kq & 15))) << 2;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// softplus (beta=1, threshold=20) as torch.nn.Softplus: odefunc.py:55 NONLINEARITIES["softplus"]
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Hot-loop variants for the CNF epilogues, built on the 1-ulp hardware v_exp_f32 / v_log_f32 / v_rcp_f32.
__device__ __forceinline__ float softplus_fast(float x)
{
    // max(x,0) + ln(1 + e^-|x|) on the raw v_exp_f32 / v_log_f32 (base 2, 1 ulp): 7 VALU ops, 2 transcendental.
    // Absolute error <= ~1e-7 everywhere (the rounding of 1+u); u below 2^-126 may flush to 0, where ln(1+u) = 0
    // in f32 anyway.  (__expf/__logf add ~10 instructions of denormal range handling that this range never needs.)
    const float u = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504088896341f);
    return fmaxf(x, 0.0f) + 0.69314718055994531f * __builtin_amdgcn_logf(1.0f + u);
}
__device__ __forceinline__ float sigmoid_fast(float x)
{
    // 1 / (1 + e^-x); e^-x overflows to +inf only for x < -88.7 where the result is 0 anyway
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// ---------------------------------------------------------------------------------------------
// All-reduce over W = 4 / 8 / 16 consecutive lanes (aligned groups inside one 16-lane DPP row) with DPP moves:
// quad_perm [1,0,3,2] (xor 1), quad_perm [2,3,0,1] (xor 2), row_half_mirror (quads 0<->1, 2<->3), row_mirror
// (halves).  A DPP move is one VALU op; the generic __shfl_xor lowers to ds_bpermute_b32 (an LDS-pipe round trip).
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, dpp_mov_i32<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = dpp_mov_i32<CTRL>((int)b), hi = dpp_mov_i32<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int W, typename T>
__device__ __forceinline__ T row_allreduce_add(T x)
{
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    if (W >= 8) x += dpp_mov<0x141>(x);
    if (W >= 16) x += dpp_mov<0x140>(x);
    return x;
}
template <int W>
__device__ __forceinline__ float row_allreduce_max(float x)
{
    x = fmaxf(x, dpp_mov<0xB1>(x));
    x = fmaxf(x, dpp_mov<0x4E>(x));
    if (W >= 8) x = fmaxf(x, dpp_mov<0x141>(x));
    if (W >= 16) x = fmaxf(x, dpp_mov<0x140>(x));
    return x;
}

// wave-wide max of a 64-bit key, result uniform (SGPRs): four DPP steps inside each 16-lane row, then one lane of each
// of the four rows read back with v_readlane -- no ds_bpermute (what __shfl_xor lowers to, ~100 cycles per dependent step)
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_mov_u64(unsigned long long v)
{
    const int lo = dpp_mov_i32<CTRL>((int)(unsigned)v), hi = dpp_mov_i32<CTRL>((int)(unsigned)(v >> 32));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long x)
{
    unsigned long long o;
    o = dpp_mov_u64<0xB1>(x);  x = o > x ? o : x;
    o = dpp_mov_u64<0x4E>(x);  x = o > x ? o : x;
    o = dpp_mov_u64<0x141>(x); x = o > x ? o : x;
    o = dpp_mov_u64<0x140>(x); x = o > x ? o : x;
    unsigned long long best = 0ull;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, row * 16);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), row * 16);
        const unsigned long long r = ((unsigned long long)hi << 32) | lo;
        best = r > best ? r : best;
    }
    return best;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// Host-side launch hygiene (the C entries carry no state a result could depend on).
//
//  * Kernels that need more than 64 KiB of dynamic LDS must opt in with hipFuncSetAttribute; that is a per-(kernel,
//    device) setting, so it is done the first time a device launches the kernel, not on every launch.  The only thing
//    remembered is the bit "this device has opted in" -- a cache of an idempotent driver call.
//  * Phase traces (s_memtime stamps) and experiment switches read from the environment exist only in a build with
//    -DCASPR_DEBUG_HOOKS (CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py -> libcaspr_hip_debug.so, used by tools/*_phase_trace.py);
//    the production library exports no caspr_debug_* symbol, calls getenv nowhere and compiles the stamps out.
// ---------------------------------------------------------------------------------------------
struct CasprLdsOptIn {
    std::atomic<unsigned long long> devices{0};
};
static inline hipError_t caspr_lds_opt_in(CasprLdsOptIn &st, const void *kernel, size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (st.devices.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) st.devices.fetch_or(bit, std::memory_order_release);
    return e;
}

#ifdef CASPR_DEBUG_HOOKS
#define CASPR_DEBUG_ENV_INT(name) (getenv(name) ? atoi(getenv(name)) : 0)
#define CASPR_IF_DEBUG(...) __VA_ARGS__
#else
#define CASPR_DEBUG_ENV_INT(name) 0
#define CASPR_IF_DEBUG(...)
#endif
