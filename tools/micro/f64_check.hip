// Companion of pk_check.hip: do OTHER instructions with 64-bit register operands show the hazard packed-f32 ones do?  f64 VALU arithmetic that consumes
// a register pair an LDS read (ds_read_b64) has just returned, evaluated twice -- right behind the read, and again after a few unrelated instructions --
// alone and beside the encoder's conv (tools/r06_pk_check.py --f64).  The two evaluations must be the same bits.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void f64_check_kernel(const float *__restrict__ pts, int n, int iters, unsigned *__restrict__ bad)
{
    __shared__ double sd[1536];
    const int tid = threadIdx.x;
    const float *p = pts + (long)blockIdx.x * n * 3;
    for (int i = tid; i < n * 3; i += 256) sd[i] = (double)p[i] * 1.0000001;
    __syncthreads();
    const double X = sd[tid * 3 + 0], Y = sd[tid * 3 + 1], Z = sd[tid * 3 + 2];
    unsigned nbad = 0, nbad_pk = 0;
    int old = 0;
    for (int j = 0; j < iters; ++j) {
        const double x1 = sd[old * 3 + 0], y1 = sd[old * 3 + 1], z1 = sd[old * 3 + 2];
        // first evaluation: straight behind the reads
        double dx = X - x1, dy = Y - y1, dz = Z - z1;
        double d1 = __builtin_fma(dz, dz, __builtin_fma(dx, dx, dy * dy));
        asm volatile("" : "+v"(d1));
        // a few unrelated instructions, then the same again on opaque copies of the SAME registers
        double xa = x1, ya = y1, za = z1;
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(xa), "+v"(ya), "+v"(za));
        double ex = X - xa, ey = Y - ya, ez = Z - za;
        double d2 = __builtin_fma(ez, ez, __builtin_fma(ex, ex, ey * ey));
        asm volatile("" : "+v"(d2));
        if (__double_as_longlong(d1) != __double_as_longlong(d2)) nbad += 1;
        old = (int)(((unsigned long long)__double_as_longlong(d2) >> 13) % (unsigned)n);
        old = __builtin_amdgcn_readfirstlane(old);
    }
    if (nbad) atomicAdd(bad, nbad);
    (void)nbad_pk;
}

extern "C" int f64_check(const float *pts, int frames, int n, int iters, unsigned *bad, void *stream)
{
    if (n > 512 || n < 256) return -1;
    f64_check_kernel<<<dim3(frames), dim3(256), 0, (hipStream_t)stream>>>(pts, n, iters, bad);
    return (int)hipGetLastError();
}
