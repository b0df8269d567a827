// Minimal self-checking probe for the round-2 finding "packed fp32 VALU results are wrong beside the RPN conv's MFMA loop"
// (determinism section).  Not part of libsecond_hip.so; built on its own by tools/pkfp32_repro.py.
//
//   k_victim     every lane evaluates r = a * b + c three ways on the same registers -- v_pk_fma_f32, v_pk_mul_f32 + v_pk_add_f32
//                (inline asm: the instructions the -O3 vectorisers emit) and two scalar v_fma_f32 / v_mul + v_add -- `iters`
//                times, and counts lanes / iterations where a packed result differs from the scalar one (bitwise).  Inputs change
//                every iteration so that no result is loop-invariant.  err[lane] += mismatches, err[64 + kind] += per-kind totals.
//   k_mfma_loop  a generic aggressor: every wave issues dense v_mfma_f32_32x32x16_bf16 back to back (four independent accumulators)
//                for `iters` rounds -- the matrix pipe as busy as a GEMM main loop makes it, no LDS, no memory traffic.
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

extern "C" __global__ __launch_bounds__(256) __attribute__((target("packed-fp32-ops"))) void k_victim(unsigned long long *err, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f2 a = {seed + 0.001f * threadIdx.x, 1.0f + 0.002f * lane};
    f2 b = {0.5f + 0.0003f * blockIdx.x, 1.25f - 0.001f * lane};
    f2 c = {0.125f * (lane + 1), -0.0625f * (lane + 3)};
    unsigned long long bad_fma = 0, bad_mul = 0, bad_add = 0;
    for (int it = 0; it < iters; ++it) {
        f2 pf, pm, pa;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa) : "v"(a), "v"(c));
        float s0, s1, m0, m1, a0, a1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(a.x), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(a.y), "v"(b.y), "v"(c.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(a.x), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(a.y), "v"(b.y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(a.x), "v"(c.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(a.y), "v"(c.y));
        bad_fma += (__float_as_uint(pf.x) != __float_as_uint(s0)) + (__float_as_uint(pf.y) != __float_as_uint(s1));
        bad_mul += (__float_as_uint(pm.x) != __float_as_uint(m0)) + (__float_as_uint(pm.y) != __float_as_uint(m1));
        bad_add += (__float_as_uint(pa.x) != __float_as_uint(a0)) + (__float_as_uint(pa.y) != __float_as_uint(a1));
        // next iteration's operands: bounded, lane-dependent, not loop-invariant
        a.x = 0.75f * a.x + 0.01f * s0 + 0.003f; a.y = 0.75f * a.y - 0.01f * m1 + 0.001f;
        b.x = 0.9f * b.x + 0.05f;                 b.y = 0.9f * b.y + 0.0625f;
        c.x = 0.5f * c.x + 0.1f * a0;             c.y = 0.5f * c.y - 0.1f * a1;
    }
    const unsigned long long bad = bad_fma + bad_mul + bad_add;
    if (bad) {
        atomicAdd(&err[lane], bad);
        atomicAdd(&err[64], bad_fma);
        atomicAdd(&err[65], bad_mul);
        atomicAdd(&err[66], bad_add);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&err[67], 1ull);      // launches that ran to completion
}

extern "C" __global__ __launch_bounds__(256) void k_mfma_loop(float *sink, int iters) {
    bf8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.01f * (threadIdx.x + i)); y[i] = (__bf16)(0.02f * (i + 1)); }
    f16v acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, acc3, 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    if (s == 12345.678f) sink[0] = s;                      // keeps the loop alive, never true
}

extern "C" __attribute__((visibility("default"))) int pk_launch_victim(unsigned long long *err, int blocks, int iters, float seed, void *stream) {
    hipLaunchKernelGGL(k_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, err, iters, seed);
    return (int)hipGetLastError();
}
extern "C" __attribute__((visibility("default"))) int pk_launch_mfma(float *sink, int blocks, int iters, void *stream) {
    hipLaunchKernelGGL(k_mfma_loop, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
    return (int)hipGetLastError();
}
