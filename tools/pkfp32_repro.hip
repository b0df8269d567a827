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
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

extern "C" __global__ __launch_bounds__(256) __attribute__((target("packed-fp32-ops"))) void k_victim(unsigned long long *err, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f2 a = {seed + 0.001f * threadIdx.x, 1.0f + 0.002f * lane};
    f2 b = {0.5f + 0.0003f * blockIdx.x, 1.25f - 0.001f * lane};
    f2 c = {0.125f * (lane + 1), -0.0625f * (lane + 3)};
    unsigned long long bad_fma = 0, bad_mul = 0, bad_add = 0, bad_inpl1 = 0, bad_inpl2 = 0, bad_dist = 0, bad_const = 0, bad_sgpr = 0;
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
        // round 4: the forms the vectorised k_nms_mask actually contains (hipcc -O3 -S of csrc/nms.hip, box_corners):
        //   v_pk_mul_f32 v[6:7], v[6:7], v[16:17] op_sel_hi:[0,1]                 destination pair == source pair, the HIGH half
        //   v_pk_mul_f32 v[4:5], v[4:5], v[8:9]  op_sel:[1,0] op_sel_hi:[0,0]     reads the LOW source element the low half writes
        // in place (q1, q2) and, as the control, with a distinct destination pair (q3: same modifiers, "=&v")
        {
            f2 q1 = a, q2 = a, q3;
            float e1l, e1h, e2l, e2h;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1l) : "v"(a.x), "v"(b.x));      // expected: lo = a.lo * b.lo
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1h) : "v"(a.x), "v"(b.y));      //           hi = a.lo * b.hi   (op_sel_hi:[0,1])
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e2l) : "v"(a.y), "v"(b.x));      // expected: lo = a.hi * b.lo   (op_sel:[1,0])
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e2h) : "v"(a.x), "v"(b.x));      //           hi = a.lo * b.lo   (op_sel_hi:[0,0])
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(q1) : "v"(b));
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,0]" : "+v"(q2) : "v"(b));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=&v"(q3) : "v"(a), "v"(b));
            // ... and its operand kinds: an inline constant broadcast with op_sel_hi, an SGPR pair that a VALU compare overwrites
            // a few instructions later (v_pk_mul_f32 v[16:17], v[14:15], s[0:1] ... v_cmp_ge_f32_e64 s[0:1], v15, v14)
            f2 q5, q6;
            float e5l, e5h, e6l, e6h, junk;
            asm volatile("v_mul_f32 %0, -0.5, %1" : "=v"(e5l) : "v"(a.x));
            asm volatile("v_mul_f32 %0, -0.5, %1" : "=v"(e5h) : "v"(a.y));
            asm volatile("v_mul_f32 %0, 0.5, %1" : "=v"(e6h) : "v"(a.y));
            e6l = e5l;
            asm volatile("v_pk_mul_f32 %0, %1, -0.5 op_sel_hi:[1,0]" : "=&v"(q5) : "v"(a));
            asm volatile("s_mov_b32 s20, -0.5\n\ts_mov_b32 s21, 0.5\n\t"
                         "v_pk_mul_f32 %0, %2, s[20:21]\n\t"
                         "v_mul_f32 %1, %3, %3\n\t"
                         "v_cmp_ge_f32_e64 s[20:21], %4, %3"
                         : "=&v"(q6), "=&v"(junk) : "v"(a), "v"(b.x), "v"(b.y) : "s20", "s21");
            bad_const += (__float_as_uint(q5.x) != __float_as_uint(e5l)) + (__float_as_uint(q5.y) != __float_as_uint(e5h));
            bad_sgpr += (__float_as_uint(q6.x) != __float_as_uint(e6l)) + (__float_as_uint(q6.y) != __float_as_uint(e6h));
            bad_inpl1 += (__float_as_uint(q1.x) != __float_as_uint(e1l)) + (__float_as_uint(q1.y) != __float_as_uint(e1h));
            bad_inpl2 += (__float_as_uint(q2.x) != __float_as_uint(e2l)) + (__float_as_uint(q2.y) != __float_as_uint(e2h));
            bad_dist += (__float_as_uint(q3.x) != __float_as_uint(e1l)) + (__float_as_uint(q3.y) != __float_as_uint(e1h));
        }
        bad_fma += (__float_as_uint(pf.x) != __float_as_uint(s0)) + (__float_as_uint(pf.y) != __float_as_uint(s1));
        bad_mul += (__float_as_uint(pm.x) != __float_as_uint(m0)) + (__float_as_uint(pm.y) != __float_as_uint(m1));
        bad_add += (__float_as_uint(pa.x) != __float_as_uint(a0)) + (__float_as_uint(pa.y) != __float_as_uint(a1));
        // next iteration's operands: bounded, lane-dependent, not loop-invariant
        a.x = 0.75f * a.x + 0.01f * s0 + 0.003f; a.y = 0.75f * a.y - 0.01f * m1 + 0.001f;
        b.x = 0.9f * b.x + 0.05f;                 b.y = 0.9f * b.y + 0.0625f;
        c.x = 0.5f * c.x + 0.1f * a0;             c.y = 0.5f * c.y - 0.1f * a1;
    }
    const unsigned long long bad = bad_fma + bad_mul + bad_add + bad_inpl1 + bad_inpl2 + bad_dist + bad_const + bad_sgpr;
    if (bad) {
        atomicAdd(&err[lane], bad);
        atomicAdd(&err[64], bad_fma);
        atomicAdd(&err[65], bad_mul);
        atomicAdd(&err[66], bad_add);
        atomicAdd(&err[68], bad_inpl1);
        atomicAdd(&err[69], bad_inpl2);
        atomicAdd(&err[70], bad_dist);
        atomicAdd(&err[71], bad_const);
        atomicAdd(&err[72], bad_sgpr);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&err[67], 1ull);      // launches that ran to completion
}

// round 4, second probe: the CONTEXT of the failing code -- in box_corners the packed multiplies are the first consumers of a
// global_load_dwordx4 (s_waitcnt vmcnt(0) directly in front of them).  Every lane loads 16 bytes, a packed multiply consumes two of
// the loaded registers right behind the wait; the same products are taken again with scalar multiplies after ~40 idle cycles and
// compared bitwise.  A load whose last lanes land after the wait released (VGPR write port contended by another wave's MFMA
// results) would show up as mismatches in the late lanes.
extern "C" __global__ __launch_bounds__(256) void k_victim_load(const float *data, int n4, unsigned long long *err, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 7919u;
    for (int it = 0; it < iters; ++it) {
        idx = idx * 1664525u + 1013904223u;
        const float *p = data + (size_t)(idx % (unsigned)n4) * 4;
        f2 q;
        float l0, l1;
        asm volatile("global_load_dwordx4 v[100:103], %3, off\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_pk_mul_f32 %0, v[100:101], -0.5 op_sel_hi:[1,0]\n\t"      // first consumer of the loaded registers, right behind the wait
                     "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mul_f32 %1, -0.5, v100\n\t"
                     "v_mul_f32 %2, -0.5, v101"
                     : "=&v"(q), "=&v"(l0), "=&v"(l1) : "v"(p) : "memory", "v100", "v101", "v102", "v103");
        bad += (__float_as_uint(q.x) != __float_as_uint(l0)) + (__float_as_uint(q.y) != __float_as_uint(l1));
    }
    if (bad) { atomicAdd(&err[lane], bad); atomicAdd(&err[73], bad); }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&err[74], 1ull);
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
extern "C" __attribute__((visibility("default"))) int pk_launch_victim_load(const float *data, int n4, unsigned long long *err, int blocks, int iters, void *stream) {
    hipLaunchKernelGGL(k_victim_load, dim3(blocks), dim3(256), 0, (hipStream_t)stream, data, n4, err, iters);
    return (int)hipGetLastError();
}
