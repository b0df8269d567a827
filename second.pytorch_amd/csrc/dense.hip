// Dense-side helpers for the RPN (second/pytorch/models/rpn.py:468-497): the convolutions themselves run
// through MIOpen in phase 1; the per-channel bias (folded BatchNorm2d) + ReLU that follows every conv is ONE
// in-place pass here instead of MIOpen's separate bias tensor-op plus a ReLU kernel (3 passes -> 1).
#include "common.hpp"

namespace sec {

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ __hip_bfloat16 from_f(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half from_f(float v) { return __float2half_rn(v); }

// x: [pixels, C] channels-last, 16-bit; 8 channels (16 bytes) per thread; C % 8 == 0
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bias_act16(T *__restrict__ x, const float *__restrict__ bias, long long n_vec,
                                                      int c_vec, int relu) {
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < n_vec; g += (long long)gridDim.x * kBlock) {
        uint4 v = reinterpret_cast<uint4 *>(x)[g];
        T *e = reinterpret_cast<T *>(&v);
        const float *b = bias + (size_t)(g % c_vec) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float f = to_f<T>(e[i]) + b[i];
            if (relu) f = f > 0.0f ? f : 0.0f;
            e[i] = from_f<T>(f);
        }
        reinterpret_cast<uint4 *>(x)[g] = v;
    }
}

__global__ __launch_bounds__(kBlock) void k_bias_act32(float *__restrict__ x, const float *__restrict__ bias, long long n_vec,
                                                      int c_vec, int relu) {
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < n_vec; g += (long long)gridDim.x * kBlock) {
        float4 v = reinterpret_cast<float4 *>(x)[g];
        const float *b = bias + (size_t)(g % c_vec) * 4;
        v.x += b[0]; v.y += b[1]; v.z += b[2]; v.w += b[3];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4 *>(x)[g] = v;
    }
}

}  // namespace sec

using namespace sec;

SEC_API int sec_bias_act_nhwc(void *x, const float *bias, size_t pixels, int channels, int relu, int dtype, void *stream) {
    if (!x || !bias || channels <= 0) return SEC_E_INVALID;
    if (pixels == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_F32) {
        if (channels % 4) return SEC_E_UNSUPPORTED;
        long long n = (long long)pixels * channels / 4;
        int blocks = div_up(n, kBlock); if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(k_bias_act32, dim3(blocks), dim3(kBlock), 0, st, (float *)x, bias, n, channels / 4, relu);
    } else {
        if (channels % 8) return SEC_E_UNSUPPORTED;
        long long n = (long long)pixels * channels / 8;
        int blocks = div_up(n, kBlock); if (blocks > 256 * 16) blocks = 256 * 16;
        if (dtype == SEC_BF16)
            hipLaunchKernelGGL(k_bias_act16<__hip_bfloat16>, dim3(blocks), dim3(kBlock), 0, st, (__hip_bfloat16 *)x, bias, n, channels / 8, relu);
        else if (dtype == SEC_F16)
            hipLaunchKernelGGL(k_bias_act16<__half>, dim3(blocks), dim3(kBlock), 0, st, (__half *)x, bias, n, channels / 8, relu);
        else return SEC_E_UNSUPPORTED;
    }
    return check_launch();
}
