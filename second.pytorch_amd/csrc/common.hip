// Library-wide helpers: error reporting and the device-wide exclusive scan used by the voxeliser and
// the rulebook builders (wave64 shuffles -> block scan -> three-launch device scan).
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace sec {

static char g_last_error[256] = "";
void set_last_error(hipError_t e) {
    strncpy(g_last_error, hipGetErrorString(e), sizeof(g_last_error) - 1);
}
static thread_local char g_last_kernel[192] = "";
void set_last_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}

__global__ __launch_bounds__(kBlock) void k_scan_reduce(const int *__restrict__ in, long long n,
                                                       int *__restrict__ block_sums) {
    __shared__ int smem[4];
    long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) s += in[base + i];
    int tot;
    block_exclusive_scan(s, smem, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void k_scan_sums(int *__restrict__ block_sums, int nblocks,
                                                     int *__restrict__ total_out) {
    __shared__ int smem[4];
    int carry = 0;
    for (int start = 0; start < nblocks; start += kBlock) {
        int i = start + threadIdx.x;
        int v = i < nblocks ? block_sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, smem, &tot);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        block_sums[nblocks] = carry;
        if (total_out) *total_out = carry;
    }
}

__global__ __launch_bounds__(kBlock) void k_scan_apply(const int *__restrict__ in, int *__restrict__ out,
                                                      long long n, const int *__restrict__ block_sums) {
    __shared__ int smem[4];
    long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan(s, smem, &tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
}

// single-workgroup scan (1024 threads, each a contiguous run): one launch instead of three for small n
constexpr int kSoloThreads = 1024;
__global__ __launch_bounds__(kSoloThreads) void k_scan_solo(const int *__restrict__ in, int *__restrict__ out, int n,
                                                           int *__restrict__ total_out) {
    __shared__ int wtot[kSoloThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n + kSoloThreads - 1) / kSoloThreads;
    const int lo = tid * per, hi = lo + per < n ? lo + per : n;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += in[i];
    int inc = wave_inclusive_scan(s);
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kSoloThreads / 64; ++i) {
        int v = wtot[i];
        if (i < wv) base += v;
        tot += v;
    }
    int run = base + inc - s;
    for (int i = lo; i < hi; ++i) {
        int v = in[i];
        out[i] = run;
        run += v;
    }
    if (tid == 0 && total_out) *total_out = tot;
}

int exclusive_scan_i32(const int *in, int *out, long long n, int *total_out, int *scratch, hipStream_t st) {
    if (n > 0 && n <= 16 * 1024) {   // tiny scans only: a thread-contiguous run is uncoalesced, 128-element runs cost more than 3 launches
        hipLaunchKernelGGL(k_scan_solo, dim3(1), dim3(kSoloThreads), 0, st, in, out, (int)n, total_out);
        return check_launch();
    }
    if (n <= 0) {
        if (total_out) return hip_ok(hipMemsetAsync(total_out, 0, sizeof(int), st));
        return SEC_OK;
    }
    int nblocks = div_up(n, kScanTile);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nblocks), dim3(kBlock), 0, st, in, n, scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, st, scratch, nblocks, total_out);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(kBlock), 0, st, in, out, n, scratch);
    return check_launch();
}

}  // namespace sec

SEC_API int sec_abi_version(void) { return SEC_ABI_VERSION; }
SEC_API const char *sec_last_error(void) { return sec::g_last_error; }
SEC_API const char *sec_last_kernel_name(void) { return sec::g_last_kernel; }
