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
        if (total_out) return fill_words(total_out, sizeof(int), 0u, st);
        return SEC_OK;
    }
    int nblocks = div_up(n, kScanTile);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nblocks), dim3(kBlock), 0, st, in, n, scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, st, scratch, nblocks, total_out);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(kBlock), 0, st, in, out, n, scratch);
    return check_launch();
}

__global__ __launch_bounds__(kBlock) void k_fill_words(unsigned *__restrict__ p, long long n, unsigned v) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = v;
}
int fill_words(void *p, size_t bytes, unsigned v, hipStream_t st) {
    if (bytes == 0) return SEC_OK;
    if ((bytes & 3) || ((size_t)p & 3)) return SEC_E_INVALID;
    const long long n = (long long)(bytes / 4);
    long long blocks = div_up(n, (long long)kBlock * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fill_words, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<unsigned *>(p), n, v);
    return check_launch();
}

// ---- content checksum of a set of tensors (sec_tensors_checksum): sums[2 i] = sum of tensor i's 32-bit words, sums[2 i + 1] = sum of
// word * (index + 1), both mod 2^64 -- order-independent accumulation (atomics), position-sensitive result.  One launch over all
// tensors; the table of pointers travels as kernel arguments.
constexpr int kCsumTensors = 168, kCsumChunk = 16384;       // words per workgroup
struct CsumArgs {
    const unsigned *p[kCsumTensors];
    unsigned words[kCsumTensors];       // 32-bit words (a trailing 16-bit half, if any, is counted as one more word)
    int blk0[kCsumTensors + 1];
    int n;
};
__global__ __launch_bounds__(kBlock) void k_tensors_checksum(CsumArgs a, int slot0, unsigned long long *__restrict__ sums) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.blk0[i + 1]) ++i;
    const unsigned w0 = (unsigned)((int)blockIdx.x - a.blk0[i]) * kCsumChunk;
    const unsigned w1 = w0 + kCsumChunk < a.words[i] ? w0 + kCsumChunk : a.words[i];
    const unsigned *p = a.p[i];
    unsigned long long s1 = 0ull, s2 = 0ull;
    for (unsigned k = w0 + threadIdx.x; k < w1; k += kBlock) {
        const unsigned long long v = p[k];
        s1 += v;
        s2 += v * (unsigned long long)(k + 1u);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s1 += __shfl_xor(s1, d, 64);
        s2 += __shfl_xor(s2, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[2 * (slot0 + i)], s1);
        atomicAdd(&sums[2 * (slot0 + i) + 1], s2);
    }
}

}  // namespace sec

SEC_API int sec_tensors_checksum(const void *const *h_ptrs, const long long *h_nbytes, int count, unsigned long long *sums, void *stream) {
    using namespace sec;
    if (count < 0 || !sums || (count > 0 && (!h_ptrs || !h_nbytes))) return SEC_E_INVALID;
    for (int i = 0; i < count; ++i)
        if (!h_ptrs[i] || h_nbytes[i] < 0 || (h_nbytes[i] & 3) || ((size_t)h_ptrs[i] & 3) || h_nbytes[i] / 4 > 0xffffffffll) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // (zeroed by a kernel, not hipMemsetAsync: the memset NODE of a captured graph did not reliably precede the atomics on replay --
    // the second replay of a drop-in session compared doubled sums)
    int rc;
    if ((rc = fill_words(sums, (size_t)count * 2 * sizeof(unsigned long long), 0u, st))) return rc;
    for (int i0 = 0; i0 < count; i0 += kCsumTensors) {
        CsumArgs a;
        a.n = count - i0 < kCsumTensors ? count - i0 : kCsumTensors;
        int blocks = 0;
        for (int j = 0; j < a.n; ++j) {
            a.p[j] = reinterpret_cast<const unsigned *>(h_ptrs[i0 + j]);
            a.words[j] = (unsigned)(h_nbytes[i0 + j] / 4);
            a.blk0[j] = blocks;
            blocks += (int)div_up((long long)a.words[j], kCsumChunk);
        }
        a.blk0[a.n] = blocks;
        if (blocks > 0) hipLaunchKernelGGL(k_tensors_checksum, dim3(blocks), dim3(kBlock), 0, st, a, i0, sums);
    }
    return check_launch();
}

SEC_API int sec_abi_version(void) { return SEC_ABI_VERSION; }
SEC_API const char *sec_last_error(void) { return sec::g_last_error; }
SEC_API const char *sec_last_kernel_name(void) { return sec::g_last_kernel; }
