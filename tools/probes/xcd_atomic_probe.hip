// Probe: throughput of returning atomicAdd on random words, device scope on one array vs XCD-local (no sc1) on a private copy per XCD
// (copy chosen by HW_REG_XCC_ID).  hipcc --offload-arch=gfx950 -O3 -o xcd_atomic_probe xcd_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__device__ __forceinline__ unsigned rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>   // 0: agent-scope returning, 1: agent-scope non-returning, 2: XCD-local returning, 3: XCD-local non-returning
__global__ __launch_bounds__(256) void k_probe(int *tab, int words, int n, int *sink, int *xcc_hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 << 11) | 20) & 7u;
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[xcc], 1);
    const int w = rnd(i * 2654435761u + 12345u) % words;
    int r = 0;
    if (MODE == 0) r = __hip_atomic_fetch_add(&tab[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) (void)__hip_atomic_fetch_add(&tab[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) r = __hip_atomic_fetch_add(&tab[(size_t)xcc * words + w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 3) (void)__hip_atomic_fetch_add(&tab[(size_t)xcc * words + w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (r == 0x7fffffff) sink[0] = r;
}
__global__ void k_sum(const int *tab, long long n, unsigned long long *out) {
    unsigned long long s = 0;
    for (long long i = blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (unsigned)tab[i];
    atomicAdd(out, s);
}
int main() {
    const int n = 1172931, words = 102579;
    int *tab, *sink, *hist; unsigned long long *tot;
    hipMalloc(&tab, (size_t)8 * words * 4); hipMalloc(&sink, 4); hipMalloc(&hist, 32); hipMalloc(&tot, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipMemset(tab, 0, (size_t)8 * words * 4); hipMemset(hist, 0, 32); hipMemset(tot, 0, 8);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, 0, tab, words, n, sink, hist);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipLaunchKernelGGL(k_sum, dim3(512), dim3(256), 0, 0, tab, (long long)8 * words, tot);
        unsigned long long t; hipMemcpy(&t, tot, 8, hipMemcpyDeviceToHost);
        int h[8]; hipMemcpy(h, hist, 32, hipMemcpyDeviceToHost);
        printf("%-34s %7.1f us  %6.1f G/s  sum=%llu (expect %d)  wg per xcc:", name, best * 1e3, n / (best * 1e-3) / 1e9, t, n);
        for (int x = 0; x < 8; ++x) printf(" %d", h[x]);
        printf("\n");
    };
    run(k_probe<0>, "agent returning");
    run(k_probe<1>, "agent non-returning");
    run(k_probe<2>, "xcd-local returning");
    run(k_probe<3>, "xcd-local non-returning");
    return 0;
}
