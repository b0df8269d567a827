// Probe: which (XCC, CU) a stream created by hipExtStreamCreateWithCUMask runs on, per mask bit range -- the bit -> CU layout of the
// 256-bit mask on MI355X (8 XCCs x 32 CUs).  Every mask used here leaves CUs enabled in EVERY XCC under both layout hypotheses
// (bit i -> XCC i % 8, or bit i -> XCC i / 32): a mask that empties an XCC might never retire the workgroups dispatched to it.
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_where(int *hist, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg((16 << 11) | (0 << 6) | 4);       // HW_ID[15:0]: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 7u; // XCC_ID[3:0]
    const long long until = clock64() + spin;
    while (clock64() < until) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) atomicAdd(&hist[xcc * 256 + ((hw >> 8) & 0xff)], 1);
}
static int g_cus[8];
static int run(const char *name, const unsigned *mask, bool masked) {
    hipStream_t st;
    if (masked) CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
    else CK(hipStreamCreate(&st));
    int *hist;
    CK(hipMalloc(&hist, 8 * 256 * 4));
    CK(hipMemsetAsync(hist, 0, 8 * 256 * 4, st));
    hipLaunchKernelGGL(k_where, dim3(16384), dim3(64), 0, st, hist, 20000);
    CK(hipStreamSynchronize(st));
    std::vector<int> h(8 * 256);
    CK(hipMemcpy(h.data(), hist, 8 * 256 * 4, hipMemcpyDeviceToHost));
    printf("%-44s", name);
    int total = 0;
    for (int x = 0; x < 8; ++x) {
        int cus = 0, wgs = 0;
        for (int c = 0; c < 256; ++c) if (h[x * 256 + c]) { ++cus; wgs += h[x * 256 + c]; }
        printf("  xcc%d: %2d CUs %5d wg", x, cus, wgs);
        g_cus[x] = cus;
        total += cus;
    }
    printf("   = %d CUs\n", total);
    if (masked && total <= 64) {      // which hardware CU ids (se/sh/cu byte) per XCC
        for (int x = 0; x < 8; ++x) {
            printf("    xcc%d ids:", x);
            for (int c = 0; c < 256; ++c) if (h[x * 256 + c]) printf(" %02x", c);
            printf("\n");
        }
    }
    CK(hipFree(hist));
    CK(hipStreamDestroy(st));
    return 0;
}
int main() {
    unsigned m[8];
    memset(m, 0xff, sizeof m);
    if (run("no mask", m, false)) return 1;
    if (run("all 256 bits", m, true)) return 1;
    memset(m, 0xff, sizeof m); m[0] = 0xffff0000u;                       // bits 0..15 off
    if (run("bits 0..15 off", m, true)) return 1;
    memset(m, 0xff, sizeof m); m[0] = 0xffffff00u;                       // bits 0..7 off
    if (run("bits 0..7 off", m, true)) return 1;
    // a quarter of every XCC under the interleaved hypothesis (bit i -> XCC i % 8, CU slot i / 8): slots 8 q .. 8 q + 7 = bits 64 q .. 64 q + 63;
    // under the contiguous hypothesis that would be two XCCs complete and six empty -- only run when the mask above took ONE CU from every XCC
    bool interleaved = true;
    for (int x = 0; x < 8; ++x) interleaved = interleaved && g_cus[x] == 31;
    printf("layout: %s\n", interleaved ? "interleaved (bit i -> XCC i % 8)" : "NOT interleaved");
    if (!interleaved) return 0;
    for (int q = 0; q < 4; ++q) {
        memset(m, 0, sizeof m); m[2 * q] = m[2 * q + 1] = 0xffffffffu;
        char name[64]; snprintf(name, sizeof name, "bits %d..%d only", 64 * q, 64 * q + 63);
        if (run(name, m, true)) return 1;
    }
    return 0;
}
