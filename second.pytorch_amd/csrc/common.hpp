// Shared device/host helpers for libsecond_hip.so (gfx950 / CDNA4 only: wave64, 256 CUs).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/second_hip.h"

#define SEC_API extern "C" __attribute__((visibility("default")))

namespace sec {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // default workgroup: 4 waves, one per SIMD
constexpr int kEmptyI32 = 0x7f7f7f7f;               // hipMemsetAsync(…, 0x7f) pattern: > any row index
constexpr unsigned long long kEmptyKey = ~0ull;     // hipMemsetAsync(…, 0xff) pattern

void set_last_error(hipError_t e);
// name (as a profiler prints it) of the kernel instantiation the dispatchers of sec_indice_conv_fwd / sec_conv2d_nhwc picked last
void set_last_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
template <typename T> inline const char *dtype_name() { return "float"; }
template <> inline const char *dtype_name<__hip_bfloat16>() { return "__hip_bfloat16"; }
template <> inline const char *dtype_name<__half>() { return "__half"; }
inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error(e); return SEC_E_LAUNCH; }
    return SEC_OK;
}
inline int hip_ok(hipError_t e) {
    if (e != hipSuccess) { set_last_error(e); return SEC_E_LAUNCH; }
    return SEC_OK;
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
inline uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// bump allocator over the caller-provided workspace
struct Arena {
    char *base;
    size_t used, cap;
    Arena(void *p, size_t c) : base((char *)p), used(0), cap(c) {}
    template <typename T> T *take(size_t n) {
        size_t off = align_up(used);
        used = off + n * sizeof(T);
        return (T *)(base + off);
    }
    bool ok() const { return used <= cap; }
};

// ---------------------------------------------------------------- hashing (open addressing, linear probe)
__device__ __forceinline__ uint32_t hash64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}
// returns the slot holding `key`, inserting it if absent
__device__ __forceinline__ uint32_t hash_insert(unsigned long long *keys, uint32_t mask, unsigned long long key) {
    uint32_t s = hash64(key) & mask;
    while (true) {
        unsigned long long prev = atomicCAS(&keys[s], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) return s;
        s = (s + 1) & mask;
    }
}
// bounded variant: gives up (-1) after `mask + 1` probes, i.e. when the table is full -- a sync-free
// pipeline with under-estimated capacity must degrade into a reported overflow, never into a hang
__device__ __forceinline__ int hash_insert_bounded(unsigned long long *keys, uint32_t mask, unsigned long long key) {
    uint32_t s = hash64(key) & mask;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        // peek first (agent-scope load, L2): most candidates of a strided conv find their cell already inserted, and a
        // returning compare-and-swap is a fabric round trip.  A slot only ever goes empty -> key, so a stale "empty"
        // merely costs the CAS we would have issued anyway and a non-empty value is final.
        unsigned long long cur = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return (int)s;
        if (cur == kEmptyKey) {
            cur = atomicCAS(&keys[s], kEmptyKey, key);
            if (cur == kEmptyKey || cur == key) return (int)s;
        }
        s = (s + 1) & mask;
    }
    return -1;
}
// returns slot or -1.  Bounded like the insert: a table that a bounded insert filled completely (reported overflow of a
// static-capacity build) has no empty slot left to stop the probe of an absent key -- give up after one sweep instead
// of spinning inside a captured graph.
__device__ __forceinline__ int hash_find(const unsigned long long *keys, uint32_t mask, unsigned long long key) {
    uint32_t s = hash64(key) & mask;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        unsigned long long cur = keys[s];
        if (cur == key) return (int)s;
        if (cur == kEmptyKey) return -1;
        s = (s + 1) & mask;
    }
    return -1;
}

// ---------------------------------------------------------------- wave / block scans (wave64)
// `ok ? p[i] : d` as an UNCONDITIONAL load from a clamped index plus a select.  hipcc compiles the conditional form to a branch with
// `s_waitcnt vmcnt(0)` behind the load, so "N independent loads in flight" in an unrolled loop become N dependent round trips (found
// with the dense weight gradient, profiles/r05_h_wgrad_pmc.txt).  p[0] must be readable (the array is not empty).
template <typename T, typename I>
__device__ __forceinline__ T ld_sel(const T *__restrict__ p, I i, bool ok, T d) {
    const T v = p[ok ? i : (I)0];
    return ok ? v : d;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}
// exclusive scan across a 256-thread block; returns exclusive prefix, *total = block sum. smem: >= 4 ints
__device__ __forceinline__ int block_exclusive_scan(int v, int *smem, int *total) {
    int inc = wave_inclusive_scan(v);
    int w = threadIdx.x >> 6;
    if (lane_id() == 63) smem[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kBlock / 64; ++i) {
        int s = smem[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------- single-pass grid scan (decoupled look-back)
// A producer kernel that already visits every element scans in place instead of writing flags for three more
// launches.  Tiles are handed out by an atomic ticket (a spinning tile only waits on tiles that are already running);
// a workgroup publishes its aggregate, then one wave inspects up to 64 predecessors per step until it meets an
// inclusive prefix.  status[t] = flag << 32 | value (1 = aggregate, 2 = inclusive prefix, 0 = not yet); the ticket and
// the status words must be zero at launch (the init kernel of the same pipeline clears them).
constexpr unsigned long long kScanAgg = 1ull << 32, kScanIncl = 2ull << 32;
inline size_t scan_ctl_words(long long n) { return 4 + 2 * (size_t)div_up(n > 0 ? n : 1, kBlock); }   // [ticket,-,-,-,status...]
__device__ __forceinline__ int scan_take_tile(int *ticket, int *s_tile) {
    if (threadIdx.x == 0) *s_tile = atomicAdd(ticket, 1);
    __syncthreads();
    return *s_tile;
}
// every thread of the 256-thread block calls this with its value; returns the exclusive prefix over the whole grid
// in tile order; the last tile stores the grand total.  smem: >= 5 ints.
__device__ __forceinline__ int scan_lookback(int v, int tile, int ntiles, unsigned long long *status, int *smem,
                                             int *total_out) {
    int total;
    const int ex = block_exclusive_scan(v, smem, &total);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(&status[tile], (tile == 0 ? kScanIncl : kScanAgg) | (unsigned)total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        int prefix = 0;
        for (int look = tile - 1; look >= 0; look -= 64) {
            const int idx = look - lane;
            unsigned long long sv = kScanIncl;                // before the first tile: inclusive prefix 0
            if (idx >= 0) {
                while (((sv = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0)
                    __builtin_amdgcn_s_sleep(1);
            }
            const unsigned long long incl = __ballot((sv >> 32) == 2);
            const int stop = incl ? __ffsll((long long)incl) - 1 : 64;   // nearest predecessor holding an inclusive prefix
            int part = lane <= stop ? (int)(unsigned)sv : 0;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
            prefix += part;
            if (incl) break;
        }
        if (lane == 0) {
            if (tile > 0)
                __hip_atomic_store(&status[tile], kScanIncl | (unsigned)(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            smem[4] = prefix;
            if (tile == ntiles - 1 && total_out) *total_out = prefix + total;
        }
    }
    __syncthreads();
    return smem[4] + ex;
}

// device-wide exclusive scan of int32 (three launches; n up to ~2^31). Scratch: scan_scratch_ints(n) ints.
constexpr int kScanItems = 8;                      // per thread
constexpr int kScanTile = kBlock * kScanItems;     // 2048 per block
inline size_t scan_scratch_ints(long long n) { return (size_t)div_up(n, kScanTile) + 8; }
int exclusive_scan_i32(const int *in, int *out, long long n, int *total_out, int *scratch, hipStream_t st);
// `bytes` (a multiple of 4) of `p` (4-byte aligned) := the 32-bit word `v`, by a KERNEL.  Use this, not hipMemsetAsync, in anything
// that may be captured into a hipGraph: on ROCm 7.2 the runtime's memset NODE misbehaves on replay (scatter.hip: a non-zero pattern
// after some tens of replays; round 6: atomics that followed it accumulated onto the previous replay's values).
int fill_words(void *p, size_t bytes, unsigned v, hipStream_t st);

// voxelize.hip: hash table left in the workspace of sec_voxelize_f32 (see sec_rulebook_subm3d_after_voxelize)
bool vox_slots_of(const void *ws, size_t bytes, int n, int batch, int max_voxels, int max_points, const int **count,
                  const int **slot_idx);
bool vox_table_of(const void *ws, size_t bytes, int n, int batch, int max_voxels, int max_points, const unsigned long long **keys,
                  const int **svid, uint32_t *mask);

}  // namespace sec
