// points_to_voxel on gfx950: a deterministic, sort-free parallel restatement of the sequential
// hard-voxelisation loop (reference: spconv VoxelGeneratorV2.generate, called at
// second/data/preprocess.py:301-316; in-repo copy of the loop: second/utils/simplevis.py:31-50).
//
// Sequential semantics to reproduce bit-exactly (per cloud):
//   voxel id   = order of first occurrence of the voxel among the points,
//   slot       = order of arrival of the point inside its voxel, first `max_points` kept,
//   cap        = at most `max_voxels` voxels (`break` or `continue` at the cap).
// Parallel formulation (all clouds of a batch in one pass, HBM/L2-bound integer work):
//   1. hash insert  key=(cloud, linear cell) -> atomicMin(point index)      [first point of each voxel]
//   2. flag first points, device-wide exclusive scan -> voxel rank = sequential voxel id
//   3. per voxel, the `max_points` smallest point indices via an atomicMin cascade
//      (position t keeps the min of everything that reaches it and forwards the loser, so position t
//       ends up holding the (t+1)-th smallest index regardless of interleaving)
//   4. voxel-major fill (coalesced stores) + optional SimpleVoxel mean epilogue.
#include "common.hpp"
#include <type_traits>

namespace sec {

constexpr int kFusedItems = 2;    // points per thread of k_vox_scan_assign_cascade
constexpr int kFusedFrames = 64;  // batches up to this size compute the per-cloud frames inside k_vox_assign

struct VoxParams {
    float lo[3], vs[3];
    int grid[3];  // x, y, z
    int num_points, num_features, batch, max_points, max_voxels, cap_mode;
    int peek;        // k_vox_hash: look before the atomics (pays when most points share their cell with an earlier point)
    uint32_t table_mask;
    // Pillar path (many points per voxel, small grid: sec_voxelize_f32 with voxels == NULL):
    int stage;       // k_vox_cell_first: 0 = every point; 1 / 2 = the EARLY points of every cloud (in-cloud index < len / kStageDiv) / the others
    int dense_cells; // > 0: cells per cloud of the DENSE slot numbering (slot = cloud * cells + linear cell; no key array, no probing)
};

// Staged first-point pass of a pillar config (dense numbering, from a quarter of a million points).  One device-scope atomicMin per
// point runs at the fabric's atomic rate, ~11 G/s: 105 us for the 1.17 M points of config 4 (the hash form with its two device-scope
// loads per point took 95).  The first eighth of every cloud's points is enough to settle most slots: after it -- a kernel boundary
// later -- a slot that any early point touched holds an index below every later point's, so the later points look with a plain,
// L2-cached load and skip the atomic (a stale value can only be LARGER: it costs the atomic it would have saved; slots no early point
// touched still take their atomicMin).  13 + 37 us for the two launches (r06_stage*), exact.
constexpr int kStageDiv = 8;
__device__ __forceinline__ bool stage_early(const int *__restrict__ offs, int b, int i) {
    const int q = i - offs[b], len = offs[b + 1] - offs[b];
    return (long long)q * kStageDiv < len;
}

__device__ __forceinline__ int frame_of(const int *__restrict__ offs, int batch, int i) {
    int lo = 0, hi = batch;  // find b with offs[b] <= i < offs[b+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void k_vox_hash(const float *__restrict__ points,
                                                    const int *__restrict__ offs, VoxParams p,
                                                    unsigned long long *__restrict__ keys,
                                                    int *__restrict__ vals, int *__restrict__ pslot) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.num_points) return;
    if (i >= offs[p.batch]) { pslot[i] = -1; return; }  // capacity rows beyond the live point count
    const float *pt = points + (size_t)i * p.num_features;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // fp32 IEEE subtract + correctly rounded divide + floor, exactly like the reference loop
        float q = floorf(__fdiv_rn(__fsub_rn(pt[j], p.lo[j]), p.vs[j]));
        if (!(q >= 0.0f) || !(q < (float)p.grid[j])) ok = false;
        c[j] = (int)q;
    }
    if (!ok) { pslot[i] = -1; return; }
    int b = frame_of(offs, p.batch, i);
    unsigned long long vol = (unsigned long long)p.grid[0] * p.grid[1] * p.grid[2];
    unsigned long long lin = ((unsigned long long)c[2] * p.grid[1] + c[1]) * p.grid[0] + c[0];
    // Peek before the atomics (agent-scope loads: L2): a slot only goes empty -> key and vals[] only decreases, so a key already in
    // place / an index already smaller is final and the returning compare-and-swap (a fabric round trip) and the atomicMin can be
    // skipped; a stale read only costs the atomic we would have issued anyway.  Pillars collect hundreds of points each
    // (nuscenes/all.pp.largea): most points find their cell taken by an earlier one.
    // (p.peek is set by the host when the points outnumber the voxel capacity more than twice -- nuScenes sweeps: 358 -> 335 us
    // for config 4's voxeliser, 230 -> 211 us for config 5's; on KITTI-like clouds, ~1 point per voxel, the extra loads cost 5 us.)
    const unsigned long long key = (unsigned long long)b * vol + lin;
    uint32_t s;
    if (p.peek) {
        s = hash64(key) & p.table_mask;
        while (true) {
            unsigned long long cur = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == key) break;
            if (cur == kEmptyKey) {
                cur = atomicCAS(&keys[s], kEmptyKey, key);
                if (cur == kEmptyKey || cur == key) break;
            }
            s = (s + 1) & p.table_mask;
        }
        if (__hip_atomic_load(&vals[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&vals[s], i);
    } else {
        s = hash_insert(keys, p.table_mask, key);
        atomicMin(&vals[s], i);
    }
    pslot[i] = (int)s;
}

// DENSE numbering (pillar grids: batch * cells fits the table the workspace holds anyway): a cell's slot is its linear index, so the
// first-point pass is ONE non-returning atomicMin per point (no key array, no compare-and-swap, no probing, nothing to wait for) and
// vals[] is 2.5 MB for config 4 instead of the 48 MB hash table of 1.17 M points.  Stage 2 looks first (plain load, see above).
__global__ __launch_bounds__(kBlock) void k_vox_cell_first(const float *__restrict__ points, const int *__restrict__ offs, VoxParams p,
                                                          int *__restrict__ vals, int *__restrict__ pslot) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.num_points) return;
    if (i >= offs[p.batch]) { if (p.stage != 2) pslot[i] = -1; return; }        // capacity rows beyond the live point count
    const int b = frame_of(offs, p.batch, i);
    if (p.stage && stage_early(offs, b, i) != (p.stage == 1)) return;           // the other launch's point
    const float *pt = points + (size_t)i * p.num_features;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float q = floorf(__fdiv_rn(__fsub_rn(pt[j], p.lo[j]), p.vs[j]));       // as k_vox_hash: the reference loop's arithmetic
        if (!(q >= 0.0f) || !(q < (float)p.grid[j])) ok = false;
        c[j] = (int)q;
    }
    if (!ok) { pslot[i] = -1; return; }
    const int s = b * p.dense_cells + (c[2] * p.grid[1] + c[1]) * p.grid[0] + c[0];
    if (p.stage != 2 || __hip_atomic_load(&vals[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > i) atomicMin(&vals[s], i);
    pslot[i] = s;
}

__global__ __launch_bounds__(kBlock) void k_vox_init(unsigned long long *__restrict__ keys, int *__restrict__ vals, long long table,
                                                    int *__restrict__ count, long long rows, int *__restrict__ slot_idx,
                                                    long long slots, int *__restrict__ ctl, long long ctl_words,
                                                    int *__restrict__ break_idx, int batch, int *__restrict__ svid,
                                                    unsigned long long *__restrict__ frame_words) {
    long long stride = (long long)gridDim.x * kBlock;
    if (blockIdx.x == 0)
        for (int b = threadIdx.x; b < batch; b += kBlock) break_idx[b] = 0x7fffffff;      // "no cloud has hit its voxel cap yet"
    if (blockIdx.x == 0 && frame_words)
        for (int b = threadIdx.x; b <= batch; b += kBlock) frame_words[b] = 0ull;         // fused scan: "rank at the cloud's first point" not published yet
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < ctl_words; i += stride) ctl[i] = 0;
    if (svid)      // fused scan: a point of a voxel spins on svid[slot] until the voxel's first point has numbered it
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < table; i += stride) svid[i] = kEmptyI32;
    if (keys)
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < table; i += stride) { keys[i] = kEmptyKey; vals[i] = kEmptyI32; }
    else       // dense numbering: `table` = batch * cells first-point words, no keys
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < table; i += stride) vals[i] = kEmptyI32;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) count[i] = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < slots; i += stride) slot_idx[i] = kEmptyI32;
}

// rank[i] = number of voxel-creating points before point i (exclusive scan of the first-touch flags), in one launch
// (single-pass scan of common.hpp); *total = number of voxels over all clouds.  Four points per thread: a quarter of the tiles, so a
// quarter of the same-address ticket atomics and status words on the look-back chain (round 3).
constexpr int kFlagItems = 4;
// ITEMS = 16 for the sweeps of the nuScenes configs (1.17 M points at batch 4): 286 tiles instead of 1146 -- the launch is as long as its
// same-address ticket atomics take (~25 ns each: 30 us with four points per thread).
constexpr int kFlagItemsBig = 16;
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void k_vox_flag_scan(const int *__restrict__ pslot, const int *__restrict__ vals, int n,
                                                         int *__restrict__ rank, unsigned long long *__restrict__ status,
                                                         int *__restrict__ ticket, int *__restrict__ total) {
    static_assert(ITEMS % 4 == 0, "int4 groups");
    __shared__ int smem[5];
    __shared__ int s_tile;
    const int tile = scan_take_tile(ticket, &s_tile);
    const int i0 = (tile * kBlock + threadIdx.x) * ITEMS;
    int f[ITEMS], v = 0;
    int s4[ITEMS];
    if (i0 + ITEMS <= n) {
#pragma unroll
        for (int g = 0; g < ITEMS / 4; ++g) {
            const int4 q = *reinterpret_cast<const int4 *>(pslot + i0 + 4 * g);
            s4[4 * g] = q.x; s4[4 * g + 1] = q.y; s4[4 * g + 2] = q.z; s4[4 * g + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) s4[j] = ld_sel(pslot, i0 + j, i0 + j < n, -1);
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        f[j] = (ld_sel(vals, s4[j], s4[j] >= 0, -1) == i0 + j) ? 1 : 0;      // unconditional loads: all ITEMS in flight at once
        v += f[j];
    }
    int ex = scan_lookback(v, tile, (int)gridDim.x, status, smem, total);
    if (i0 + ITEMS <= n) {
#pragma unroll
        for (int g = 0; g < ITEMS / 4; ++g) {
            int4 r;
            r.x = ex; ex += f[4 * g];
            r.y = ex; ex += f[4 * g + 1];
            r.z = ex; ex += f[4 * g + 2];
            r.w = ex; ex += f[4 * g + 3];
            *reinterpret_cast<int4 *>(rank + i0 + 4 * g) = r;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (i0 + j < n) rank[i0 + j] = ex;
            ex += f[j];
        }
    }
}

// one thread: per-cloud voxel counts (capped) -> voxel_offsets; rank base per cloud.  Only for empty inputs and batches > 64:
// otherwise every workgroup of k_vox_assign derives the same few numbers itself (one launch less on the latency chain).
__global__ void k_vox_frames(const int *__restrict__ offs, const int *__restrict__ rank,
                             const int *__restrict__ total, VoxParams p, int *__restrict__ base,
                             int *__restrict__ break_idx, int *__restrict__ voxel_offsets) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int tot = *total;
    int acc = 0;
    voxel_offsets[0] = 0;
    int prev = offs[0] < p.num_points ? rank[offs[0]] : tot;
    base[0] = prev;
    for (int b = 0; b < p.batch; ++b) {
        int o = offs[b + 1];
        int nxt = o < p.num_points ? rank[o] : tot;
        base[b + 1] = nxt;
        int cnt = nxt - prev;
        if (cnt > p.max_voxels) cnt = p.max_voxels;
        acc += cnt;
        voxel_offsets[b + 1] = acc;
        prev = nxt;
    }
}

__global__ __launch_bounds__(kBlock) void k_vox_assign(const int *__restrict__ offs,
                                                      const int *__restrict__ pslot,
                                                      const int *__restrict__ vals,
                                                      const unsigned long long *__restrict__ keys,
                                                      const int *__restrict__ rank,
                                                      int *__restrict__ base,
                                                      int *__restrict__ voxel_offsets, VoxParams p,
                                                      int *__restrict__ svid, int *__restrict__ break_idx,
                                                      int *__restrict__ coors, const int *__restrict__ total) {
    // total != NULL: base[] / voxel_offsets[] (what k_vox_frames computes) are derived here, per workgroup, in LDS; workgroup 0
    // also stores them for the kernels that follow
    __shared__ int s_base[kFusedFrames + 1], s_voff[kFusedFrames + 1];
    if (total) {
        const int t = threadIdx.x;
        if (t <= p.batch) {
            const int o = offs[t];
            s_base[t] = o < p.num_points ? rank[o] : *total;
        }
        __syncthreads();
        if (t == 0) {
            int acc = 0;
            s_voff[0] = 0;
            for (int b = 0; b < p.batch; ++b) {
                int cnt = s_base[b + 1] - s_base[b];
                if (cnt > p.max_voxels) cnt = p.max_voxels;
                acc += cnt;
                s_voff[b + 1] = acc;
            }
        }
        __syncthreads();
        if (blockIdx.x == 0 && t <= p.batch) { base[t] = s_base[t]; voxel_offsets[t] = s_voff[t]; }
    }
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.num_points) return;
    int s = pslot[i];
    if (s < 0 || vals[s] != i) return;  // only the first point of a voxel acts
    int b = frame_of(offs, p.batch, i);
    int r = rank[i] - (total ? s_base[b] : base[b]);
    if (r < p.max_voxels) {
        int vid = (total ? s_voff[b] : voxel_offsets[b]) + r;
        svid[s] = vid;
        unsigned long long vol = (unsigned long long)p.grid[0] * p.grid[1] * p.grid[2];
        unsigned long long lin = p.dense_cells ? (unsigned long long)(s - b * p.dense_cells) : keys[s] - (unsigned long long)b * vol;
        int x = (int)(lin % p.grid[0]);
        unsigned long long t = lin / p.grid[0];
        int y = (int)(t % p.grid[1]);
        int z = (int)(t / p.grid[1]);
        int4 c = make_int4(b, z, y, x);
        *reinterpret_cast<int4 *>(coors + (size_t)vid * 4) = c;
    } else {
        svid[s] = -1;
        if (r == p.max_voxels && p.cap_mode == 0) break_idx[b] = i;  // sequential loop `break`s here
    }
}

__global__ __launch_bounds__(kBlock) void k_vox_cascade(const int *__restrict__ offs,
                                                       const int *__restrict__ pslot,
                                                       const int *__restrict__ svid,
                                                       const int *__restrict__ break_idx, VoxParams p,
                                                       int *__restrict__ count, int *__restrict__ slot_idx) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.num_points) return;
    int s = pslot[i];
    if (s < 0) return;
    int vid = svid[s];
    if (vid < 0) return;
    if (p.cap_mode == 0) {
        int b = frame_of(offs, p.batch, i);
        if (i >= break_idx[b]) return;
    }
    atomicAdd(&count[vid], 1);
    int v = i;
    int *row = slot_idx + (size_t)vid * p.max_points;
    // Slot values only ever DECREASE, so a relaxed peek that already shows a smaller index than the one in hand is final for this
    // thread: the atomicMin there would change nothing and hand back something smaller (v stays), and a LAST slot that is already
    // smaller means max_points smaller indices exist -- this point cannot be among the first max_points of its voxel at all.
    // With 60-point pillars (nuscenes/all.pp.largea: a near pillar collects hundreds of points) the late points used to walk all 60
    // positions with returning device-scope atomics -- 1.8 ms of the 2.7 ms step; a stale (larger) peek only costs the atomic it
    // would have saved.
    if (__hip_atomic_load(&row[p.max_points - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) return;
    for (int t = 0; t < p.max_points; ++t) {
        if (__hip_atomic_load(&row[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) continue;
        int old = atomicMin(&row[t], v);
        if (old == kEmptyI32) break;
        v = old > v ? old : v;
    }
}

// k_vox_flag_scan + k_vox_assign + k_vox_cascade in ONE launch (batches <= kFusedFrames, max_points <= kCascadeMaxPoints: car.fhd).
// The three kernels were each a near-empty grid paying its own launch boundary on the step's latency chain (7.5 + 7.7 + 8.3 us for
// 136 k points).  What ties them together is ordering information that the ticketed tiles of the single-pass scan already carry:
//   * a tile knows the global rank of its points after its look-back; the rank at each cloud's first point (`base`) is published by
//     the tile that owns that point as one 64-bit word (flag << 32 | rank), and a tile waits only for the words of the clouds its own
//     points belong to -- all of them published by tiles with SMALLER tickets, i.e. tiles that are already running;
//   * a voxel is numbered by its first point (smallest index => an earlier or the same tile); the other points of the voxel spin on
//     svid[slot] (initialised to "empty") until that happened -- again only ever waiting for a smaller ticket;
//   * the `break` position needs no break_idx round trip: point i comes after the break iff more than max_voxels voxel-creating
//     points of its cloud have an index <= i, which its own inclusive rank says.
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void k_vox_scan_assign_cascade(const float *__restrict__ points, const int *__restrict__ offs,
                                                                   const int *__restrict__ pslot, const int *__restrict__ vals,
                                                                   int n, VoxParams p, unsigned long long *__restrict__ status,
                                                                   int *__restrict__ ticket, int *__restrict__ total,
                                                                   unsigned long long *__restrict__ frame_words, int *__restrict__ base,
                                                                   int *__restrict__ voxel_offsets, int *__restrict__ svid,
                                                                   int *__restrict__ break_idx, int *__restrict__ coors,
                                                                   int *__restrict__ count, int *__restrict__ slot_idx) {
    __shared__ int smem[5];
    __shared__ int s_tile, s_end;
    __shared__ int s_rank[kBlock * ITEMS];
    __shared__ int s_base[kFusedFrames + 1], s_voff[kFusedFrames + 1];
    const int tile = scan_take_tile(ticket, &s_tile), ntiles = (int)gridDim.x, tid = threadIdx.x;
    const int i0 = (tile * kBlock + tid) * ITEMS;
    int f[ITEMS], s4[ITEMS], r[ITEMS], v = 0;
    if (i0 + ITEMS <= n) {
        if constexpr (ITEMS == 4) {
            const int4 q = *reinterpret_cast<const int4 *>(pslot + i0);
            s4[0] = q.x; s4[1] = q.y; s4[2] = q.z; s4[3] = q.w;
        } else if constexpr (ITEMS == 2) {
            const int2 q = *reinterpret_cast<const int2 *>(pslot + i0);
            s4[0] = q.x; s4[1] = q.y;
        } else {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) s4[j] = pslot[i0 + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) s4[j] = ld_sel(pslot, i0 + j, i0 + j < n, -1);
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        f[j] = (ld_sel(vals, s4[j], s4[j] >= 0, -1) == i0 + j) ? 1 : 0;      // unconditional loads: all ITEMS in flight at once
        v += f[j];
    }
    int ex = scan_lookback(v, tile, ntiles, status, smem, total);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { r[j] = ex; s_rank[tid * ITEMS + j] = ex; ex += f[j]; }
    if (tid == kBlock - 1) s_end = ex;                 // rank behind this tile's last point
    __syncthreads();
    const int t0 = tile * kBlock * ITEMS;
    const int t1 = min(t0 + kBlock * ITEMS, n);
    if (tid <= p.batch) {                              // publish the rank at every cloud start that lies in this tile
        const int o = offs[tid];
        int val = -1;
        if (o >= t0 && o < t1) val = s_rank[o - t0];
        else if (o >= n && tile == ntiles - 1) val = s_end;
        if (val >= 0) __hip_atomic_store(&frame_words[tid], (1ull << 32) | (unsigned)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool last = tile == ntiles - 1;
    const int b_hi = last ? p.batch : frame_of(offs, p.batch, t1 - 1);      // clouds 0 .. b_hi are needed here (the last tile: all + the end)
    if (tid <= b_hi) {
        unsigned long long wv;
        while (((wv = __hip_atomic_load(&frame_words[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0)
            __builtin_amdgcn_s_sleep(1);
        s_base[tid] = (int)(unsigned)wv;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        s_voff[0] = 0;
        for (int b = 0; b < b_hi; ++b) {
            int cnt = s_base[b + 1] - s_base[b];
            if (cnt > p.max_voxels) cnt = p.max_voxels;
            acc += cnt;
            s_voff[b + 1] = acc;
        }
    }
    __syncthreads();
    if (last && tid <= p.batch) { base[tid] = s_base[tid]; voxel_offsets[tid] = s_voff[tid]; }
    // number the voxels whose first point is here
    int vid4[ITEMS], fb[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        vid4[j] = -1;
        fb[j] = 0;
        const int i = i0 + j, s = s4[j];
        if (s < 0) continue;
        const int b = frame_of(offs, p.batch, i);
        fb[j] = b;
        if (!f[j]) continue;
        const int rr = r[j] - s_base[b];
        if (rr < p.max_voxels) {
            const int vid = s_voff[b] + rr;
            vid4[j] = vid;
            __hip_atomic_store(&svid[s], vid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the cell of the voxel's first point, recomputed exactly as k_vox_hash computed it (a coalesced load of the point
            // instead of a random 8-byte read of the hash key and two 64-bit divisions)
            const float *pt = points + (size_t)i * p.num_features;
            int c[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) c[d] = (int)floorf(__fdiv_rn(__fsub_rn(pt[d], p.lo[d]), p.vs[d]));
            *reinterpret_cast<int4 *>(coors + (size_t)vid * 4) = make_int4(b, c[2], c[1], c[0]);
            slot_idx[(size_t)vid * p.max_points] = i;       // slot 0 IS the first point (the smallest index of the voxel): no atomics
        } else {
            __hip_atomic_store(&svid[s], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rr == p.max_voxels && p.cap_mode == 0) break_idx[b] = i;       // (the sort path of large max_points reads it)
        }
    }
    __syncthreads();     // first points of this tile have stored their svid before any lane of the tile starts to wait
    // slots: the max_points smallest point indices of every voxel (the atomicMin cascade of k_vox_cascade)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int i = i0 + j, s = s4[j];
        if (s < 0) continue;
        int vid = vid4[j];
        if (!f[j]) {
            while ((vid = __hip_atomic_load(&svid[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kEmptyI32) __builtin_amdgcn_s_sleep(1);
        }
        if (vid < 0) continue;
        if (p.cap_mode == 0 && r[j] + f[j] - s_base[fb[j]] > p.max_voxels) continue;     // at or behind the sequential loop's `break`
        atomicAdd(&count[vid], 1);
        if (f[j] || p.max_points < 2) continue;             // the first point sits in slot 0 already; the others cascade through 1 ..
        int pv = i;
        int *row = slot_idx + (size_t)vid * p.max_points;
        if (__hip_atomic_load(&row[p.max_points - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pv) continue;
        for (int t = 1; t < p.max_points; ++t) {
            if (__hip_atomic_load(&row[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pv) continue;
            const int old = atomicMin(&row[t], pv);
            if (old == kEmptyI32) break;
            pv = old > pv ? old : pv;
        }
    }
}

// ---- many points per voxel (pillars, max_points 60): slots by a stable sort instead of the cascade ------------------------------
// The cascade is O(points x occupied slots) returning atomics on the voxel's slot row, and the hundreds of points of a near pillar
// all walk the same 60 addresses at the same time: 2.0 ms of a 2.7 ms nuscenes/all.pp.largea step (profiles/r03_h_*).  Here every
// point gets the key `voxel row` (invalid / dropped points: one past the last row), the pairs (key, point index) are sorted with a
// STABLE radix sort -- points start in index order, so inside a voxel they stay in arrival order -- and a point's slot is its
// distance from the start of its voxel's run: the first max_points of every run are exactly the reference loop's slots.
constexpr int kCascadeMaxPoints = 8;       // up to here the cascade wins (car.fhd: 5 points per voxel, 7 us)

// (1) k_vox_group_rank: every kept point takes a place in its voxel's run: counted per workgroup in an LDS table, ONE returning
//     atomicAdd(count[row], points of the row in this workgroup) per (workgroup, voxel) hands the group a block of places.  The order
//     inside a run is whatever the atomics make it.  count[row] ends as the voxel's uncapped point count.
// (2) k_vox_count_scan: run_start = exclusive scan of the counts (single pass, decoupled look-back).
// (3) k_vox_run_scatter: run[run_start[row] + place] = point index.
// (4) k_vox_run_select: one wave per voxel puts the run's max_points SMALLEST indices into slot order.  Up to 64 * R points: ranked
//     by counting (n readlane rounds) and permuted through LDS; longer runs: the first 64 * R sorted the same way, every later element
//     below the current largest kept one is inserted (ballot -> position, shift by one lane) -- few qualify once the kept set has
//     settled.  (Measured alternatives, profiles/r05_n_pillar_voxeliser.txt: a dense cell grid with a returning atomic per point or per
//     (workgroup, cell) instead of the hash -- 215 / 303 us for that launch against 96 + 83; sixteen lanes per short run plus a list
//     of long ones -- the list's one counter cost 99 us; a thread per point ranking by counting -- 77 + 31 us.)
constexpr int kRankBlock = 1024, kRankTable = 2048;
__global__ __launch_bounds__(kRankBlock) void k_vox_group_rank(const int *__restrict__ offs, const int *__restrict__ pslot,
                                                              const int *__restrict__ svid, const int *__restrict__ break_idx, VoxParams p,
                                                              int *__restrict__ count, int *__restrict__ pvid, int *__restrict__ prank,
                                                              int *__restrict__ ctl, long long ctl_words) {
    // The points of ONE voxel inside a workgroup are counted in LDS first (a small hash table keyed by the voxel row), so a voxel costs
    // one returning global atomic per WORKGROUP that touches it: the near pillars of a nuScenes sweep hold thousands of points, and a
    // returning atomic per point (or per wave) on their counters serialised the launch -- 116 us for 293 k points, 8 us this way.
    // (Random point order -- the synthetic sweeps -- leaves ~1000 distinct voxels per 1024-point workgroup, i.e. about one returning atomic
    // per point again: 78 us for 1.17 M points, the fabric's rate for atomics.  Placing the early eighth of every cloud first and dropping
    // the later points of voxels it already fills was measured in round 6: only 2 % of the bench cloud's points live in such pillars
    // (max 758 points per pillar), two launches 21 + 68 us -- rejected, DESIGN_APPENDIX.)
    __shared__ int t_key[kRankTable], t_cnt[kRankTable], t_base[kRankTable];
    const int tid = threadIdx.x, i = blockIdx.x * kRankBlock + tid;
    for (int j = tid; j < kRankTable; j += kRankBlock) { t_key[j] = -1; t_cnt[j] = 0; }
    for (long long j = i; j < ctl_words; j += (long long)gridDim.x * kRankBlock) ctl[j] = 0;      // ticket + status words of the scan behind this launch
    int vid = -1;
    if (i < p.num_points) {
        const int s = pslot[i];
        if (s >= 0) {
            const int v = svid[s];
            bool ok = v >= 0;
            if (ok && p.cap_mode == 0) ok = i < break_idx[frame_of(offs, p.batch, i)];
            if (ok) vid = v;
        }
    }
    __syncthreads();
    int slot = 0, local = 0;
    if (vid >= 0) {
        unsigned h = ((unsigned)vid * 2654435761u) >> 21;     // 11 bits
        while (true) {
            const int prev = atomicCAS(&t_key[h], -1, vid);
            if (prev == -1 || prev == vid) break;
            h = (h + 1) & (kRankTable - 1);                   // at most 1024 distinct rows in 2048 places: the probe ends
        }
        slot = (int)h;
        local = atomicAdd(&t_cnt[h], 1);
    }
    __syncthreads();
    for (int j = tid; j < kRankTable; j += kRankBlock)
        if (t_key[j] >= 0) t_base[j] = atomicAdd(&count[t_key[j]], t_cnt[j]);
    __syncthreads();
    if (i < p.num_points) { pvid[i] = vid; prank[i] = vid >= 0 ? t_base[slot] + local : 0; }
}

constexpr int kCountItems = 4;
__global__ __launch_bounds__(kBlock) void k_vox_count_scan(const int *__restrict__ count, int rows, int *__restrict__ run_start,
                                                          unsigned long long *__restrict__ status, int *__restrict__ ticket) {
    __shared__ int smem[5];
    __shared__ int s_tile;
    const int tile = scan_take_tile(ticket, &s_tile);
    const int i0 = (tile * kBlock + threadIdx.x) * kCountItems;
    int c[kCountItems], v = 0;
#pragma unroll
    for (int j = 0; j < kCountItems; ++j) { c[j] = ld_sel(count, i0 + j, i0 + j < rows, 0); v += c[j]; }
    int ex = scan_lookback(v, tile, (int)gridDim.x, status, smem, (int *)nullptr);
#pragma unroll
    for (int j = 0; j < kCountItems; ++j) {
        if (i0 + j < rows) run_start[i0 + j] = ex;
        ex += c[j];
    }
}

__global__ __launch_bounds__(kBlock) void k_vox_run_scatter(const int *__restrict__ pvid, const int *__restrict__ prank, int n,
                                                           const int *__restrict__ run_start, int *__restrict__ run) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int v = pvid[i];
    if (v >= 0) run[run_start[v] + prank[i]] = i;
}

template <int R>     // max_points <= 64 * R; one WAVE per run: n entries at run[s0 ..], of which `nvalid` are point indices (the others holes)
__device__ __forceinline__ void vox_select_wave(int v, int n, int s0, int nvalid, int (*s_sorted)[64 * R], const int *__restrict__ run,
                                                int max_points, int *__restrict__ slot_idx) {
    constexpr int CAP = 64 * R, BIG = 0x7fffffff;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = n < CAP ? n : CAP;                          // the first chunk: ranked by counting
    int x[R], rk[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = ld_sel(run, s0 + r * 64 + lane, r * 64 + lane < m, BIG); rk[r] = 0; }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {                           // every element of the chunk, broadcast: smaller ones (or equal and earlier) count
        const int lim = m - rr * 64 < 64 ? m - rr * 64 : 64;   // elements of register rr
        for (int k = 0; k < lim; ++k) {
            const int o = __builtin_amdgcn_readlane(x[rr], k);      // k is wave-uniform: one v_readlane instead of a ds_bpermute round trip per element
            const int pos = rr * 64 + k;
#pragma unroll
            for (int r = 0; r < R; ++r) rk[r] += (o < x[r] || (o == x[r] && pos < r * 64 + lane)) ? 1 : 0;
        }
    }
    // padding lanes (BIG) rank behind every element and among themselves by position: a permutation of 0 .. CAP - 1
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int rank = rk[r];
        if (r * 64 + lane >= m) rank = r * 64 + lane;          // m real elements took ranks 0 .. m - 1; padding keeps its own place
        s_sorted[wv][rank] = x[r];
    }
    __builtin_amdgcn_wave_barrier();
    int S[R];
#pragma unroll
    for (int r = 0; r < R; ++r) S[r] = s_sorted[wv][r * 64 + lane];
    if (n > CAP) {
        int T = __builtin_amdgcn_readlane(S[R - 1], 63);       // the largest kept index
        for (int base = CAP; base < n; base += 64) {
            const int y = ld_sel(run, s0 + base + lane, base + lane < n, BIG);
            unsigned long long pend = __ballot(y < T);
            while (pend) {
                const int l = __ffsll((long long)pend) - 1;
                const int val = __builtin_amdgcn_readlane(y, l);
                pend &= pend - 1;
                if (val >= T) continue;                        // T fell since the ballot
                int pos = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) pos += (int)__popcll(__ballot(S[r] < val));
                int carry = val;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int pr = pos - r * 64;               // insertion lane in this register: < 0 everything shifts, >= 64 nothing does
                    if (pr < 64) {
                        const int last = __builtin_amdgcn_readlane(S[r], 63);
                        const int up = __shfl_up(S[r], 1, 64);
                        const int at = pr < 0 ? 0 : pr;
                        S[r] = lane > at ? up : (lane == at ? carry : S[r]);
                        carry = last;
                    }
                }
                T = __builtin_amdgcn_readlane(S[R - 1], 63);
            }
        }
    }
    int *row = slot_idx + (size_t)v * max_points;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = r * 64 + lane;
        if (t < max_points && t < nvalid) row[t] = S[r];
    }
}


template <int R>
__global__ __launch_bounds__(kBlock) void k_vox_run_select(const int *__restrict__ voxel_offsets, int batch, const int *__restrict__ count,
                                                          const int *__restrict__ run_start, const int *__restrict__ run, int max_points,
                                                          int *__restrict__ slot_idx, int *__restrict__ num_points_per_voxel) {
    __shared__ int s_sorted[kBlock / 64][64 * R];
    const int v = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (v >= voxel_offsets[batch]) return;                    // (whole waves leave: no barrier below)
    const int n = count[v];
    // callers without a voxel tensor (point lists only) get the capped counts here: one launch (k_vox_counts) less on the pillar path
    if (num_points_per_voxel && (threadIdx.x & 63) == 0) num_points_per_voxel[v] = n > max_points ? max_points : n;
    vox_select_wave<R>(v, n, run_start[v], n, s_sorted, run, max_points, slot_idx);
}

__global__ __launch_bounds__(kBlock) void k_vox_fill(const float *__restrict__ points,
                                                    const int *__restrict__ voxel_offsets,
                                                    const int *__restrict__ count,
                                                    const int *__restrict__ slot_idx, VoxParams p,
                                                    float *__restrict__ voxels,
                                                    int *__restrict__ num_points_per_voxel) {
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;  // (vid, t)
    int total = voxel_offsets[p.batch];
    int vid = (int)(g / p.max_points);
    int t = (int)(g % p.max_points);
    if (vid >= total) return;
    int n = count[vid];
    if (n > p.max_points) n = p.max_points;
    if (t == 0) num_points_per_voxel[vid] = n;
    float *dst = voxels + (size_t)g * p.num_features;
    if (t < n) {
        const float *src = points + (size_t)slot_idx[g] * p.num_features;
        for (int f = 0; f < p.num_features; ++f) dst[f] = src[f];
    } else {
        for (int f = 0; f < p.num_features; ++f) dst[f] = 0.0f;
    }
}

// k_vox_fill + k_vox_mean in one launch for the common shape (4 point features, a handful of points per voxel): one thread per
// voxel copies its points and accumulates the SimpleVoxel sums in the same slot order k_vox_mean uses (padded slots add +0.0f,
// exactly as there), so the results are bit-identical to the two-kernel path; one launch less on the latency chain.
template <typename OT>
__global__ __launch_bounds__(kBlock) void k_vox_fill_mean4(const float *__restrict__ points, const int *__restrict__ voxel_offsets,
                                                          const int *__restrict__ count, const int *__restrict__ slot_idx,
                                                          VoxParams p, int mean_features, float *__restrict__ voxels,
                                                          int *__restrict__ num_points_per_voxel, OT *__restrict__ mean) {
    const int vid = blockIdx.x * kBlock + threadIdx.x;
    if (vid >= voxel_offsets[p.batch]) return;
    int n = count[vid];
    if (n > p.max_points) n = p.max_points;
    num_points_per_voxel[vid] = n;
    const float4 *pts = reinterpret_cast<const float4 *>(points);
    float4 *dst = reinterpret_cast<float4 *>(voxels) + (size_t)vid * p.max_points;
    const int *slots = slot_idx + (size_t)vid * p.max_points;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 q[kCascadeMaxPoints];
    int sl[kCascadeMaxPoints];
#pragma unroll
    for (int t = 0; t < kCascadeMaxPoints; ++t) sl[t] = ld_sel(slots, t, t < n, 0);
#pragma unroll
    for (int t = 0; t < kCascadeMaxPoints; ++t)       // all gathers in flight together (max_points <= kCascadeMaxPoints here): ld_sel, not `?:`
        q[t] = ld_sel(pts, sl[t], t < n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
#pragma unroll
    for (int t = 0; t < kCascadeMaxPoints; ++t) {          // (no `break`: an early exit between the gathers made hipcc wait for each one in turn)
        if (t < p.max_points) dst[t] = q[t];
        s.x = __fadd_rn(s.x, q[t].x); s.y = __fadd_rn(s.y, q[t].y); s.z = __fadd_rn(s.z, q[t].z); s.w = __fadd_rn(s.w, q[t].w);
    }
    const float inv = (float)n;
    const float m[4] = {__fdiv_rn(s.x, inv), __fdiv_rn(s.y, inv), __fdiv_rn(s.z, inv), __fdiv_rn(s.w, inv)};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        if (f >= mean_features) break;
        if constexpr (std::is_same<OT, float>::value) mean[(size_t)vid * mean_features + f] = m[f];
        else if constexpr (std::is_same<OT, __half>::value) mean[(size_t)vid * mean_features + f] = __float2half_rn(m[f]);
        else mean[(size_t)vid * mean_features + f] = __float2bfloat16(m[f]);
    }
}

// SimpleVoxel.forward (second/pytorch/models/voxel_encoder.py:220-225): sum over slots / num_points; stored as fp32 or, for a
// 16-bit sparse stack, directly in its dtype (the cast torch ran as two extra kernels inside every captured step)
template <typename OT>
__global__ __launch_bounds__(kBlock) void k_vox_mean(const float *__restrict__ voxels,
                                                    const int *__restrict__ voxel_offsets,
                                                    const int *__restrict__ num_points_per_voxel,
                                                    VoxParams p, int mean_features,
                                                    OT *__restrict__ mean) {
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;  // (vid, f)
    int vid = (int)(g / mean_features);
    int f = (int)(g % mean_features);
    if (vid >= voxel_offsets[p.batch]) return;
    const float *src = voxels + (size_t)vid * p.max_points * p.num_features + f;
    float s = 0.0f;
    for (int t = 0; t < p.max_points; ++t) s = __fadd_rn(s, src[(size_t)t * p.num_features]);
    const float m = __fdiv_rn(s, (float)num_points_per_voxel[vid]);
    if constexpr (std::is_same<OT, float>::value) mean[g] = m;
    else if constexpr (std::is_same<OT, __half>::value) mean[g] = __float2half_rn(m);
    else mean[g] = __float2bfloat16(m);
}

struct VoxWorkspace {
    unsigned long long *keys;
    int *vals, *svid, *pslot, *rank, *ctl, *base, *break_idx, *total, *count, *slot_idx;
    // sort path (max_points > kCascadeMaxPoints): appended BEHIND everything else, so the offsets vox_table_of relies on never move
    int *sval_in, *sval_out, *run_start, *run, *ctl2;
    long long ctl2_words;
    int sort_bits;
    uint32_t table;
    unsigned long long *frame_words;
    size_t bytes;
};

static VoxWorkspace carve_vox(void *ws, size_t cap, int n, int batch, int max_voxels, int max_points) {
    VoxWorkspace w;
    Arena a(ws, cap);
    w.table = next_pow2((uint32_t)(n > 512 ? 2 * (size_t)n : 1024));
    w.keys = a.take<unsigned long long>(w.table);
    w.vals = a.take<int>(w.table);
    w.svid = a.take<int>(w.table);
    w.pslot = a.take<int>(n);
    w.rank = a.take<int>(n);
    w.ctl = a.take<int>(scan_ctl_words(n));   // ticket + tile status of the single-pass scan
    w.base = a.take<int>(batch + 1);
    w.break_idx = a.take<int>(batch);
    w.total = a.take<int>(1);
    w.count = a.take<int>((size_t)batch * max_voxels);
    w.slot_idx = a.take<int>((size_t)batch * max_voxels * max_points);
    w.sval_in = w.sval_out = w.run_start = w.run = w.ctl2 = nullptr;
    w.ctl2_words = 0;
    w.sort_bits = 0;
    if (max_points > kCascadeMaxPoints && n > 0) {
        // many points per voxel: group rank -> count scan -> run scatter -> per-voxel select (k_vox_group_rank ...)
        long long rows = (long long)batch * max_voxels;
        if (rows > n) rows = n;
        w.sort_bits = 1;                                      // "the run path is carved"
        w.sval_in = a.take<int>(n);                           // pvid
        w.sval_out = a.take<int>(n);                          // prank
        w.run = a.take<int>(n);
        w.run_start = a.take<int>((size_t)rows + 1);
        w.ctl2_words = (long long)scan_ctl_words((rows + kCountItems - 1) / kCountItems);
        w.ctl2 = a.take<int>((size_t)w.ctl2_words);
    }
    w.frame_words = a.take<unsigned long long>((size_t)batch + 1);     // fused scan (appended: earlier offsets never move)
    w.bytes = align_up(a.used);
    return w;
}

// the voxel hash table of a finished sec_voxelize_f32 call (cell key -> slot, svid[slot] = voxel row): the first SubM rulebook
// looks its sites up here instead of hashing them again (sec_rulebook_subm3d_after_voxelize)
bool vox_table_of(const void *ws, size_t bytes, int n, int batch, int max_voxels, int max_points, const unsigned long long **keys,
                  const int **svid, uint32_t *mask) {
    VoxWorkspace w = carve_vox(const_cast<void *>(ws), bytes, n, batch, max_voxels, max_points);
    if (!ws || w.bytes > bytes) return false;
    *keys = w.keys;
    *svid = w.svid;
    *mask = w.table - 1;
    return true;
}

// the per-voxel point lists of a finished sec_voxelize_f32 call: count[row] points (uncapped), slot_idx[row * max_points + t] = index
// of the t-th of them in the caller's point array (the first max_points, in arrival order) -- what k_vox_fill copies from.  A
// consumer that walks these lists itself (sec_pfn_fwd_slots) needs no [rows, max_points, F] tensor at all.
bool vox_slots_of(const void *ws, size_t bytes, int n, int batch, int max_voxels, int max_points, const int **count,
                  const int **slot_idx) {
    VoxWorkspace w = carve_vox(const_cast<void *>(ws), bytes, n, batch, max_voxels, max_points);
    if (!ws || w.bytes > bytes) return false;
    *count = w.count;
    *slot_idx = w.slot_idx;
    return true;
}

// sec_voxelize_f32 with voxels == NULL: only the per-voxel point counts are written (the point lists stay in the workspace)
__global__ __launch_bounds__(kBlock) void k_vox_counts(const int *__restrict__ voxel_offsets, const int *__restrict__ count, VoxParams p,
                                                      int *__restrict__ num_points_per_voxel) {
    const int vid = blockIdx.x * kBlock + threadIdx.x;
    if (vid >= voxel_offsets[p.batch]) return;
    const int n = count[vid];
    num_points_per_voxel[vid] = n > p.max_points ? p.max_points : n;
}

}  // namespace sec

using namespace sec;

static constexpr int kVoxMaxPoints = 256;      // largest max_points (points kept per voxel) sec_voxelize_f32 serves

SEC_API size_t sec_voxelize_workspace_bytes(int num_points, int batch, int max_voxels, int max_points) {
    if (num_points < 0 || batch <= 0 || max_voxels <= 0 || max_points <= 0) return 0;
    if (max_points > kVoxMaxPoints) return 0;          // no kernel for it (sec_voxelize_f32 returns SEC_E_UNSUPPORTED): no size to report
    return carve_vox(nullptr, 0, num_points, batch, max_voxels, max_points).bytes;
}

SEC_API int sec_voxelize_f32(const float *points, const int *point_offsets, int num_points,
                             int num_features, int batch, const float *h_range6,
                             const float *h_voxel_size3, int max_points, int max_voxels, int cap_mode,
                             float *voxels, int *coors, int *num_points_per_voxel, int *voxel_offsets,
                             void *mean, int mean_features, int mean_dtype, void *workspace, size_t workspace_bytes,
                             void *stream) {
    if (num_points < 0 || num_features < 3 || batch <= 0 || max_points <= 0 || max_voxels <= 0 ||
        !h_range6 || !h_voxel_size3 || (!voxels && mean) || !coors || !num_points_per_voxel || !voxel_offsets ||
        (mean && (mean_features <= 0 || mean_features > num_features || mean_dtype < SEC_F32 || mean_dtype > SEC_BF16)))
        return SEC_E_INVALID;
    // the per-voxel point selection keeps 64 x 4 candidates in registers (k_vox_run_select<4>); the reference's configs use <= 100
    // points per pillar.  Refused HERE, before anything is enqueued.
    if (max_points > kVoxMaxPoints) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    VoxWorkspace w = carve_vox(workspace, workspace_bytes, num_points, batch, max_voxels, max_points);
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;
    VoxParams p;
    for (int j = 0; j < 3; ++j) {
        p.lo[j] = h_range6[j];
        p.vs[j] = h_voxel_size3[j];
        // grid = round-half-even((max - min) / size) in fp32 (np.round), second/utils/simplevis.py:26-29
        p.grid[j] = (int)__builtin_rintf((h_range6[3 + j] - h_range6[j]) / h_voxel_size3[j]);
        if (p.grid[j] <= 0) return SEC_E_INVALID;
    }
    p.num_points = num_points; p.num_features = num_features; p.batch = batch;
    p.max_points = max_points; p.max_voxels = max_voxels; p.cap_mode = cap_mode;
    p.table_mask = w.table - 1;
    p.peek = (long long)num_points > 2ll * batch * max_voxels ? 1 : 0;

    int rc;
    // scan + numbering + slots in one launch (k_vox_scan_assign_cascade) for the shapes of the sparse-middle configs
    const bool fused_scan = num_points > 0 && batch <= kFusedFrames && max_points <= kCascadeMaxPoints;
    // Pillar path (sort path + voxels == NULL: the consumers read the point lists, nobody looks cells up in the hash table afterwards --
    // sec_rulebook_*_after_voxelize needs the voxel tensor's configs): dense slot numbering when batch * cells fits the carved table,
    // and the staged first-point pass (kStageDiv) from a quarter of a million points.
    const long long cells = (long long)p.grid[0] * p.grid[1] * p.grid[2];
    const bool dense = w.sort_bits && !voxels && num_points > 0 && (long long)batch * cells <= (long long)w.table && (long long)batch * cells < 0x7fffffffll;   // slots are ints
    const bool staged_first = dense && num_points >= 256 * 1024;
    p.dense_cells = dense ? (int)cells : 0;
    p.stage = 0;
    {   // one init launch instead of four memset nodes; only the rows that can be live (#voxels <= #points) are touched
        long long cap_rows = (long long)batch * max_voxels;
        if (cap_rows > num_points) cap_rows = num_points > 0 ? num_points : 1;
        const long long table = dense ? (long long)batch * cells : (long long)w.table;
        int blocks = div_up(table, kBlock);
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_vox_init, dim3(blocks), dim3(kBlock), 0, st, dense ? (unsigned long long *)nullptr : w.keys, w.vals, table, w.count, cap_rows,
                           w.slot_idx, w.sort_bits ? 0ll : cap_rows * max_points /* the run path (k_vox_run_select) writes every slot a consumer reads (t < count): 29 MB of fill less for config 4 */, w.ctl, (long long)scan_ctl_words(num_points), w.break_idx, batch,
                           fused_scan ? w.svid : (int *)nullptr, fused_scan ? w.frame_words : (unsigned long long *)nullptr);
    }
    const bool fused_frames = num_points > 0 && batch <= kFusedFrames;
    int nb = div_up(num_points > 0 ? num_points : 1, kBlock);
    if (fused_scan) {
        hipLaunchKernelGGL(k_vox_hash, dim3(nb), dim3(kBlock), 0, st, points, point_offsets, p, w.keys, w.vals, w.pslot);
        // two points per thread: ~270 tiles for car.fhd's 136 k points (four: 133 workgroups = one wave per SIMD on half the chip, every
        // dependent round of the tile paid in full; one: a look-back chain four times as long)
        hipLaunchKernelGGL(k_vox_scan_assign_cascade<kFusedItems>, dim3(div_up(num_points, kBlock * kFusedItems)), dim3(kBlock), 0, st, points, point_offsets,
                           w.pslot, w.vals, num_points, p, reinterpret_cast<unsigned long long *>(w.ctl + 4), w.ctl, w.total,
                           w.frame_words, w.base, voxel_offsets, w.svid, w.break_idx, coors, w.count, w.slot_idx);
    } else if (num_points > 0) {
        if (dense) {
            for (int stg = staged_first ? 1 : 0; stg <= (staged_first ? 2 : 0); ++stg) {
                p.stage = stg;
                hipLaunchKernelGGL(k_vox_cell_first, dim3(nb), dim3(kBlock), 0, st, points, point_offsets, p, w.vals, w.pslot);
            }
            p.stage = 0;
        } else {
            hipLaunchKernelGGL(k_vox_hash, dim3(nb), dim3(kBlock), 0, st, points, point_offsets, p, w.keys, w.vals, w.pslot);
        }
        if (num_points >= 512 * 1024)
            hipLaunchKernelGGL(k_vox_flag_scan<kFlagItemsBig>, dim3(div_up(num_points, kBlock * kFlagItemsBig)), dim3(kBlock), 0, st, w.pslot, w.vals, num_points,
                               w.rank, reinterpret_cast<unsigned long long *>(w.ctl + 4), w.ctl, w.total);
        else
            hipLaunchKernelGGL(k_vox_flag_scan<kFlagItems>, dim3(div_up(num_points, kBlock * kFlagItems)), dim3(kBlock), 0, st, w.pslot, w.vals, num_points,
                               w.rank, reinterpret_cast<unsigned long long *>(w.ctl + 4), w.ctl, w.total);
    } else if ((rc = fill_words(w.total, sizeof(int), 0u, st))) return rc;
    if (!fused_frames)
        hipLaunchKernelGGL(k_vox_frames, dim3(1), dim3(64), 0, st, point_offsets, w.rank, w.total, p, w.base,
                           w.break_idx, voxel_offsets);
    if (num_points > 0 && !fused_scan) {
        hipLaunchKernelGGL(k_vox_assign, dim3(nb), dim3(kBlock), 0, st, point_offsets, w.pslot, w.vals, w.keys,
                           w.rank, w.base, voxel_offsets, p, w.svid, w.break_idx, coors,
                           fused_frames ? w.total : (const int *)nullptr);
        if (w.sort_bits) {
            long long rows = (long long)batch * max_voxels;
            if (rows > num_points) rows = num_points;
            hipLaunchKernelGGL(k_vox_group_rank, dim3(div_up(num_points, kRankBlock)), dim3(kRankBlock), 0, st, point_offsets, w.pslot, w.svid, w.break_idx, p, w.count,
                               w.sval_in, w.sval_out, w.ctl2, w.ctl2_words);
            hipLaunchKernelGGL(k_vox_count_scan, dim3(div_up(rows, kBlock * kCountItems)), dim3(kBlock), 0, st, w.count, (int)rows, w.run_start,
                               reinterpret_cast<unsigned long long *>(w.ctl2 + 4), w.ctl2);
            hipLaunchKernelGGL(k_vox_run_scatter, dim3(nb), dim3(kBlock), 0, st, w.sval_in, w.sval_out, num_points, w.run_start, w.run);
            const dim3 gs(div_up(rows, kBlock / 64));
            int *counts_here = voxels ? (int *)nullptr : num_points_per_voxel;
            if (max_points <= 64)
                hipLaunchKernelGGL(k_vox_run_select<1>, gs, dim3(kBlock), 0, st, voxel_offsets, batch, w.count, w.run_start, w.run, max_points, w.slot_idx, counts_here);
            else if (max_points <= 128)
                hipLaunchKernelGGL(k_vox_run_select<2>, gs, dim3(kBlock), 0, st, voxel_offsets, batch, w.count, w.run_start, w.run, max_points, w.slot_idx, counts_here);
            else
                hipLaunchKernelGGL(k_vox_run_select<4>, gs, dim3(kBlock), 0, st, voxel_offsets, batch, w.count, w.run_start, w.run, max_points, w.slot_idx, counts_here);
        } else {
            hipLaunchKernelGGL(k_vox_cascade, dim3(nb), dim3(kBlock), 0, st, point_offsets, w.pslot, w.svid,
                               w.break_idx, p, w.count, w.slot_idx);
        }
    }
    long long cap = (long long)batch * max_voxels;
    long long bound = num_points < cap ? num_points : cap;  // #voxels <= #points
    if (bound > 0 && !voxels) {
        // the caller consumes the point lists in the workspace directly (sec_pfn_fwd_slots): no [rows, max_points, F] tensor
        if (!w.sort_bits)      // (the run path wrote them in k_vox_run_select)
            hipLaunchKernelGGL(k_vox_counts, dim3(div_up(bound, kBlock)), dim3(kBlock), 0, st, voxel_offsets, w.count, p, num_points_per_voxel);
    } else if (bound > 0 && mean && num_features == 4 && max_points <= kCascadeMaxPoints &&
        (reinterpret_cast<uintptr_t>(points) & 15) == 0 && (reinterpret_cast<uintptr_t>(voxels) & 15) == 0) {
        const dim3 gf(div_up(bound, kBlock));
        if (mean_dtype == SEC_F32)
            hipLaunchKernelGGL(k_vox_fill_mean4<float>, gf, dim3(kBlock), 0, st, points, voxel_offsets, w.count, w.slot_idx, p, mean_features, voxels, num_points_per_voxel, (float *)mean);
        else if (mean_dtype == SEC_F16)
            hipLaunchKernelGGL(k_vox_fill_mean4<__half>, gf, dim3(kBlock), 0, st, points, voxel_offsets, w.count, w.slot_idx, p, mean_features, voxels, num_points_per_voxel, (__half *)mean);
        else
            hipLaunchKernelGGL(k_vox_fill_mean4<__hip_bfloat16>, gf, dim3(kBlock), 0, st, points, voxel_offsets, w.count, w.slot_idx, p, mean_features, voxels, num_points_per_voxel,
                               (__hip_bfloat16 *)mean);
    } else if (bound > 0) {
        hipLaunchKernelGGL(k_vox_fill, dim3(div_up(bound * max_points, kBlock)), dim3(kBlock), 0, st, points,
                           voxel_offsets, w.count, w.slot_idx, p, voxels, num_points_per_voxel);
        if (mean) {
            const dim3 gm(div_up(bound * mean_features, kBlock));
            if (mean_dtype == SEC_F32)
                hipLaunchKernelGGL(k_vox_mean<float>, gm, dim3(kBlock), 0, st, voxels, voxel_offsets, num_points_per_voxel, p, mean_features, (float *)mean);
            else if (mean_dtype == SEC_F16)
                hipLaunchKernelGGL(k_vox_mean<__half>, gm, dim3(kBlock), 0, st, voxels, voxel_offsets, num_points_per_voxel, p, mean_features, (__half *)mean);
            else
                hipLaunchKernelGGL(k_vox_mean<__hip_bfloat16>, gm, dim3(kBlock), 0, st, voxels, voxel_offsets, num_points_per_voxel, p, mean_features,
                                   (__hip_bfloat16 *)mean);
        }
    }
    return check_launch();
}

namespace sec {
// SimpleVoxel.forward (second/pytorch/models/voxel_encoder.py:220-225) on a voxel tensor that ARRIVES from the host side of the
// boundary (the example dict of VoxelNet.forward: `voxels` [N, T, F], `num_points` [N]) -- the drop-in sessions' first step.  Same
// arithmetic as k_vox_mean above (slot order, __fadd_rn / __fdiv_rn): the features equal the fused voxeliser epilogue's, bit for
// bit.  Rows at or past the live count (`num_dev`) are written as zeros.
template <typename OT>
__global__ __launch_bounds__(kBlock) void k_simple_voxel(const float *__restrict__ voxels, const int *__restrict__ num_points, int n,
                                                        const int *__restrict__ num_dev, int T, int F, int nf, OT *__restrict__ mean) {
    const long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= (long long)n * nf) return;
    const int vid = (int)(g / nf), f = (int)(g - (long long)vid * nf);
    const int live = num_dev ? (*num_dev < n ? *num_dev : n) : n;
    float m = 0.0f;
    if (vid < live) {
        const float *src = voxels + (size_t)vid * T * F + f;
        float s = 0.0f;
        for (int t = 0; t < T; ++t) s = __fadd_rn(s, src[(size_t)t * F]);
        m = __fdiv_rn(s, (float)num_points[vid]);
    }
    if constexpr (std::is_same<OT, float>::value) mean[g] = m;
    else if constexpr (std::is_same<OT, __half>::value) mean[g] = __float2half_rn(m);
    else mean[g] = __float2bfloat16(m);
}

// flag[0] |= 1 when any of the `rows` rows of `a` [rows][n] differs from `b` [n] (bit compare: NaN equals the same NaN).  The drop-in
// sessions' check of the example's anchors against the anchor table a graph was captured with -- one launch, one pass over `a`.
__global__ __launch_bounds__(kBlock) void k_rows_differ(const unsigned *__restrict__ a, const unsigned *__restrict__ b, unsigned n,
                                                       int *__restrict__ flag) {
    // blockIdx.y = row; 16-byte loads where the row length allows it (no 64-bit modulo per element: the first form spent 25 us on it)
    const unsigned *row = a + (size_t)blockIdx.y * n;
    const unsigned stride = gridDim.x * kBlock;
    unsigned diff = 0u;
    if ((n & 3u) == 0u && ((((size_t)a) | ((size_t)b)) & 15) == 0) {
        const uint4 *r4 = reinterpret_cast<const uint4 *>(row), *b4 = reinterpret_cast<const uint4 *>(b);
        for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < (n >> 2); i += stride) {
            const uint4 x = r4[i], y = b4[i];
            diff |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
        }
    } else {
        for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) diff |= row[i] ^ b[i];
    }
    if (__ballot(diff != 0u) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
}  // namespace sec

SEC_API int sec_simple_voxel_f32(const float *voxels, const int *num_points, int n, const int *num_dev, int max_points, int num_features,
                                 int mean_features, void *mean, int mean_dtype, void *stream) {
    if (n < 0 || max_points <= 0 || num_features <= 0 || mean_features <= 0 || mean_features > num_features || !mean ||
        (n > 0 && (!voxels || !num_points)))
        return SEC_E_INVALID;
    if (mean_dtype < SEC_F32 || mean_dtype > SEC_BF16) return SEC_E_UNSUPPORTED;
    if (n == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(div_up((long long)n * mean_features, kBlock));
    if (mean_dtype == SEC_F32)
        hipLaunchKernelGGL(k_simple_voxel<float>, grid, dim3(kBlock), 0, st, voxels, num_points, n, num_dev, max_points, num_features, mean_features, (float *)mean);
    else if (mean_dtype == SEC_F16)
        hipLaunchKernelGGL(k_simple_voxel<__half>, grid, dim3(kBlock), 0, st, voxels, num_points, n, num_dev, max_points, num_features, mean_features, (__half *)mean);
    else
        hipLaunchKernelGGL(k_simple_voxel<__hip_bfloat16>, grid, dim3(kBlock), 0, st, voxels, num_points, n, num_dev, max_points, num_features, mean_features,
                           (__hip_bfloat16 *)mean);
    return check_launch();
}

SEC_API int sec_rows_differ_f32(const float *a, long long rows, const float *b, long long n, int *flag, void *stream) {
    if (rows < 0 || n <= 0 || n > 0x7fffffffll || rows > 65535 || !flag || (rows > 0 && (!a || !b))) return SEC_E_INVALID;
    if (rows == 0) return SEC_OK;
    // rows * n * 4 bytes must be 16-byte aligned per row for the vector path: n % 4 == 0 and a 16-byte aligned base (torch allocations are)
    long long bx = div_up(n / 4 > 0 ? n / 4 : n, (long long)kBlock * 2);
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_rows_differ, dim3((unsigned)bx, (unsigned)rows), dim3(kBlock), 0, (hipStream_t)stream, reinterpret_cast<const unsigned *>(a),
                       reinterpret_cast<const unsigned *>(b), (unsigned)n, flag);
    return check_launch();
}
