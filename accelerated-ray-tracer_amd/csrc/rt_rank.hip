// rt_rank.hip -- the ranking between the parts of a split frame (cost-aware schedule, rt_abi.hip), on the device.
//
// Part 1 of a split frame parks every pixel with the rays it cost so far (rt_pixel_state.cost) and adds them up per 8x8
// tile (tile_cost).  Before the next part starts, three small kernels turn that into the next launch's schedule:
//   1. rt_rank_tiles_kernel   (one workgroup)  tiles ordered dearest first; cost thresholds from the mean cost per pixel;
//   2. rt_collect_heavy_kernel (whole grid)    every pixel at or above the heavy threshold -> (cost, pixel) list;
//   3. rt_rank_heavy_kernel   (one workgroup)  the list ordered dearest first, cut into tiers 0 / 1 / 2, workgroups per tier.
// Everything stays in device memory (rt_rank_info), so a frame is one stream enqueue: rt_render(blocking = 0) returns
// while the first part is still running (include/rt_abi.h).  Round 1 did this on the host with four stream
// synchronisations per ranking.
//
// The two orderings are bucket sorts by one workgroup: 2048 linear cost buckets, counted and scattered through LDS
// atomics.  Within a bucket the order is whatever the atomics give -- this is scheduling only: which wave renders a
// pixel never changes the pixel (its samples are one XORWOW stream, picked up where the previous part left it).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace {

enum { RANK_THREADS = 1024, RANK_BUCKETS = 2048 };

// workgroup-wide max / min of a per-thread value through LDS (all threads call; result in every thread)
__device__ unsigned int wg_reduce_max(unsigned int v, unsigned int* scratch) {
    if (threadIdx.x == 0) *scratch = 0u;
    __syncthreads();
    atomicMax(scratch, v);
    __syncthreads();
    const unsigned int r = *scratch;
    __syncthreads();
    return r;
}
__device__ unsigned int wg_reduce_min(unsigned int v, unsigned int* scratch) {
    if (threadIdx.x == 0) *scratch = 0xFFFFFFFFu;
    __syncthreads();
    atomicMin(scratch, v);
    __syncthreads();
    const unsigned int r = *scratch;
    __syncthreads();
    return r;
}

// Descending bucket sort by one workgroup: out[k] = value(i) for the items i in [0, n), dearest first.
// hist: RANK_BUCKETS + 1 words of LDS; scratch: one word of LDS.
// On return hist[b] = number of items in buckets 0..b (the dearest b + 1 buckets) and *hi_out / *span_out describe the
// buckets: bucket b holds costs in (hi - (b + 1) * span / RANK_BUCKETS, hi - b * span / RANK_BUCKETS].
template <class Key, class Val>
__device__ void wg_bucket_sort_desc(unsigned int n, Key key, Val value, unsigned int* out, unsigned int* hist, unsigned int* scratch,
                                    unsigned int* hi_out = nullptr, unsigned long long* span_out = nullptr) {
    unsigned int lo = 0xFFFFFFFFu, hi = 0u;
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) { const unsigned int k = key(i); lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
    hi = wg_reduce_max(hi, scratch);
    lo = wg_reduce_min(lo, scratch);
    const unsigned long long span = (unsigned long long)(hi >= lo ? hi - lo : 0u) + 1ull;
    for (unsigned int b = threadIdx.x; b <= RANK_BUCKETS; b += blockDim.x) hist[b] = 0u;
    __syncthreads();
    // bucket 0 holds the dearest items
    auto bucket = [&](unsigned int k) -> unsigned int { return (unsigned int)(((unsigned long long)(hi - k) * RANK_BUCKETS) / span); };
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[bucket(key(i))], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {   // exclusive scan; 2048 additions by one thread are a few microseconds
        unsigned int run = 0u;
        for (unsigned int b = 0; b < RANK_BUCKETS; ++b) { const unsigned int c = hist[b]; hist[b] = run; run += c; }
        hist[RANK_BUCKETS] = run;
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) out[atomicAdd(&hist[bucket(key(i))], 1u)] = value(i);
    if (threadIdx.x == 0) { if (hi_out) *hi_out = hi; if (span_out) *span_out = span; }
    __syncthreads();
}

__global__ void __launch_bounds__(RANK_THREADS) rt_rank_tiles_kernel(rt_rank_params rp) {
    __shared__ unsigned int hist[RANK_BUCKETS + 1];
    __shared__ unsigned int scratch;
    if (threadIdx.x == 0) {
        // cost thresholds of the heavy list and its tiers, from the mean cost per pixel so far
        const double mean = (double)*rp.ray_counter / (double)rp.n_pixels;
        rt_rank_info inf;
        inf.heavy_items = 0u; inf.heavy_threshold = 0xFFFFFFFFu; inf.tier1_items = 0u; inf.tier2_items = 0u;
        inf.tier1_wgs = 0; inf.main_skip_wgs = 0; inf.sparse_wgs = 0; inf.sparse_stride = 1; inf.semi_wgs = 0; inf.semi_stride = 1;
        inf.threshold1 = (unsigned int)(mean * (double)rp.tier1_factor + 0.999);
        inf.threshold2 = (unsigned int)(mean * (double)rp.sparse_factor + 0.999);
        inf.collected = 0u;
        if (rp.sparse_stride > 0) inf.heavy_threshold = (unsigned int)(mean * (double)rp.heavy_factor + 0.999);
        *rp.info = inf;
    }
    const unsigned int* cost = rp.tile_cost;
    wg_bucket_sort_desc(rp.n_tiles, [cost](unsigned int i) { return cost[i]; }, [](unsigned int i) { return i; }, rp.tile_order, hist, &scratch);
}

// every pixel whose cost estimate reaches the heavy threshold is appended as (estimate << 32 | pixel).  The estimate is the
// pixel's own rays so far, or -- a pixel's cost over 32 samples is a noisy predictor of its cost over 500 when its paths
// go through glass -- smooth_percent of its dearest 4-neighbour's, whichever is larger.
__global__ void rt_collect_heavy_kernel(rt_rank_params rp) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.n_pixels) return;
    const unsigned int threshold = rp.info->heavy_threshold;
    const unsigned int own = rp.state[i].cost & 0x7FFFFFFFu;
    unsigned int c = own;
    if (rp.smooth_percent > 0 && rp.nx > 0) {
        const unsigned int x = i % (unsigned int)rp.nx;
        unsigned int m = 0u;
        if (x > 0u) m = rp.state[i - 1].cost & 0x7FFFFFFFu;
        if (x + 1u < (unsigned int)rp.nx && i + 1u < rp.n_pixels) { const unsigned int v = rp.state[i + 1].cost & 0x7FFFFFFFu; m = v > m ? v : m; }
        if (i >= (unsigned int)rp.nx) { const unsigned int v = rp.state[i - rp.nx].cost & 0x7FFFFFFFu; m = v > m ? v : m; }
        if (i + (unsigned int)rp.nx < rp.n_pixels) { const unsigned int v = rp.state[i + rp.nx].cost & 0x7FFFFFFFu; m = v > m ? v : m; }
        m &= 0x7FFFFFFFu;   // (a neighbour may already carry this ranking's list flag)
        const unsigned int sm = (unsigned int)(((unsigned long long)m * (unsigned int)rp.smooth_percent) / 100ull);
        c = sm > c ? sm : c;
    }
    // bit 31 of the parked cost says "listed": the render kernel's tile queue skips exactly these pixels
    rp.state[i].cost = own | (c >= threshold ? 0x80000000u : 0u);
    if (c >= threshold) {
        const unsigned int at = atomicAdd(&rp.info->collected, 1u);
        if (at < rp.heavy_cap) rp.heavy_list[at] = ((unsigned long long)c << 32) | i;
    }
}

__global__ void __launch_bounds__(RANK_THREADS) rt_rank_heavy_kernel(rt_rank_params rp) {
    __shared__ unsigned int hist[RANK_BUCKETS + 1];
    __shared__ unsigned int scratch;
    __shared__ unsigned int n_tier1, n_tier2, sort_hi;
    __shared__ unsigned long long sort_span;
    rt_rank_info inf = *rp.info;
    const unsigned int count = inf.collected;
    // no list: nothing collected, more than the list holds, or so many (half the pixels) that "heavy" has lost its meaning
    const bool usable = rp.sparse_stride > 0 && count > 0u && count <= rp.heavy_cap && (unsigned long long)count * 2ull < (unsigned long long)rp.n_pixels;
    if (!usable) {
        if (threadIdx.x == 0) { inf.heavy_items = 0u; inf.heavy_threshold = 0xFFFFFFFFu; *rp.info = inf; }
        return;
    }
    const unsigned long long* list = rp.heavy_list;
    wg_bucket_sort_desc(count, [list](unsigned int i) { return (unsigned int)(list[i] >> 32); },
                        [list](unsigned int i) { return (unsigned int)(list[i] & 0xFFFFFFFFull); }, rp.heavy_pixels, hist, &scratch, &sort_hi, &sort_span);
    if (threadIdx.x == 0) { n_tier1 = 0u; n_tier2 = 0u; }
    __syncthreads();
    unsigned int c1 = 0u, c2 = 0u;
    for (unsigned int i = threadIdx.x; i < count; i += blockDim.x) {
        const unsigned int c = (unsigned int)(list[i] >> 32);
        c1 += c >= inf.threshold1 ? 1u : 0u;
        c2 += c >= inf.threshold2 ? 1u : 0u;
    }
    if (c1) atomicAdd(&n_tier1, c1);
    if (c2) atomicAdd(&n_tier2, c2);
    __syncthreads();
    if (threadIdx.x != 0) return;
    // The tiers trade throughput for latency (a sparse wave has an eighth of its lanes live), which only pays while they
    // hold a small part of the frame's WORK: the dearest pixels are admitted while their rays so far stay within
    // sparse_work_percent of all rays (bucket by bucket, from the sort's histogram); everything after them is tier 3.  In
    // the Book-2 final scene 6 % of the pixels cost four times the mean and a quarter of the rays; in the random scene
    // 0.1 % and 0.5 %.
    {
        const double budget = (double)*rp.ray_counter * (double)rp.sparse_work_percent / 100.0;
        double work = 0.0;
        unsigned int admitted = 0u, prev = 0u;
        for (unsigned int b = 0; b < RANK_BUCKETS; ++b) {
            const unsigned int upto = hist[b], nb = upto - prev;
            prev = upto;
            if (nb == 0u) continue;
            const double cost_b = (double)sort_hi - ((double)b + 0.5) * (double)sort_span / (double)RANK_BUCKETS;
            work += (double)nb * (cost_b > 0.0 ? cost_b : 0.0);
            if (work > budget) break;
            admitted = upto;
        }
        if (n_tier1 > admitted) n_tier1 = admitted;
        if (n_tier2 > admitted) n_tier2 = admitted;
    }
    // ---- workgroups per tier.  Tier 1 = the tier kernel's (one pixel per wave at a time, workgroups of four waves), tier 2 =
    // 64 / sparse_stride live lanes per wave of the main kernel; each queue is served dearest first and whatever exceeds its
    // workgroups waits.
    const unsigned int cap_wgs = rp.max_grid * (unsigned int)rp.sparse_percent / 100u;
    unsigned int tier1_items = rp.tier_possible ? n_tier1 : 0u;
    if (tier1_items > (unsigned int)rp.tier1_pixels) tier1_items = (unsigned int)rp.tier1_pixels;
    // a tier-1 wave takes its pixels one after the other from the tier's queue (dearest first): tier1_depth of them on
    // average, fewer workgroups than that only if the tier kernel's grid is used up
    const unsigned int depth1 = rp.tier1_depth > 0 ? (unsigned int)rp.tier1_depth : 1u;
    const unsigned int tier_waves_per_wg = RT_TIER_THREADS / 64u;
    unsigned int tier1_wgs = (tier1_items + tier_waves_per_wg * depth1 - 1u) / (tier_waves_per_wg * depth1);
    if (tier1_wgs > (unsigned int)rp.tier_wgs_cap) tier1_wgs = (unsigned int)rp.tier_wgs_cap;
    if (tier1_wgs == 0u) tier1_items = 0u;
    // tier 2 = what is left of the pixels at or above the sparse threshold; tier 3 = the rest of the list
    unsigned int tier2_items = n_tier2 > tier1_items ? n_tier2 - tier1_items : 0u;
    if (tier1_items + tier2_items > count) tier2_items = count - tier1_items;
    const unsigned int per_wg2 = rp.waves_per_wg * (64u / (unsigned int)rp.sparse_stride);
    unsigned int tier2_wgs = (tier2_items + per_wg2 - 1u) / per_wg2;
    if (tier2_wgs > cap_wgs) tier2_wgs = cap_wgs;
    if (tier2_items > tier2_wgs * per_wg2) tier2_items = tier2_wgs * per_wg2;   // what the sparse workgroups cannot hold at once joins tier 3
    const unsigned int sparse_wgs = tier2_wgs;
    // tier 3 on workgroups of their own with every semi_stride-th lane live (a lane's rays advance faster the fewer lanes
    // its wave has), as many as hold the whole tier at once
    unsigned int semi_wgs = 0u;
    if (rp.semi_stride > 0) {
        const unsigned int tier3_items = count - tier1_items - tier2_items;
        const unsigned int per_wg3 = rp.waves_per_wg * (64u / (unsigned int)rp.semi_stride);
        semi_wgs = (tier3_items + per_wg3 - 1u) / per_wg3;
        const unsigned int room = cap_wgs > sparse_wgs ? cap_wgs - sparse_wgs : 0u;
        if (semi_wgs > room) semi_wgs = room;
    }
    unsigned int total = rp.normal_need + sparse_wgs + semi_wgs;
    if (total > rp.max_grid) total = rp.max_grid;
    // Where a tier workgroup does not fit beside a full main grid (tier_waves_per_main_wg > 0: that many tier waves fit into
    // the slot of one main workgroup), main workgroups make room: as many as the tier kernel's waves need beyond the slots
    // the main grid leaves empty anyway.
    unsigned int skip = 0u;
    if (rp.tier_waves_per_main_wg > 0 && tier1_wgs > 0u) {
        const unsigned int slots = (tier1_wgs * tier_waves_per_wg + (unsigned int)rp.tier_waves_per_main_wg - 1u) / (unsigned int)rp.tier_waves_per_main_wg;
        const unsigned int spare = rp.max_grid > total ? rp.max_grid - total : 0u;
        skip = slots > spare ? slots - spare : 0u;
        if (skip + 1u > total) skip = total > 1u ? total - 1u : 0u;   // at least one main workgroup keeps working
    }
    if (total > sparse_wgs + semi_wgs) {
        inf.semi_wgs = (int32_t)semi_wgs; inf.semi_stride = rp.semi_stride > 0 ? rp.semi_stride : 1;
        inf.heavy_items = count; inf.tier1_items = tier1_items; inf.tier2_items = tier2_items;
        inf.tier1_wgs = (int32_t)tier1_wgs; inf.main_skip_wgs = (int32_t)skip; inf.sparse_wgs = (int32_t)sparse_wgs;
        inf.sparse_stride = rp.sparse_stride;
    } else {
        inf.heavy_items = 0u; inf.heavy_threshold = 0xFFFFFFFFu;
    }
    *rp.info = inf;
}

// The cost prior of a ranked FIRST part: before a single sample of the frame is rendered, every pixel gets the rays its place
// in the scene's calibration frame cost (rt_scene_create traces a small frame through the scene's own camera and keeps
// the per-pixel ray counts), the maximum over the 3 x 3 calibration pixels around it -- dear regions (glass, the inside
// of a medium) are contiguous, their edges are not where a 256-pixel-wide frame puts them.  The ranking then lists and
// tiers the pixels exactly as it does from measured costs, so that the dearest chains start on tier waves at sample 0
// instead of running the first part at an ordinary lane's pace.  Scheduling only.
__global__ void rt_prior_kernel(rt_prior_params pp) {
    const unsigned int p = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n = (unsigned int)pp.local_rows * (unsigned int)pp.nx;
    unsigned int est = 0u;
    if (p < n) {
        const int lrow = (int)(p / (unsigned int)pp.nx), i = (int)(p - (unsigned int)lrow * (unsigned int)pp.nx);
        const int t = lrow / pp.tile_rows;
        const int j = (pp.tile_first + t * pp.tile_stride) * pp.tile_rows + (lrow - t * pp.tile_rows);
        const int ci = min(pp.cal_nx - 1, (int)(((long long)i * pp.cal_nx) / pp.nx)), cj = min(pp.cal_ny - 1, (int)(((long long)j * pp.cal_ny) / pp.ny));
        for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) {
                const int x = ci + di, y = cj + dj;
                if (x < 0 || y < 0 || x >= pp.cal_nx || y >= pp.cal_ny) continue;
                const unsigned int c = pp.cal_cost[(size_t)y * pp.cal_nx + x];
                est = c > est ? c : est;
            }
        pp.state[p].cost = est;
        atomicAdd(&pp.tile_cost[(lrow >> 3) * pp.tiles_x + (i >> 3)], est);
    }
    unsigned long long sum = est;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(pp.total, sum);
}

}  // namespace

hipError_t rt_launch_prior(const rt_prior_params& pp, hipStream_t st) {
    const unsigned int n = (unsigned int)pp.local_rows * (unsigned int)pp.nx;
    hipLaunchKernelGGL(rt_prior_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, pp);
    return hipGetLastError();
}

hipError_t rt_launch_rank(const rt_rank_params& rp, hipStream_t st) {
    hipLaunchKernelGGL(rt_rank_tiles_kernel, dim3(1), dim3(RANK_THREADS), 0, st, rp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rt_collect_heavy_kernel, dim3((rp.n_pixels + 255u) / 256u), dim3(256), 0, st, rp);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(rt_rank_heavy_kernel, dim3(1), dim3(RANK_THREADS), 0, st, rp);
    return hipGetLastError();
}
