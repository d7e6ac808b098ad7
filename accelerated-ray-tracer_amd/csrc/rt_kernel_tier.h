// rt_kernel_tier.h -- the tier kernel: ONE pixel per wave; instantiated by rt_tier_*.hip.
//
// A pixel's samples are one sequential chain (one XORWOW stream, main.cu:116-126), and the dearest pixels of a frame
// trace ten times the mean number of rays: their chains, not throughput, bound the frame and every multi-GPU share of
// it.  The ranking (rt_rank.hip) hands those pixels -- "tier 1" of its list -- to this kernel, which runs beside the
// main render kernel on a stream of its own: every wave takes one listed pixel at a time and traces each of its rays
// with all 64 lanes (trace_wave below), which makes a ray several times faster than on a lane of the main kernel's
// state machine.  Every lane of a wave carries the same pixel and computes the same values; only the traversal is
// shared out.  The reference's loop nest (main.cu:107-133, color() main.cu:44-87) is otherwise kept as it stands.
// A second kind of launch (fp.tail_mode) follows every main-kernel launch: it serves the queue of pixels the main kernel
// handed off near its end (rt_device.h, "tail hand-off"), each from the sample and the state its lane left it with.
//
// Why its own kernel (round 3): inside the main kernel the one-pixel loops cost every ordinary wave their register
// budget (128 VGPRs + 112 B of scratch instead of 94 + 0, profiles/r03_kernel_resources.md), and a general-scene
// version (quads, boxes, instances, media; up to 4096 leaves) would not have fitted at all.  Workgroups are four waves,
// one per SIMD.  The lean family's main kernel leaves 96 registers per SIMD free (4 x 104 of 512), so a tier workgroup (76)
// is co-resident with a full main grid; for the other families the first main_skip_wgs main workgroups leave at once
// and the tier workgroups take their slots (tools/ubench/concurrent_kernels.hip measures both cases).
#pragma once
#include "rt_device_funcs.h"

// the scene's leaves as the tier kernel reads them (rt_scene_dev: leaf_lo / leaf_hi / slot_ranges), in LDS
struct TierView {
    const float4* lo;        // [n_slots * 64]: (bmin, prim as int bits; -1 = padding)
    const float4* hi;        // [n_slots * 64]: (bmax, 0)
    const float* ranges;     // [n_slots][8]: union box of the slot's 64 leaves
    int n_slots;
    int n_media;             // constant_medium leaves (0, 1 or 2) ...
    int med_ord0, med_ord1;  // ... and their ordinals in the leaf order (INT_MAX when absent)
};

// minimum over the wave of `v` among the lanes where `have` (bit patterns of non-negative floats order like the floats);
// FLT_MAX when no lane has one.  Spheres-only scenes: a ray's candidates are few (its hits), a scalar loop over them is short.
// General scenes: they can be many (every face the ray crosses); a butterfly of six exchanges instead.
template <bool FEW>
DEV float wave_min_of(float v, bool have) {
    if (!FEW) {
        float r = have ? v : FLT_MAX;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) r = fminf(r, __shfl_xor(r, off, 64));
        return r;
    }
    unsigned long long m = __ballot(have);
    float r = FLT_MAX;
    while (m != 0ull) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        r = fminf(r, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)));
    }
    return r;
}

// One ray, one wave.  Leaf q of the depth-first leaf order belongs to lane q % 64 in "slot" q / 64; a slot is 64
// consecutive leaves, i.e. a compact patch of the scene, and lane k first tests slot k's union box so that slots the ray
// cannot meet are skipped (exact: a box inside a box that fails cannot pass).  In a surviving slot every
// lane tests its leaf's own box WITHOUT a limit, then -- where that passes -- the object without a limit.  Why the
// answer is bvh_node::hit's (bvh.cuh:95-106) for a ray with finite 1/d:
//   * bvh_node::hit keeps a running closest hit; it reaches an object iff every box above it and the object's own box
//     pass against that limit.  Every leaf it tests therefore passes its own box with no limit: the set tested here is a
//     superset.  A candidate hit c that the reference never reached was cut off by a limit b <= T_c (the entry distance of
//     its own box or of a box around it) and has t_c > T_c (checked: "anomaly"), so t_c > b: it is not the closest, and it
//     never lowers a limit below what the reference had.
//   * sphere::hit accepts t < limit (sphere.cuh:66), quad::hit t <= limit (quad.cuh:67; compound6 scans its faces the same
//     way, quad.cuh:124-139), so of candidates with EQUAL t the reference ends with the last quad-type one in visiting
//     order if there is one, else with the first.  Visiting order = leaf ordinal.  The reduction below applies that rule.
//   * constant_medium::hit (constant_medium.cuh:36-64) clips its interval to the limit, so its result depends on the
//     limit's VALUE at the moment of its visit: the minimum t over the candidates of smaller ordinal (its own box tested
//     against that limit, the medium evaluated with exactly it).  Media are taken in ordinal order after the solids; each
//     lane keeps the minimum of its candidates below each medium's ordinal for that purpose.  A medium hit replaces the
//     record whenever it hits (it is compared like a quad).
// A candidate at or before its own box's entry distance (rounding on a grazing ray), or a zero direction component, hands
// the ray to the reference's walk (trace(), over the walk array in memory).  All arguments are wave-uniform and the result is
// the same in every lane.
template <bool SPHERES_ONLY>
DEV bool trace_wave(const TierView& tv, const SceneView& sc, const Ray& r, HitInfo& best) {
    const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const float tmin = 0.001f;
    const float a = dot(r.d, r.d);
    const int lane = (int)(threadIdx.x & 63u);
    // a zero direction component (wave-uniform), or -- found below -- a grazing hit at or before its own box's entry: the
    // reference's walk decides (one call site at the end: the walk and everything it inlines exist once in the ray loop)
    const bool finite = inv_is_finite(inv);
    bool meets = false;
    if (finite && lane < tv.n_slots) {
        const float* q = tv.ranges + lane * 8;
        float t_enter, t_exit;
        slab_interval(make_float4(q[0], q[1], q[2], 0.f), make_float4(q[3], q[4], q[5], 0.f), r.o, inv, tmin, t_enter, t_exit);
        meets = !(t_exit <= t_enter);
    }
    unsigned long long slot_mask = __ballot(meets);
    // this lane's running candidate, merged in ordinal order by the reference's acceptance rule
    float bt = FLT_MAX;
    int bord = -1;
    int32_t bleaf = -1, binst = -1;
    bool bincl = false, anomaly = false;
    float pre0 = FLT_MAX, pre1 = FLT_MAX;          // min t of this lane's candidates below medium 0's / medium 1's ordinal
    if (SPHERES_ONLY) {
        // Spheres only: the record is the minimum of (t, ordinal) whatever the order of the tests, so the box tests of all
        // surviving slots run first -- two slots per LDS round trip -- and every lane notes the (at most two; more: tested on
        // the spot) leaves whose box passes; the sphere tests then run once or twice for the whole wave instead of once per
        // slot that has a passing lane.
        int np = 0;
        int32_t p0 = -1, p1 = -1;
        int o0 = 0, o1 = 0;
        float e0 = 0.f, e1 = 0.f;
        auto sphere_candidate = [&](int32_t prim, int ord, float t_enter) {
            float t;
            if (sphere_test_a(sc.spheres[RT_PRIM_INDEX(prim)], r, a, tmin, FLT_MAX, t)) {
                if (!(t > t_enter)) anomaly = true;
                if (t < bt || (t == bt && ord < bord)) { bt = t; bord = ord; bleaf = prim; }
            }
        };
        auto note = [&](bool pass, int32_t prim, int ord, float t_enter) {
            const bool overflow = pass && np >= 2;
            if (pass && np == 0) { p0 = prim; o0 = ord; e0 = t_enter; }
            if (pass && np == 1) { p1 = prim; o1 = ord; e1 = t_enter; }
            if (pass) ++np;
            if (__ballot(overflow) != 0ull) { if (overflow) sphere_candidate(prim, ord, t_enter); }
        };
        while (slot_mask != 0ull) {
            const int k0 = __ffsll((long long)slot_mask) - 1;
            slot_mask &= slot_mask - 1ull;
            const bool two = slot_mask != 0ull;                       // wave-uniform
            const int k1 = two ? __ffsll((long long)slot_mask) - 1 : k0;
            if (two) slot_mask &= slot_mask - 1ull;
            const int ord0 = k0 * 64 + lane, ord1 = k1 * 64 + lane;
            const float4 lo0 = tv.lo[ord0], hi0 = tv.hi[ord0], lo1 = tv.lo[ord1], hi1 = tv.hi[ord1];
            float t_enter, t_exit;
            slab_interval(lo0, hi0, r.o, inv, tmin, t_enter, t_exit);
            note(__float_as_int(lo0.w) >= 0 && !(t_exit <= t_enter), __float_as_int(lo0.w), ord0, t_enter);
            if (two) {
                slab_interval(lo1, hi1, r.o, inv, tmin, t_enter, t_exit);
                note(__float_as_int(lo1.w) >= 0 && !(t_exit <= t_enter), __float_as_int(lo1.w), ord1, t_enter);
            }
        }
        if (__ballot(np >= 1) != 0ull) { if (np >= 1) sphere_candidate(p0, o0, e0); }
        if (__ballot(np >= 2) != 0ull) { if (np >= 2) sphere_candidate(p1, o1, e1); }
    } else {
    // General scenes.  A lane's leaf is a sphere or a quad (lo.w), possibly under an instance (hi.w), or a box = compound6
    // (quad.cuh:124-139: a closest-hit scan over six faces with quad::hit's t <= limit, i.e. six quads that happen to share
    // one bounding box; lo.w is the first face, bit 30 of hi.w says so).  Boxes are not scanned face after face by the lane
    // that holds them -- a ray over Book-2 final's ground passes dozens of them, six quad tests each on a few lanes while
    // the others idle -- but shared out: face f of the T-th passing box of the slot is task 6 T + f, task j runs on lane
    // j % 64.  A candidate's ordinal is 8 * (leaf ordinal) + face; a lane meets its candidates in no particular order, so it
    // merges them by the rule that the reduction at the end uses (equal t: the last quad-type one if there is one, else the first).
    auto candidate = [&](int32_t prim, int32_t inst1, int ord8, float t_enter) {
        Ray q = r;
        if (inst1 != 0) q = to_object_space(sc.instances[inst1 - 1], r);
        float t;
        const bool is_quad = RT_PRIM_KIND(prim) == RT_PRIM_QUAD;
        const bool hit = is_quad ? quad_test(sc.quads[RT_PRIM_INDEX(prim)], q, tmin, FLT_MAX, t) : sphere_test(sc.spheres[RT_PRIM_INDEX(prim)], q, tmin, FLT_MAX, t);
        if (hit) {
            if (!(t > t_enter)) anomaly = true;
            const bool better = t < bt || (t == bt && (is_quad ? (!bincl || ord8 > bord) : (!bincl && ord8 < bord)));
            if (better) { bt = t; bord = ord8; bleaf = prim; binst = inst1 - 1; bincl = is_quad; }
            if ((ord8 >> 3) < tv.med_ord0) pre0 = fminf(pre0, t);
            if ((ord8 >> 3) < tv.med_ord1) pre1 = fminf(pre1, t);
        }
    };
    // (Noting the passing leaves of all slots first and running the object tests once per note level instead of once per slot
    // -- what the spheres-only path does -- measured the same on Book-2 final and 11 % slower on the Cornell box: not kept.)
    while (slot_mask != 0ull) {
        const int k = __ffsll((long long)slot_mask) - 1;
        slot_mask &= slot_mask - 1ull;
        const int ord = k * 64 + lane;
        const float4 lo4 = tv.lo[ord], hi4 = tv.hi[ord];
        const int32_t prim = __float_as_int(lo4.w), xw = __float_as_int(hi4.w);
        float t_enter, t_exit;
        slab_interval(lo4, hi4, r.o, inv, tmin, t_enter, t_exit);
        const bool pass = prim >= 0 && !(t_exit <= t_enter) && RT_PRIM_KIND(prim) != RT_PRIM_MEDIUM;
        const bool six = (xw & (1 << 30)) != 0;
        const unsigned long long m1 = __ballot(pass && !six), m6 = __ballot(pass && six);
        if (m1 != 0ull) { if (pass && !six) candidate(prim, xw & 0x3FFFFFFF, ord * 8, t_enter); }
        if (m6 != 0ull) {
            const int need = 6 * __popcll(m6);
            for (int base = 0; base < need; base += 64) {      // (wave-uniform trip count: one round up to ten boxes)
                const int j = base + lane;
                const int T = j / 6, f = j - 6 * T;
                // the lane holding the T-th passing box: the smallest l with more than T set bits of m6 at or below it
                int lo_l = 0, hi_l = 63;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int mid = (lo_l + hi_l) >> 1;
                    const bool enough = __popcll(m6 & ((2ull << mid) - 1ull)) > T;
                    hi_l = enough ? mid : hi_l;
                    lo_l = enough ? lo_l : mid + 1;
                }
                const int src = lo_l & 63;
                const int32_t p0 = __shfl(prim, src, 64), x0 = __shfl(xw, src, 64);
                const float te = __shfl(t_enter, src, 64);
                if (j < need) candidate(p0 + f, x0 & 0x3FFFFFFF, (k * 64 + src) * 8 + f, te);
            }
        }
    }
    }
    if (!finite || __ballot(anomaly) != 0ull) return trace<SPHERES_ONLY>(sc, r, best);
    // ---- media, in ordinal order, each against the limit the reference has when it gets there (wave-uniform values)
    float tm[2] = {FLT_MAX, FLT_MAX};
    bool hm[2] = {false, false};
    if (!SPHERES_ONLY) {
        for (int m = 0; m < tv.n_media; ++m) {
            const int ord = m == 0 ? tv.med_ord0 : tv.med_ord1;
            float limit = wave_min_of<false>(m == 0 ? pre0 : pre1, (m == 0 ? pre0 : pre1) < FLT_MAX);
            if (m == 1 && hm[0]) limit = fminf(limit, tm[0]);
            const float4 lo4 = tv.lo[ord], hi4 = tv.hi[ord];               // same address in every lane
            const int32_t prim = __builtin_amdgcn_readfirstlane(__float_as_int(lo4.w));
            float t_enter, t_exit;
            slab_interval(lo4, hi4, r.o, inv, tmin, t_enter, t_exit);
            if (!(fminf(t_exit, limit) <= t_enter)) {                       // aabb::hit(tmin, limit) of the medium's own box
                const rt_medium med = uniform_load(sc.media + RT_PRIM_INDEX(prim));
                float t;
                if (medium_test<true>(sc, med, r, tmin, limit, t)) { tm[m] = t; hm[m] = true; }
            }
        }
    }
    // ---- the closest candidate; among equal t the last quad-type one in leaf order if there is one, else the first
    const float ts = wave_min_of<SPHERES_ONLY>(bt, bord >= 0);
    const float tmin_all = fminf(ts, fminf(tm[0], tm[1]));
    best.inst = -1;
    if (!(tmin_all < FLT_MAX)) { best.t = FLT_MAX; best.prim = -1; return false; }
    unsigned long long tie = __ballot(bord >= 0 && bt == tmin_all);
    int w_ord = -1, w_lane = -1;
    bool w_incl = false;
    while (tie != 0ull) {
        const int l = __ffsll((long long)tie) - 1;
        tie &= tie - 1ull;
        const int o = __builtin_amdgcn_readlane(bord, l);
        const bool ic = __builtin_amdgcn_readlane(bincl ? 1 : 0, l) != 0;
        // first by ordinal unless a quad-type candidate exists, then the last quad-type one
        const bool take = w_ord < 0 || (ic ? (!w_incl || o > w_ord) : (!w_incl && o < w_ord));
        if (take) { w_ord = o; w_lane = l; w_incl = ic; }
    }
    int w_medium = -1;
    if (!SPHERES_ONLY) {
        for (int m = 0; m < tv.n_media; ++m) {
            if (!hm[m] || tm[m] != tmin_all) continue;
            const int o = (m == 0 ? tv.med_ord0 : tv.med_ord1) * 8;      // (candidates' ordinals are 8 * leaf + face)
            if (w_ord < 0 || !w_incl || o > w_ord) { w_ord = o; w_incl = true; w_medium = m; w_lane = -1; }
        }
    }
    best.t = tmin_all;
    if (w_medium >= 0) {
        const float4 lo4 = tv.lo[SPHERES_ONLY ? w_ord : (w_ord >> 3)];       // (general scenes: ordinals are 8 * leaf + face)
        best.prim = __builtin_amdgcn_readfirstlane(__float_as_int(lo4.w));
        return true;
    }
    best.prim = __builtin_amdgcn_readlane(bleaf, w_lane);
    best.inst = SPHERES_ONLY ? -1 : __builtin_amdgcn_readlane(binst, w_lane);
    return true;
}

// LDS_SCENE: the workgroup's LDS image also holds the spheres, materials and textures (a template parameter, not a run-time
// flag, so that those reads are LDS instructions and not flat loads through a pointer of unknown address space)
// Register budgets: 128 for the lean family (co-resident with the main kernel), 168 = the slot of a 256-thread main workgroup
// for the others.  The general variants spill 208-224 B per lane at 168; a 256-register variant without spills (two instead of
// three tier workgroups in the slot a 768-thread main workgroup vacates) measured WORSE -- Book-2 final's 1/8 share 236 -> 316 ms,
// profiles/r03_share_sweep_final_eighth.log: the number of tier waves matters more than their spills -- and was removed.
template <bool SPHERES_ONLY, int TEX, bool NEED_UV, bool LDS_SCENE>
__global__ void __launch_bounds__(RT_TIER_THREADS, (SPHERES_ONLY && TEX < 2) ? 4 : 3) rt_tier_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // tail_mode: the launch after the main kernel that finishes the pixels it handed off (rt_device.h, "tail hand-off")
    const bool tail = (fp.tail_mode & 1) != 0;
    const rt_rank_info* q = fp.rank;
    const uint32_t n_items = tail ? (fp.work_counter[RT_WC_PUSHED] < fp.handoff_cap ? fp.work_counter[RT_WC_PUSHED] : fp.handoff_cap) : q->tier1_items;   // (slots taken beyond the queue's end were not filled)
    const int my_wgs = tail ? (int)((n_items + RT_TIER_THREADS / 64 - 1) / (RT_TIER_THREADS / 64)) : q->tier1_wgs;
    if ((int)blockIdx.x >= my_wgs) return;        // the grid is fixed before the ranking has sized the tier
    // these waves are the frame's critical path: they win instruction-issue arbitration on their SIMD (option tier_priority)
    if (fp.tier_priority >= 3) __builtin_amdgcn_s_setprio(3);
    else if (fp.tier_priority == 2) __builtin_amdgcn_s_setprio(2);
    else if (fp.tier_priority == 1) __builtin_amdgcn_s_setprio(1);

    // ---- the workgroup's LDS image: leaf arrays, slot unions, then (LDS_SCENE: the host's plan) spheres, materials, textures
    SceneView sc;
    sc.nodes = sd.nodes; sc.spheres = sd.spheres; sc.quads = sd.quads; sc.boxes = sd.boxes; sc.instances = sd.instances;
    sc.media = sd.media; sc.materials = sd.materials; sc.textures = sd.textures; sc.images = sd.images; sc.n_nodes = sd.n_nodes;
    TierView tv;
    {
        const int n_pad = sd.n_slots * 64;
        float4* dlo = reinterpret_cast<float4*>(lds);
        float4* dhi = dlo + n_pad;
        float4* dr = dhi + n_pad;                  // n_slots * 2 float4
        for (int k = (int)threadIdx.x; k < n_pad; k += (int)blockDim.x) { dlo[k] = sd.leaf_lo[k]; dhi[k] = sd.leaf_hi[k]; }
        const float4* sr = reinterpret_cast<const float4*>(sd.slot_ranges);
        for (int k = (int)threadIdx.x; k < sd.n_slots * 2; k += (int)blockDim.x) dr[k] = sr[k];
        if (LDS_SCENE) {
            float4* dsph = dr + sd.n_slots * 2;
            const float4* s4 = reinterpret_cast<const float4*>(sd.spheres);
            for (int k = (int)threadIdx.x; k < sd.n_spheres * 2; k += (int)blockDim.x) dsph[k] = s4[k];
            sc.spheres = reinterpret_cast<const rt_sphere*>(dsph);
            float4* dmat = dsph + sd.n_spheres * 2;
            const float4* m4 = reinterpret_cast<const float4*>(sd.materials);
            for (int k = (int)threadIdx.x; k < sd.n_materials * 2; k += (int)blockDim.x) dmat[k] = m4[k];
            sc.materials = reinterpret_cast<const rt_material*>(dmat);
            float4* dtex = dmat + sd.n_materials * 2;
            const float4* t4 = reinterpret_cast<const float4*>(sd.textures);
            for (int k = (int)threadIdx.x; k < sd.n_textures * 4; k += (int)blockDim.x) dtex[k] = t4[k];
            sc.textures = reinterpret_cast<const rt_texture*>(dtex);
        }
        __syncthreads();
        tv.lo = dlo; tv.hi = dhi; tv.ranges = reinterpret_cast<const float*>(dr); tv.n_slots = sd.n_slots;
        tv.n_media = SPHERES_ONLY ? 0 : sd.n_media_leaves;
        tv.med_ord0 = tv.n_media > 0 ? sd.media_ord[0] : 0x7FFFFFFF;
        tv.med_ord1 = tv.n_media > 1 ? sd.media_ord[1] : 0x7FFFFFFF;
    }

    unsigned int rays = 0;
#ifdef RT_DIAG   // diagnostic build only (tools/diag_tier_pace.py): cycles per ray in the traversal and in resolve + shade
    unsigned long long diag_trace = 0, diag_shade = 0, diag_rays = 0;
    const unsigned long long diag_t0 = __builtin_readcyclecounter();
#endif
    for (;;) {
        uint32_t idx = 0;
        if ((threadIdx.x & 63) == 0) idx = atomicAdd(fp.work_counter + (tail ? RT_WC_TAIL_HEAD : 2), 1u);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx >= n_items) break;
        const unsigned long long item = tail ? fp.handoff_queue[idx] : (unsigned long long)fp.heavy_pixels[idx];
        const uint32_t pix = (uint32_t)item;
        const int first_sample = tail ? (int)(item >> 32) : fp.sample_begin;
        if (tail && (threadIdx.x & 63) == 0) {     // rt_debug_handoff
            atomicAdd(fp.ray_counter + 28, 1ull); atomicAdd(fp.ray_counter + 29, (unsigned long long)(fp.sample_end - first_sample));
        }
        const int lrow = (int)(pix / (uint32_t)fp.nx), i = (int)(pix - (uint32_t)lrow * (uint32_t)fp.nx);
        const int j = local_to_global_row(fp, lrow);
        rt_xorwow pg;
        f3 pcol;
        uint32_t cost_before;
        {
            const rt_pixel_state st = tail ? fp.handoff_state[pix] : fp.state_in[pix];
            if (fp.fresh && !tail) {               // first part: the state only carries the cost prior and the list flag
                rt_xorwow_seed(pg, fp.seed_base + (uint64_t)(j * fp.nx + i));   // render_init, main.cu:101-104
                pcol = mk3(0, 0, 0);
                cost_before = st.cost & 0x80000000u;
            } else {
                pg.v0 = st.rng[0]; pg.v1 = st.rng[1]; pg.v2 = st.rng[2]; pg.v3 = st.rng[3]; pg.v4 = st.rng[4]; pg.d = st.rng[5];
                pcol = mk3(st.col[0], st.col[1], st.col[2]);
                cost_before = st.cost;
            }
        }
        unsigned int pixel_rays = 0;
        for (int sidx = first_sample; sidx < fp.sample_end; ++sidx) {                    // main.cu:119-125
            const float u = ((float)i + rt_xorwow_uniform(pg)) / (float)fp.nx;
            const float v = ((float)j + rt_xorwow_uniform(pg)) / (float)fp.ny;
            Ray r = camera_get_ray(sd.camera, u, v, pg);
            f3 thr = mk3(1, 1, 1), rad = mk3(0, 0, 0);
            for (int depth = 0; depth < 50; ++depth) {                                   // main.cu:54-84
                HitInfo h;
                ++pixel_rays;
#ifdef RT_DIAG
                const unsigned long long dg0 = __builtin_readcyclecounter();
#endif
                const bool hit_something = trace_wave<SPHERES_ONLY>(tv, sc, r, h);
#ifdef RT_DIAG
                const unsigned long long dg1 = __builtin_readcyclecounter();
                diag_trace += dg1 - dg0; ++diag_rays;
#endif
                if (!hit_something) { rad = fma3(thr, miss_color(fp, r), rad); break; }
                const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, r, h);
                f3 emitted, attenuation;
                Ray scattered;
                const bool go_on = shade<TEX>(sc, r, rec, pg, emitted, attenuation, scattered);
#ifdef RT_DIAG
                diag_shade += __builtin_readcyclecounter() - dg1;
#endif
                rad = fma3(thr, emitted, rad);
                if (!go_on) break;
                thr = thr * attenuation;
                r = scattered;
            }
            pcol = pcol + rad;
        }
        if ((threadIdx.x & 63) == 0) {
            if (fp.state_out) {   // a first or middle part of a split frame: park the pixel again
                rt_pixel_state so;
                so.rng[0] = pg.v0; so.rng[1] = pg.v1; so.rng[2] = pg.v2; so.rng[3] = pg.v3; so.rng[4] = pg.v4; so.rng[5] = pg.d;
                so.col[0] = pcol.x; so.col[1] = pcol.y; so.col[2] = pcol.z;
                so.cost = cost_before + pixel_rays;   // (bit 31, "listed", stays: this launch's tile queue must keep skipping the pixel)
                fp.state_out[pix] = so;
                if (fp.tile_cost) atomicAdd(&fp.tile_cost[(lrow >> 3) * fp.tiles_x + (i >> 3)], pixel_rays);
            } else {
                store_pixel(fp, i, lrow, pcol);
            }
            rays += pixel_rays;
        }
    }
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(fp.ray_counter, (unsigned long long)rays);
#ifdef RT_DIAG
    if (threadIdx.x == 0 && diag_rays) {   // first wave of each tier workgroup (rt_debug_counters slots 14 / 15, rt_debug_stage_cycles 8 / 9)
        atomicAdd(fp.ray_counter + 15, diag_trace); atomicAdd(fp.ray_counter + 16, diag_rays);
        atomicAdd(fp.ray_counter + 25, diag_shade); atomicAdd(fp.ray_counter + 26, (unsigned long long)(__builtin_readcyclecounter() - diag_t0));
    }
#endif
}

template <bool SO, int TX, bool UV, bool LS>
static hipError_t rt_launch_tier_variant(const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st) {
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_tier_kernel<SO, TX, UV, LS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((rt_tier_kernel<SO, TX, UV, LS>), grid, dim3(RT_TIER_THREADS), lds, st, sd, fp);
    return hipGetLastError();
}
template <bool SO, int TX, bool UV>
static hipError_t rt_launch_tier_one(const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st) {
    return fp.tier_lds_scene ? rt_launch_tier_variant<SO, TX, UV, true>(sd, fp, grid, lds, st) : rt_launch_tier_variant<SO, TX, UV, false>(sd, fp, grid, lds, st);
}
