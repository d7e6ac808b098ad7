// rt_kernel_staged.h -- kernel 3 ("staged"), the kernel that ships; instantiated by rt_staged_*.hip.
#pragma once
#ifndef RT_STEP_UNROLL
#define RT_STEP_UNROLL 2
#endif
#include "rt_device_funcs.h"

// =============================================================================
// Kernel D ("staged"): kernel C with the shading block cut into stages.
//
// Counters for kernel C (profiles/r01_bench_n1_parked_summary.txt, pmcsweep) put
// ~60 % of all issued VALU instructions in the shading block: ~1500
// instructions covering every material, the dielectric path, pixel
// finalisation and camera-ray generation, executed for ~24 waiting lanes of
// which each needs a fraction.  A microbenchmark (tools/ubench/valu_rate.hip)
// shows the SIMDs are close to issue-bound at ~3-4 cycles per VALU
// wave-instruction, so instructions issued for idle lanes are the cost.
//
// Here a lane's state is encoded in `node`:
//   0 <= node < n      walking the BVH (next box test)
//   node < 0           parked at a leaf; ~node is where the walk resumes
//   node == n + 0      traversal finished, hit/miss not yet classified
//   node == n + 1      path ended: needs accumulate + next sample / pixel + camera ray
//   node == n + 2      dielectric hit waiting for the (rare, long) dielectric stage
//   node == n + 3      new ray ready, needs per-ray setup (1/d etc.)
//   node == n + 4      no more work
// and each stage runs when a ballot finds enough lanes for it, or when no lane
// can walk any more.  Lambertian, metal and isotropic share one
// random_in_unit_sphere loop.  Stages only re-order work between lanes: every
// pixel still draws its own XORWOW stream in the reference's order.
// =============================================================================
// Diagnostic build only (-DRT_DIAG): per-stage execution counts, written to fp.diag (never to an output).
#ifdef RT_DIAG
#define DIAG_ADD(slot, value) do { const unsigned long long v_ = (unsigned long long)(value); if ((threadIdx.x & 63) == 0) diag_local[slot] += v_; } while (0)
#define DIAG_T(slot) do { const unsigned long long n_ = __builtin_readcyclecounter(); diag_time[slot] += n_ - diag_t_last; diag_t_last = n_; } while (0)
#else
#define DIAG_ADD(slot, value) do { } while (0)
#define DIAG_T(slot) do { } while (0)
#endif

template <bool SPHERES_ONLY, int TEX, bool NEED_UV, int LDS_MODE>
__global__ void __launch_bounds__((SPHERES_ONLY && TEX < 2) ? RT_LEAN_MAX_THREADS : RT_HEAVY_MAX_THREADS, (SPHERES_ONLY && TEX < 2) ? RT_LEAN_MIN_WAVES : RT_HEAVY_MIN_WAVES) rt_render_staged_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef RT_DIAG
    unsigned long long diag_local[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long diag_time[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long diag_shade_cycles = 0;
    unsigned long long diag_t_last = 0;
    const unsigned long long diag_t_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    if ((threadIdx.x & 63) == 0) atomicMin(fp.ray_counter + RT_DIAG_T0_SLOT, diag_t_start);
    unsigned long long diag_fetch_t = 0ull, diag_done_t = 0ull;   // when this lane took its last pixel / ran out of work
    unsigned int diag_last_px = 0u, diag_last_src = 0u;          // that pixel, and where it came from (0 tile, 2 sparse list, 3 tier 3)
#endif
    // LDS_MODE 4 = "scan": a scene of a few nodes (Cornell box: its 10 leaves, rt_abi.hip leaves_only_if_scanned) is not walked lane by lane.  All
    // lanes that have a ray go through the depth-first array together, node by node; node and object records have
    // wave-uniform addresses (scalar cache, no LDS), the object kind is a scalar branch, and a lane only carries the index
    // below which it skips (a failed interior box).  Same tests against the same limits in the same order as the walk.
    constexpr bool SCAN = LDS_MODE == 4;
    // tier sizes of a ranked launch, left in device memory by the ranking kernels (rt_rank.hip); wave-uniform
    rt_rank_info rk;
    rk.heavy_items = 0u; rk.heavy_threshold = 0xFFFFFFFFu; rk.tier1_items = 0u; rk.tier2_items = 0u;
    rk.tier1_wgs = 0; rk.main_skip_wgs = 0; rk.sparse_wgs = 0; rk.sparse_stride = 1; rk.semi_wgs = 0; rk.semi_stride = 1;
    if (fp.rank) {
        const rt_rank_info* q = fp.rank;
        rk.heavy_items = q->heavy_items; rk.heavy_threshold = q->heavy_threshold; rk.tier1_items = q->tier1_items;
        rk.tier2_items = q->tier2_items; rk.semi_wgs = q->semi_wgs; rk.semi_stride = q->semi_stride;
        rk.main_skip_wgs = q->main_skip_wgs; rk.sparse_wgs = q->sparse_wgs; rk.sparse_stride = q->sparse_stride;
    }
    // Tier 1 of the list -- the dearest pixels, one per wave -- is the tier kernel's (rt_kernel_tier.h), which runs beside
    // this one.  Where its workgroups do not fit next to a full grid of this kernel (register budgets, rt_device.h) the first
    // main_skip_wgs workgroups leave at once and their slots are taken by tier workgroups.
    const int wg = (int)blockIdx.x - rk.main_skip_wgs;
    if (wg < 0) return;
    const SceneView sc = stage_scene<LDS_MODE>(sd, lds);
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const int n_nodes = sc.n_nodes;
    const int ST_DONE = n_nodes, ST_NEWPATH = n_nodes + 1, ST_DIEL = n_nodes + 2, ST_SETUP = n_nodes + 3, ST_DEAD = n_nodes + 4;
    const float tmin = 0.001f;

    rt_xorwow g = {0, 0, 0, 0, 0, 0};
    int px_i = 0, px_lrow = 0, px_j = 0, sample = 0, bounce = 0;
    f3 col = mk3(0, 0, 0), throughput = mk3(1, 1, 1), radiance = mk3(0, 0, 0);
    Ray cur; cur.o = mk3(0, 0, 0); cur.d = mk3(0, 0, 1); cur.tm = 0.f;
    f3 inv = mk3(1, 1, 1);
    HitInfo best; best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    int node = ST_NEWPATH;       // every lane starts by asking for a pixel
    int32_t parked = -1;         // leaf node the lane stopped at (node < 0)
    int32_t pend = -1;           // leaf node whose box passed during the walk and whose object test is still due
    float cur_a = 1.f;           // dot(d, d) of the current ray (sphere.cuh:58), hoisted out of the sphere tests
    bool have_pixel = false, first = true, finite_inv = true;
    bool tier3_open = true;      // this lane has not yet seen the end of the tier-3 queue
    bool leave = false;          // tail hand-off (rt_device.h): the frame's last pixels leave for the tail launch at their next sample boundary.  Wave-uniform
    unsigned int wave_dead = 0;  // ... lanes of this wave counted in work_counter[RT_WC_DEAD]
    unsigned long long last_poll = 0ull;
    bool dry_seen = false;       // a lane of this wave has found its queue empty: from then on the wave looks at the count now and then
    const uint32_t total_lanes = (uint32_t)((int)gridDim.x - rk.main_skip_wgs) * blockDim.x;
    unsigned int rays = 0, rays_at_pixel_start = 0;
    // Sparse mode (see rt_abi.hip, "heavy tiles"): the first sparse_wgs workgroups start by serving tier 2 of the list with
    // only every sparse_stride-th lane, because a lane's rays advance ~2x faster in a wave with few live lanes and those
    // pixels' sequential chains bound the frame time.  When that queue is drained and the wave's own heavy pixels are
    // finished it becomes an ordinary wave.  Wave-uniform.
    bool sparse = wg < rk.sparse_wgs;
    // "semi" workgroups come next: tier 3 (listed pixels too cheap for a sparse wave) with every semi_stride-th lane live
    bool semi = !sparse && wg < rk.sparse_wgs + rk.semi_wgs;
#ifdef RT_DIAG
    const bool diag_was_sparse = sparse;
#endif
    // A sparse wave's few lanes are on the frame's critical path: let it win instruction-issue arbitration against the
    // three ordinary waves sharing its SIMD (priority outranks age, MI355X_MICROARCH.md "Two waves per SIMD").
    if (semi && fp.semi_priority > 0) {   // the same for the workgroups that serve tier 3 (pixels just below the sparse tier's cost)
        if (fp.semi_priority == 1) __builtin_amdgcn_s_setprio(1);
        else if (fp.semi_priority == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
    if (sparse && fp.sparse_priority > 0) {
        if (fp.sparse_priority == 1) __builtin_amdgcn_s_setprio(1);
        else if (fp.sparse_priority == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }

#ifdef RT_DIAG
    diag_t_last = __builtin_readcyclecounter();
#endif
    for (;;) {
        DIAG_ADD(0, 1);
        DIAG_T(7);
        if (SCAN) {
            // ---------------- stages A + B, scan form (see LDS_MODE 4 above)
            const bool scanning = (unsigned)node < (unsigned)n_nodes;      // a ray set up in stage F and not yet traced
            if (__ballot(scanning) != 0ull) {
                const bool ref_form = __ballot(scanning && !finite_inv) != 0ull;   // a zero direction component somewhere: aabb.cuh's own form for all
                int resume = 0;                                            // this lane skips nodes below this index
                for (int k = 0; k < n_nodes; ++k) {
                    const float4 a = uniform_load(nodes4 + 2 * k), b = uniform_load(nodes4 + 2 * k + 1);
                    const int32_t prim = __builtin_amdgcn_readfirstlane(__float_as_int(b.w));
                    const int skip = RT_NODE_SKIP(__builtin_amdgcn_readfirstlane(__float_as_int(a.w)));
                    const bool active = scanning && k >= resume;
                    DIAG_ADD(1, 1); DIAG_ADD(2, __popcll(__ballot(active)));
                    const bool pass = active && (ref_form ? slab_test(a, b, cur.o, inv, tmin, best.t) : slab_test_finite(a, b, cur.o, inv, tmin, best.t));
                    if (prim < 0) {
                        if (active && !pass) resume = skip;
                        DIAG_T(0);
                    } else if (__ballot(pass) != 0ull) {
                        DIAG_T(0);
                        DIAG_ADD(3, 1); DIAG_ADD(4, __popcll(__ballot(pass)));
                        if (pass) leaf_test<SPHERES_ONLY, true>(sc, prim, cur, tmin, best);
                        DIAG_T(1);
                    }
                }
                if (scanning) node = ST_DONE;
            }
        } else {
        // ---------------- stage A: node steps
        // A lane whose box test passes at a leaf does not stop there: it notes the leaf (`pend`) and walks on with the
        // limit it has; only a second leaf while the first is still due stops it (node = ~resume, `parked` = that leaf).
        // Stage B tests the noted leaves in visiting order, each against its own box AGAIN with the limit of that
        // moment.  This is the reference's walk exactly: bvh_node::hit (bvh.cuh:95-106) reaches an object iff every
        // ancestor box and the object's own box (its single-object node, bvh.cuh:38-43) pass with the closest hit so
        // far, and for a finite 1/d the slab test is monotone in the box and in the limit, so "the own box passes with
        // the limit at the time of the object test" already implies every ancestor passed earlier with its larger
        // limit.  Walking with a stale (larger) limit therefore visits a superset of the leaves, and the re-test in
        // stage B keeps exactly the reference's.  Rays with a zero direction component (no monotonicity: 0 * inf) stop
        // at every leaf, as before.
        if (__ballot(!finite_inv && (unsigned)node < (unsigned)n_nodes) == 0ull) {
            // two node steps per check: "nobody left walking" (all stopped or finished) ends the trip -- this is what
            // keeps the latency of a wave's last few live lanes near one node step per step (end of frame, small
            // multi-GPU partitions); checking every other step halves the loop's scalar overhead
            const int trip_pairs = ((sparse ? 2 * fp.steps_per_trip : fp.steps_per_trip) + RT_STEP_UNROLL - 1) / RT_STEP_UNROLL;
            // the loop's box test is the widened one-fma-per-bound form (rt_device_funcs.h, slab_test_loose): a superset of
            // aabb::hit's passes, which is all the walk needs -- stage B tests every noted leaf's own box again, exactly
            const LooseRay lr = loose_setup(inv, cur.o, sd.bound);
            for (int pair = 0; pair < trip_pairs; ++pair) {
                if (__ballot((unsigned)node < (unsigned)n_nodes) == 0ull) break;
#pragma unroll
                for (int half = 0; half < RT_STEP_UNROLL; ++half) {
                    DIAG_ADD(1, 1); DIAG_ADD(2, __popcll(__ballot((unsigned)node < (unsigned)n_nodes)));
                    if ((unsigned)node < (unsigned)n_nodes) {
                        const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                        const bool pass = slab_test_loose(a, b, inv, lr, tmin, best.t);
                        // device encoding of the links (rt_device.h, RT_NODE_SKIP): a.w = ~skip, b.w = ~(node + 1) for
                        // an interior node and the object id (>= 0) at a leaf -- "where next" is one select and one
                        // NOT, and the stopped form ~skip is a.w as stored
                        const int32_t link = __float_as_int(b.w), nskip = __float_as_int(a.w);
                        const bool at_leaf = pass && link >= 0;
                        const bool stop = at_leaf && pend >= 0;
                        const int next = ~((pass && link < 0) ? link : nskip);
                        pend = (at_leaf && pend < 0) ? node : pend;
                        parked = stop ? node : parked;
                        node = stop ? nskip : next;
                    }
                }
            }
        } else {   // a lane's ray has a zero direction component: the reference's own slab form for this trip
            for (int step = 0; step < fp.steps_per_trip; ++step) {
                if (__ballot((unsigned)node < (unsigned)n_nodes) == 0ull) break;
                if ((unsigned)node < (unsigned)n_nodes) {
                    const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                    const bool pass = slab_test(a, b, cur.o, inv, tmin, best.t);
                    const int32_t link = __float_as_int(b.w), nskip = __float_as_int(a.w);
                    const bool at_leaf = pass && link >= 0;
                    const bool stop = at_leaf && (pend >= 0 || !finite_inv);
                    const int next = ~((pass && link < 0) ? link : nskip);
                    pend = (at_leaf && !stop) ? node : pend;
                    parked = stop ? node : parked;
                    node = stop ? nskip : next;
                }
            }
        }
        DIAG_T(0);
        // ---------------- stage B: leaf pass -- one object test per lane that has one due (the earliest noted leaf).
        // A lane stopped at a second leaf resumes with that one noted; if its walk ends right there and nobody else can
        // step either, the pass runs again at once (a wave must never reach the stage logic below with a test due and
        // nothing walking: "nothing walking, no stage ran" is the loop's exit condition).
        for (;;) {
            const int cand = pend >= 0 ? pend : (node < 0 ? parked : -1);
            const unsigned long long due_mask = __ballot(cand >= 0);
            const bool nobody_steps = __ballot((unsigned)node < (unsigned)n_nodes) == 0ull;
            const int live_b = __popcll(__ballot(node != ST_DEAD));
            if (due_mask != 0ull && (nobody_steps || sparse || __popcll(due_mask) >= 1 + ((fp.leaf_threshold - 1) * live_b >> 6))) {
                DIAG_ADD(3, 1); DIAG_ADD(4, __popcll(due_mask));
                int kind = -1;
                int32_t prim = -1;
                bool box_ok = false;
                if (cand >= 0) {
                    const float4 a = nodes4[2 * cand], b = nodes4[2 * cand + 1];
                    prim = __float_as_int(b.w);
                    kind = SPHERES_ONLY ? RT_PRIM_SPHERE : RT_PRIM_KIND(prim);
                    // the leaf's own box against the limit of THIS moment (see stage A); a ray with a zero direction
                    // component stopped right at the leaf, so its walk-time result still stands
                    box_ok = finite_inv ? slab_test_finite(a, b, cur.o, inv, tmin, best.t) : true;
                }
                // the tested leaf is done: a lane stopped at a second leaf walks on with that one noted
                bool tested = false;
                if (SPHERES_ONLY) {
                    if (cand >= 0) {
                        float t;
                        if (box_ok && sphere_test_a(sc.spheres[RT_PRIM_INDEX(prim)], cur, cur_a, tmin, best.t, t)) { best.t = t; best.prim = prim; best.inst = -1; }
                        tested = true;
                    }
                } else {
                    // General scenes: spheres and quads are served by every pass; boxes / instances (six quad tests, a
                    // transform) and media (two boundary tests, a private XORWOW, a logarithm) are long, so their lanes
                    // wait until a ballot finds enough of them -- or nobody is left who could step.
                    if (kind == RT_PRIM_SPHERE || kind == RT_PRIM_QUAD) {
                        float t;
                        const bool hit = box_ok && (kind == RT_PRIM_SPHERE ? sphere_test_a(sc.spheres[RT_PRIM_INDEX(prim)], cur, cur_a, tmin, best.t, t)
                                                                           : quad_test(sc.quads[RT_PRIM_INDEX(prim)], cur, tmin, best.t, t));
                        if (hit) { best.t = t; best.prim = prim; best.inst = -1; }
                        tested = true;
                    }
                    const unsigned long long box_mask = __ballot(kind == RT_PRIM_BOX || kind == RT_PRIM_INSTANCE);
                    if (box_mask != 0ull && (nobody_steps || __popcll(box_mask) >= 1 + ((fp.box_threshold - 1) * live_b >> 6))) {
                        if (kind == RT_PRIM_BOX || kind == RT_PRIM_INSTANCE) {
                            float t;
                            int32_t leaf = prim, inst = -1;
                            if (box_ok && solid_test(sc, prim, cur, tmin, best.t, t, leaf, inst)) { best.t = t; best.prim = leaf; best.inst = inst; }
                            tested = true;
                        }
                    }
                    const unsigned long long med_mask = __ballot(kind == RT_PRIM_MEDIUM);
                    if (med_mask != 0ull && (nobody_steps || __popcll(med_mask) >= 1 + ((fp.medium_threshold - 1) * live_b >> 6))) {
                        if (kind == RT_PRIM_MEDIUM) {
                            float t;
                            if (box_ok && medium_test(sc, sc.media[RT_PRIM_INDEX(prim)], cur, tmin, best.t, t)) { best.t = t; best.prim = prim; best.inst = -1; }
                            tested = true;
                        }
                    }
                }
                if (tested) {
                    if (node < 0) { pend = pend >= 0 ? parked : -1; parked = -1; node = ~node; }
                    else pend = -1;
                }
                if (nobody_steps && __ballot((unsigned)node < (unsigned)n_nodes) == 0ull && __ballot(pend >= 0 || node < 0) != 0ull) continue;
            }
            break;
        }
        }
        DIAG_T(1);
        // a finished walk that hit nothing (main.cu:57-68) needs no stage: add the background and end the path now
        if (node == ST_DONE && pend < 0 && best.prim < 0) {
            radiance = fma3(throughput, miss_color(fp, cur), radiance);
            node = ST_NEWPATH;
        }
        const unsigned long long walking = __ballot(node < n_nodes);
        const bool force = walking == 0ull;
        const int n_done = __popcll(__ballot(node == ST_DONE && pend < 0));
        bool ran_stage = false;
        // The thresholds are fractions of the lanes that still have work: a wave whose lanes are running out of pixels
        // (end of the frame, or a small row partition on a multi-GPU run) must not wait for 24 lanes it no longer has.
        const int live = __popcll(__ballot(node != ST_DEAD));
        const int shade_need = 1 + ((fp.shade_threshold - 1) * live >> 6);
        const int diel_need = 1 + ((fp.diel_threshold - 1) * live >> 6);
        const int newpath_need = 1 + ((fp.newpath_threshold - 1) * live >> 6);
        const bool eager = sparse && fp.sparse_eager;     // sparse waves trade their own throughput for latency

        DIAG_T(6);
        // ---------------- stage C: classify + resolve + diffuse/metal/isotropic scatter
        if (n_done > 0 && (n_done >= shade_need || force || eager)) {
            ran_stage = true;
            DIAG_ADD(5, 1); DIAG_ADD(6, n_done);
            DIAG_ADD(11, __popcll(__ballot(node == ST_DONE && pend < 0 && best.prim >= 0)));
            if (node == ST_DONE && pend < 0) {
                {
                    const HitRec rec = resolve_hit<SPHERES_ONLY, NEED_UV>(sc, cur, best);
                    const rt_material m = sc.materials[rec.mat];
                    if (m.kind == RT_MAT_DIELECTRIC) {
                        if (sparse) {
                            // a sparse wave's pixels are mostly glass: scatter here instead of queueing for stage D
                            // (saves a second resolve_hit and a stage round trip on the frame's critical chain)
                            const f3 dir = dielectric_direction(cur.d, rec.n, m.ior, g);
                            ++bounce;
                            if (bounce >= 50) node = ST_NEWPATH;
                            else { cur.o = rec.p; cur.d = dir; node = ST_SETUP; }   // attenuation (1,1,1): throughput unchanged
                        } else {
                            node = ST_DIEL;
                        }
                    } else if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
                        const f3 emitted = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                        radiance = fma3(throughput, emitted, radiance);         // main.cu:71, scatter() false
                        node = ST_NEWPATH;
                    } else {
                        // lambertian / metal / isotropic: one shared rejection loop (material.cuh:12-18)
                        const f3 rs = random_in_unit_sphere(g);
                        f3 dir, attenuation;
                        bool go_on = true;
                        if (m.kind == RT_MAT_METAL) {                       // material.cuh:99-109
                            const f3 reflected = reflect(unit_vector(cur.d), rec.n);
                            dir = fma3(m.fuzz, rs, reflected);
                            attenuation = ld3(m.albedo);
                            go_on = dot(dir, rec.n) > 0.0f;
                        } else {
                            if (m.kind == RT_MAT_LAMBERTIAN) {              // material.cuh:75-86
                                const f3 target = (rec.p + rec.n) + rs;
                                dir = target - rec.p;
                            } else {                                        // isotropic, material.cuh:193-200
                                dir = rs;
                            }
                            attenuation = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                        }
                        ++bounce;
                        if (!go_on || bounce >= 50) node = ST_NEWPATH;      // main.cu:54,76-80
                        else {
                            throughput = throughput * attenuation;
                            cur.o = rec.p; cur.d = dir;                     // time carried over
                            node = ST_SETUP;
                        }
                    }
                }
            }
        }
        DIAG_T(2);
        // ---------------- stage D: dielectric scatter (material.cuh:119-159)
        {
            const int n_diel = __popcll(__ballot(node == ST_DIEL));
            if (n_diel > 0 && (n_diel >= diel_need || force || eager)) {
                ran_stage = true;
                DIAG_ADD(7, 1); DIAG_ADD(8, n_diel);
                if (node == ST_DIEL) {
                    const HitRec rec = resolve_hit<SPHERES_ONLY, false>(sc, cur, best);   // a dielectric reads no (u, v)
                    const f3 dir = dielectric_direction(cur.d, rec.n, sc.materials[rec.mat].ior, g);
                    ++bounce;
                    if (bounce >= 50) node = ST_NEWPATH;
                    else { throughput = throughput * mk3(1.0f, 1.0f, 1.0f); cur.o = rec.p; cur.d = dir; node = ST_SETUP; }
                }
            }
        }
        DIAG_T(3);
        // ---------------- stage E: path end -> next sample / next pixel -> camera ray (main.cu:119-132)
        {
            const int n_new = __popcll(__ballot(node == ST_NEWPATH));
            if (n_new > 0 && (n_new >= newpath_need || force || eager)) {
                ran_stage = true;
                DIAG_ADD(9, 1); DIAG_ADD(10, n_new);
                // tail hand-off: once one of its lanes has found its queue empty, a wave looks now and then (option handoff_poll_us)
                // at how many lanes of the launch still have work.  The look is a load that misses every cache and stalls the
                // wave: on the headline frame every 8th time round (0.15 ms) cost 2 ms more than every 64th (profiles/r03_handoff.log).
                if (fp.handoff_queue && !leave && dry_seen) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();   // 100 MHz
                    if (now - last_poll >= (unsigned long long)fp.handoff_poll_ticks) {
                        last_poll = now;
                        const uint32_t dead = __hip_atomic_load(fp.work_counter + RT_WC_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        leave = __builtin_amdgcn_readfirstlane((int)(total_lanes - dead)) <= fp.handoff_pixels;
                    }
                }
                bool ran_dry = false;      // this lane found its queue empty
                if (node == ST_NEWPATH) {
                    if (!first) { col = col + radiance; ++sample; }
                    first = false;
                    bool alive = true;
                    const bool part_done = sample >= fp.sample_end;
                    // hand-off: samples [sample, sample_end) go to the tail launch.  The lane takes a slot of the queue first; the queue has
                    // one per resident lane, which a lane exceeds only if its wave turns ordinary again and again with the threshold set
                    // absurdly high -- a lane that finds the queue full keeps its pixel
                    bool hand = have_pixel && !part_done && leave;
                    uint32_t slot = 0u;
                    if (hand) { slot = atomicAdd(fp.work_counter + RT_WC_PUSHED, 1u); hand = slot < fp.handoff_cap; }
                    if (have_pixel && (part_done || hand)) {
                        if (fp.state_out || hand) {
                            // first part of a split frame: park the pixel at this sample boundary (no path is in flight
                            // here, so the XORWOW state and the colour sum are the whole state) and record what it cost
                            const unsigned int c = rays - rays_at_pixel_start;
                            const size_t at = (size_t)px_lrow * fp.nx + px_i;
                            rt_pixel_state st;
                            st.rng[0] = g.v0; st.rng[1] = g.v1; st.rng[2] = g.v2; st.rng[3] = g.v3; st.rng[4] = g.v4; st.rng[5] = g.d;
                            st.col[0] = col.x; st.col[1] = col.y; st.col[2] = col.z;
                            st.cost = c + (fp.state_in ? (fp.fresh ? (fp.state_in[at].cost & 0x80000000u) : fp.state_in[at].cost) : 0u);   // a middle part adds to what the pixel cost before; bit 31 ("listed") stays; a fresh part drops the prior
                            (hand ? fp.handoff_state : fp.state_out)[at] = st;
                            if (fp.tile_cost) atomicAdd(&fp.tile_cost[(px_lrow >> 3) * fp.tiles_x + (px_i >> 3)], c);
                            if (hand) {
                                fp.handoff_queue[slot] = ((unsigned long long)(uint32_t)sample << 32) | (unsigned long long)at;
                                alive = false;
                            } else if (fp.store_parked) store_pixel(fp, px_i, px_lrow, col);   // a progressive window: the frame so far
                        } else {
                            store_pixel(fp, px_i, px_lrow, col);
                        }
                        have_pixel = false;
                    }
                    while (!have_pixel && alive) {
                        bool ok;
                        if (sparse) {
                            // tier 2 of the heavy list (sorted by descending cost; its first tier1_items entries are the tier kernel's)
                            if (((threadIdx.x & 63) % (unsigned)rk.sparse_stride) != 0u) { alive = false; break; }
                            const uint32_t k2 = atomicAdd(fp.work_counter + 1, 1u);
                            if (k2 >= rk.tier2_items) { alive = false; ran_dry = true; break; }
                            const uint32_t at = rk.tier1_items + k2;
                            const uint32_t pix = fp.heavy_pixels[at];
                            px_lrow = (int)(pix / (uint32_t)fp.nx); px_i = (int)(pix - (uint32_t)px_lrow * (uint32_t)fp.nx);
                            ok = true;
                        } else if (tier3_open && (semi || rk.semi_wgs == 0)) {
                            // tier 3: listed pixels too cheap for a sparse wave start at once, dearest first -- on the semi
                            // workgroups' live lanes, or (no semi workgroups) on any ordinary lane before it takes a tile
                            if (semi && ((threadIdx.x & 63) % (unsigned)rk.semi_stride) != 0u) { alive = false; break; }
                            const uint32_t k3 = atomicAdd(fp.work_counter + 4, 1u);
                            const uint32_t first3 = rk.tier1_items + rk.tier2_items;
                            if (first3 + k3 >= rk.heavy_items) { tier3_open = false; if (semi) { alive = false; ran_dry = true; break; } continue; }
                            const uint32_t pix = fp.heavy_pixels[first3 + k3];
                            px_lrow = (int)(pix / (uint32_t)fp.nx); px_i = (int)(pix - (uint32_t)px_lrow * (uint32_t)fp.nx);
                            ok = true;
                        } else {
                            const uint32_t w = atomicAdd(fp.work_counter, 1u);
                            if (w >= fp.work_items) { alive = false; ran_dry = true; break; }
                            ok = work_to_pixel(fp, w, px_i, px_lrow);
                            // pixels in the heavy list belong to the tiers
                            if (ok && rk.heavy_items && (fp.state_in[(size_t)px_lrow * fp.nx + px_i].cost & 0x80000000u) != 0u) ok = false;
                        }
                        if (ok) {
#ifdef RT_DIAG
                            diag_fetch_t = __builtin_amdgcn_s_memrealtime(); diag_last_px = (unsigned int)px_lrow * (unsigned int)fp.nx + (unsigned int)px_i;
                            diag_last_src = sparse ? 2u : (tier3_open ? 3u : 0u);
#endif
                            px_j = local_to_global_row(fp, px_lrow);
                            if (fp.state_in && !fp.fresh) {   // a later part of a split frame: pick the pixel up where the previous part left it
                                const rt_pixel_state st = fp.state_in[(size_t)px_lrow * fp.nx + px_i];
                                g.v0 = st.rng[0]; g.v1 = st.rng[1]; g.v2 = st.rng[2]; g.v3 = st.rng[3]; g.v4 = st.rng[4]; g.d = st.rng[5];
                                col = mk3(st.col[0], st.col[1], st.col[2]);
                            } else {
                                rt_xorwow_seed(g, fp.seed_base + (uint64_t)(px_j * fp.nx + px_i));
                                col = mk3(0, 0, 0);
                            }
                            sample = fp.sample_begin; have_pixel = true;
                            rays_at_pixel_start = rays;
                        }
                    }
                    if (alive) {
                        const float u = ((float)px_i + rt_xorwow_uniform(g)) / (float)fp.nx;
                        const float v = ((float)px_j + rt_xorwow_uniform(g)) / (float)fp.ny;
                        cur = camera_get_ray(sd.camera, u, v, g);
                        throughput = mk3(1, 1, 1); radiance = mk3(0, 0, 0); bounce = 0;
                        node = ST_SETUP;
                    } else {
                        node = ST_DEAD;
#ifdef RT_DIAG
                        diag_done_t = __builtin_amdgcn_s_memrealtime();
#endif
                    }
                }
                if (fp.handoff_queue) {   // lanes of this launch that have run out of work: work_counter[RT_WC_DEAD]
                    dry_seen = dry_seen || __ballot(ran_dry) != 0ull;
                    const unsigned int n_died = (unsigned int)__popcll(__ballot(node == ST_DEAD));
                    if (n_died != wave_dead) {
                        if ((threadIdx.x & 63) == 0) atomicAdd(fp.work_counter + RT_WC_DEAD, n_died - wave_dead);   // (no return value: nothing waits for it)
                        wave_dead = n_died;
                    }
                }
            }
        }
        DIAG_T(4);
        // ---------------- stage F: per-ray setup
        if (ran_stage) {
            DIAG_ADD(12, 1); DIAG_ADD(13, __popcll(__ballot(node == ST_SETUP)));
            if (node == ST_SETUP) {
                inv = mk3(1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z);
                finite_inv = inv_is_finite(inv) && loose_ok(inv, cur.o, sd.bound);   // (else: the reference's own slab form, leaf by leaf)
                cur_a = dot(cur.d, cur.d);
                best.t = FLT_MAX; best.prim = -1; best.inst = -1;
                node = n_nodes > 0 ? 0 : ST_DONE;
                ++rays;
            }
        } else if (force) {
            // nothing walking, nothing waiting: every lane is ST_DEAD
            if (!sparse && !semi) break;
            sparse = false; semi = false;           // our queue is drained and our pixels are done: become an ordinary wave
            if (fp.handoff_queue && wave_dead) {    // its lanes are alive again
                if ((threadIdx.x & 63) == 0) atomicAdd(fp.work_counter + RT_WC_DEAD, 0u - wave_dead);
                wave_dead = 0;
            }
            __builtin_amdgcn_s_setprio(0);
            node = ST_NEWPATH; first = true; have_pixel = false;
        }
    }
    unsigned long long r64 = rays;
    for (int off = 32; off > 0; off >>= 1) r64 += __shfl_down(r64, off, 64);
    if ((threadIdx.x & 63) == 0 && r64) atomicAdd(fp.ray_counter, r64);
#ifdef RT_DIAG
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 16; ++k) atomicAdd(fp.ray_counter + 1 + k, diag_local[k]);
        for (int k = 0; k < 8; ++k) atomicAdd(fp.ray_counter + 17 + k, diag_time[k]);
        if (diag_shade_cycles) atomicAdd(fp.ray_counter + 25, diag_shade_cycles);
        const unsigned long long t0 = *reinterpret_cast<volatile unsigned long long*>(fp.ray_counter + RT_DIAG_T0_SLOT);
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        unsigned long long bin = now > t0 ? (now - t0) / 100000ull : 0ull;   // 1 ms bins
        if (bin >= (unsigned long long)RT_DIAG_BINS) bin = RT_DIAG_BINS - 1;
        atomicAdd(fp.ray_counter + RT_DIAG_HIST_SLOT + (diag_was_sparse ? RT_DIAG_BINS : 0) + bin, 1ull);
    }
    {   // the lane that ran out of work last: when it took its last pixel, which pixel, from which queue
        unsigned long long best_done = diag_done_t;
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(best_done, off, 64); best_done = o > best_done ? o : best_done; }
        const unsigned long long t0 = *reinterpret_cast<volatile unsigned long long*>(fp.ray_counter + RT_DIAG_T0_SLOT);
        const unsigned int wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (diag_done_t == best_done && best_done != 0ull && wave_id < (unsigned int)RT_DIAG_MAX_WAVES) {
            const unsigned long long f_us = diag_fetch_t > t0 ? (diag_fetch_t - t0) / 100ull : 0ull, d_us = best_done > t0 ? (best_done - t0) / 100ull : 0ull;
            fp.ray_counter[RT_DIAG_WAVE_SLOT + 2 * wave_id] = (d_us << 32) | (f_us & 0xFFFFFFFFull);
            fp.ray_counter[RT_DIAG_WAVE_SLOT + 2 * wave_id + 1] = ((unsigned long long)diag_last_src << 60) | ((unsigned long long)(diag_was_sparse ? 1u : 0u) << 59) | diag_last_px;
        }
    }
#endif
}

// One specialisation family per translation unit (rt_staged_*.hip), so that the families compile in parallel.
template <bool SO, int TX, bool UV>
static hipError_t rt_launch_staged_family(int lds_mode, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                                          size_t lds, hipStream_t st) {
#define RT_STAGED_LAUNCH(LM)                                                                                              \
    do {                                                                                                                  \
        if (lds > 65536) {                                                                                                \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_staged_kernel<SO, TX, UV, LM>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
            if (e_ != hipSuccess) return e_;                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL((rt_render_staged_kernel<SO, TX, UV, LM>), grid, block, lds, st, sd, fp);                      \
        return hipGetLastError();                                                                                         \
    } while (0)
    if (lds_mode == 4) RT_STAGED_LAUNCH(4);
    if (lds_mode == 3) RT_STAGED_LAUNCH(3);
    if (lds_mode == 2) RT_STAGED_LAUNCH(2);
    if (lds_mode == 1) RT_STAGED_LAUNCH(1);
    RT_STAGED_LAUNCH(0);
#undef RT_STAGED_LAUNCH
}
