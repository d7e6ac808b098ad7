// rt_kernel_group.h -- the group kernel: 64 / G pixels per wave, G lanes each, all advancing one ray per iteration in
// lockstep; instantiated by rt_group_spheres.hip.
//
// Between the main kernel's lanes (a ray every ~27 us: one lane walks the tree node by node) and the tier kernel's waves (a
// ray every ~2.5 us, but a whole wave per pixel) this is the middle gear for "tier 2" of the ranking -- pixels a few times
// dearer than the mean, too many for a wave each, whose chains would otherwise end the launch (and bound every multi-GPU
// share of a frame).  A group of G lanes owns ONE pixel and traces each of its rays through a G-ary hierarchy of union boxes
// over the reference's leaf sequence (rt_scene_dev: grp_lo / grp_hi), G boxes per step, one per lane:
//   * level 0 = the leaves, in the reference's depth-first order; level k + 1 = unions of G consecutive level-k boxes; the top
//     level has at most G boxes and is tested first.  A group walks this tree in FIXED order (patch by patch, ascending leaf
//     ordinals) with one bit mask of passed children per level -- no dependent chain of single-node visits: a ray of the
//     headline scene is ~12 steps instead of ~23, and all 64 lanes of the wave are busy in every one.
//   * Exactness, as for the walk array and trace_wave (rt_kernel_tier.h): interior boxes only ever cull, and a
//     union contains what is below it, so testing them against a stale (larger) limit visits a superset of the leaves.  The G
//     leaves of a patch are tested together against the limit the group has when it reaches the patch: own box, then the
//     sphere with that limit (sphere.cuh:66, t < limit).  The reference, walking the same leaves one by one, ends the patch
//     with the minimum of (t, ordinal) over exactly those candidates -- a leaf it skips inside the patch was cut off by an
//     earlier candidate b <= T_leaf < t_leaf (the entry distance of its own box; "t > T" is checked and a violation hands the
//     ray to the reference's walk) and is not the minimum.  Patches are taken in ordinal order with the true running limit, so
//     a later equal t never replaces an earlier one.
//   * Every lane of a group carries the same pixel and computes the same values (XORWOW stream, colour sum, ray); the groups of
//     a wave differ in data only, so shading and camera rays cost one pass over each material kind present, not one per pixel.
// Spheres-only scenes (this file); rays with a zero direction component take the reference's walk.
#pragma once
#include "rt_device_funcs.h"

#define RT_GROUP_THREADS 256
#define RT_GROUP_MAX_LEVELS 4

// x from the lane of this lane's row / group given by a DPP control (gfx9: quad_perm, row_half_mirror, row_mirror)
template <int CTRL> DEV float dpp_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false)); }
template <int CTRL> DEV int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false); }
// minimum over the G lanes of a group (G = 8 or 16: groups are aligned halves / whole rows of a 16-lane DPP row); every lane
// of the group gets the result
template <int G> DEV float group_min_f(float x) {
    x = fminf(x, dpp_f<0xB1>(x));     // quad_perm [1,0,3,2]
    x = fminf(x, dpp_f<0x4E>(x));     // quad_perm [2,3,0,1]
    x = fminf(x, dpp_f<0x141>(x));    // row_half_mirror: lane i <-> 7 - i within each 8
    if (G >= 16) x = fminf(x, dpp_f<0x140>(x));   // row_mirror: lane i <-> 15 - i
    return x;
}
template <int G> DEV int group_min_i(int x) {
    x = min(x, dpp_i<0xB1>(x));
    x = min(x, dpp_i<0x4E>(x));
    x = min(x, dpp_i<0x141>(x));
    if (G >= 16) x = min(x, dpp_i<0x140>(x));
    return x;
}

// the group tree in LDS
struct GroupTree {
    const float4* lo;
    const float4* hi;
    int top;               // highest level
    int off[RT_GROUP_MAX_LEVELS];
};

// closest hit for the ray of every group of the wave (`active`: this lane's group has a ray to trace)
template <int G>
DEV void trace_groups(const GroupTree& gt, const SceneView& sc, bool active, const Ray& r, HitInfo& best) {
    const float tmin = 0.001f;
    const int lane = (int)(threadIdx.x & 63u), l = lane & (G - 1), gbase = lane & ~(G - 1);
    const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const float a = dot(r.d, r.d);
    best.t = FLT_MAX; best.prim = -1; best.inst = -1;
    const bool finite = inv_is_finite(inv);
    bool walking = active && finite, anomaly = false;
    // walk state: per level the mask of children that passed and are still to be visited, and the index (at the level above)
    // of the node they belong to; level top + 1 is a virtual root with one child 0
    uint32_t m1 = 0u, m2 = 0u, m3 = 0u, m4 = 0u;
    int b1 = 0, b2 = 0, b3 = 0;
    int lvl = gt.top + 1;
    { const uint32_t one = 1u; if (lvl == 1) m1 = one; else if (lvl == 2) m2 = one; else if (lvl == 3) m3 = one; else m4 = one; }
    while (__ballot(walking) != 0ull) {
        // ---- pop to the next level that still has a child to visit
        if (walking) {
#pragma unroll
            for (int k = 1; k <= RT_GROUP_MAX_LEVELS; ++k) {
                const uint32_t mk = lvl == 1 ? m1 : (lvl == 2 ? m2 : (lvl == 3 ? m3 : m4));
                if (mk == 0u && lvl <= gt.top + 1) ++lvl;
            }
            if (lvl > gt.top + 1) walking = false;
        }
        if (walking) {
            // ---- take the first remaining child of that level and test ITS children (one level down), one per lane
            uint32_t mk = lvl == 1 ? m1 : (lvl == 2 ? m2 : (lvl == 3 ? m3 : m4));
            const int c = __ffs((int)mk) - 1;
            mk &= mk - 1u;
            if (lvl == 1) m1 = mk; else if (lvl == 2) m2 = mk; else if (lvl == 3) m3 = mk; else m4 = mk;
            const int parent = lvl == 1 ? b1 : (lvl == 2 ? b2 : (lvl == 3 ? b3 : 0));
            const int idx = parent * G + c;                        // the node, at level lvl (the virtual root: 0)
            const int down = lvl - 1;                              // its children's level
            const int off = down == 0 ? gt.off[0] : (down == 1 ? gt.off[1] : (down == 2 ? gt.off[2] : gt.off[3]));
            const int child = idx * G + l;
            const float4 lo4 = gt.lo[off + child], hi4 = gt.hi[off + child];
            float t_enter, t_exit;
            slab_interval(lo4, hi4, r.o, inv, tmin, t_enter, t_exit);
            const bool pass = hi4.w != 0.0f && !(fminf(t_exit, best.t) <= t_enter);      // aabb::hit(tmin, closest so far)
            if (down == 0) {
                // leaves: the sphere against the same limit; the patch's closest candidate, lowest ordinal first
                const int32_t prim = __float_as_int(lo4.w);
                float t = FLT_MAX;
                bool hit = false;
                if (pass && prim >= 0) hit = sphere_test_a(sc.spheres[RT_PRIM_INDEX(prim)], r, a, tmin, best.t, t);
                if (hit && !(t > t_enter)) anomaly = true;
                const float tg = group_min_f<G>(hit ? t : FLT_MAX);
                if (tg < FLT_MAX) {
                    const int wl = group_min_i<G>((hit && t == tg) ? l : G);      // the first lane = the lowest ordinal
                    best.t = tg;
                    best.prim = __shfl(prim, gbase + wl, 64);
                }
            } else {
                const uint32_t mask = (uint32_t)(__ballot(pass) >> gbase) & ((1u << G) - 1u);
                if (down == 1) { m1 = mask; b1 = idx; } else if (down == 2) { m2 = mask; b2 = idx; } else { m3 = mask; b3 = idx; }
                lvl = down;
            }
        }
    }
    // a grazing hit at or before its own box's entry somewhere in the group, or a zero direction component: the reference's walk
    const bool redo = active && (!finite || group_min_i<G>(anomaly ? 0 : 1) == 0);
    if (__ballot(redo) != 0ull) { if (redo) (void)trace<true>(sc, r, best); }
}

// LDS_SCENE: as in the tier kernel (spheres, materials and textures in the workgroup's LDS image besides the group tree)
template <int TEX, bool NEED_UV, bool LDS_SCENE, int G>
__global__ void __launch_bounds__(RT_GROUP_THREADS, TEX < 2 ? 4 : 3) rt_group_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const rt_rank_info* q = fp.rank;
    if ((int)blockIdx.x >= q->group_wgs) return;       // the grid is fixed before the ranking has sized the tier
    const uint32_t first_item = q->tier1_items, n_items = q->tier2_items;
    if (fp.tier_priority >= 3) __builtin_amdgcn_s_setprio(2);
    else if (fp.tier_priority >= 1) __builtin_amdgcn_s_setprio(1);

    SceneView sc;
    sc.nodes = sd.nodes; sc.spheres = sd.spheres; sc.quads = sd.quads; sc.boxes = sd.boxes; sc.instances = sd.instances;
    sc.media = sd.media; sc.materials = sd.materials; sc.textures = sd.textures; sc.images = sd.images; sc.n_nodes = sd.n_nodes;
    GroupTree gt;
    {
        float4* dlo = reinterpret_cast<float4*>(lds);
        float4* dhi = dlo + sd.grp_total;
        for (int k = (int)threadIdx.x; k < sd.grp_total; k += (int)blockDim.x) { dlo[k] = sd.grp_lo[k]; dhi[k] = sd.grp_hi[k]; }
        if (LDS_SCENE) {
            float4* dsph = dhi + sd.grp_total;
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
        gt.lo = dlo; gt.hi = dhi; gt.top = sd.grp_top;
        for (int k = 0; k < RT_GROUP_MAX_LEVELS; ++k) gt.off[k] = sd.grp_off[k];
    }

    const int lane = (int)(threadIdx.x & 63u), gbase = lane & ~(G - 1);
    const bool leader = (lane & (G - 1)) == 0;
    // the group's pixel: every lane of the group holds the same values
    rt_xorwow pg = {0, 0, 0, 0, 0, 0};
    f3 pcol = mk3(0, 0, 0), thr = mk3(1, 1, 1), rad = mk3(0, 0, 0);
    Ray r; r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1); r.tm = 0.f;
    int px_i = 0, px_j = 0, px_lrow = 0, sample = 0, depth = 0;
    uint32_t pix = 0u, cost_before = 0u;
    unsigned int pixel_rays = 0, rays = 0;
    bool alive = true, have_pixel = false, new_path = true, first = true;
    for (;;) {
        // ---- groups whose path ended: next sample / next pixel / camera ray (main.cu:119-132)
        if (alive && new_path) {
            if (!first) { pcol = pcol + rad; ++sample; }
            first = false;
            if (have_pixel && sample >= fp.sample_end) {
                if (leader) {
                    if (fp.state_out) {   // a first or middle part of a split frame: park the pixel again
                        rt_pixel_state so;
                        so.rng[0] = pg.v0; so.rng[1] = pg.v1; so.rng[2] = pg.v2; so.rng[3] = pg.v3; so.rng[4] = pg.v4; so.rng[5] = pg.d;
                        so.col[0] = pcol.x; so.col[1] = pcol.y; so.col[2] = pcol.z;
                        so.cost = cost_before + pixel_rays;   // (bit 31, "listed", stays)
                        fp.state_out[pix] = so;
                        atomicAdd(&fp.tile_cost[(px_lrow >> 3) * fp.tiles_x + (px_i >> 3)], pixel_rays);
                    } else {
                        store_pixel(fp, px_i, px_lrow, pcol);
                    }
                    rays += pixel_rays;
                }
                have_pixel = false;
            }
            if (!have_pixel) {
                uint32_t idx = 0u;
                if (leader) idx = atomicAdd(fp.work_counter + 1, 1u);
                idx = (uint32_t)__shfl((int)idx, gbase, 64);
                if (idx >= n_items) alive = false;
                else {
                    pix = fp.heavy_pixels[first_item + idx];
                    px_lrow = (int)(pix / (uint32_t)fp.nx); px_i = (int)(pix - (uint32_t)px_lrow * (uint32_t)fp.nx);
                    px_j = local_to_global_row(fp, px_lrow);
                    const rt_pixel_state st = fp.state_in[pix];
                    if (fp.fresh) {
                        rt_xorwow_seed(pg, fp.seed_base + (uint64_t)(px_j * fp.nx + px_i));
                        pcol = mk3(0, 0, 0);
                        cost_before = st.cost & 0x80000000u;
                    } else {
                        pg.v0 = st.rng[0]; pg.v1 = st.rng[1]; pg.v2 = st.rng[2]; pg.v3 = st.rng[3]; pg.v4 = st.rng[4]; pg.d = st.rng[5];
                        pcol = mk3(st.col[0], st.col[1], st.col[2]);
                        cost_before = st.cost;
                    }
                    sample = fp.sample_begin; have_pixel = true; pixel_rays = 0;
                }
            }
            if (alive) {
                const float u = ((float)px_i + rt_xorwow_uniform(pg)) / (float)fp.nx;
                const float v = ((float)px_j + rt_xorwow_uniform(pg)) / (float)fp.ny;
                r = camera_get_ray(sd.camera, u, v, pg);
                thr = mk3(1, 1, 1); rad = mk3(0, 0, 0); depth = 0;
                new_path = false;
            }
        }
        if (__ballot(alive) == 0ull) break;
        // ---- one ray per group (main.cu:54-84)
        HitInfo h;
        trace_groups<G>(gt, sc, alive, r, h);
        if (alive) {
            ++pixel_rays; ++depth;
            if (h.prim < 0) { rad = fma3(thr, miss_color(fp, r), rad); new_path = true; }
            else {
                const HitRec rec = resolve_hit<true, NEED_UV>(sc, r, h);
                f3 emitted, attenuation;
                Ray scattered;
                const bool go_on = shade<TEX>(sc, r, rec, pg, emitted, attenuation, scattered);
                rad = fma3(thr, emitted, rad);
                if (!go_on || depth >= 50) new_path = true;
                else { thr = thr * attenuation; r = scattered; }
            }
        }
    }
    if (leader && rays) atomicAdd(fp.ray_counter, (unsigned long long)rays);
}

template <int TX, bool UV, bool LS, int G>
static hipError_t rt_launch_group_variant(const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st) {
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_group_kernel<TX, UV, LS, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((rt_group_kernel<TX, UV, LS, G>), grid, dim3(RT_GROUP_THREADS), lds, st, sd, fp);
    return hipGetLastError();
}
template <int TX, bool UV>
static hipError_t rt_launch_group_one(int lanes, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, size_t lds, hipStream_t st) {
    if (lanes == 16) return fp.group_lds_scene ? rt_launch_group_variant<TX, UV, true, 16>(sd, fp, grid, lds, st) : rt_launch_group_variant<TX, UV, false, 16>(sd, fp, grid, lds, st);
    return fp.group_lds_scene ? rt_launch_group_variant<TX, UV, true, 8>(sd, fp, grid, lds, st) : rt_launch_group_variant<TX, UV, false, 8>(sd, fp, grid, lds, st);
}
