// rt_kernel_wavefront.hip -- kernel 4 ("wavefront"), experimental; see below.
#include "rt_device_funcs.h"

// =============================================================================
// Kernel E ("wavefront"): bulk-synchronous wavefront path tracing inside one workgroup.
//
// Stage counters for kernel D (profiles/r01_diag_staged_stage_counts.txt): box
// steps are ~55 % of all issued instructions and run with 22.5 of 64 lanes,
// because a lane owns one pixel and sits idle while its ray waits for a stage.
// Here no lane owns anything.  The workgroup keeps P ray slots in LDS (ray,
// 1/d, traversal result, path throughput/radiance, and the slot's pixel:
// XORWOW state, colour sum, sample count) and compact lists of slot ids.  An
// iteration is three phases separated by __syncthreads():
//   T  lanes pop slots from READY (in a strided order, so that one wave does
//      not get 64 neighbouring pixels), walk the BVH (node steps + parked leaf
//      pass as in kernel C), push hits to DONE and misses to NEWPATH, and pop
//      again; when READY is empty and a wave runs low on walking lanes it
//      saves its rays' walk state and re-queues them for the next iteration;
//   C  lanes pop DONE slots: light -> NEWPATH; lambertian / metal / isotropic /
//      dielectric scatter -> READY(next) (NEWPATH when absorbed or at depth 50);
//   E  lanes pop NEWPATH slots: add the background for misses, accumulate,
//      next sample or next pixel, camera ray -> READY(next).
// Within a phase every list is either consumed (through an atomic cursor) or
// appended to (through an atomic count), never both, counters are reset in a
// phase that does not touch them, and a slot is touched only by the lane that
// popped it: no spin waits, nothing to deadlock on.  As in kernels B-D this
// only re-orders work: a slot's pixel draws its own XORWOW stream in the
// reference's order, so frames are bit-identical.
//
// Spheres-only scenes with inline / solid / checker textures (the headline
// random scene); other scenes use kernel D.
// =============================================================================
namespace {

enum { F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_TM, F_BEST_T, F_IX, F_IY, F_IZ, F_THR, F_RAD = F_THR + 3, F_COL = F_RAD + 3, F_COUNT = F_COL + 3 };
enum { I_NODE, I_BEST_PRIM, I_BOUNCE, I_SAMPLE, I_PX, I_RNG, I_COUNT = I_RNG + 6 };
enum { C_READY_POS, C_READY_COUNT0, C_READY_COUNT1, C_DONE_COUNT, C_DONE_POS, C_NEW_COUNT0, C_NEW_COUNT1, C_NEW_POS0, C_NEW_POS1, C_WALKING, C_NUM };

// Reserve one list entry for every lane with `want` set: one LDS atomic per wave.  Must be called with all
// 64 lanes active.  Returns this lane's index (valid only where want).
DEV int wave_reserve(unsigned int* counter, bool want) {
    const unsigned long long m = __ballot(want);
    const int cnt = __popcll(m);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && cnt) base = (int)atomicAdd(counter, (unsigned int)cnt);
    base = __shfl(base, 0, 64);
    const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    return base + below;
}

}  // namespace

#ifdef RT_DIAG
#define WF_STAMP(slot_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); wf_t[slot_] += now_ - wf_mark; wf_mark = now_; } while (0)
#else
#define WF_STAMP(slot_) do { } while (0)
#endif

template <int TEX, int LDS_MODE>
__global__ void __launch_bounds__(1024) rt_render_wavefront_kernel(rt_scene_dev sd, rt_frame_params fp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const SceneView sc = stage_scene<LDS_MODE>(sd, lds);
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const int n_nodes = sc.n_nodes;
    const float tmin = 0.001f;
    const int P = fp.wf_slots;
    const int lane = threadIdx.x & 63;

    // ---- carve the slot pool (structure of arrays) and the lists out of LDS behind the staged scene
    size_t off = 0;
    if (LDS_MODE >= 1) off += (size_t)sd.n_nodes * sizeof(rt_node);
    if (LDS_MODE >= 2) off += (size_t)sd.n_spheres * sizeof(rt_sphere);
    off = (off + 15) & ~(size_t)15;
    float* const slot_f = reinterpret_cast<float*>(lds + off); off += (size_t)P * 4 * F_COUNT;
    int32_t* const slot_i = reinterpret_cast<int32_t*>(lds + off); off += (size_t)P * 4 * I_COUNT;
#define SF(field, slot_) slot_f[(field) * P + (slot_)]
#define SI(field, slot_) slot_i[(field) * P + (slot_)]
    uint16_t* const ready_a = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const ready_b = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const done = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const new_a = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    uint16_t* const new_b = reinterpret_cast<uint16_t*>(lds + off); off += (size_t)P * 2;
    off = (off + 15) & ~(size_t)15;
    unsigned int* const ctr = reinterpret_cast<unsigned int*>(lds + off);

    // every slot starts by asking for a pixel (sample = -1 marks "no pixel yet")
    for (int s = threadIdx.x; s < P; s += blockDim.x) { new_a[s] = (uint16_t)s; SI(I_SAMPLE, s) = -1; }
    if (threadIdx.x < C_NUM) ctr[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) ctr[C_NEW_COUNT0] = (unsigned int)P;
    __syncthreads();

    unsigned int rays = 0;
#ifdef RT_DIAG
    unsigned long long wf_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // T, barrier, C, barrier, E, barrier
    unsigned long long wf_mark = __builtin_amdgcn_s_memtime();
#endif
    // Every iteration retires at least one ray segment per live slot or advances a paused walk, so the frame needs
    // far fewer iterations than the cap; hitting it means a scheduling bug, reported through the error flag
    // (ray_counter[31]) instead of hanging the GPU.
    for (unsigned int iteration = 0;; ++iteration) {
        if (iteration >= fp.wf_max_iterations) { if (threadIdx.x == 0) atomicAdd(fp.ray_counter + 31, 1ull); break; }
        const int cur = (int)(iteration & 1u);
        const int n_ready = (int)ctr[C_READY_COUNT0 + cur];     // written before the previous iteration's last barrier
        if (iteration > 0 && n_ready == 0) break;                // every live slot is in READY at this point
        unsigned int* const ready_count_next = &ctr[C_READY_COUNT0 + (cur ^ 1)];
        uint16_t* const ready_cur = cur ? ready_b : ready_a;
        uint16_t* const ready_next = cur ? ready_a : ready_b;
        uint16_t* const newpath = cur ? new_b : new_a;           // NEWPATH is double-buffered by iteration parity
        unsigned int* const new_count = &ctr[C_NEW_COUNT0 + cur];
        unsigned int* const new_pos = &ctr[C_NEW_POS0 + cur];

        // =============================== T phase ===============================
        if (threadIdx.x == 0) { ctr[C_NEW_COUNT0 + (cur ^ 1)] = 0; ctr[C_NEW_POS0 + (cur ^ 1)] = 0; }   // last iteration's list is dead
        if (n_ready > 0) {
            // stride for the pop order: a prime that does not divide n_ready, so idx -> (idx * stride) % n_ready is a bijection
            const int stride = (n_ready % 37) ? 37 : ((n_ready % 41) ? 41 : 43);
            int slot = -1, node = n_nodes;
            int32_t parked = -1;
            f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(1, 1, 1);
            float tm = 0.f;
            HitInfo best; best.t = FLT_MAX; best.prim = -1; best.inst = -1;
            bool finite_inv = true, exhausted = false;
            // Re-queueing is only worth it (and only safe against livelock) when plenty of rays are around and this
            // wave has advanced its rays at least one trip since it last looked.
            const bool may_pause = 2 * n_ready >= (int)blockDim.x;
            int trips = 0;
            for (;;) {
                // -- finished walks: hits -> DONE, misses -> NEWPATH (they only need the background, added in E)
                {
                    const bool fin = slot >= 0 && node >= n_nodes;
                    if (__ballot(fin) != 0ull) {
                        const bool hit = fin && best.prim >= 0, miss = fin && best.prim < 0;
                        const int idx = wave_reserve(&ctr[C_DONE_COUNT], hit);
                        const int idx2 = wave_reserve(new_count, miss);
                        if (hit) { SF(F_BEST_T, slot) = best.t; SI(I_BEST_PRIM, slot) = best.prim; done[idx] = (uint16_t)slot; }
                        if (miss) { SI(I_BEST_PRIM, slot) = -2; newpath[idx2] = (uint16_t)slot; }   // -2: "missed, background pending"
                        if (fin) { slot = -1; ++rays; }      // one finished world->hit call (main.cu:57)
                        if (lane == 0) atomicSub(&ctr[C_WALKING], (unsigned int)__popcll(__ballot(fin)));
                    }
                }
                // -- refill idle lanes from READY
                if (!exhausted) {
                    const bool want = slot < 0;
                    const unsigned long long wm = __ballot(want);
                    if (wm != 0ull) {
                        const int idx = wave_reserve(&ctr[C_READY_POS], want);
                        const int first_idx = __shfl(idx, __ffsll((long long)wm) - 1, 64);
                        if (first_idx + __popcll(wm) >= n_ready) exhausted = true;
                        const unsigned long long got = __ballot(want && idx < n_ready);
                        if (lane == 0 && got != 0ull) atomicAdd(&ctr[C_WALKING], (unsigned int)__popcll(got));
                        if (want && idx < n_ready) {
                            slot = ready_cur[(idx * stride) % n_ready];
                            o = mk3(SF(F_OX, slot), SF(F_OY, slot), SF(F_OZ, slot));
                            d = mk3(SF(F_DX, slot), SF(F_DY, slot), SF(F_DZ, slot));
                            tm = SF(F_TM, slot);
                            node = SI(I_NODE, slot);
                            best.t = SF(F_BEST_T, slot); best.prim = SI(I_BEST_PRIM, slot);
                            inv = mk3(SF(F_IX, slot), SF(F_IY, slot), SF(F_IZ, slot));   // 1/d, stored by whoever made the ray
                            finite_inv = inv_is_finite(inv);
                            parked = -1;
                        }
                    }
                }
                const unsigned long long active = __ballot(slot >= 0);
                if (active == 0ull) break;
                // Once READY is empty the whole workgroup should stop walking at about the same time (the barrier waits
                    // for the slowest wave), so the test is on the workgroup's walking-lane count, not this wave's.
                const int wg_walking = (int)*reinterpret_cast<volatile unsigned int*>(&ctr[C_WALKING]);
                if (exhausted && may_pause && trips > 0 && wg_walking * 64 < fp.wf_pause_lanes * (int)blockDim.x) {
                    // -- READY is empty and the workgroup is running dry: save the walks, re-queue them
                    const bool have = slot >= 0;
                    if (lane == 0) atomicSub(&ctr[C_WALKING], (unsigned int)__popcll(active));
                    const int idx = wave_reserve(ready_count_next, have);
                    if (have) {
                        SI(I_NODE, slot) = node;
                        SF(F_BEST_T, slot) = best.t; SI(I_BEST_PRIM, slot) = best.prim;
                        ready_next[idx] = (uint16_t)slot;
                    }
                    break;
                }
                // -- node steps
                ++trips;
                Ray cur_ray; cur_ray.o = o; cur_ray.d = d; cur_ray.tm = tm;
                if (__ballot(!finite_inv && slot >= 0) == 0ull) {
                    for (int step = 0; step < fp.steps_per_trip; ++step) {
                        if ((unsigned)node < (unsigned)n_nodes) {
                            const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                            const bool pass = slab_test_finite(a, b, o, inv, tmin, best.t);
                            const int32_t prim = __float_as_int(b.w);
                            const int skip = RT_NODE_SKIP(__float_as_int(a.w));
                            const bool at_leaf = pass && prim >= 0;
                            const int next = (pass && prim < 0) ? node + 1 : skip;
                            parked = at_leaf ? prim : parked;
                            node = at_leaf ? ~next : next;
                        }
                    }
                } else {
                    for (int step = 0; step < fp.steps_per_trip; ++step) {
                        if ((unsigned)node < (unsigned)n_nodes) {
                            const float4 a = nodes4[2 * node], b = nodes4[2 * node + 1];
                            const bool pass = slab_test(a, b, o, inv, tmin, best.t);
                            const int32_t prim = __float_as_int(b.w);
                            const int skip = RT_NODE_SKIP(__float_as_int(a.w));
                            const bool at_leaf = pass && prim >= 0;
                            const int next = (pass && prim < 0) ? node + 1 : skip;
                            parked = at_leaf ? prim : parked;
                            node = at_leaf ? ~next : next;
                        }
                    }
                }
                // -- leaf pass
                if (node < 0) {
                    leaf_test<true>(sc, parked, cur_ray, tmin, best);
                    parked = -1;
                    node = ~node;
                }
            }
        }
        WF_STAMP(0);
        __syncthreads();
        WF_STAMP(1);

        // =============================== C phase ===============================
        if (threadIdx.x == 0) { ctr[C_READY_POS] = 0; ctr[C_READY_COUNT0 + cur] = 0; }   // READY(cur) was consumed in T
        {
            const int n_done = (int)ctr[C_DONE_COUNT];
            for (;;) {
                int base = 0;
                if (lane == 0) base = (int)atomicAdd(&ctr[C_DONE_POS], 64u);
                base = __shfl(base, 0, 64);
                if (base >= n_done) break;
                const bool have = base + lane < n_done;
                const int slot = have ? (int)done[base + lane] : 0;
                int dest = 0;   // 1 -> READY(next), 2 -> NEWPATH
                if (have) {
                    Ray r;
                    r.o = mk3(SF(F_OX, slot), SF(F_OY, slot), SF(F_OZ, slot));
                    r.d = mk3(SF(F_DX, slot), SF(F_DY, slot), SF(F_DZ, slot));
                    r.tm = SF(F_TM, slot);
                    HitInfo h; h.t = SF(F_BEST_T, slot); h.prim = SI(I_BEST_PRIM, slot); h.inst = -1;
                    const f3 thr = mk3(SF(F_THR, slot), SF(F_THR + 1, slot), SF(F_THR + 2, slot));
                    const HitRec rec = resolve_hit<true, false>(sc, r, h);
                    const rt_material m = sc.materials[rec.mat];
                    if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
                        const f3 emitted = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                        // main.cu:71; scatter() is false
                        SF(F_RAD, slot) = fmaf(thr.x, emitted.x, SF(F_RAD, slot)); SF(F_RAD + 1, slot) = fmaf(thr.y, emitted.y, SF(F_RAD + 1, slot));
                        SF(F_RAD + 2, slot) = fmaf(thr.z, emitted.z, SF(F_RAD + 2, slot));
                        dest = 2;
                    } else {
                        rt_xorwow g;
                        g.v0 = (uint32_t)SI(I_RNG, slot); g.v1 = (uint32_t)SI(I_RNG + 1, slot); g.v2 = (uint32_t)SI(I_RNG + 2, slot);
                        g.v3 = (uint32_t)SI(I_RNG + 3, slot); g.v4 = (uint32_t)SI(I_RNG + 4, slot); g.d = (uint32_t)SI(I_RNG + 5, slot);
                        f3 dir, attenuation;
                        bool go_on = true;
                        if (m.kind == RT_MAT_DIELECTRIC) {                  // material.cuh:119-159
                            dir = dielectric_direction(r.d, rec.n, m.ior, g);
                            attenuation = mk3(1.0f, 1.0f, 1.0f);
                        } else {
                            const f3 rs = random_in_unit_sphere(g);         // shared by lambertian / metal / isotropic
                            if (m.kind == RT_MAT_METAL) {
                                const f3 reflected = reflect(unit_vector(r.d), rec.n);
                                dir = fma3(m.fuzz, rs, reflected);
                                attenuation = ld3(m.albedo);
                                go_on = dot(dir, rec.n) > 0.0f;
                            } else {
                                if (m.kind == RT_MAT_LAMBERTIAN) {
                                    const f3 target = (rec.p + rec.n) + rs;
                                    dir = target - rec.p;
                                } else {
                                    dir = rs;
                                }
                                attenuation = (TEX > 0 && m.tex >= 0) ? texture_value<TEX>(sc, m.tex, rec.u, rec.v, rec.p) : ld3(m.albedo);
                            }
                        }
                        SI(I_RNG, slot) = (int32_t)g.v0; SI(I_RNG + 1, slot) = (int32_t)g.v1; SI(I_RNG + 2, slot) = (int32_t)g.v2;
                        SI(I_RNG + 3, slot) = (int32_t)g.v3; SI(I_RNG + 4, slot) = (int32_t)g.v4; SI(I_RNG + 5, slot) = (int32_t)g.d;
                        const int bounce = SI(I_BOUNCE, slot) + 1;
                        SI(I_BOUNCE, slot) = bounce;
                        if (!go_on || bounce >= 50) dest = 2;               // main.cu:54,76-80
                        else {
                            const f3 t2 = thr * attenuation;
                            SF(F_THR, slot) = t2.x; SF(F_THR + 1, slot) = t2.y; SF(F_THR + 2, slot) = t2.z;
                            SF(F_OX, slot) = rec.p.x; SF(F_OY, slot) = rec.p.y; SF(F_OZ, slot) = rec.p.z;
                            SF(F_DX, slot) = dir.x; SF(F_DY, slot) = dir.y; SF(F_DZ, slot) = dir.z;
                            SF(F_IX, slot) = 1.0f / dir.x; SF(F_IY, slot) = 1.0f / dir.y; SF(F_IZ, slot) = 1.0f / dir.z;
                            SI(I_NODE, slot) = 0; SF(F_BEST_T, slot) = FLT_MAX; SI(I_BEST_PRIM, slot) = -1;
                            dest = 1;
                        }
                    }
                }
                { const int idx = wave_reserve(ready_count_next, dest == 1); if (dest == 1) ready_next[idx] = (uint16_t)slot; }
                { const int idx = wave_reserve(new_count, dest == 2); if (dest == 2) newpath[idx] = (uint16_t)slot; }
            }
        }
        WF_STAMP(2);
        __syncthreads();
        WF_STAMP(3);

        // =============================== E phase ===============================
        if (threadIdx.x == 0) { ctr[C_DONE_COUNT] = 0; ctr[C_DONE_POS] = 0; }   // DONE was consumed in C
        {
            const int n_new = (int)*new_count;
            for (;;) {
                int base = 0;
                if (lane == 0) base = (int)atomicAdd(new_pos, 64u);
                base = __shfl(base, 0, 64);
                if (base >= n_new) break;
                const bool have = base + lane < n_new;
                const int slot = have ? (int)newpath[base + lane] : 0;
                bool alive = false;
                if (have) {
                    alive = true;
                    int sample = SI(I_SAMPLE, slot);
                    int px = SI(I_PX, slot);
                    f3 col = mk3(0, 0, 0);
                    rt_xorwow g = {0, 0, 0, 0, 0, 0};
                    bool have_pixel = sample >= 0;
                    if (have_pixel) {
                        f3 rad = mk3(SF(F_RAD, slot), SF(F_RAD + 1, slot), SF(F_RAD + 2, slot));
                        if (SI(I_BEST_PRIM, slot) == -2) {                                      // the path ended on a miss (main.cu:57-68)
                            Ray mr; mr.o = mk3(0, 0, 0); mr.tm = 0.f;
                            mr.d = mk3(SF(F_DX, slot), SF(F_DY, slot), SF(F_DZ, slot));
                            const f3 thr = mk3(SF(F_THR, slot), SF(F_THR + 1, slot), SF(F_THR + 2, slot));
                            rad = fma3(thr, miss_color(fp, mr), rad);                                 // radiance += throughput * bg
                        }
                        col = mk3(SF(F_COL, slot) + rad.x, SF(F_COL + 1, slot) + rad.y, SF(F_COL + 2, slot) + rad.z);   // col += color(...), main.cu:124
                        ++sample;
                        g.v0 = (uint32_t)SI(I_RNG, slot); g.v1 = (uint32_t)SI(I_RNG + 1, slot); g.v2 = (uint32_t)SI(I_RNG + 2, slot);
                        g.v3 = (uint32_t)SI(I_RNG + 3, slot); g.v4 = (uint32_t)SI(I_RNG + 4, slot); g.d = (uint32_t)SI(I_RNG + 5, slot);
                        if (sample >= fp.ns) { store_pixel(fp, px & 0xFFFF, (int)((unsigned)px >> 16), col); have_pixel = false; }
                    }
                    int px_i = px & 0xFFFF, px_lrow = (int)((unsigned)px >> 16);
                    while (!have_pixel && alive) {
                        const uint32_t w = atomicAdd(fp.work_counter, 1u);
                        if (w >= fp.work_items) { alive = false; break; }
                        if (work_to_pixel(fp, w, px_i, px_lrow)) {
                            const int px_j = local_to_global_row(fp, px_lrow);
                            rt_xorwow_seed(g, fp.seed_base + (uint64_t)(px_j * fp.nx + px_i));
                            col = mk3(0, 0, 0); sample = 0; have_pixel = true;
                            px = px_i | (px_lrow << 16);
                        }
                    }
                    if (alive) {
                        const int px_j = local_to_global_row(fp, px_lrow);
                        const float u = ((float)px_i + rt_xorwow_uniform(g)) / (float)fp.nx;
                        const float v = ((float)px_j + rt_xorwow_uniform(g)) / (float)fp.ny;
                        const Ray r = camera_get_ray(sd.camera, u, v, g);
                        SF(F_OX, slot) = r.o.x; SF(F_OY, slot) = r.o.y; SF(F_OZ, slot) = r.o.z;
                        SF(F_DX, slot) = r.d.x; SF(F_DY, slot) = r.d.y; SF(F_DZ, slot) = r.d.z;
                        SF(F_IX, slot) = 1.0f / r.d.x; SF(F_IY, slot) = 1.0f / r.d.y; SF(F_IZ, slot) = 1.0f / r.d.z;
                        SF(F_TM, slot) = r.tm;
                        SF(F_THR, slot) = 1.f; SF(F_THR + 1, slot) = 1.f; SF(F_THR + 2, slot) = 1.f;
                        SF(F_RAD, slot) = 0.f; SF(F_RAD + 1, slot) = 0.f; SF(F_RAD + 2, slot) = 0.f;
                        SF(F_COL, slot) = col.x; SF(F_COL + 1, slot) = col.y; SF(F_COL + 2, slot) = col.z;
                        SI(I_BOUNCE, slot) = 0; SI(I_SAMPLE, slot) = sample; SI(I_PX, slot) = px;
                        SI(I_RNG, slot) = (int32_t)g.v0; SI(I_RNG + 1, slot) = (int32_t)g.v1; SI(I_RNG + 2, slot) = (int32_t)g.v2;
                        SI(I_RNG + 3, slot) = (int32_t)g.v3; SI(I_RNG + 4, slot) = (int32_t)g.v4; SI(I_RNG + 5, slot) = (int32_t)g.d;
                        SI(I_NODE, slot) = 0; SF(F_BEST_T, slot) = FLT_MAX; SI(I_BEST_PRIM, slot) = -1;
                    }
                }
                { const int idx = wave_reserve(ready_count_next, alive); if (alive) ready_next[idx] = (uint16_t)slot; }
            }
        }
        WF_STAMP(4);
        __syncthreads();
        WF_STAMP(5);
    }
    unsigned long long r64 = rays;
    for (int off2 = 32; off2 > 0; off2 >>= 1) r64 += __shfl_down(r64, off2, 64);
    if (lane == 0 && r64) atomicAdd(fp.ray_counter, r64);
#ifdef RT_DIAG
    if (lane == 0) { for (int q = 0; q < 8; ++q) atomicAdd(fp.ray_counter + 1 + q, wf_t[q]); atomicAdd(fp.ray_counter + 9, 1ull); }
#endif
#undef SF
#undef SI
}

hipError_t rt_launch_wavefront(int lds_mode, int tex_level, const rt_scene_dev& sd, const rt_frame_params& fp, dim3 grid, dim3 block,
                               size_t lds_bytes, hipStream_t st) {
#define WF_LAUNCH(TX, LM)                                                                                              \
    do {                                                                                                               \
        const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_render_wavefront_kernel<TX, LM>),  \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);         \
        if (e_ != hipSuccess) return e_;                                                                               \
        hipLaunchKernelGGL((rt_render_wavefront_kernel<TX, LM>), grid, block, lds_bytes, st, sd, fp);                  \
        return hipGetLastError();                                                                                      \
    } while (0)
    if (tex_level == 0) { if (lds_mode == 2) WF_LAUNCH(0, 2); if (lds_mode == 1) WF_LAUNCH(0, 1); WF_LAUNCH(0, 0); }
    if (lds_mode == 2) WF_LAUNCH(1, 2);
    if (lds_mode == 1) WF_LAUNCH(1, 1);
    WF_LAUNCH(1, 0);
#undef WF_LAUNCH
}
