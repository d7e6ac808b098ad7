"""Parity tests proper: the HIP render path, called through the C ABI, against the CPU oracle.

Tolerance (north_star): per-pixel max-abs error < 1e-4.  What is actually asserted is stronger: the frame is
bit-identical to the oracle, because both sides evaluate every fp32 + - * / sqrt as one IEEE operation in the
reference's order and every transcendental as the correctly rounded fp32 value (see oracle/rt_oracle.cpp header and
DESIGN.md).  Ray counts (world->hit calls, main.cu:57) must match exactly as well.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-4
KERNELS = [0, 3]   # pixel (the reference's loop nest on the reference's full tree: GPU-side cross-check), staged + tier kernel (the default)


def assert_frames_equal(got, ref, what=""):
    assert got.shape == ref.shape, what
    assert not np.isnan(got).any(), what
    err = float(np.abs(got - ref).max())
    assert err < TOL, f"{what}: max abs err {err}"
    same = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=-1)
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} pixels are not bit-identical (max err {err})"


DEFAULT_KERNEL = 3


def render(art, hs, kernel=DEFAULT_KERNEL, opts=None, **frame_kw):
    """One frame through the C ABI with the shipped defaults (rt_reset_options) plus `opts`."""
    art.reset_options()
    art.set_option("kernel", kernel)
    opts = dict(opts or {})
    if "tier_auto" not in opts and any(k.startswith(("tier1_", "heavy_", "sparse_wg")) for k in opts):
        opts["tier_auto"] = 0          # hand-set tier sizes only apply with the automatic sizing off
    for k, v in opts.items():
        art.set_option(k, v)
    ds = art.DeviceScene(hs)
    try:
        fb, st = ds.render(hs.frame(**frame_kw))
    finally:
        ds.close()
        art.reset_options()
    return fb, st


@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_fixtures(gpu, kernel):
    """Committed oracle frames (tests/golden/oracle_frames.npz), every scene, every kernel."""
    gold = np.load(os.path.join(HERE, "golden", "oracle_frames.npz"))
    for key in gold.files:
        name, nx, ny, ns = key.rsplit("_", 3)
        hs = gpu.HostScene(name, int(nx), int(ny))
        fb, _ = render(gpu, hs, kernel, ns=int(ns))
        assert_frames_equal(fb, gold[key], f"{key} kernel {kernel}")


@pytest.mark.parametrize("name,nx,ny,ns", [("two_spheres", 200, 100, 4), ("bouncing", 160, 96, 16), ("book1", 160, 96, 8),
                                           ("cornell", 96, 96, 16), ("cornell_smoke", 96, 96, 16), ("final", 80, 80, 8),
                                           ("degenerate", 32, 16, 8), ("checker", 96, 48, 8), ("earth", 96, 48, 8), ("perlin", 64, 32, 4),
                                           ("quads", 96, 48, 8), ("simple_light", 96, 48, 16), ("original", 64, 64, 8)])
def test_scene_matches_oracle(gpu, orc, name, nx, ny, ns):
    img, iw, ih = gpu.default_texture(name)
    hs = gpu.HostScene(name, nx, ny, img, iw, ih)
    ref, cnt = orc.OracleScene(name, nx, ny, img, iw, ih).render(ns)
    fb, st = render(gpu, hs, DEFAULT_KERNEL, ns=ns)
    assert st.rays == cnt["rays"], (st.rays, cnt["rays"])
    assert st.samples == nx * ny * ns
    assert_frames_equal(fb, ref, name)


@pytest.mark.parametrize("name,ny", [("bouncing", 171), ("cornell", 256), ("final", 256)])
def test_calibration_pass_counts_are_the_oracles(gpu, orc, name, ny):
    """rt_scene_create measures, per node of the reference's tree, how often its box test passes (kernel 0 with counters
    on, a 256-pixel-wide 4-spp frame through the scene's camera, seeds 1984 + pixel; collected in LDS per workgroup).  The
    same frame through the oracle must give the same counts, because the traversal is the same to the bit: checked
    through what the planner derives from them -- its box tests per ray before the collapse equal the oracle's counter,
    and the walk array it builds from the oracle's counts on the host has the size the device built."""
    nx = 256                                                                       # the calibration frame of a scene of this aspect
    img, iw, ih = gpu.default_texture(name)
    hs = gpu.HostScene(name, nx, ny, img, iw, ih)
    gpu.reset_options()
    gpu.set_option("bvh_collapse", 2)                                             # the plan over the reference's own tree
    ds = gpu.DeviceScene(hs)
    try:
        info = ds.walk_info()
    finally:
        ds.close()
        gpu.reset_options()
    o = orc.OracleScene(name, nx, ny, img, iw, ih)
    passes, rays, box_tests = o.node_passes(4, threads=8)
    assert abs(info["tests_before"] * rays - box_tests) < 0.5, (info["tests_before"] * rays, box_tests)
    walk, before, after = gpu.plan_walk_array(hs.nodes(), passes, rays)
    # (an array small enough to be scanned in lockstep -- option scan_nodes, 24 -- is reduced to its leaves after planning)
    expect = int((walk["prim"] >= 0).sum()) if len(walk) <= 24 else len(walk)
    assert expect == info["nodes_walked"], (len(walk), expect, info)
    assert abs(after - info["tests_after"]) < 1e-9 * max(1.0, after)


@pytest.mark.parametrize("name,nx,ny,ns", [("bouncing", 160, 96, 8), ("cornell", 64, 64, 8), ("cornell_smoke", 64, 64, 8), ("final", 64, 64, 4), ("degenerate", 32, 16, 8)])
def test_walk_array_is_invisible(gpu, orc, name, nx, ny, ns):
    """rt_scene_create drops the interior nodes of the reference's tree whose box test does not pay (option bvh_collapse:
    0 = none, 1 = chosen by surface area, 2 = by measured pass counts, 3 = as 2 or a regrouping of the same leaves by
    surface-area cost, whichever predicts fewer box tests).  Interior boxes never change what bvh_node::hit
    returns (bvh.cuh:95-106; see rt_abi.hip "Collapse"), so all three must give the oracle's frame and ray count, and the
    walk array must keep every leaf in the reference's order."""
    img, iw, ih = gpu.default_texture(name)
    hs = gpu.HostScene(name, nx, ny, img, iw, ih)
    ref, cnt = orc.OracleScene(name, nx, ny, img, iw, ih).render(ns)
    sizes = {}
    for mode in (0, 1, 2, 3):
        gpu.reset_options()
        gpu.set_option("bvh_collapse", mode)
        ds = gpu.DeviceScene(hs)
        try:
            info = ds.walk_info()
            gpu.reset_options()
            fb, st = ds.render(hs.frame(ns=ns))
        finally:
            ds.close()
        assert st.rays == cnt["rays"], (mode, st.rays, cnt["rays"])
        assert_frames_equal(fb, ref, f"{name} bvh_collapse={mode}")
        assert info["nodes_reference"] == hs.desc.n_nodes
        sizes[mode] = info["nodes_walked"]
        n_leaves = int((hs.nodes()["prim"] >= 0).sum())
        assert n_leaves <= info["nodes_walked"] <= info["nodes_reference"]
        if mode and info["nodes_walked"] < info["nodes_reference"]:
            # (<=: an array small enough to be scanned in lockstep is reduced to its leaves even where the planner kept every node)
            assert info["tests_after"] <= info["tests_before"]
    assert sizes[0] == hs.desc.n_nodes
    print(name, sizes)


def test_ragged_sizes_and_edge_frames(gpu, orc):
    """Frame sizes that are not multiples of the 8x8 work tiles, single rows/columns, 1 spp, gamma 1."""
    for nx, ny, ns, gamma in [(37, 19, 3, None), (1, 1, 5, None), (65, 1, 2, None), (3, 70, 1, None), (40, 24, 4, 1.0)]:
        hs = gpu.HostScene("bouncing", nx, ny)
        o = orc.OracleScene("bouncing", nx, ny)
        ref, cnt = o.render(ns, gamma=gamma)
        for kernel in KERNELS:
            fb, st = render(gpu, hs, kernel, ns=ns, gamma=gamma)
            assert st.rays == cnt["rays"]
            assert_frames_equal(fb, ref, f"{nx}x{ny}@{ns} kernel {kernel}")


def test_row_partition_is_invisible(gpu, orc):
    """Multi-GPU row tiling (8(e)): any partition reproduces the 1-GPU frame bit for bit, because the per-pixel seed is
    seed_base + global pixel index (main.cu:101-104)."""
    nx, ny, ns = 72, 50, 4
    hs = gpu.HostScene("bouncing", nx, ny)
    whole, st_whole = render(gpu, hs, DEFAULT_KERNEL, ns=ns)
    for tile_rows, world in [(4, 2), (4, 8), (8, 3), (1, 5), (64, 2)]:
        full = np.full((ny, nx, 3), np.nan, np.float32)
        rays = 0
        for r in range(world):
            f_kw = dict(ns=ns, tile_rows=tile_rows, tile_first=r, tile_stride=world)
            part, st = render(gpu, hs, DEFAULT_KERNEL, **f_kw)
            rows = gpu.local_rows_to_global(hs.frame(**f_kw))
            assert part.shape[0] == len(rows) == st.local_rows
            if len(rows):
                full[rows] = part
            rays += st.rays
        assert rays == st_whole.rays
        assert np.array_equal(full.view(np.uint32), whole.view(np.uint32)), (tile_rows, world)
    ref, _ = orc.OracleScene("bouncing", nx, ny).render(ns)
    assert_frames_equal(whole, ref, "whole frame")


def test_partitions_with_automatic_tier_sizing(gpu, orc):
    """What each rank of a 2-, 4- or 8-GPU run does: the cost-aware schedule sizes its tiers from the share of the frame
    (the emptier the machine, the more pixels get a wave of the tier kernel).  Still the whole frame's pixels, bit for bit."""
    nx, ny, ns = 240, 192, 16
    hs = gpu.HostScene("bouncing", nx, ny)
    ref, cnt = orc.OracleScene("bouncing", nx, ny).render(ns)
    opts = {"tier_auto": 1, "split_samples": 4, "presplit_samples": 2}
    for world in (1, 2, 4, 8):
        full = np.full((ny, nx, 3), np.nan, np.float32)
        rays = 0
        for r in range(world):
            f_kw = dict(ns=ns, tile_rows=4 if world > 1 else ny, tile_first=r, tile_stride=world)
            part, st = render(gpu, hs, DEFAULT_KERNEL, opts, **f_kw)
            full[gpu.local_rows_to_global(hs.frame(**f_kw))] = part
            rays += st.rays
        assert rays == cnt["rays"], world
        assert_frames_equal(full, ref, f"world {world}")


def test_empty_partition(gpu):
    hs = gpu.HostScene("two_spheres", 16, 4)
    fb, st = render(gpu, hs, DEFAULT_KERNEL, ns=1, tile_rows=4, tile_first=3, tile_stride=4)   # only one tile exists
    assert fb.shape[0] == 0 and st.rays == 0 and st.local_rows == 0


def test_scheduling_knobs_do_not_change_pixels(gpu):
    """Everything the scheduler does is re-ordering: LDS residency, trip length, thresholds, workgroup shape."""
    hs = gpu.HostScene("bouncing", 128, 80)
    base, st0 = render(gpu, hs, 0, ns=6)   # ns = 6 with split_samples <= 3 exercises the split-frame schedule
    variants = [(3, {}), (3, {"lpt": 0}), (3, {"sparse_stride": 0}), (3, {"sparse_stride": 64, "heavy_factor_x10": 10, "sparse_wg_percent": 100}),
                (3, {"sparse_stride": 2, "heavy_factor_x10": 12, "sparse_priority": 0, "sparse_eager": 1}), (3, {"tier1_pixels": 0, "sparse_stride": 16}),
                (3, {"split_samples": 1, "heavy_factor_x10": 10, "tier1_factor_x10": 10, "tier1_pixels": 4096}), (3, {"split_samples": 3, "tier1_factor_x10": 15}),
                (3, {"split_samples": 3, "presplit_samples": 1, "heavy_factor_x10": 10, "tier1_factor_x10": 12}), (3, {"split_samples": 2, "presplit_samples": 1, "tier1_pixels": 0}),
                # the tier kernel off (tier 1 empty), the cost prior off (an unranked first part), both
                (3, {"split_samples": 3, "tier_kernel": 0, "heavy_factor_x10": 10}), (3, {"split_samples": 2, "prior": 0}), (3, {"split_samples": 2, "presplit_samples": 1, "prior": 0, "tier_kernel": 0, "heavy_factor_x10": 10}),
                (3, {"split_samples": 3, "presplit_samples": 1, "heavy_factor_x10": 10, "tier1_factor_x10": 10, "tier1_pixels": 8192, "tier1_depth": 1}),
                (3, {"lds_mode": 0}), (3, {"lds_mode": 3}), (3, {"lds_mode": 1}), (3, {"steps_per_trip": 1, "shade_threshold": 1, "diel_threshold": 1, "newpath_threshold": 1}),
                (3, {"steps_per_trip": 11, "shade_threshold": 64, "diel_threshold": 64, "newpath_threshold": 64}), (3, {"threads": 256, "wg_per_cu": 3}),
                (3, {"threads": 64, "wg_per_cu": 8, "shade_threshold": 40, "newpath_threshold": 3}),
                (0, {"lds_mode": 0}), (3, {"bvh_collapse": 0}), (3, {"bvh_collapse": 1}), (3, {"bvh_collapse": 1, "leaf_threshold": 1, "steps_per_trip": 3}),
                (3, {"bvh_collapse": 0, "leaf_threshold": 64, "steps_per_trip": 20}), (3, {"leaf_threshold": 1}), (3, {"leaf_threshold": 33, "steps_per_trip": 2}),
                # tier 3 (listed pixels below the sparse threshold): on ordinary lanes, on semi workgroups, none; a second ranking
                (3, {"split_samples": 2, "heavy_factor_x10": 10, "sparse_factor_x10": 30, "semi_stride": 0}), (3, {"split_samples": 2, "heavy_factor_x10": 10, "sparse_factor_x10": 30, "semi_stride": 2}),
                (3, {"split_samples": 2, "presplit_samples": 1, "heavy_factor_x10": 12, "sparse_factor_x10": 20, "semi_stride": 4, "tier1_pixels": 0}),
                (3, {"split_samples": 1, "resplit_samples": 3, "heavy_factor_x10": 10, "sparse_factor_x10": 15}),
                (3, {"split_samples": 2, "heavy_factor_x10": 10, "sparse_factor_x10": 20, "cost_smooth_percent": 90, "tier1_pixels": 64, "tier1_factor_x10": 20, "tier1_depth": 4}),
                (3, {"split_samples": 2, "presplit_samples": 1, "heavy_factor_x10": 15, "cost_smooth_percent": 100, "sparse_work_percent": 1}), (3, {"split_samples": 3, "heavy_factor_x10": 10, "sparse_work_percent": 100}), (3, {"split_samples": 2, "resplit_samples": 3, "presplit_samples": 1, "semi_stride": 8, "heavy_factor_x10": 10})]
    for kernel, opts in variants:
        fb, st = render(gpu, hs, kernel, opts, ns=6)
        assert st.rays == st0.rays, (kernel, opts)
        assert np.array_equal(fb.view(np.uint32), base.view(np.uint32)), (kernel, opts)
    hs2 = gpu.HostScene("cornell_smoke", 48, 48)
    base2, _ = render(gpu, hs2, 0, ns=4)
    for kernel, opts in [(3, {}), (3, {"split_samples": 2, "presplit_samples": 1, "heavy_factor_x10": 10, "tier1_factor_x10": 10, "tier1_pixels": 4096}), (3, {"split_samples": 2, "tier_kernel": 0}), (3, {"lds_mode": 0, "diel_threshold": 1}), (3, {"lds_mode": 1, "box_threshold": 1, "medium_threshold": 1}), (3, {"steps_per_trip": 2, "shade_threshold": 60, "box_threshold": 64, "medium_threshold": 64}),
                         (3, {"bvh_collapse": 0}), (3, {"bvh_collapse": 1, "leaf_threshold": 1}), (3, {"bvh_collapse": 2, "leaf_threshold": 40, "steps_per_trip": 5})]:
        fb, _ = render(gpu, hs2, kernel, opts, ns=4)
        assert np.array_equal(fb.view(np.uint32), base2.view(np.uint32)), (kernel, opts)


def test_headline_frame_properties(gpu, orc):
    """BASELINE size (1200x800 random scene): size-independent checks -- determinism, kernel-independence, seed
    sensitivity, no NaN, exact ray count per sample band -- plus the oracle on a band of rows."""
    nx, ny, ns = 1200, 800, 10
    hs = gpu.HostScene("random_scene", nx, ny)
    a, st_a = render(gpu, hs, DEFAULT_KERNEL, ns=ns)
    b, st_b = render(gpu, hs, DEFAULT_KERNEL, ns=ns)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and st_a.rays == st_b.rays      # idempotent
    c, st_c = render(gpu, hs, 0, ns=ns)
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32)) and st_a.rays == st_c.rays      # scheduling-independent
    d, _ = render(gpu, hs, DEFAULT_KERNEL, ns=ns, seed_base=2024)
    assert not np.array_equal(a, d)                                                             # the seed matters
    assert not np.isnan(a).any() and (a >= 0).all()
    rps = st_a.rays / st_a.samples
    assert 2.0 < rps < 2.25, rps                     # 2.12 rays per sample on this scene
    o = orc.OracleScene("bouncing", nx, ny)
    ref, cnt = o.render(ns, row0=400, row1=408)
    assert_frames_equal(a[400:408], ref[400:408], "rows 400..407 of the headline frame")
    band, st_band = render(gpu, hs, DEFAULT_KERNEL, ns=ns, tile_rows=8, tile_first=400 // 8, tile_stride=10 ** 6)
    assert np.array_equal(band.view(np.uint32), a[400:408].view(np.uint32))
    assert st_band.rays == cnt["rays"]


def test_device_pointer_output(gpu):
    """The boundary also takes a device pointer (what bench.py and a multi-GPU caller pass)."""
    import torch
    hs = gpu.HostScene("cornell", 40, 40)
    host, st = render(gpu, hs, DEFAULT_KERNEL, ns=3)
    ds = gpu.DeviceScene(hs)
    buf = torch.zeros((40, 40, 3), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        _, st2 = ds.render(hs.frame(ns=3), out=buf.data_ptr(), stream=s.cuda_stream, blocking=False)
        st3 = ds.finish()
    s.synchronize()
    ds.close()
    assert st3.rays == st.rays and st3.ms_render > 0
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), host.view(np.uint32))


def test_malformed_scene_is_rejected_on_the_host(gpu):
    """Bad indices must come back as RT_ERR_INVALID from rt_scene_create, never reach a kernel."""
    hs = gpu.HostScene("bouncing", 32, 32)
    L = gpu.rt_lib()

    def try_create(mutate):
        desc = gpu.RtSceneDesc.from_buffer_copy(hs.desc)
        keep = mutate(desc)
        p = C.c_void_p()
        st = L.rt_scene_create(C.byref(desc), C.byref(p))
        if st == 0:
            L.rt_scene_destroy(p)
        return st, keep

    def bad_skip(desc):
        n = hs.nodes(); n["skip"][5] = 2; desc.nodes = n.ctypes.data; return n
    def bad_prim(desc):
        n = hs.nodes(); leaf = np.flatnonzero(n["prim"] >= 0)[0]; n["prim"][leaf] = 10 ** 6; desc.nodes = n.ctypes.data; return n
    def bad_mat(desc):
        s = hs.spheres(); s["mat"][3] = 99999; desc.spheres = s.ctypes.data; return s
    def bad_count(desc):
        desc.n_spheres = -1; return None
    for m in (bad_skip, bad_prim, bad_mat, bad_count):
        st, _ = try_create(m)
        assert st == 1, m.__name__          # RT_ERR_INVALID
    st, _ = try_create(lambda d: None)
    assert st == 0
    with pytest.raises(gpu.RtError):
        ds = gpu.DeviceScene(hs)
        try:
            f = hs.frame(ns=0)
            ds.render(f)
        finally:
            ds.close()


def test_smoke_entry_point(gpu):
    import __graft_entry__ as g
    g.smoke()


def test_drop_in_executable_writes_the_reference_ppm(gpu, orc, tmp_path):
    """lib/rayTracer is the drop-in for src/main.cu's main(): ASCII P3 on stdout, rows top to bottom, int(255.99*c)
    unclamped (main.cu:715-727), progress + timing on stderr (main.cu:668-669,712)."""
    import subprocess
    exe = os.path.join(gpu.LIB_DIR, "rayTracer")
    assert os.path.exists(exe)
    for name, nx, ny, ns, dbl in [("bouncing", 40, 24, 3, True), ("cornell", 32, 32, 4, False)]:
        r = subprocess.run([exe, "--scene", name, "--nx", str(nx), "--ny", str(ny), "--ns", str(ns)], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        err = r.stderr.decode()
        assert f"Rendering a {nx}x{ny} image in 8x8 blocks." in err and "took " in err and " seconds." in err
        toks = r.stdout.decode().split()
        assert toks[:4] == ["P3", str(nx), str(ny), "255"]
        got = np.array(toks[4:], np.int64).reshape(ny, nx, 3)
        ref, _ = orc.OracleScene(name, nx, ny).render(ns)
        scale = np.float64(255.99) if dbl else np.float32(255.99)
        want = (scale * ref[::-1].astype(np.float64 if dbl else np.float32)).astype(np.int64)   # top row first, truncation, no clamp
        assert np.array_equal(got, want), name
    r = subprocess.run([exe, "--scene", "nope"], capture_output=True, timeout=60)
    assert r.returncode != 0


@pytest.mark.parametrize("name,nx,ny,ns", [("bouncing", 160, 96, 16), ("cornell", 96, 96, 12), ("cornell_smoke", 96, 96, 12), ("final", 80, 80, 8),
                                           ("degenerate", 96, 64, 8), ("simple_light", 96, 64, 8)])
def test_split_frame_schedule_matches_oracle(gpu, orc, name, nx, ny, ns):
    """The cost-aware schedule (frame split at sample boundaries, pixels parked and resumed, heavy pixels on sparse waves
    and on the tier kernel's one-pixel waves: trace_wave over spheres, quads, boxes, instances and media) is scheduling
    only: with aggressive settings every kind of wave is exercised and the frame must still equal the oracle bit for bit."""
    img, iw, ih = gpu.default_texture(name)
    hs = gpu.HostScene(name, nx, ny, img, iw, ih)
    ref, cnt = orc.OracleScene(name, nx, ny, img, iw, ih).render(ns)
    for opts in ({"split_samples": 4, "heavy_factor_x10": 15, "tier1_factor_x10": 25, "tier1_pixels": 64, "sparse_stride": 8},
                 {"split_samples": 2, "heavy_factor_x10": 10, "tier1_factor_x10": 10, "tier1_pixels": 4096, "sparse_wg_percent": 100},
                 {"split_samples": 3, "heavy_factor_x10": 12, "tier1_pixels": 0, "sparse_stride": 32, "sparse_priority": 0},
                 # three parts: a ranked middle part parks every pixel again, from ordinary, sparse and single-pixel waves
                 {"split_samples": 4, "presplit_samples": 2, "heavy_factor_x10": 12, "tier1_factor_x10": 20, "tier1_pixels": 64},
                 {"split_samples": 3, "presplit_samples": 1, "heavy_factor_x10": 10, "tier1_factor_x10": 10, "tier1_pixels": 4096, "sparse_wg_percent": 100},
                 {"split_samples": 4, "presplit_samples": 3, "heavy_factor_x10": 11, "tier1_pixels": 0, "sparse_stride": 16},
                 # every pixel on the tier kernel (list from 1.0 x the mean, no cap to speak of), one pixel per wave and several
                 {"split_samples": 4, "presplit_samples": 2, "heavy_factor_x10": 10, "tier1_factor_x10": 10, "tier1_pixels": 65536, "tier1_depth": 1, "sparse_work_percent": 100},
                 {"split_samples": 2, "heavy_factor_x10": 10, "tier1_factor_x10": 10, "tier1_pixels": 65536, "tier1_depth": 16, "sparse_work_percent": 100, "threads": 256, "wg_per_cu": 3},
                 # without the cost prior (unranked first part), without the tier kernel
                 {"split_samples": 4, "presplit_samples": 2, "prior": 0, "heavy_factor_x10": 12, "tier1_factor_x10": 15, "tier1_pixels": 4096},
                 {"split_samples": 4, "tier_kernel": 0, "heavy_factor_x10": 12}):
        fb, st = render(gpu, hs, 3, opts, ns=ns)
        assert st.rays == cnt["rays"], (opts, st.rays, cnt["rays"])
        assert_frames_equal(fb, ref, f"{name} {opts}")


@pytest.mark.parametrize("name,nx,ny,ns", [("bouncing", 160, 96, 16), ("cornell", 96, 96, 12), ("cornell_smoke", 96, 96, 12), ("final", 80, 80, 8),
                                           ("simple_light", 96, 64, 8)])
def test_tail_handoff_matches_oracle(gpu, orc, name, nx, ny, ns):
    """The tail hand-off (rt_device.h: once few pixels are in flight the main kernel parks them at their next sample boundary and a
    launch of the tier kernel after it finishes them, one per wave) is scheduling only.  Thresholds from "never" to "every pixel,
    after its first sample", polls from every microsecond to never within the frame, with and without tiers, on a row band: the
    frame must equal the oracle bit for bit and count the same rays.  The general scenes go through trace_wave's shared-out box
    faces (Cornell box: two rotated boxes; final: the ground), the lockstep-scanned ones (lds_mode 4) through handoff_scan."""
    img, iw, ih = gpu.default_texture(name)
    hs = gpu.HostScene(name, nx, ny, img, iw, ih)
    ref, cnt = orc.OracleScene(name, nx, ny, img, iw, ih).render(ns)
    L = gpu.rt_lib()
    L.rt_debug_handoff.argtypes = [C.c_void_p, C.c_void_p]
    handed = {}
    for tag, opts in (("off", {"handoff": 0}), ("auto", {}), ("never", {"handoff_pixels": 0}),
                      ("all", {"handoff_pixels": 1 << 24, "handoff_poll_us": 1, "split_samples": 4, "presplit_samples": 2}),
                      ("all, no tiers", {"handoff_pixels": 1 << 24, "handoff_poll_us": 1, "split_samples": 2, "tier_kernel": 0}),
                      ("all, no prior", {"handoff_pixels": 1 << 24, "handoff_poll_us": 1, "split_samples": 4, "prior": 0}),
                      ("some", {"handoff_pixels": nx * ny // 4, "handoff_poll_us": 5, "split_samples": 3, "heavy_factor_x10": 12, "tier1_factor_x10": 20, "tier1_pixels": 64}),
                      ("no scan", {"handoff_scan": 0})):
        gpu.reset_options()
        for k, v in opts.items():
            gpu.set_option(k, v)
        if any(k.startswith(("tier1_", "heavy_")) for k in opts):
            gpu.set_option("tier_auto", 0)
        ds = gpu.DeviceScene(hs)
        try:
            fb, st = ds.render(hs.frame(ns=ns))
            h = np.zeros(2, np.uint64)
            assert L.rt_debug_handoff(ds._p, h.ctypes.data) == 0
            handed[tag] = int(h[0])
        finally:
            ds.close()
            gpu.reset_options()
        assert st.rays == cnt["rays"], (tag, st.rays, cnt["rays"])
        assert_frames_equal(fb, ref, f"{name} hand-off {tag}")
    assert handed["off"] == 0 and handed["never"] == 0
    assert handed["all"] > 0, handed      # the path under test did run
    # a row band (what one rank of a multi-GPU run renders) with everything handed off
    gpu.reset_options()
    gpu.set_option("handoff_pixels", 1 << 24); gpu.set_option("handoff_poll_us", 1); gpu.set_option("split_samples", 4)
    ds = gpu.DeviceScene(hs)
    try:
        f = hs.frame(ns=ns, tile_rows=4, tile_first=1, tile_stride=3)
        fb, st = ds.render(f)
    finally:
        ds.close()
        gpu.reset_options()
    rows = gpu.local_rows_to_global(f)
    assert len(rows) == st.local_rows > 0
    assert_frames_equal(fb, ref[rows], f"{name} hand-off on a row band")


@pytest.mark.parametrize("name,nx,ny", [("bouncing", 96, 64), ("cornell_smoke", 64, 64), ("final", 48, 48)])
def test_progressive_windows_equal_one_shot(gpu, orc, name, nx, ny):
    """rt_render_window (progressive accumulation on the carried per-pixel RNG state, main.cu:126): windows
    [0, 5), [5, 12), [12, 24) leave after each window the frame a one-shot render of that many samples gives, bit for bit --
    and the last one equals the oracle; ray counts add up; a window out of sequence is refused."""
    img, iw, ih = gpu.default_texture(name)
    hs = gpu.HostScene(name, nx, ny, img, iw, ih)
    gpu.reset_options()
    ds = gpu.DeviceScene(hs)
    try:
        ref, cnt = orc.OracleScene(name, nx, ny, img, iw, ih).render(24)
        prog = ds.progressive(hs.frame(ns=24))
        rays = 0
        for begin, end in ((0, 5), (5, 12), (12, 24)):
            fb, st = prog.render(begin, end)
            rays += st.rays
            one, _ = ds.render(hs.frame(ns=end))
            assert np.array_equal(fb.view(np.uint32), one.view(np.uint32)), (name, begin, end)
        assert rays == cnt["rays"]
        assert_frames_equal(fb, ref, f"{name} progressive")
        with pytest.raises(gpu.RtError):
            prog.render(30, 40)
        with pytest.raises(gpu.RtError):
            prog.render(24, 24)
        prog.close()
    finally:
        ds.close()


def test_tail_handoff_queue_cannot_overflow(gpu):
    """The hand-off queue holds one entry per resident lane.  With the threshold set absurdly high on a frame of more pixels than
    lanes (960 000 against 262 144), waves that turn ordinary again fetch and hand off over and over: lanes that find the queue
    full must keep their pixels.  The frame equals the one rendered without a hand-off, bit for bit, and counts the same rays."""
    nx, ny, ns = 1200, 800, 64
    hs = gpu.HostScene("random_scene", nx, ny)
    base, st0 = render(gpu, hs, DEFAULT_KERNEL, {"handoff": 0}, ns=ns)
    L = gpu.rt_lib()
    L.rt_debug_handoff.argtypes = [C.c_void_p, C.c_void_p]
    for opts in ({"handoff_pixels": 1 << 24, "handoff_poll_us": 1}, {"handoff_pixels": 1 << 24, "handoff_poll_us": 1, "sparse_wg_percent": 100, "heavy_factor_x10": 10, "tier_auto": 0}):
        gpu.reset_options()
        for k, v in opts.items():
            gpu.set_option(k, v)
        ds = gpu.DeviceScene(hs)
        try:
            fb, st = ds.render(hs.frame(ns=ns))
            h = np.zeros(2, np.uint64)
            assert L.rt_debug_handoff(ds._p, h.ctypes.data) == 0
        finally:
            ds.close()
            gpu.reset_options()
        assert st.rays == st0.rays, (opts, st.rays, st0.rays)
        assert np.array_equal(fb.view(np.uint32), base.view(np.uint32)), opts
        assert int(h[0]) > 0, opts
