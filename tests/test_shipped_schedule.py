"""The configuration that ships, at the sizes BASELINE.json quotes, against the oracle.

Every test here runs with the library's default options (rt_reset_options: the cost-aware schedule -- samples
[0, 16) ranked on the cost prior, [16, ns) on measured costs --, automatic tier sizing, device-side ranking, tail hand-off) -- exactly what bench.py times -- and compares fp32
bits with the CPU oracle on row bands (a whole 1200x800 @ 500 frame is 1.0e9 rays; a band of 8 rows is ~1e7, seconds
on the host), plus the ray counts of those bands.  Reference path: render_init + render, src/main.cu:96-133.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def render_default(art, ds, hs, **frame_kw):
    art.reset_options()
    return ds.render(hs.frame(**frame_kw))


def render_partitioned(art, ds, hs, world, tile_rows=4, **frame_kw):
    """What the ranks of a `world`-GPU run render (bench.py: 4-row tiles dealt round-robin), one after the other on
    this GPU, assembled into the reference's frame layout."""
    ny, nx = frame_kw.get("ny", hs.ny), frame_kw.get("nx", hs.nx)
    full = np.full((ny, nx, 3), np.nan, np.float32)
    rays = 0
    for r in range(world):
        kw = dict(frame_kw, tile_rows=tile_rows, tile_first=r, tile_stride=world)
        part, st = render_default(art, ds, hs, **kw)
        rows = art.local_rows_to_global(hs.frame(**kw))
        assert part.shape[0] == len(rows) == st.local_rows
        full[rows] = part
        rays += st.rays
    return full, rays


# the dearest rows of the headline frame (408..415: glass sphere, ~23 rays per sample in the worst pixel), the bottom
# and top bands (ground / sky), and one through the small spheres
HEADLINE_BANDS = [0, 200, 408, 792]


def test_bench_configuration_matches_oracle(gpu, orc):
    """bench.py's exact workload: random_scene 1200x800 @ 500 spp, default options, whole frame in one call."""
    nx, ny, ns = 1200, 800, 500
    hs = gpu.HostScene("random_scene", nx, ny)
    ds = gpu.DeviceScene(hs)
    try:
        fb, st = render_default(gpu, ds, hs, ns=ns)
        again, st2 = render_default(gpu, ds, hs, ns=ns)
        assert st.samples == nx * ny * ns
        assert st.rays == st2.rays and bits_equal(fb, again)                     # idempotent
        assert st.kernel_variant // 1000 == 3                                      # the staged kernel ran
        o = orc.OracleScene("bouncing", nx, ny)
        for row0 in HEADLINE_BANDS:
            ref, cnt = o.render(ns, row0=row0, row1=row0 + 8)
            assert bits_equal(fb[row0:row0 + 8], ref[row0:row0 + 8]), f"rows {row0}..{row0 + 7} differ from the oracle"
            band, st_band = render_default(gpu, ds, hs, ns=ns, tile_rows=8, tile_first=row0 // 8, tile_stride=10 ** 6)
            assert st_band.rays == cnt["rays"], (row0, st_band.rays, cnt["rays"])
            assert bits_equal(band, ref[row0:row0 + 8])
    finally:
        ds.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rank_shares_of_the_bench_frame(gpu, orc, world):
    """Each rank's share of the 1200x800 @ 500 frame in a 2-, 4- and 8-GPU run (full-size frame, so the tiers the default
    sizing uses -- 1, 2, 3 -- are populated; tier 0 is covered by the hand-set cases of test_gpu_parity.py), default options: assembled frame == oracle on the bands, total rays == whole-frame rays."""
    nx, ny, ns = 1200, 800, 500
    hs = gpu.HostScene("random_scene", nx, ny)
    ds = gpu.DeviceScene(hs)
    try:
        whole, st = render_default(gpu, ds, hs, ns=ns)
        full, rays = render_partitioned(gpu, ds, hs, world, ns=ns)
        assert rays == st.rays
        assert bits_equal(full, whole)
        o = orc.OracleScene("bouncing", nx, ny)
        for row0 in (408, 792):
            ref, _ = o.render(ns, row0=row0, row1=row0 + 8)
            assert bits_equal(full[row0:row0 + 8], ref[row0:row0 + 8]), (world, row0)
    finally:
        ds.close()


def test_config4_1920x1080_row_tiled_across_8(gpu, orc):
    """BASELINE configs[3]: the Book-1 final scene at 1920x1080 @ 500 spp, row-tiled across 8 ranks (here: the eight
    shares rendered on one GPU).  Whole frame == assembled shares; bands == oracle; ray counts add up."""
    nx, ny, ns = 1920, 1080, 500
    hs = gpu.HostScene("random_scene", nx, ny)
    ds = gpu.DeviceScene(hs)
    try:
        whole, st = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns)
        full, rays = render_partitioned(gpu, ds, hs, 8, nx=nx, ny=ny, ns=ns)
        assert rays == st.rays and bits_equal(full, whole)
        o = orc.OracleScene("bouncing", nx, ny)
        for row0 in (8, 544, 1072):
            ref, cnt = o.render(ns, row0=row0, row1=row0 + 8)
            assert bits_equal(whole[row0:row0 + 8], ref[row0:row0 + 8]), row0
            band, st_band = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns, tile_rows=8, tile_first=row0 // 8, tile_stride=10 ** 6)
            assert st_band.rays == cnt["rays"] and bits_equal(band, ref[row0:row0 + 8])
    finally:
        ds.close()


@pytest.mark.parametrize("name,nx,ny,ns,bands", [("cornell", 600, 600, 1000, (0, 296, 592)), ("final", 800, 800, 128, (96, 400, 704))])
def test_other_baseline_scenes_default_schedule(gpu, orc, name, nx, ny, ns, bands):
    """BASELINE configs[2] (Cornell 600x600 @ 1000) and the configs[4] scene (Book-2 final, 800x800, at 128 spp so the
    oracle bands stay in seconds): default options, whole frame, 8-row bands and their ray counts against the oracle."""
    img, iw, ih = gpu.default_texture(name)
    hs = gpu.HostScene(name, nx, ny, img, iw, ih)
    ds = gpu.DeviceScene(hs)
    try:
        fb, st = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns)
        o = orc.OracleScene(name, nx, ny, img, iw, ih)
        for row0 in bands:
            ref, cnt = o.render(ns, row0=row0, row1=row0 + 8)
            assert bits_equal(fb[row0:row0 + 8], ref[row0:row0 + 8]), (name, row0)
            band, st_band = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns, tile_rows=8, tile_first=row0 // 8, tile_stride=10 ** 6)
            assert st_band.rays == cnt["rays"], (name, row0)
    finally:
        ds.close()


def test_nonblocking_render_returns_before_the_frame_is_done(gpu):
    """include/rt_abi.h: with blocking = 0 the work is only enqueued on `stream`.  On an ns = 500 frame (>100 ms of
    device time) the call must return long before the kernels finish, and the frame must be the blocking one's."""
    import time
    import torch
    nx, ny, ns = 1200, 800, 500
    hs = gpu.HostScene("random_scene", nx, ny)
    ds = gpu.DeviceScene(hs)
    try:
        gpu.reset_options()
        ref = torch.zeros((ny, nx, 3), dtype=torch.float32, device="cuda")
        _, st_ref = ds.render(hs.frame(ns=ns), out=ref.data_ptr(), blocking=True)
        buf = torch.zeros_like(ref)
        s = torch.cuda.Stream()
        done = torch.cuda.Event()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _ = ds.render(hs.frame(ns=ns), out=buf.data_ptr(), stream=s.cuda_stream, blocking=False)
        t_call = time.perf_counter() - t0
        done.record(s)
        still_running = not done.query()
        st = ds.finish()
        s.synchronize()
        assert still_running, "rt_render(blocking=0) returned only after the frame had finished"
        assert t_call * 1e3 < 0.25 * st.ms_render, (t_call * 1e3, st.ms_render)
        assert st.rays == st_ref.rays
        assert torch.equal(buf.view(torch.int32), ref.view(torch.int32))
    finally:
        ds.close()


@pytest.mark.parametrize("force_rccl", [0, 1])
def test_multi_device_entry_points_on_one_gpu(gpu, force_rccl):
    """rt_multi_* (one host thread, N devices, RCCL gather + un-interleave) with N = 1, which is all a one-GPU box can run:
    plain (no gather) and with the gather forced through RCCL (multi_force_rccl: ncclCommInitAll on one device,
    ncclGather of the 4-row-tile buffer, un-interleave kernel).  Both must reproduce rt_render's frame bit for bit."""
    hs = gpu.HostScene("random_scene", 240, 136)
    gpu.reset_options()
    ds = gpu.DeviceScene(hs)
    ref, st_ref = ds.render(hs.frame(ns=64))
    ds.close()
    gpu.set_option("multi_force_rccl", force_rccl)
    ms = gpu.MultiScene(hs, 1)
    try:
        fb, st = ms.render(hs.frame(ns=64), tile_rows=4)
    finally:
        ms.close()
        gpu.reset_options()
    assert st.rays == st_ref.rays and st.samples == st_ref.samples and st.local_rows == 136
    assert bits_equal(fb, ref)


def test_raytracer_gpus_flag(gpu, tmp_path):
    """`rayTracer --gpus 1` is byte-identical to the plain run (the multi-device path is only taken for N > 1), and
    asking for more devices than the box has fails with the reference's exit code 99 instead of rendering something else."""
    import subprocess
    exe = os.path.join(gpu.LIB_DIR, "rayTracer")
    base = subprocess.run([exe, "--scene", "bouncing", "--nx", "64", "--ny", "40", "--ns", "4"], capture_output=True, timeout=120)
    one = subprocess.run([exe, "--scene", "bouncing", "--nx", "64", "--ny", "40", "--ns", "4", "--gpus", "1"], capture_output=True, timeout=120)
    assert base.returncode == 0 and one.returncode == 0 and base.stdout == one.stdout
    import torch
    too_many = torch.cuda.device_count() + 1
    bad = subprocess.run([exe, "--scene", "bouncing", "--nx", "64", "--ny", "40", "--ns", "4", "--gpus", str(too_many)], capture_output=True, timeout=120)
    assert bad.returncode == 99, bad.stderr.decode()


@pytest.mark.parametrize("world,ny", [(3, 50), (8, 136), (8, 37)])
def test_reassembly_for_worlds_a_one_gpu_box_cannot_run(gpu, world, ny):
    """rt_multi_render's un-interleave step (rt_uninterleave_kernel: staging[rank][local_row] -> frame[global_row]) on a
    synthetic `world`: the staging buffer is filled on this GPU with the rank-local renders of every rank, padded to the
    common max_rows exactly as the gather would deliver them -- including ny that is no multiple of tile_rows * world --
    and the reassembled frame must be the whole frame bit for bit."""
    import torch
    import ctypes as C
    nx, ns, tile_rows = 96, 8, 4
    hs = gpu.HostScene("random_scene", nx, ny)
    ds = gpu.DeviceScene(hs)
    try:
        whole, _ = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns)
        parts, max_rows = [], 0
        for r in range(world):
            part, st = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns, tile_rows=tile_rows, tile_first=r, tile_stride=world)
            parts.append(part)
            max_rows = max(max_rows, part.shape[0])
        staging = torch.full((world, max_rows, nx, 3), float("nan"), dtype=torch.float32, device="cuda")
        for r, part in enumerate(parts):
            if part.shape[0]:
                staging[r, : part.shape[0]] = torch.from_numpy(part).cuda()
        frame = torch.zeros((ny, nx, 3), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        st = gpu.rt_lib().rt_multi_debug_uninterleave(C.c_void_p(staging.data_ptr()), C.c_void_p(frame.data_ptr()), nx, ny, tile_rows, world, max_rows)
        assert st == 0, gpu.rt_lib().rt_last_error_detail()
        assert bits_equal(frame.cpu().numpy(), whole)
        for j in range(ny):   # ... and the host-side owner arithmetic names the same rows
            d, l = gpu.row_owner(j, tile_rows, world)
            assert bits_equal(parts[d][l], whole[j])
    finally:
        ds.close()


def test_book1_full_size_default_schedule(gpu, orc):
    """BASELINE configs[1] as the book defines it (`book1`: the book's material rules, static spheres, gradient sky) at its
    full size, 1200x800 @ 100 spp, default options: 8-row bands and their ray counts against the oracle."""
    nx, ny, ns = 1200, 800, 100
    hs = gpu.HostScene("book1", nx, ny)
    ds = gpu.DeviceScene(hs)
    try:
        fb, st = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns)
        assert st.samples == nx * ny * ns and st.kernel_variant // 1000 == 3
        o = orc.OracleScene("book1", nx, ny)
        for row0 in (0, 296, 408, 792):
            ref, cnt = o.render(ns, row0=row0, row1=row0 + 8)
            assert bits_equal(fb[row0:row0 + 8], ref[row0:row0 + 8]), row0
            band, st_band = render_default(gpu, ds, hs, nx=nx, ny=ny, ns=ns, tile_rows=8, tile_first=row0 // 8, tile_stride=10 ** 6)
            assert st_band.rays == cnt["rays"] and bits_equal(band, ref[row0:row0 + 8]), row0
    finally:
        ds.close()
