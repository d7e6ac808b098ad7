"""The N > 1 path on CPU: world_size-2 gloo processes run the same partition / gather / un-interleave logic
bench.py uses over RCCL.  Each rank fills its compact local buffer from a function of the GLOBAL row index, so the
assembled frame proves every row went to the right place exactly once (no GPU, no rendering)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ny, nx, tile_rows, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = bench.RowPlan(ny, tile_rows, world)
        rows = plan.rows_of(rank)
        local = torch.zeros((plan.max_rows, nx, 3), dtype=torch.float32)
        for k, j in enumerate(rows):
            local[k] = float(j) + torch.arange(nx, dtype=torch.float32)[:, None] * 0.001 + torch.tensor([0.0, 0.25, 0.5])
        full = bench.gather_rows(local, plan, rank, world, torch.device("cpu"))
        if rank == 0:
            ret.put(full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ny,tile_rows,world", [(800, 4, 2), (50, 8, 2), (7, 4, 2)])
def test_gather_reassembles_frame(ny, tile_rows, world):
    nx = 12
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ny, nx, tile_rows, ret)) for r in range(world)]
    for p in procs:
        p.start()
    full = ret.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.arange(ny, dtype=np.float32)[:, None, None] + np.arange(nx, dtype=np.float32)[None, :, None] * np.float32(0.001) \
        + np.array([0.0, 0.25, 0.5], np.float32)[None, None, :]
    assert np.array_equal(full, expect.astype(np.float32))


def test_row_plan_matches_c_abi():
    sys.path.insert(0, ROOT)
    import bench
    import accelerated_ray_tracer_amd as art
    hs = art.HostScene("two_spheres")
    for ny, tile, world in [(800, 4, 8), (101, 8, 4), (7, 4, 2), (600, 600, 1)]:
        plan = bench.RowPlan(ny, tile, world)
        for r in range(world):
            f = hs.frame(nx=8, ny=ny, ns=1, tile_rows=tile, tile_first=r, tile_stride=world)
            assert plan.rows_of(r) == art.local_rows_to_global(f).tolist()


def test_c_abi_row_partition_round_trips():
    """rt_multi_render (single process, N devices) deals 4-row tiles round-robin; its un-interleave kernel uses
    rt_multi_row_owner, which must be the inverse of rt_local_to_global_row for every row of every rank -- the same
    partition bench.py's RowPlan describes."""
    sys.path.insert(0, ROOT)
    import bench
    import accelerated_ray_tracer_amd as art
    hs = art.HostScene("two_spheres")
    for ny, tile, world in [(800, 4, 8), (1080, 4, 8), (101, 8, 3), (7, 4, 2), (600, 600, 1), (5, 1, 4)]:
        plan = bench.RowPlan(ny, tile, world)
        seen = np.zeros(ny, np.int32)
        for r in range(world):
            f = hs.frame(nx=8, ny=ny, ns=1, tile_rows=tile, tile_first=r, tile_stride=world)
            rows = art.local_rows_to_global(f)
            assert rows.tolist() == plan.rows_of(r)
            for k, j in enumerate(rows):
                assert art.row_owner(int(j), tile, world) == (r, k)
                seen[j] += 1
        assert (seen == 1).all()


def test_missing_rccl_library_is_an_error_status_not_a_crash():
    """rt_multi_render loads RCCL with dlopen when a gather is needed; a box without the library must get RT_ERR_HIP and the
    loader's message (the message was once built from a second, null dlerror()).  Host only, no device."""
    import accelerated_ray_tracer_amd as art
    L = art.rt_lib()
    st = L.rt_multi_probe_rccl(b"librccl_this_name_does_not_exist.so.9")
    assert st == 3, st                                                   # RT_ERR_HIP
    detail = L.rt_last_error_detail().decode()
    assert "cannot load librccl.so" in detail and "librccl_this_name_does_not_exist.so.9" in detail, detail
